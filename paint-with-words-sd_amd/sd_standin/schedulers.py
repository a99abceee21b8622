"""Scheduler stand-ins (diffusers==0.10.0 is pinned by the reference but not installable here).

``LMSDiscreteScheduler`` is the scheduler every reference entry point defaults to
(paint_with_words/paint_with_words.py:130, :400) and the only one its loop can run: the loop reads
``scheduler.sigmas[step_index]`` with ``step_index = (timesteps == t).nonzero().item()``
(:473-474). The published K-LMS algorithm is restated here: scaled-linear betas,
``sigma = sqrt((1 - abar) / abar)`` interpolated at ``linspace(0, T-1, n)[::-1]``, order-4 linear
multistep coefficients by numerical integration (scipy ``quad``, epsrel 1e-4).

``PLMSScheduler`` is the PNDM/PLMS scheduler (``skip_prk_steps=True``, ``steps_offset=1``: the SD
configuration) that BASELINE.json's headline config names. The reference cannot run it
(SURVEY.md section 8 a-note: no ``.sigmas`` and a repeated timestep), so its PwW sigma is DEFINED
here as ``sigma_t = sqrt((1 - abar_t) / abar_t)`` at the step's timestep and the step index is the
loop counter (``sigmas[i]`` / ``timesteps[i]``), which this class exposes with the same attribute
names so the same sampling loop drives both.
"""
from types import SimpleNamespace

import numpy as np
import torch
from scipy import integrate


def _scaled_linear_alphas_cumprod(beta_start, beta_end, num_train_timesteps):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class _Config(dict):
    """dict with attribute access (the reference calls ``scheduler.config.get("steps_offset", 0)``, :435)."""
    __getattr__ = dict.get


class LMSDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear"):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError("stand-in implements the scaled_linear schedule SD uses")
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule)
        self.alphas_cumprod = _scaled_linear_alphas_cumprod(beta_start, beta_end, num_train_timesteps)
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.concatenate([sigmas[::-1], [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self.init_noise_sigma = self.sigmas.max()
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=float)[::-1].copy())
        self.derivatives = []
        self._coeff_cache = {}

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        timesteps = np.linspace(0, self.config["num_train_timesteps"] - 1, num_inference_steps, dtype=float)[::-1].copy()
        sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()
        sigmas = np.interp(timesteps, np.arange(0, len(sigmas)), sigmas)
        sigmas = np.concatenate([sigmas, [0.0]]).astype(np.float32)
        self.sigmas = torch.from_numpy(sigmas)
        self.timesteps = torch.from_numpy(timesteps)
        if device is not None:
            self.sigmas = self.sigmas.to(device)
            self.timesteps = self.timesteps.to(device)
        self.derivatives = []
        self._coeff_cache = {}

    def _index(self, timestep):
        if torch.is_tensor(timestep):
            timestep = timestep.to(self.timesteps.device)
        return int((self.timesteps == timestep).nonzero().item())

    def scale_model_input(self, sample, timestep):
        sigma = self.sigmas[self._index(timestep)]
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def get_lms_coefficient(self, order, t, current_order):
        key = (order, t, current_order)
        if key not in self._coeff_cache:
            def lms_derivative(tau):
                prod = 1.0
                for k in range(order):
                    if current_order == k:
                        continue
                    prod *= (tau - self.sigmas[t - k]) / (self.sigmas[t - current_order] - self.sigmas[t - k])
                return prod
            self._coeff_cache[key] = integrate.quad(lms_derivative, self.sigmas[t], self.sigmas[t + 1], epsrel=1e-4)[0]
        return self._coeff_cache[key]

    def step(self, model_output, timestep, sample, order=4):
        step_index = self._index(timestep)
        sigma = self.sigmas[step_index]
        pred_original_sample = sample - sigma * model_output           # epsilon prediction
        derivative = (sample - pred_original_sample) / sigma
        self.derivatives.append(derivative)
        if len(self.derivatives) > order:
            self.derivatives.pop(0)
        order = min(step_index + 1, order)
        coeffs = [self.get_lms_coefficient(order, step_index, o) for o in range(order)]
        prev_sample = sample + sum(c * d for c, d in zip(coeffs, reversed(self.derivatives)))
        return SimpleNamespace(prev_sample=prev_sample, pred_original_sample=pred_original_sample)

    def add_noise(self, original_samples, noise, timesteps):
        sigmas = self.sigmas.to(device=original_samples.device, dtype=original_samples.dtype)
        idx = [self._index(t) for t in timesteps]
        sigma = sigmas[idx].flatten()
        while sigma.dim() < original_samples.dim():
            sigma = sigma.unsqueeze(-1)
        return original_samples + noise * sigma

    def __len__(self):
        return self.config["num_train_timesteps"]


class PLMSScheduler:
    """PNDM with skip_prk_steps (pseudo linear multistep, Liu et al. 2022), SD settings."""
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1):
        if beta_schedule != "scaled_linear":
            raise NotImplementedError("stand-in implements the scaled_linear schedule SD uses")
        self.config = _Config(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                              beta_schedule=beta_schedule, steps_offset=steps_offset, skip_prk_steps=True)
        self.alphas_cumprod = _scaled_linear_alphas_cumprod(beta_start, beta_end, num_train_timesteps)
        self.final_alpha_cumprod = self.alphas_cumprod[0]        # set_alpha_to_one=False
        self.init_noise_sigma = torch.tensor(1.0)
        self.timesteps = None
        self.sigmas = None
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        self.step_ratio = self.config["num_train_timesteps"] // num_inference_steps
        base = (np.arange(0, num_inference_steps) * self.step_ratio).round() + self.config["steps_offset"]
        # the second-to-last timestep is visited twice (the PLMS warm-up evaluation)
        plms = np.concatenate([base[:-1], base[-2:-1], base[-1:]])[::-1].copy()
        self.timesteps = torch.from_numpy(plms.astype(np.int64))
        ac = self.alphas_cumprod[self.timesteps]
        # PwW sigma of each step (defined here, see module docstring); trailing 0 mirrors LMS.sigmas
        self.sigmas = torch.cat([((1 - ac) / ac) ** 0.5, torch.zeros(1)]).to(torch.float32)
        if device is not None:
            self.timesteps = self.timesteps.to(device)
        self.ets = []
        self.counter = 0
        self.cur_sample = None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _get_prev_sample(self, sample, timestep, prev_timestep, model_output):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        sample_coeff = (a_prev / a_t) ** 0.5
        denom = a_t * b_prev ** 0.5 + (a_t * b_t * a_prev) ** 0.5
        return float(sample_coeff) * sample - float((a_prev - a_t) / denom) * model_output

    def step(self, model_output, timestep, sample):
        timestep = int(timestep)
        prev_timestep = timestep - self.step_ratio
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(model_output)
        else:
            prev_timestep = timestep
            timestep = timestep + self.step_ratio
        if len(self.ets) == 1 and self.counter == 0:
            self.cur_sample = sample
        elif len(self.ets) == 1 and self.counter == 1:
            model_output = (model_output + self.ets[-1]) / 2
            sample = self.cur_sample
            self.cur_sample = None
        elif len(self.ets) == 2:
            model_output = (3 * self.ets[-1] - self.ets[-2]) / 2
        elif len(self.ets) == 3:
            model_output = (23 * self.ets[-1] - 16 * self.ets[-2] + 5 * self.ets[-3]) / 12
        else:
            model_output = (1 / 24) * (55 * self.ets[-1] - 59 * self.ets[-2] + 37 * self.ets[-3] - 9 * self.ets[-4])
        prev_sample = self._get_prev_sample(sample, timestep, prev_timestep, model_output)
        self.counter += 1
        return SimpleNamespace(prev_sample=prev_sample)

    def __len__(self):
        return self.config["num_train_timesteps"]

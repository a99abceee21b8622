"""Denoising driver around the HIP attention path (reference loop: paint_with_words/paint_with_words.py
:431-506; inpaint variant paint_with_words_inpaint.py:230-266).

Three execution modes over the SAME arithmetic:
  "eager"   two batch-1 UNet calls per step (cond dict, then uncond dict), exactly the reference's
            call pattern (:483-499) -- the parity mode.
  "folded"  cond and uncond rows of all images in ONE UNet call ([cond rows..., uncond rows...]); the
            PwW bias is gated per row by the `_PWW_ROW_GATE` coefficients (kernel arg bias_coeff), and
            the qk reductions stay per image (QKProxy).
  "graph"   "folded", with each step's UNet call captured once into a hipGraph (torch.cuda.CUDAGraph
            on ROCm) and replayed: at batch 1 the UNet is launch-bound (~1.5k kernels per call), so
            replay removes the host launch cost. One graph per step index because the user's
            weight_function bakes the step's sigma into kernel arguments as a Python float.
"""
import torch

from . import ops
from .attention import install, refresh_kv_cache, refresh_orig_cache, ROW_GATE, KV_CACHE


def initial_latents(seed, in_channels, height, width, region_masks=None, extra_seeds=None, batch_seeds=None):
    """:445-455. CPU generator, exactly as the reference, so results do not depend on the device or
    the number of GPUs. `region_masks`: callable(dtype, size) -> list of [1,1,h,w] masks for the seeded
    regions (the reference's _get_binary_mask). `batch_seeds`: one latent per seed (batched mode)."""
    seeds = [seed] if batch_seeds is None else list(batch_seeds)
    size = (1, in_channels, height // 8, width // 8)
    out = []
    for s in seeds:
        latents = torch.randn(size, generator=torch.manual_seed(s))
        if extra_seeds:
            print("Use region based seeding: ", extra_seeds)
            multi = [torch.randn(size, generator=torch.manual_seed(_s)) for _s in extra_seeds.values()]
            masks = region_masks(latents[0].dtype, size[-2:])
            foreground = (sum(masks) > 0).squeeze()
            summed = sum(l * m for l, m in zip(multi, masks))
            latents[:, :, foreground] = summed[:, :, foreground]
        out.append(latents)
    return torch.cat(out, dim=0)


def _as_list(x, n):
    """One context dict per image: a single dict serves every image of the batch."""
    if isinstance(x, (list, tuple)):
        if len(x) not in (1, n):
            raise ValueError("got %d context dicts for %d images" % (len(x), n))
        return list(x) if len(x) == n else [x[0]] * n
    return [x] * n


def _fold_context(cond, uncond, n_images, device):
    """Build the single dict of a folded call: rows [cond x n, uncond x n]. `cond` / `uncond`: one dict, or one dict per
    image (paint_with_words_batch: per-image prompts and color maps). Per-image weight maps are stacked to
    [2n, 1, N, 77] -- the kernels take the image axis through bias_stride[0] -- with zeros for the unconditional rows
    (their gate is 0, the values are never used); a shared map stays [N, 77]."""
    conds, unconds = _as_list(cond, n_images), _as_list(uncond, n_images)
    shared = all(c is conds[0] for c in conds)

    def rows(dicts):
        if all(d is dicts[0] for d in dicts):
            t = dicts[0]["CONTEXT_TENSOR"]
            return t.expand(n_images, -1, -1) if t.shape[0] == 1 else t
        return torch.cat([d["CONTEXT_TENSOR"] for d in dicts], dim=0)

    folded = dict(conds[0])
    folded[KV_CACHE] = {}
    folded.pop("_PWW_ORIG_CACHE", None)
    folded["CONTEXT_TENSOR"] = torch.cat([rows(conds), rows(unconds)], dim=0).contiguous()
    folded[ROW_GATE] = torch.cat([torch.ones(n_images), torch.zeros(n_images)]).to(device=device, dtype=torch.float32)
    if not shared:
        for key in [k for k in conds[0] if k.startswith("CROSS_ATTENTION_WEIGHT_")]:
            maps = [c[key] for c in conds]
            if not all(torch.is_tensor(m) for m in maps):
                raise ValueError("per-image contexts must all carry a tensor for %s" % key)
            w = torch.stack(maps, dim=0)                       # [n, N, 77]  (ORIG: [n, H, W, 77])
            w = torch.cat([w, torch.zeros_like(w)], dim=0)
            folded[key] = w if key.endswith("_ORIG") else w.unsqueeze(1)
    return folded


def weight_function_signature(f):
    """What a captured hipGraph depends on in a weight function: its code and the constants it can see (closure cells,
    defaults, numeric globals it names). The reference's callers pass a FRESH lambda per request (runner.py:104,
    gradio_pww.py:43), so identity is useless as a cache key; two lambdas with the same code and constants replay the
    same graphs, and a changed constant (the graphs bake `c0 * g(sigma)` into kernel arguments) re-captures. Values that
    are not plain numbers / strings are keyed by identity."""
    def atom(v):
        if isinstance(v, (int, float, str, bool, bytes, type(None))):
            return ("v", type(v).__name__, v)
        if isinstance(v, tuple) and all(isinstance(x, (int, float, str, bool, type(None))) for x in v):
            return ("t", v)
        return ("id", id(v))
    code = getattr(f, "__code__", None)
    if code is None:       # callable object / builtin: identity
        return ("callable", id(f))
    consts = tuple(atom(c) if not hasattr(c, "co_code") else ("code", c.co_code, c.co_names) for c in code.co_consts)
    cells = tuple(atom(c.cell_contents) for c in (f.__closure__ or ()))
    defaults = tuple(atom(v) for v in (f.__defaults__ or ())) + tuple(sorted((k, atom(v)) for k, v in (f.__kwdefaults__ or {}).items()))
    g = getattr(f, "__globals__", {})
    globs = tuple((n, atom(g[n])) for n in code.co_names if n in g and isinstance(g[n], (int, float, str, bool)))
    return ("code", code.co_code, consts, code.co_names, code.co_varnames[:code.co_argcount], cells, defaults, globs)


class _GraphedUNet:
    """Capture-once / replay-many wrapper of one folded UNet call per step index."""

    def __init__(self, unet):
        self.unet = unet
        self.graphs = {}
        self.pool = None
        self.static_in = None
        self.static_t = None

    def reset(self):
        """Drop every captured graph AND the memory pool they shared: once the last graph of a pool is gone the
        allocator retires the pool, and capturing into the stale handle trips an internal assert."""
        self.graphs.clear()
        self.pool = None

    def __call__(self, key, x, t_value, context):
        if self.static_in is None or self.static_in.shape != x.shape or self.static_in.dtype != x.dtype:
            self.reset()
            self.static_in = torch.empty_like(x)
            self.static_t = torch.zeros((), dtype=torch.float32, device=x.device)
        self.static_in.copy_(x)
        self.static_t.fill_(float(t_value))
        entry = self.graphs.get(key)
        if entry is None:
            # warm-up on a side stream (required before capture), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                self.unet(self.static_in, self.static_t, encoder_hidden_states=context)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            with torch.cuda.graph(g, pool=self.pool):
                out = self.unet(self.static_in, self.static_t, encoder_hidden_states=context).sample
            entry = (g, out)
            self.graphs[key] = entry
        g, out = entry
        g.replay()
        return out


class PwWSampler:
    """Runs the reference's sampling loop for one or many images on one GPU."""

    def __init__(self, unet, scheduler, mode="eager"):
        if mode not in ("eager", "folded", "graph"):
            raise ValueError("mode must be eager | folded | graph")
        self.unet, self.scheduler, self.mode = unet, scheduler, mode
        install(unet)
        self._graphed = _GraphedUNet(unet) if mode == "graph" else None
        self._graph_sig = None
        self._static_folded = None

    def _static_context(self, folded, weight_function, latents, timesteps):
        """Captured graphs read the context tensors by ADDRESS: keep one set of static tensors alive
        and copy each new request's values into them; rebuild the graphs when the geometry, the
        weight function (code or constants: they are baked into kernel arguments) or the schedule changes."""
        def tensor_sig(d):
            return tuple(sorted((k, tuple(v.shape), v.dtype) for k, v in d.items() if torch.is_tensor(v)))
        sig = (weight_function_signature(weight_function), tuple(latents.shape), tuple(float(t) for t in timesteps), tensor_sig(folded))
        if sig != self._graph_sig:
            self._graphed.reset()
            self._graph_sig = sig
            self._static_folded = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in folded.items()}
        else:
            for k, v in folded.items():
                if torch.is_tensor(v):
                    self._static_folded[k].copy_(v)
            refresh_kv_cache(self._static_folded)     # new prompt embedding -> new K|V, same addresses
            refresh_orig_cache(self._static_folded)   # new color map -> new fallback maps, same addresses
        return self._static_folded

    def _sigma_and_index(self, i, t):
        sch = self.scheduler
        ts = sch.timesteps
        # the reference looks the index up by timestep equality (:473); PLMS repeats a timestep, so a
        # scheduler that is not 1:1 falls back to the loop counter (SURVEY.md 8 a-note)
        hits = (ts == t).nonzero()
        idx = int(hits.item()) if hits.numel() == 1 else i
        return sch.sigmas[idx], idx

    @torch.no_grad()
    def sample(self, cond, uncond, latents, timesteps, guidance_scale, weight_function, extra_channels=None,
               on_step=None):
        """cond / uncond: the two context dicts of the PwW protocol, or one dict per image (per-image prompts / maps).
        latents: [n_images, C, h, w] already scaled by init_noise_sigma (or noised for img2img).
        extra_channels: inpaint's cat([mask, masked_image_latents]) ([n or 1, 5, h, w]) or None."""
        sch, unet, dev = self.scheduler, self.unet, latents.device
        install(unet)      # the plug is a class-level patch (:193-195): put it back if somebody removed it since __init__
        n = latents.shape[0]
        udt = unet.dtype if hasattr(unet, "dtype") else next(unet.parameters()).dtype
        conds, unconds = _as_list(cond, n), _as_list(uncond, n)
        if self.mode == "eager":   # per-request caches of the fused K|V projections (prompt constant over the steps)
            for d in conds + unconds:
                d.setdefault(KV_CACHE, {}).clear()
        folded = None
        if self.mode != "eager":
            folded = _fold_context(cond, uncond, n, dev)
            if self._graphed is not None:
                folded = self._static_context(folded, weight_function, latents, timesteps)
        if extra_channels is not None and extra_channels.shape[0] != n:
            extra_channels = extra_channels.expand(n, -1, -1, -1)
        for i, t in enumerate(timesteps):
            sigma, _ = self._sigma_and_index(i, t)
            x = sch.scale_model_input(latents, t)
            if extra_channels is not None:
                x = torch.cat([x, extra_channels.to(x.dtype)], dim=1)
            if self.mode == "eager":
                eps_c, eps_u = [], []
                for j in range(n):   # the reference is batch-1 (:445); images are independent
                    conds[j].update({"SIGMA": sigma, "WEIGHT_FUNCTION": weight_function})
                    eps_c.append(unet(x[j:j + 1], t, encoder_hidden_states=conds[j]).sample)
                    unconds[j].update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
                    eps_u.append(unet(x[j:j + 1], t, encoder_hidden_states=unconds[j]).sample)
                eps_c, eps_u = torch.cat(eps_c), torch.cat(eps_u)
            else:
                folded.update({"SIGMA": sigma, "WEIGHT_FUNCTION": weight_function})
                x2 = torch.cat([x, x], dim=0).to(udt)
                if self.mode == "graph":
                    out = self._graphed(i, x2, float(t), folded)
                else:
                    out = unet(x2, t, encoder_hidden_states=folded).sample
                eps_c, eps_u = out[:n], out[n:]
            if eps_c.dtype in (torch.float16, torch.bfloat16):
                noise_pred = ops.cfg_combine(eps_c, eps_u, guidance_scale)     # fp32, :501-503
            else:
                noise_pred = eps_u + guidance_scale * (eps_c - eps_u)
            latents = sch.step(noise_pred, t, latents).prev_sample
            if on_step is not None:
                on_step(i, t, latents)
        return latents

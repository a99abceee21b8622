"""Shared, seeded test-case definitions (test infrastructure).

Used by oracle/make_golden.py (which runs the REAL reference on them, in the build container) and by
the tests (which run the oracle restatement and the HIP path on the same inputs), so every side sees
identical inputs. Settings mirror the reference's own examples (runner.py:9-72) and BASELINE.md.
"""
import math
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "paint-with-words-sd_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

# runner.py:9-20 (EXAMPLE_SETTING_1) -- the 5-region example of BASELINE configs 1/2
RUNNER_CONTEXT = {
    (0, 0, 0): "cat,1.0",
    (255, 255, 255): "dog,1.0",
    (13, 255, 0): "tree,1.5",
    (90, 206, 255): "sky,0.2",
    (74, 18, 1): "ground,0.2",
}
RUNNER_PROMPT = "realistic photo of a dog, cat, tree, with beautiful sky, on sandy ground"

# runner.py:61-72 (EXAMPLE_SETTING_4_seed): seed grammar, one region seeded
AURORA_SEED_CONTEXT = {
    (7, 9, 182): "aurora,0.5,-1",
    (136, 178, 92): "full moon,1.5,-1",
    (51, 193, 217): "mountains,0.4,-1",
    (61, 163, 35): "a half-frozen lake,0.3,-1",
    (89, 102, 255): "boat,2.0,2077",
}
AURORA_PROMPT = ("A digital painting of a half-frozen lake near mountains under a full moon and aurora. "
                 "A boat is in the middle of the lake. Highly detailed.")


def weight_fn_runner(w, sigma, qk):          # runner.py:104
    return 0.4 * w * math.log(1 + sigma) * qk.max()


def weight_fn_default(w, sigma, qk):         # paint_with_words.py:402-405
    return 0.1 * w * math.log(sigma + 1) * qk.max()


def weight_fn_std(w, sigma, qk):             # README.md:152
    return 0.4 * w * math.log(sigma ** 2 + 1) * qk.std()


WEIGHT_FUNCTIONS = {"runner": weight_fn_runner, "default": weight_fn_default, "std": weight_fn_std}


def stripes_case(n_regions=8, size=512):
    """BASELINE config 3: n vertical stripes, colors (32 i, 255 - 32 i, (97 i) % 256), strengths
    0.2 + 0.2 i, single-word phrases all present in the prompt."""
    words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet",
             "kilo", "lima", "mike", "november", "oscar", "papa"][:n_regions]
    img = np.zeros((size, size, 3), dtype=np.uint8)
    ctx = {}
    w = size // n_regions
    for i in range(n_regions):
        color = ((32 * i) % 256, (255 - 32 * i) % 256, (97 * i) % 256)
        img[:, i * w:(i + 1) * w] = color
        ctx[color] = f"{words[i]},{0.2 + 0.2 * i:.1f}"
    prompt = "a photo of " + " ".join(words)
    return img, ctx, prompt


def stripes_batch_case(j, n_regions=8, size=512):
    """Image j of BASELINE config 3's batch: the stripes map rotated by j stripes (so every image of the batch has its
    OWN weight maps -- the per-image bias path), same color_context and prompt, seed j."""
    img, ctx, prompt = stripes_case(n_regions, size)
    return np.ascontiguousarray(np.roll(img, j * (size // n_regions), axis=1)), ctx, prompt


def grid_case(rows=3, cols=4, height=768, width=768, seeds=True):
    """BASELINE config 5: rows x cols grid, per-region seeds 1000 + i."""
    words = ["alpha", "bravo", "charlie", "delta", "echo", "foxtrot", "golf", "hotel", "india", "juliet",
             "kilo", "lima"]
    img = np.zeros((height, width, 3), dtype=np.uint8)
    ctx = {}
    k = 0
    for r in range(rows):
        for c in range(cols):
            color = (20 * k + 5, 255 - 20 * k, (53 * k) % 256)
            img[r * height // rows:(r + 1) * height // rows, c * width // cols:(c + 1) * width // cols] = color
            ctx[color] = f"{words[k]},{0.3 + 0.1 * k:.1f}" + (f",{1000 + k}" if seeds else "")
            k += 1
    return img, ctx, "a painting of " + " ".join(words[: rows * cols])


def grid_batch_case(j, rows=3, cols=4, height=768, width=768):
    """Image j of a config-5 style batch with PER-IMAGE maps: the grid rolled by j cells to the right (so the weight maps and
    the region-seed placement differ per image), same color_context and prompt, seed j. j = 0 is bench.py's request."""
    img, ctx, prompt = grid_case(rows, cols, height, width, seeds=True)
    return np.ascontiguousarray(np.roll(img, j * (width // cols), axis=1)), ctx, prompt


def load_example_rgb():
    """The reference's contents/example_input.png, re-saved under tests/golden/ by make_golden.py."""
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, "example_input.png")).convert("RGB"))


def load_aurora_rgb():
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, "aurora_1.png")).convert("RGB"))


def synthetic_init_image(size=512, seed=81):
    """BASELINE config 4: init image = randn-derived uint8 (the reference's aurora_1_output.png is a
    real-weights artifact)."""
    g = torch.Generator().manual_seed(seed)
    low = torch.randn(1, 3, size // 16, size // 16, generator=g)
    img = torch.nn.functional.interpolate(low, size=(size, size), mode="bilinear", align_corners=False)[0]
    img = ((img * 0.25 + 0.5).clamp(0, 1) * 255).round().to(torch.uint8).permute(1, 2, 0).contiguous().numpy()
    return img


def load_moon_mask():
    from PIL import Image
    return Image.open(os.path.join(GOLDEN, "moon_mask_L.png"))


def weight_fn_inpaint(w, sigma, qk):         # runner_inpaint.py:87
    return 0.15 * w * math.log(1 + sigma) * qk.max()


INPAINT_CONTEXT = {k: ",".join(v.split(",")[:2]) for k, v in list(AURORA_SEED_CONTEXT.items())[:4]}


# ---- seeded attention-module cases (SURVEY.md section 8c pin (1)) -------------------------------
# name: (N tokens, channels C, heads, ctx_dim)
ATTN_SHAPES = {
    "sd15_n4096": (4096, 320, 8, 768),
    "sd15_n1024": (1024, 640, 8, 768),
    "sd15_n256": (256, 1280, 8, 768),
    "sd15_n64": (64, 1280, 8, 768),
    "sd21_n576": (576, 1280, 20, 1024),
}
ATTN_MODES = ("self", "cond", "uncond", "tensor")


def make_attention_case(shape_name, seed=0, qk_gain=3.0):
    """Seeded CrossAttention stand-in + inputs. qk_gain widens the logit spread so softmax is not
    near-uniform (default nn init gives logits with std ~0.3)."""
    from sd_standin import CrossAttention
    N, C, H, ctx_dim = ATTN_SHAPES[shape_name]
    g = torch.Generator().manual_seed(1000 + seed)
    state = torch.random.get_rng_state()
    torch.manual_seed(2000 + seed)
    try:
        attn_self = CrossAttention(C, None, H, C // H)
        attn_cross = CrossAttention(C, ctx_dim, H, C // H)
    finally:
        torch.random.set_rng_state(state)
    for m in (attn_self, attn_cross):
        m.requires_grad_(False)
        m.to_q.weight.mul_(qk_gain)
    hidden = torch.randn(1, N, C, generator=g)
    ctx = torch.randn(1, 77, ctx_dim, generator=g)
    w = (torch.rand(N, 77, generator=g) < 0.15).float() * torch.rand(N, 77, generator=g) * 1.5
    w[:, 20:] = 0.0   # only prompt positions < 20 carry region weight, like a real prompt
    return dict(attn_self=attn_self, attn_cross=attn_cross, hidden=hidden, ctx=ctx, w=w, N=N, C=C, H=H)


def attention_context(case, mode, weight_fn, sigma=7.8399):
    N = case["N"]
    if mode == "self":
        return None
    if mode == "tensor":
        return case["ctx"]
    d = {"CONTEXT_TENSOR": case["ctx"], "SIGMA": torch.tensor(sigma)}
    if mode == "cond":
        d[f"CROSS_ATTENTION_WEIGHT_{N}"] = case["w"]
        d["WEIGHT_FUNCTION"] = weight_fn
    else:
        d[f"CROSS_ATTENTION_WEIGHT_{N}"] = 0
        d["CROSS_ATTENTION_WEIGHT_ORIG"] = 0
        d["WEIGHT_FUNCTION"] = lambda w, sigma, qk: 0.0
    return d


def subsample_rows(n):
    """Row indices stored in the golden files (keeps fixtures small)."""
    step = max(1, n // 64)
    return np.arange(0, n, step)


# ---- full-loop cases ---------------------------------------------------------------------------

def build_tools(config_name="tiny", dtype=torch.float32, device="cpu", scheduler="lms", qk_gain=2.0):
    """(vae, unet, text_encoder, tokenizer, scheduler) stand-ins with the BASELINE seeds."""
    from sd_standin import (build_unet, SD15_CONFIG, SD15_INPAINT_CONFIG, SD21_CONFIG, TINY_CONFIG, TINY_SD2_CONFIG,
                            HashTokenizer, TinyTextEncoder, TinyVAE, LMSDiscreteScheduler, PLMSScheduler)
    cfg = {"tiny": TINY_CONFIG, "tiny_sd2": TINY_SD2_CONFIG, "sd15": SD15_CONFIG, "sd15_inpaint": SD15_INPAINT_CONFIG, "sd21": SD21_CONFIG,
           "tiny_inpaint": dict(TINY_CONFIG, in_channels=9)}[config_name]
    unet = build_unet(cfg, seed=1234, dtype=dtype, device=device, qk_gain=qk_gain)
    text = TinyTextEncoder(cfg["cross_attention_dim"], seed=1235).to(device=device, dtype=dtype)
    vae = TinyVAE(4, seed=1236).to(device=device, dtype=dtype)
    tok = HashTokenizer()
    if scheduler == "lms":
        sch = LMSDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                   num_train_timesteps=1000)
    else:
        sch = PLMSScheduler()
    return vae, unet, text, tok, sch


# ---- pipeline-class cases: keyword arguments shared by the reference's classes (AST-loaded, oracle/make_golden.py
# pipelines) and this repo's classes -- both have the reference's __call__ signature (paint_with_words.py:629-650,
# paint_with_words_inpaint.py:340-362). Every case exercises something the pipeline classes do DIFFERENTLY from the function API.
PIPE_SEED_SIGMA_CONTEXT = {(0, 0, 0): "cat,1.0,42,4.0", (255, 255, 255): "dog,1.0,7", (13, 255, 0): "tree,1.5,-1,9.5",
                           (90, 206, 255): "sky,0.2", (74, 18, 1): "ground,0.2"}
PIPE_INPAINT_CONTEXT = dict(INPAINT_CONTEXT)
PIPE_INPAINT_CONTEXT[(136, 178, 92)] = "full moon,1.5,-1,6.0"       # a blur sigma the class parses and drops (:574)


def pipe_case(name):
    """(unet config, class kind, call kwargs, global torch seed or None) of a pipeline-class case.
      txt2img  region seeds used, blur sigmas IGNORED (:574), negative_prompt, callback every 2nd step
      hw384    height = width = 384 with a 512 x 512 color map: the latent is sized by height / width (:756), the weight maps by
               the color map -> every layer takes the CROSS_ATTENTION_WEIGHT_ORIG fallback (:96-101)
      img2img  image= + eta as the img2img strength (:735); the noise comes from the GLOBAL generator (:771)
      inpaint  the inpaint class: eta as strength (paint_with_words_inpaint.py:441), seed-driven noise (:467-473), callback"""
    from PIL import Image
    ex, au = Image.fromarray(load_example_rgb()), Image.fromarray(load_aurora_rgb())
    init = Image.fromarray(synthetic_init_image())
    if name == "txt2img":
        return "tiny", "txt2img", dict(prompt=RUNNER_PROMPT, color_map_image=ex, color_context=dict(PIPE_SEED_SIGMA_CONTEXT),
                                       weight_function=weight_fn_runner, num_inference_steps=6, guidance_scale=7.5,
                                       negative_prompt="blurry, low quality", seed=3, callback_steps=2), None
    if name == "hw384":
        return "tiny", "txt2img", dict(prompt=RUNNER_PROMPT, color_map_image=ex, color_context=dict(RUNNER_CONTEXT),
                                       weight_function=weight_fn_runner, height=384, width=384, num_inference_steps=4, guidance_scale=5.0,
                                       seed=5), None
    if name == "img2img":
        return "tiny", "txt2img", dict(prompt=RUNNER_PROMPT, color_map_image=ex, color_context=dict(RUNNER_CONTEXT),
                                       weight_function=weight_fn_default, num_inference_steps=6, eta=0.6, seed=9, image=init), 123
    if name == "inpaint":
        return "tiny_inpaint", "inpaint", dict(prompt=AURORA_PROMPT, image=init, mask_image=load_moon_mask(), color_map_image=au,
                                               color_context=dict(PIPE_INPAINT_CONTEXT), weight_function=weight_fn_inpaint,
                                               num_inference_steps=8, guidance_scale=7.5, eta=0.75, seed=81, callback_steps=3), None
    raise KeyError(name)


PIPE_CASES = ("txt2img", "hw384", "img2img", "inpaint")

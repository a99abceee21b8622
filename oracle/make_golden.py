"""Generate tests/golden/* by running the REAL reference (AST-loaded, unmodified) in this container.

TEST INFRASTRUCTURE. Run once here (`python oracle/make_golden.py [--full]`); the outputs are small
fixtures that travel to the GPU box, where /root/reference does not exist. The reference ships no
golden vectors of its own; these pin oracle/pww_oracle.py and, through it, the
HIP path.

Fixtures written:
  kat.json                 always_round / _extract_seed_and_sigma_from_context known answers
  example_input.png, aurora_1.png   the reference's input assets, re-encoded (inputs, not code)
  masks_<case>.npz         W tensors of _tokens_img_attention_weight (non-zero columns only)
  attn_<shape>.npz         inj_forward outputs (row subsample) for seeded CrossAttention stand-ins
  loop_<case>.npz          final latents of the reference's paint_with_words loop on stand-in modules
"""
import copy
import json
import math
import os
import sys
import time

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

import pww_cases as cases  # noqa: E402
from oracle import ref_loader  # noqa: E402

GOLDEN = cases.GOLDEN


def sparse_cols(w):
    w = np.asarray(w, np.float32)
    nz = np.nonzero(np.abs(w).sum(axis=0))[0]
    return nz.astype(np.int32), w[:, nz].copy()


def gen_kat(ref):
    xs = [0.5, 1.5, 2.5, 3.5, 63.5, 8.4999, 64.0, 7.5, 0.49, 96.0, 23.99, 12.5]
    kat = {"always_round": [[x, ref["always_round"](x)] for x in xs], "extract": []}
    for ctx in ({"a": "boat,2.0,2077", "b": "full moon,1.5,7,3.0", "c": "sky,0.2", "d": "x,1.0,-1", "e": "lake,0.3,-1,2.5"},
                dict((str(k), v) for k, v in cases.AURORA_SEED_CONTEXT.items())):
        c = dict(ctx)
        out, seeds, sigmas = ref["_extract_seed_and_sigma_from_context"](c)
        kat["extract"].append({"input": ctx, "output": dict(out), "seeds": {str(k): v for k, v in seeds.items()},
                               "sigmas": {str(k): v for k, v in sigmas.items()}})
    json.dump(kat, open(os.path.join(GOLDEN, "kat.json"), "w"), indent=1)
    print("kat.json", kat["always_round"])


def gen_masks(ref):
    from sd_standin import HashTokenizer
    tok = HashTokenizer()
    ex = np.array(Image.open(os.path.join(ref_loader.REFERENCE_ROOT, "contents", "example_input.png")).convert("RGB"))
    au = np.array(Image.open(os.path.join(ref_loader.REFERENCE_ROOT, "contents", "aurora_1.png")).convert("RGB"))
    Image.fromarray(ex).save(os.path.join(GOLDEN, "example_input.png"), optimize=True)
    Image.fromarray(au).save(os.path.join(GOLDEN, "aurora_1.png"), optimize=True)
    stripes, sctx, sprompt = cases.stripes_case()
    grid, gctx, gprompt = cases.grid_case(seeds=False)
    todo = {
        "example": (ex, dict(cases.RUNNER_CONTEXT), cases.RUNNER_PROMPT),
        "aurora": (au, {k: ",".join(v.split(",")[:2]) for k, v in cases.AURORA_SEED_CONTEXT.items()}, cases.AURORA_PROMPT),
        "stripes8": (stripes, sctx, sprompt),
        "grid768": (grid, gctx, gprompt),
        "nonsquare": (ex[:384, :448].copy(), dict(cases.RUNNER_CONTEXT), cases.RUNNER_PROMPT),
        "odd500": (ex[:500, :500].copy(), dict(cases.RUNNER_CONTEXT), cases.RUNNER_PROMPT + " dog"),  # repeated phrase
        "blur": (ex, {(0, 0, 0): "cat,1.0,-1,4.0", (255, 255, 255): "dog,1.0", (13, 255, 0): "tree,1.5,-1,9.5"}, cases.RUNNER_PROMPT),
    }
    for name, (img, ctx, prompt) in todo.items():
        pil = Image.fromarray(img)
        text_input = tok([prompt], padding="max_length", max_length=77, truncation=True, return_tensors="pt")
        ctx2, seeds, sigmas = ref["_extract_seed_and_sigma_from_context"](dict(ctx))
        sep, w, h = ref["_image_context_seperator"](pil, ctx2, tok)
        if sigmas:
            sep = ref["_blur_image_mask"](sep, sigmas)
        out = {"width": w, "height": h, "token_ids": text_input["input_ids"][0].numpy()}
        for r in (8, 16, 32, 64):
            wt = ref["_tokens_img_attention_weight"](sep, text_input, ratio=r).numpy()
            cols, vals = sparse_cols(wt)
            out[f"cols_{r}"], out[f"vals_{r}"], out[f"shape_{r}"] = cols, vals, np.array(wt.shape)
        w1 = ref["_tokens_img_attention_weight"](sep, text_input, ratio=1, original_shape=True).numpy()
        out["orig_colsum"] = w1.sum(axis=(0, 1), dtype=np.float64)
        out["orig_shape"] = np.array(w1.shape)
        np.savez_compressed(os.path.join(GOLDEN, f"masks_{name}.npz"), **out)
        print(f"masks_{name}: cols {out['cols_8'].tolist()} colsum8 {out['vals_8'].sum():.3f}")
    # _get_binary_mask (region seeding, :300-304) on the aurora seed case
    ctx2, seeds, _ = ref["_extract_seed_and_sigma_from_context"](dict(cases.AURORA_SEED_CONTEXT))
    sep, w, h = ref["_image_context_seperator"](Image.fromarray(au), ctx2, tok)
    bm = ref["_get_binary_mask"](sep, seeds, torch.float32, (64, 64))
    np.savez_compressed(os.path.join(GOLDEN, "binary_mask_aurora.npz"), mask=torch.stack(bm).numpy()[:, 0, 0],
                        seeds=np.array(list(seeds.items())))


def gen_attention(ref):
    inj = ref["inj_forward"]
    import warnings
    warnings.filterwarnings("ignore")
    for shape in cases.ATTN_SHAPES:
        case = cases.make_attention_case(shape)
        rows = cases.subsample_rows(case["N"])
        out = {"rows": rows}
        for mode in cases.ATTN_MODES:
            for wname, wf in (cases.WEIGHT_FUNCTIONS.items() if mode == "cond" else [("none", None)]):
                mod = case["attn_self"] if mode == "self" else case["attn_cross"]
                ctx = cases.attention_context(case, mode, wf)
                t0 = time.time()
                y = inj(mod, case["hidden"], ctx)
                key = mode if mode != "cond" else f"cond_{wname}"
                out[key] = y[0, rows].numpy()
                out[key + "_absmean"] = np.float64(y.abs().double().mean().item())
                print(f"attn_{shape} {key}: {time.time() - t0:.2f}s absmean {out[key + '_absmean']:.6f}")
        np.savez_compressed(os.path.join(GOLDEN, f"attn_{shape}.npz"), **out)
    # _ORIG fallback (:96-101): token count with no per-resolution key
    case = cases.make_attention_case("sd15_n256")
    torch.manual_seed(5)
    w_orig = (torch.rand(128, 128, 77) < 0.1).float() * 1.3
    ctx = {"CONTEXT_TENSOR": case["ctx"], "SIGMA": torch.tensor(3.0), "WEIGHT_FUNCTION": cases.weight_fn_runner,
           "CROSS_ATTENTION_WEIGHT_ORIG": w_orig}
    y = inj(case["attn_cross"], case["hidden"], ctx)
    np.savez_compressed(os.path.join(GOLDEN, "attn_orig_fallback.npz"), out=y[0, cases.subsample_rows(256)].numpy())


def _run_loop(ref, config, steps, img, ctx, prompt, wf, seed=0, scheduler="lms"):
    tools = cases.build_tools(config, scheduler=scheduler)
    for m in tools[1].modules():   # what pww_load_tools does at :193-195
        if m.__class__.__name__ == "CrossAttention":
            m.__class__.__call__ = ref["inj_forward"]
    captured = {}
    orig = ref["_pil_from_latents"]

    def grab(vae, latents):
        captured["latents"] = latents.detach().clone()
        return [Image.new("RGB", (8, 8))]
    ref["_pil_from_latents"] = grab   # `ref` IS the globals dict of the exec'd reference functions
    try:
        t0 = time.time()
        ref["paint_with_words"](color_context=dict(ctx), color_map_image=Image.fromarray(img), input_prompt=prompt,
                                num_inference_steps=steps, guidance_scale=7.5, seed=seed, device="cpu",
                                weight_function=wf, preloaded_utils=tools)
        dt = time.time() - t0
    finally:
        ref["_pil_from_latents"] = orig
        from sd_standin import CrossAttention
        if "__call__" in CrossAttention.__dict__:
            del CrossAttention.__call__
    return captured["latents"].numpy(), dt


def gen_loops(ref, full):
    ex = np.array(Image.open(os.path.join(ref_loader.REFERENCE_ROOT, "contents", "example_input.png")).convert("RGB"))
    au = np.array(Image.open(os.path.join(ref_loader.REFERENCE_ROOT, "contents", "aurora_1.png")).convert("RGB"))
    todo = [("tiny_example_lms10", "tiny", 10, ex, cases.RUNNER_CONTEXT, cases.RUNNER_PROMPT, "runner", 0),
            ("tiny_aurora_seed_std6", "tiny", 6, au, cases.AURORA_SEED_CONTEXT, cases.AURORA_PROMPT, "std", 3)]
    grid, gctx, gprompt = cases.grid_case(seeds=True)     # BASELINE config 5 inputs on the reduced SD2-style UNet
    todo.append(("tiny_sd2_grid768_std5", "tiny_sd2", 5, grid, gctx, gprompt, "std", 11))
    if full:
        todo.append(("sd15_example_lms10", "sd15", 10, ex, cases.RUNNER_CONTEXT, cases.RUNNER_PROMPT, "runner", 0))
    only = os.environ.get("PWW_GOLDEN_ONLY")
    for name, config, steps, img, ctx, prompt, wname, seed in todo:
        if only and only not in name:
            continue
        lat, dt = _run_loop(ref, config, steps, img, ctx, prompt, cases.WEIGHT_FUNCTIONS[wname], seed)
        np.savez_compressed(os.path.join(GOLDEN, f"loop_{name}.npz"), latents=lat, seconds=np.float64(dt),
                            steps=steps, threads=torch.get_num_threads())
        print(f"loop_{name}: {dt:.1f}s latents std {lat.std():.4f} absmean {np.abs(lat).mean():.4f}")


def gen_inpaint():
    """paint_with_words_inpaint (reference paint_with_words_inpaint.py:137-270) on the tiny 9-channel UNet:
    aurora_1.png color map, moon_mask.png, a seeded synthetic init image (BASELINE config 4 inputs, reduced UNet)."""
    ref = ref_loader.load_reference_inpaint()
    au = np.array(Image.open(os.path.join(ref_loader.REFERENCE_ROOT, "contents", "aurora_1.png")).convert("RGB"))
    mask = Image.open(os.path.join(ref_loader.REFERENCE_ROOT, "contents", "moon_mask.png"))
    mask.convert("L").save(os.path.join(GOLDEN, "moon_mask_L.png"), optimize=True)
    init = cases.synthetic_init_image()
    tools = cases.build_tools("tiny_inpaint")
    for m in tools[1].modules():
        if m.__class__.__name__ == "CrossAttention":
            m.__class__.__call__ = ref["inj_forward"] if "inj_forward" in ref else ref_loader.load_reference()["inj_forward"]
    captured = {}

    def grab(vae, latents):
        captured["latents"] = latents.detach().clone()
        return [Image.new("RGB", (8, 8))]
    ref["_pil_from_latents"] = grab
    try:
        ctx = {k: v for k, v in list(cases.AURORA_SEED_CONTEXT.items())[:4]}
        ctx = {k: ",".join(v.split(",")[:2]) for k, v in ctx.items()}
        ref["paint_with_words_inpaint"](color_context=dict(ctx), color_map_image=Image.fromarray(au), mask_image=Image.open(os.path.join(GOLDEN, "moon_mask_L.png")),
                                        init_image=Image.fromarray(init), input_prompt=cases.AURORA_PROMPT, num_inference_steps=8,
                                        guidance_scale=7.5, seed=81, device="cpu",
                                        weight_function=lambda w, sigma, qk: 0.15 * w * math.log(1 + sigma) * qk.max(),
                                        preloaded_utils=tools, strength=1.0)
    finally:
        from sd_standin import CrossAttention
        if "__call__" in CrossAttention.__dict__:
            del CrossAttention.__call__
    lat = captured["latents"].numpy()
    np.savez_compressed(os.path.join(GOLDEN, "loop_tiny_inpaint8.npz"), latents=lat)
    print("loop_tiny_inpaint8: latents std %.4f" % lat.std())


if __name__ == "__main__":
    assert ref_loader.available(), "reference not found at %s" % ref_loader.REFERENCE_ROOT
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    ref = ref_loader.load_reference()
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or ["kat", "masks", "attn", "loops", "inpaint"]
    if "kat" in which:
        gen_kat(ref)
    if "masks" in which:
        gen_masks(ref)
    if "attn" in which:
        gen_attention(ref)
    if "loops" in which:
        gen_loops(ref, "--full" in sys.argv)
    if "inpaint" in which:
        gen_inpaint()

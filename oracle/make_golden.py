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
import torch.nn.functional as F
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


def gen_img2img(ref):
    """Function-API img2img (reference paint_with_words.py:434-441 the shortened schedule, :459-468 vae.encode -> 0.18215 x -> add_noise
    with noise from the GLOBAL torch RNG): the reference's own function on the tiny UNet, 10 LMS steps, strength 0.5 and 0.8, the global
    generator seeded with 7 right before the call (the only random input of this path; `seed` is not used by it). VERDICT round 4 item 3:
    only the pipeline-class img2img was pinned before."""
    ex = np.array(Image.open(os.path.join(ref_loader.REFERENCE_ROOT, "contents", "example_input.png")).convert("RGB"))
    init = cases.synthetic_init_image(512, 3)
    out = {}
    for strength in (0.5, 0.8):
        tools = cases.build_tools("tiny")
        for m in tools[1].modules():   # what pww_load_tools does at :193-195
            if m.__class__.__name__ == "CrossAttention":
                m.__class__.__call__ = ref["inj_forward"]
        captured = {}
        orig = ref["_pil_from_latents"]

        def grab(vae, latents):
            captured["latents"] = latents.detach().clone()
            return [Image.new("RGB", (8, 8))]
        ref["_pil_from_latents"] = grab
        try:
            torch.manual_seed(7)
            ref["paint_with_words"](color_context=dict(cases.RUNNER_CONTEXT), color_map_image=Image.fromarray(ex), input_prompt=cases.RUNNER_PROMPT,
                                    num_inference_steps=10, guidance_scale=7.5, seed=0, device="cpu", weight_function=cases.weight_fn_runner,
                                    preloaded_utils=tools, init_image=Image.fromarray(init), strength=strength)
        finally:
            ref["_pil_from_latents"] = orig
            from sd_standin import CrossAttention
            if "__call__" in CrossAttention.__dict__:
                del CrossAttention.__call__
        lat = captured["latents"].numpy()
        out["latents_s%02d" % int(strength * 10)] = lat
        print(f"loop_tiny_img2img strength {strength}: latents std {lat.std():.4f} absmean {np.abs(lat).mean():.4f}")
    assert np.abs(out["latents_s05"] - out["latents_s08"]).max() > 1e-2
    np.savez_compressed(os.path.join(GOLDEN, "loop_tiny_img2img_lms10.npz"), global_seed=7, steps=10, **out)


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


# ---- round-2 fixtures ---------------------------------------------------------------------------

SEED_SIGMA_CONTEXT = {(0, 0, 0): "cat,1.0,42,4.0", (255, 255, 255): "dog,1.0,7", (13, 255, 0): "tree,1.5,-1,9.5",
                      (90, 206, 255): "sky,0.2", (74, 18, 1): "ground,0.2"}


def gen_seed_sigma(ref):
    """Regions that carry BOTH a seed and a sigma ("text,strength,seed,sigma"): the reference blurs the masks in place
    (:338-340) BEFORE thresholding them for region seeding (:300-304, :451), so the seeded area is the dilated one."""
    from sd_standin import HashTokenizer
    tok = HashTokenizer()
    ex = np.array(Image.open(os.path.join(GOLDEN, "example_input.png")).convert("RGB"))
    ctx2, seeds, sigmas = ref["_extract_seed_and_sigma_from_context"](dict(SEED_SIGMA_CONTEXT))
    sep, w, h = ref["_image_context_seperator"](Image.fromarray(ex), ctx2, tok)
    sep = ref["_blur_image_mask"](sep, sigmas)
    bm = ref["_get_binary_mask"](sep, seeds, torch.float32, (64, 64))
    np.savez_compressed(os.path.join(GOLDEN, "binary_mask_seed_sigma.npz"), mask=torch.stack(bm).numpy()[:, 0, 0],
                        seeds=np.array(list(seeds.items())), sigmas=np.array(list(sigmas.items())))
    lat, dt = _run_loop(ref, "tiny", 5, ex, SEED_SIGMA_CONTEXT, cases.RUNNER_PROMPT, cases.weight_fn_runner, seed=2)
    np.savez_compressed(os.path.join(GOLDEN, "loop_tiny_seed_sigma5.npz"), latents=lat, steps=5)
    print("seed+sigma: binary masks", [float(b.sum()) for b in bm], "loop latents std %.4f (%.1fs)" % (lat.std(), dt))


def gen_prep():
    """prepare_mask_and_masked_image (paint_with_words_inpaint.py:20-106, PIL branch) + the latent-size mask (:115)."""
    ref = ref_loader.load_reference_inpaint()
    init = cases.synthetic_init_image()
    mask = Image.open(os.path.join(GOLDEN, "moon_mask_L.png"))
    m, mi = ref["prepare_mask_and_masked_image"](Image.fromarray(init), mask)
    ml = F.interpolate(m, size=(64, 64))
    np.savez_compressed(os.path.join(GOLDEN, "inpaint_prep.npz"), mask=np.packbits(m.numpy()[0, 0] > 0.5), masked_rows=mi.numpy()[0, :, ::8],
                        masked_sum=mi.double().sum(dim=(2, 3)).numpy()[0], masked_abs_sum=mi.abs().double().sum().item(),
                        mask_lat=ml.numpy()[0, 0])
    print("inpaint_prep: mask ones %d, latent mask ones %d, masked sum %s" % (int(m.sum()), int(ml.sum()), mi.double().sum().item()))


def _install_ref(ref, unet):
    for m in unet.modules():
        if m.__class__.__name__ == "CrossAttention":
            m.__class__.__call__ = ref["inj_forward"]


def _uninstall_ref():
    from sd_standin import CrossAttention
    if "__call__" in CrossAttention.__dict__:
        del CrossAttention.__call__


def gen_forwards(ref):
    """Single UNet forwards of the FULL-SIZE stand-ins with the reference's inj_forward plugged in (fp32, CPU):
    the shapes BASELINE configs 4 and 5 run at -- SD1.5-inpainting (9 input channels, 4 regions, moon mask) and
    SD2.1 at 768x768 (N = 9216 tokens, head dim 64, 12 regions, 0.4 w log(1+sigma^2) qk.std())."""
    import warnings
    warnings.filterwarnings("ignore")
    only = os.environ.get("PWW_GOLDEN_ONLY", "")
    if "sd21" not in only:
        au = np.array(Image.open(os.path.join(GOLDEN, "aurora_1.png")).convert("RGB"))
        tools = cases.build_tools("sd15_inpaint")
        vae, unet, text, tok, sch = tools
        _install_ref(ref, unet)
        try:
            _, _, cond, uncond = ref["_encode_text_color_inputs"](text, tok, "cpu", Image.fromarray(au), dict(cases.INPAINT_CONTEXT),
                                                                  cases.AURORA_PROMPT, "")
            sch.set_timesteps(30)
            i = 4
            t, sigma = sch.timesteps[i], sch.sigmas[i]
            out = {"step_index": i, "sigma": float(sigma)}
            g = torch.Generator().manual_seed(81)
            x8 = torch.randn(8, 9, 64, 64, generator=g)
            x8[:, 4] = (x8[:, 4] > 0).float()          # channel 4 is the binary mask
            out["x"] = x8.numpy()
            for j in (0, 5):
                x = sch.scale_model_input(x8[j:j + 1, :4], t)
                x = torch.cat([x, x8[j:j + 1, 4:]], dim=1)
                cond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": cases.weight_fn_inpaint})
                t0 = time.time()
                out[f"eps_cond_{j}"] = unet(x, t, encoder_hidden_states=cond).sample.numpy()
                uncond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
                out[f"eps_uncond_{j}"] = unet(x, t, encoder_hidden_states=uncond).sample.numpy()
                print("fwd sd15_inpaint image %d: %.1fs |eps| %.4f" % (j, time.time() - t0, np.abs(out[f"eps_cond_{j}"]).mean()))
        finally:
            _uninstall_ref()
        np.savez_compressed(os.path.join(GOLDEN, "fwd_sd15_inpaint.npz"), **out)
    if "inpaint" not in only:
        grid, gctx, gprompt = cases.grid_case(seeds=True)
        tools = cases.build_tools("sd21")
        vae, unet, text, tok, sch = tools
        _install_ref(ref, unet)
        try:
            _, _, cond, uncond = ref["_encode_text_color_inputs"](text, tok, "cpu", Image.fromarray(grid), dict(gctx), gprompt, "")
            sch.set_timesteps(30)
            i = 4
            t, sigma = sch.timesteps[i], sch.sigmas[i]
            g = torch.Generator().manual_seed(11)
            x0 = torch.randn(1, 4, 96, 96, generator=g) * sch.init_noise_sigma
            x = sch.scale_model_input(x0, t)
            out = {"step_index": i, "sigma": float(sigma), "x": x0.numpy()}
            cond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": cases.weight_fn_std})
            t0 = time.time()
            out["eps_cond"] = unet(x, t, encoder_hidden_states=cond).sample.numpy()
            uncond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
            out["eps_uncond"] = unet(x, t, encoder_hidden_states=uncond).sample.numpy()
            print("fwd sd21 768x768: %.1fs |eps| %.4f" % (time.time() - t0, np.abs(out["eps_cond"]).mean()))
        finally:
            _uninstall_ref()
        np.savez_compressed(os.path.join(GOLDEN, "fwd_sd21_grid768.npz"), **out)


def gen_config3(ref):
    """BASELINE configs[2] pinned end to end: full-size SD1.5 stand-in, 8 stripes, image j of the
    batch = the stripes map rotated by j stripes (per-image weight maps), seed j.
      fwd_sd15_stripes8.npz   ONE forward (step 4 of 50 LMS) of the reference's inj_forward for images 0, 3, 6: cond + uncond eps
      loop_sd15_stripes8_lms50.npz   final latents of the reference's own 50-step paint_with_words loop for images 0 and 5"""
    import warnings
    warnings.filterwarnings("ignore")
    which = os.environ.get("PWW_GOLDEN_ONLY", "fwd loop")
    if "fwd" in which:
        tools = cases.build_tools("sd15")
        vae, unet, text, tok, sch = tools
        _install_ref(ref, unet)
        try:
            sch.set_timesteps(50)
            i = 4
            t, sigma = sch.timesteps[i], sch.sigmas[i]
            out = {"step_index": i, "sigma": float(sigma), "images": np.array([0, 3, 6])}
            for j in (0, 3, 6):
                img, ctx, prompt = cases.stripes_batch_case(j)
                _, _, cond, uncond = ref["_encode_text_color_inputs"](text, tok, "cpu", Image.fromarray(img), dict(ctx), prompt, "")
                x0 = torch.randn((1, 4, 64, 64), generator=torch.manual_seed(j)) * sch.init_noise_sigma
                x = sch.scale_model_input(x0, t)
                t0 = time.time()
                cond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": cases.weight_fn_runner})
                out[f"eps_cond_{j}"] = unet(x, t, encoder_hidden_states=cond).sample.numpy()
                uncond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
                out[f"eps_uncond_{j}"] = unet(x, t, encoder_hidden_states=uncond).sample.numpy()
                print("fwd sd15 stripes8 image %d: %.1fs |eps| %.4f gap %.4f" % (j, time.time() - t0, np.abs(out[f"eps_cond_{j}"]).mean(),
                                                                              np.abs(out[f"eps_cond_{j}"] - out[f"eps_uncond_{j}"]).mean()))
        finally:
            _uninstall_ref()
        np.savez_compressed(os.path.join(GOLDEN, "fwd_sd15_stripes8.npz"), **out)
    if "loop" in which:
        out = {"steps": 50, "images": np.array([0, 5])}
        for j in (0, 5):
            img, ctx, prompt = cases.stripes_batch_case(j)
            lat, dt = _run_loop(ref, "sd15", 50, img, ctx, prompt, cases.weight_fn_runner, seed=j)
            out[f"latents_{j}"] = lat
            print("loop sd15 stripes8 lms50 image %d: %.1fs latents std %.4f" % (j, dt, lat.std()))
        np.savez_compressed(os.path.join(GOLDEN, "loop_sd15_stripes8_lms50.npz"), **out)


def gen_plms():
    """ORACLE-generated (not reference-generated: the reference cannot run PNDM/PLMS -- no `.sigmas`, a repeated timestep,
    SURVEY.md section 8 a-note): final latent of BASELINE configs[1] -- full-size SD1.5 stand-in, 30 PLMS steps
    (31 UNet evaluations x {cond, uncond}), CFG 7.5, 5-region example -- through oracle/pww_oracle.py in fp32 on the CPU.
    The oracle itself is pinned to the reference by every other fixture; this file pins the HIP path on the benchmarked
    configuration to the oracle without 9 minutes of CPU time inside the GPU test."""
    from oracle import pww_oracle as O
    ex = np.array(Image.open(os.path.join(GOLDEN, "example_input.png")).convert("RGB"))
    vae, unet, text, tok, sch = cases.build_tools("sd15", scheduler="plms")
    O.install_oracle_attention(unet)
    try:
        t0 = time.time()
        lat = O.paint_with_words_latents(dict(cases.RUNNER_CONTEXT), ex, cases.RUNNER_PROMPT, unet, text, tok, sch,
                                         num_inference_steps=30, guidance_scale=7.5, seed=0, weight_function=cases.weight_fn_runner)
        dt = time.time() - t0
    finally:
        _uninstall_ref()
    np.savez_compressed(os.path.join(GOLDEN, "loop_sd15_example_plms30_oracle.npz"), latents=lat.numpy(), seconds=np.float64(dt),
                        threads=torch.get_num_threads())
    print("loop_sd15_example_plms30_oracle: %.1fs latents std %.4f" % (dt, lat.std()))


def gen_ref_timing(ref):
    """CPU baseline as SURVEY.md section 8(d) specifies it: the AST-loaded, unmodified reference `paint_with_words`
    (config 1: full-size SD1.5 stand-in fp32, 512x512, 10 LMS steps, CFG 7.5, 5-region example) on this build box's
    cores, with the time spent inside its `inj_forward` accumulated separately. bench.py reports this file's figures
    next to the oracle port it times on the GPU box's own host cores."""
    ex = np.array(Image.open(os.path.join(GOLDEN, "example_input.png")).convert("RGB"))
    inner = ref["inj_forward"]
    acc = {"s": 0.0, "calls": 0}

    def timed(self, hidden_states, context=None, mask=None):
        t0 = time.perf_counter()
        try:
            return inner(self, hidden_states, context, mask)
        finally:
            acc["s"] += time.perf_counter() - t0
            acc["calls"] += 1
    ref["inj_forward"] = timed
    try:
        lat, dt = _run_loop(ref, "sd15", 10, ex, cases.RUNNER_CONTEXT, cases.RUNNER_PROMPT, cases.weight_fn_runner, 0)
    finally:
        ref["inj_forward"] = inner
    golden = np.load(os.path.join(GOLDEN, "loop_sd15_example_lms10.npz"))["latents"]
    rec = {"what": "AST-loaded reference paint_with_words, fp32, device=cpu, 10 LMS steps (20 UNet forwards), 512x512, 5-region example, stand-in SD1.5 UNet (seed 1234)",
           "wall_s": round(dt, 2), "s_per_unet_forward": round(dt / 20, 3), "s_inside_inj_forward": round(acc["s"], 2),
           "inj_forward_calls": acc["calls"], "threads": torch.get_num_threads(), "cpu_count": os.cpu_count(),
           "images_per_s_30_steps": round(1.0 / (dt / 10 * 30), 6), "max_abs_diff_vs_golden": float(np.abs(lat - golden).max())}
    json.dump(rec, open(os.path.join(GOLDEN, "ref_cpu_timing.json"), "w"), indent=1)
    print("ref_cpu_timing", rec)


# ---- round-4 fixtures: BASELINE configs 4 and 5 pinned END TO END at full size, pipeline classes ------------------

def gen_config4():
    """BASELINE configs[3] end to end (missing 2): the reference's own `paint_with_words_inpaint`
    (paint_with_words_inpaint.py:137-270) on the FULL-SIZE SD1.5-inpainting stand-in (9 input channels), aurora_1.png +
    the 4-region context + moon_mask.png + the synthetic init image, 30 LMS steps, strength 1.0, fp32 CPU; final latents of
    images 0 and 5 of bench.py's batch of 8 (seeds 81 and 86)."""
    import warnings
    warnings.filterwarnings("ignore")
    ref = ref_loader.load_reference_inpaint()
    inj = ref_loader.load_reference()["inj_forward"]
    au = np.array(Image.open(os.path.join(GOLDEN, "aurora_1.png")).convert("RGB"))
    init = cases.synthetic_init_image()
    out = {"steps": 30, "seeds": np.array([81, 86])}
    captured = {}

    def grab(vae, latents):
        captured["latents"] = latents.detach().clone()
        return [Image.new("RGB", (8, 8))]
    ref["_pil_from_latents"] = grab
    for seed in (81, 86):
        tools = cases.build_tools("sd15_inpaint")
        _install_ref({"inj_forward": inj}, tools[1])
        try:
            t0 = time.time()
            ref["paint_with_words_inpaint"](color_context=dict(cases.INPAINT_CONTEXT), color_map_image=Image.fromarray(au),
                                            mask_image=Image.open(os.path.join(GOLDEN, "moon_mask_L.png")), init_image=Image.fromarray(init),
                                            input_prompt=cases.AURORA_PROMPT, num_inference_steps=30, guidance_scale=7.5, seed=seed,
                                            device="cpu", weight_function=cases.weight_fn_inpaint, preloaded_utils=tools, strength=1.0)
            dt = time.time() - t0
        finally:
            _uninstall_ref()
        out[f"latents_{seed}"] = captured["latents"].numpy()
        out[f"seconds_{seed}"] = np.float64(dt)
        print("loop sd15_inpaint lms30 seed %d: %.1fs latents std %.4f" % (seed, dt, out[f"latents_{seed}"].std()), flush=True)
        np.savez_compressed(os.path.join(GOLDEN, "loop_sd15_inpaint_lms30.npz"), **out)


def gen_config5(ref):
    """BASELINE configs[4] end to end: the reference's own `paint_with_words` (paint_with_words.py:393-510, region seeds
    :445-457) on the FULL-SIZE SD2.1 stand-in at 768x768 (N = 9216, head dim 64), 12-region grid with per-region seeds
    1000+k, 0.4 w log(1+sigma^2) qk.std() (README.md:152), 30 LMS steps, fp32 CPU.
      image 0: bench.py's request (seed 0);
      image 2: the same grid rolled by one cell to the right (cases.grid_batch_case(2): its own weight maps and its own
               region-seed placement -- the per-image path of paint_with_words_batch at this size), seed 2.
    (With the shared grid every latent pixel is overwritten by a region seed (:451-455), so the images of bench.py's batch
    are identical by construction; the rolled map is what makes a second image informative.)"""
    import warnings
    warnings.filterwarnings("ignore")
    path = os.path.join(GOLDEN, "loop_sd21_grid768_lms30.npz")
    out = dict(np.load(path)) if os.path.isfile(path) else {"steps": 30}
    for j in (0, 2):
        if f"latents_{j}" in out:
            continue
        img, ctx, prompt = cases.grid_batch_case(j)
        lat, dt = _run_loop(ref, "sd21", 30, img, ctx, prompt, cases.weight_fn_std, seed=j)
        out[f"latents_{j}"] = lat
        out[f"seconds_{j}"] = np.float64(dt)
        print("loop sd21 grid768 lms30 image %d: %.1fs latents std %.4f" % (j, dt, lat.std()), flush=True)
        np.savez_compressed(path, **out)


def gen_pipelines():
    """The reference's two pipeline CLASSES (paint_with_words.py:513-842, paint_with_words_inpaint.py:273-575), AST-loaded and exec'd
    unmodified over a stand-in of the diffusers base class (ref_loader._StableDiffusionPipelineStub), on the tiny UNets: final
    latents (grabbed at `decode_latents`) and the (step, timestep) pairs the callback saw, for the cases of cases.pipe_case."""
    import warnings
    from types import SimpleNamespace
    warnings.filterwarnings("ignore")
    ns_inp = ref_loader.load_reference_inpaint(classes=True)
    ns = ns_inp["_base_namespace"]
    out = {}
    for name in cases.PIPE_CASES:
        config, kind, kwargs, gseed = cases.pipe_case(name)
        vae, unet, text, tok, sch = cases.build_tools(config)
        vae.config = SimpleNamespace(block_out_channels=(1, 1, 1, 1), latent_channels=4)       # what the diffusers base class reads
        cls = ns["PaintWithWord_StableDiffusionPipeline"] if kind == "txt2img" else ns_inp["PaintWithWord_StableDiffusionInpaintPipeline"]
        try:
            pipe = cls(vae=vae, text_encoder=text, tokenizer=tok, unet=unet, scheduler=sch, safety_checker=None, feature_extractor=None)
            captured, calls = {}, []
            decode = pipe.decode_latents

            def grab(latents):
                captured["latents"] = latents.detach().clone()
                return decode(latents)
            pipe.decode_latents = grab
            if gseed is not None:
                torch.manual_seed(gseed)
            t0 = time.time()
            res = pipe(callback=lambda i, t, lat: calls.append((int(i), float(t))), output_type="np", **kwargs)
            dt = time.time() - t0
        finally:
            _uninstall_ref()
        out[f"{name}_latents"] = captured["latents"].numpy()
        out[f"{name}_callbacks"] = np.array(calls, dtype=np.float64).reshape(-1, 2)
        out[f"{name}_image_mean"] = np.float64(np.asarray(res.images).mean())
        print("pipeline %s: %.1fs latents %s std %.4f callbacks %s" % (name, dt, tuple(captured["latents"].shape), captured["latents"].std(),
                                                                      [c[0] for c in calls]), flush=True)
    np.savez_compressed(os.path.join(GOLDEN, "pipeline_classes.npz"), **out)


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
    if "img2img" in which:
        gen_img2img(ref)
    if "inpaint" in which:
        gen_inpaint()
    if "seedsigma" in which:
        gen_seed_sigma(ref)
    if "prep" in which:
        gen_prep()
    if "fwd" in which:
        gen_forwards(ref)
    if "plms" in which:
        gen_plms()
    if "config3" in which:
        gen_config3(ref)
    if "reftime" in which:
        gen_ref_timing(ref)
    if "pipelines" in which:
        gen_pipelines()
    if "config4" in which:
        gen_config4()
    if "config5" in which:
        gen_config5(ref)

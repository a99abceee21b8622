"""Round-2 GPU parity: the statistic-in-the-same-launch cross-attention kernel, the mask-extras kernels (blur, _ORIG
resize, inpaint prep), hot-logit behaviour of the folded-reference kernel, the batched public entry points, hipGraph
re-use with the reference's fresh-lambda call pattern, and the full-size configurations BASELINE names (PLMS-30 bf16
SD1.5, SD1.5-inpainting batch 8, SD2.1 at 768x768)."""
import math
import os

import numpy as np
import pytest
import torch
from PIL import Image

import pww_cases as cases
from gpu_util import TOL, install_unfused, uninstall_all, rel_l2
from oracle import pww_oracle as O

pytestmark = pytest.mark.gpu
G = cases.GOLDEN


def _mode(mode):
    import importlib
    mod = importlib.import_module("paint_with_words.paint_with_words")

    class _Ctx:
        def __enter__(self):
            self.old = mod.DEFAULT_MODE
            mod.DEFAULT_MODE = mode

        def __exit__(self, *a):
            mod.DEFAULT_MODE = self.old
    return _Ctx()


# ---- a3: the _ORIG fallback through the HIP plug ---------------------------------------------------------------------

def test_orig_fallback_through_hip_plug(gpu_device):
    """No CROSS_ATTENTION_WEIGHT_<N> key -> KeyError -> CROSS_ATTENTION_WEIGHT_ORIG resized to N tokens
    (paint_with_words.py:96-101), driven through pww_hip.inj_forward against the reference's own output."""
    import pww_hip
    from pww_hip import ops
    g = np.load(os.path.join(G, "attn_orig_fallback.npz"))
    case = cases.make_attention_case("sd15_n256")
    torch.manual_seed(5)
    w_orig = (torch.rand(128, 128, 77) < 0.1).float() * 1.3
    # the resize kernel alone: bit-exact with the oracle's restatement
    got = ops.resize_tokens(w_orig.to(gpu_device), 256).cpu().numpy()
    assert np.array_equal(got, O.orig_weight_fallback(w_orig.numpy(), 256))
    got = ops.resize_tokens(w_orig[:100, :72].contiguous().to(gpu_device), 99).cpu().numpy()       # ragged sizes
    assert np.array_equal(got, O.orig_weight_fallback(w_orig[:100, :72].numpy(), 99))
    for dtype in (torch.float16, torch.bfloat16):
        mod = case["attn_cross"].to(gpu_device, dtype)
        ctx = {"CONTEXT_TENSOR": case["ctx"].to(gpu_device, dtype), "SIGMA": torch.tensor(3.0), "WEIGHT_FUNCTION": cases.weight_fn_runner,
               "CROSS_ATTENTION_WEIGHT_ORIG": w_orig.to(gpu_device)}
        y = pww_hip.inj_forward(mod, case["hidden"].to(gpu_device, dtype), ctx).float().cpu().numpy()
        assert 256 in ctx["_PWW_ORIG_CACHE"]
        ref = g["out"]
        err = np.abs(y[0, cases.subsample_rows(256)] - ref).max() / np.abs(ref).max()
        print(f"_ORIG fallback {dtype}: max err / max|O| = {err:.3e}")
        assert err <= 2.5 * TOL[dtype]      # projections in half precision on top of the attention bar


# ---- f-4 kernels -----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("sigma", [1.5, 4.0, 9.5, 25.0])
def test_gauss_blur_kernel(gpu_device, sigma):
    from pww_hip import ops
    rgb = cases.load_example_rgb()
    mask = ((rgb == np.array([13, 255, 0], np.uint8)).all(-1).astype(np.float32) * 1.5)
    ref = O.gaussian_blur(mask, sigma)
    got = ops.gauss_blur(torch.from_numpy(mask).to(gpu_device), sigma).cpu().numpy()
    err = np.abs(got - ref).max()
    print(f"blur sigma {sigma}: max abs err {err:.2e}")
    assert err <= 1e-6
    # non-square, reflect padding active on every side
    small = mask[100:160, 200:290].copy()
    assert np.abs(ops.gauss_blur(torch.from_numpy(small).to(gpu_device), sigma).cpu().numpy() - O.gaussian_blur(small, sigma)).max() <= 1e-6


def test_blurred_masks_match_reference(gpu_device):
    """masks_blur.npz (two blurred regions, sigma 4.0 / 9.5) through the product path at the 1e-6 bar of BASELINE.md section 4
    (round 1 needed 2e-5 with torch's conv2d on the GPU)."""
    from pww_hip.conditioning import _encode_text_color_inputs
    from sd_standin import HashTokenizer, TinyTextEncoder
    g = np.load(os.path.join(G, "masks_blur.npz"))
    ctx = {(0, 0, 0): "cat,1.0,-1,4.0", (255, 255, 255): "dog,1.0", (13, 255, 0): "tree,1.5,-1,9.5"}
    text = TinyTextEncoder(64, seed=1235).to(gpu_device)
    _, _, cond, _ = _encode_text_color_inputs(text, HashTokenizer(), gpu_device, cases.load_example_rgb(), ctx, cases.RUNNER_PROMPT, "")
    for r in (8, 16, 32, 64):
        ref = np.zeros(tuple(g[f"shape_{r}"]), np.float32)
        ref[:, g[f"cols_{r}"]] = g[f"vals_{r}"]
        err = np.abs(cond[f"CROSS_ATTENTION_WEIGHT_{ref.shape[0]}"].cpu().numpy() - ref).max()
        print(f"blurred maps ratio {r}: max abs err {err:.2e}")
        assert err <= 1e-6


def test_inpaint_prep_kernel(gpu_device):
    from pww_hip import ops
    from paint_with_words.paint_with_words_inpaint import prepare_mask_and_masked_image
    g = np.load(os.path.join(G, "inpaint_prep.npz"))
    init, mask_l = cases.synthetic_init_image(), np.array(cases.load_moon_mask().convert("L"))
    m, masked, ml = ops.inpaint_prep(torch.from_numpy(init).to(gpu_device), torch.from_numpy(mask_l).to(gpu_device), 64, 64)
    assert np.array_equal(np.packbits(m.cpu().numpy()[0, 0] > 0.5), g["mask"])
    assert np.array_equal(masked.cpu().numpy()[0, :, ::8], g["masked_rows"])        # bit-exact with the reference's fp32 arithmetic
    assert np.array_equal(ml.cpu().numpy()[0, 0], g["mask_lat"])
    om, omi = O.prepare_mask_and_masked_image(init, mask_l)
    assert torch.equal(m.cpu(), om) and torch.equal(masked.cpu(), omi)
    # the public function with PIL inputs (the reference's callers) goes through the same kernel
    m2, mi2 = prepare_mask_and_masked_image(Image.fromarray(init), cases.load_moon_mask(), device=gpu_device)
    assert torch.equal(m2, m) and torch.equal(mi2, masked)
    # tensor inputs: validated and thresholded like the reference's first branch
    mt, it = prepare_mask_and_masked_image(omi.clamp(-1, 1)[0], (om[0, 0] * 0.7))
    assert mt.shape == (1, 1, 512, 512) and it.shape == (1, 3, 512, 512) and set(mt.unique().tolist()) <= {0.0, 1.0}
    with pytest.raises(ValueError):
        prepare_mask_and_masked_image(omi[0] * 3.0, om[0, 0])
    with pytest.raises(TypeError):
        prepare_mask_and_masked_image(omi[0], mask_l)


# ---- region seeds + sigma ---------------------------------------------------------------------------

SEED_SIGMA_CONTEXT = {(0, 0, 0): "cat,1.0,42,4.0", (255, 255, 255): "dog,1.0,7", (13, 255, 0): "tree,1.5,-1,9.5",
                      (90, 206, 255): "sky,0.2", (74, 18, 1): "ground,0.2"}


def test_region_seed_with_sigma(gpu_device):
    import paint_with_words as pw
    from pww_hip.conditioning import _encode_text_color_inputs, _get_binary_mask
    g = np.load(os.path.join(G, "binary_mask_seed_sigma.npz"))
    tools = cases.build_tools("tiny", dtype=torch.float16, device=gpu_device)
    seeds, info, _, _ = _encode_text_color_inputs(tools[2], tools[3], gpu_device, cases.load_example_rgb(), dict(SEED_SIGMA_CONTEXT),
                                                  cases.RUNNER_PROMPT, "")
    masks = torch.stack(_get_binary_mask(info, seeds, torch.float32, (64, 64))).numpy()[:, 0, 0]
    assert seeds == {0: 42, 1: 7} and np.abs(masks - g["mask"]).max() <= 1e-6
    gl = np.load(os.path.join(G, "loop_tiny_seed_sigma5.npz"))
    try:
        with _mode("eager"):
            lat = pw.paint_with_words(color_context=dict(SEED_SIGMA_CONTEXT), color_map_image=Image.fromarray(cases.load_example_rgb()),
                                      input_prompt=cases.RUNNER_PROMPT, num_inference_steps=5, guidance_scale=7.5, seed=2, device=str(gpu_device),
                                      weight_function=cases.weight_fn_runner, preloaded_utils=tools, return_latents=True)
    finally:
        uninstall_all()
    d = rel_l2(lat, gl["latents"])
    print(f"seed+sigma tiny loop fp16: rel-L2 vs reference {d:.3e}")
    assert d <= 2e-2


# ---- the fused statistic + attention launch ---------------------------------------------------------------------------

@pytest.mark.parametrize("shape,dtype", [("sd15_n4096", torch.bfloat16), ("sd15_n1024", torch.float16), ("sd15_n256", torch.bfloat16),
                                         ("sd15_n64", torch.float16), ("sd21_n576", torch.bfloat16)])
def test_fused_cross_equals_two_launches(gpu_device, shape, dtype):
    """pww_cross_attn_fwd_fused == pww_qk_reduce + pww_cross_attn_fwd_stat bit for bit (output and statistics), for a
    CFG-folded batch (second row gated out), eagerly and replayed from a hipGraph (the state words must come back to
    zero by themselves)."""
    from pww_hip import ops
    case = cases.make_attention_case(shape)
    N, C, H = case["N"], case["C"], case["H"]
    g = torch.Generator().manual_seed(3)
    B = 4
    q = (torch.randn(B, N, C, generator=g) * 1.2).to(gpu_device, dtype)
    k = torch.randn(B, 77, C, generator=g).to(gpu_device, dtype)
    v = torch.randn(B, 77, C, generator=g).to(gpu_device, dtype)
    w = case["w"].to(gpu_device)
    gate = torch.tensor([1.0, 1.0, 0.0, 0.0], device=gpu_device)
    scale = (C // H) ** -0.5
    for kind in (ops.STAT_MAX, ops.STAT_STD):
        stats = ops.qk_stats(q, k, H)
        two = ops.attention(q, k, v, H, scale, bias=w, bias_coeff=gate, stat=(stats, kind, 0.3))
        scratch = ops.FusedScratch()
        st_out = torch.zeros(B, 4, dtype=torch.float64, device=gpu_device)
        one = ops.attention(q, k, v, H, scale, bias=w, bias_coeff=gate, stat=(None, kind, 0.3), scratch=scratch, stats_out=st_out)
        assert torch.equal(one, two) and torch.equal(st_out[:2], stats[:2]) and not scratch.error()
        assert int(scratch.state.abs().sum().item()) == 0
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            ops.attention(q, k, v, H, scale, bias=w, bias_coeff=gate, stat=(None, kind, 0.3), scratch=scratch)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(graph):
            outs = [ops.attention(q, k, v, H, scale, bias=w, bias_coeff=gate, stat=(None, kind, 0.3), scratch=scratch) for _ in range(3)]
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(o, two) for o in outs) and not scratch.error()
    # per-image maps (paint_with_words_batch): bias [B, 1, N, 77]
    wb = torch.stack([w * (i + 1) for i in range(B)]).unsqueeze(1).contiguous()
    stats = ops.qk_stats(q, k, H)
    two = ops.attention(q, k, v, H, scale, bias=wb, bias_coeff=gate, stat=(stats, ops.STAT_MAX, 0.3))
    one = ops.attention(q, k, v, H, scale, bias=wb, bias_coeff=gate, stat=(None, ops.STAT_MAX, 0.3), scratch=ops.FusedScratch())
    assert torch.equal(one, two)
    ref, _ = O.attention_core(*(O.split_heads(t[:1].float().cpu(), H) for t in (q, k, v)),
                              (0.3 * float(stats[0, 0]) * wb[0, 0].cpu())[None], scale)
    err = (O.merge_heads(ref, H) - one[:1].float().cpu()).abs().max().item() / ref.abs().max().item()
    assert err <= TOL[dtype], err


# ---- hot logits (VERDICT round 1, weak 2 / ADVICE low) ------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("n,d,heads", [(4096, 40, 8), (1000, 40, 4), (1024, 80, 8), (2304, 64, 5)])
def test_hot_logits(gpu_device, dtype, n, d, heads):
    """Trained SD layers produce scaled logits with std ~4 and row maxima >= 30 (natural units); the random-init UNet of
    the other tests stays below 1. The folded-reference d = 40 kernel rounds Q * scale * log2(e) to half precision and
    re-references lazily: pin it (and the plain kernels) at the per-call bar on such inputs, against fp64 on the same
    rounded inputs."""
    from pww_hip import ops
    g = torch.Generator().manual_seed(11)
    B = 2
    q = (torch.randn(B, n, heads * d, generator=g) * 2.0).to(dtype)
    k = (torch.randn(B, n, heads * d, generator=g) * 2.0).to(dtype)
    k[:, 777 % n] *= 2.5                                   # one outlier key per image: spikes far above the running reference
    v = torch.randn(B, n, heads * d, generator=g).to(dtype)
    out = ops.attention(q.to(gpu_device), k.to(gpu_device), v.to(gpu_device), heads, d ** -0.5).float().cpu()
    rows = torch.arange(0, n, max(1, n // 48))
    qh, kh, vh = (O.split_heads(t.double(), heads) for t in (q, k, v))
    logits = torch.matmul(qh[:, rows], kh.transpose(-1, -2)) * d ** -0.5
    ref = O.merge_heads(torch.matmul(logits.softmax(-1), vh), heads)
    err = (out[:, rows].double() - ref).abs().max().item() / ref.abs().max().item()
    print(f"hot logits {dtype} N={n} d={d}: scaled-logit std {logits.std():.2f} max {logits.max():.1f}; max err / max|O| = {err:.3e} (bar {TOL[dtype]:.1e})")
    assert logits.std() > 3.5 and logits.max() >= 30
    assert torch.isfinite(out).all() and err <= TOL[dtype]


# ---- f-2: the batched public entry point --------------------------------------------------------------------------------

def _three_requests():
    ex, au = cases.load_example_rgb(), cases.load_aurora_rgb()
    stripes, sctx, sprompt = cases.stripes_case(8, 512)
    return ([dict(cases.RUNNER_CONTEXT), dict(cases.AURORA_SEED_CONTEXT), dict(sctx)], [Image.fromarray(ex), Image.fromarray(au), Image.fromarray(stripes)],
            [cases.RUNNER_PROMPT, cases.AURORA_PROMPT, sprompt], [0, 3, 9], [ex, au, stripes])


@pytest.mark.parametrize("mode", ["folded", "graph"])
def test_batch_entry_point_equals_single_calls(gpu_device, mode):
    """paint_with_words_batch with per-image color maps, color_contexts (one with a region seed), prompts and seeds:
    image i equals paint_with_words on request i (gradio_pww.py:24-45 generates samples one call at a time), and the
    reference's own result for that request."""
    import paint_with_words as pw
    tools = cases.build_tools("tiny", dtype=torch.float16, device=gpu_device)
    ctxs, maps, prompts, seeds, rgbs = _three_requests()
    kw = dict(num_inference_steps=6, guidance_scale=7.5, device=str(gpu_device), weight_function=cases.weight_fn_std, preloaded_utils=tools,
              return_latents=True)
    try:
        with _mode(mode):
            batch = pw.paint_with_words_batch([dict(c) for c in ctxs], maps, prompts, seeds, **kw)
        with _mode("eager"):
            single = torch.cat([pw.paint_with_words(color_context=dict(c), color_map_image=m, input_prompt=p, seed=s, **kw)
                                for c, m, p, s in zip(ctxs, maps, prompts, seeds)])
    finally:
        uninstall_all()
    assert batch.shape == (3, 4, 64, 64)
    for i in range(3):
        d = rel_l2(batch[i], single[i])
        print(f"batch[{i}] vs single call ({mode}): rel-L2 {d:.3e}")
        assert d <= 1.5e-2
    # request 1 is the reference's own region-seed + qk.std() golden (loop_tiny_aurora_seed_std6: same inputs, seed 3, 6 steps)
    g = np.load(os.path.join(G, "loop_tiny_aurora_seed_std6.npz"))
    assert rel_l2(batch[1:2], g["latents"]) <= 2e-2
    # shared request, several seeds (the bench's shape): maps stay [N, 77]
    try:
        with _mode(mode):
            shared = pw.paint_with_words_batch(dict(cases.RUNNER_CONTEXT), maps[0], prompts[0], [4, 5], **kw)
            imgs = pw.paint_with_words_batch(dict(cases.RUNNER_CONTEXT), maps[0], prompts[0], [4, 5], **dict(kw, return_latents=False))
        with _mode("eager"):
            one = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), color_map_image=maps[0], input_prompt=prompts[0], seed=5, **kw)
    finally:
        uninstall_all()
    assert rel_l2(shared[1:2], one) <= 1.5e-2 and len(imgs) == 2 and isinstance(imgs[0], Image.Image)
    with pytest.raises(ValueError):
        pw.paint_with_words_batch(dict(cases.RUNNER_CONTEXT), [maps[0], maps[0].resize((256, 256))], prompts[0], [1, 2], **kw)


def test_inpaint_batch_equals_single_calls(gpu_device):
    import paint_with_words as pw
    tools = cases.build_tools("tiny_inpaint", dtype=torch.float16, device=gpu_device)
    au, mask, init = Image.fromarray(cases.load_aurora_rgb()), cases.load_moon_mask(), Image.fromarray(cases.synthetic_init_image())
    kw = dict(num_inference_steps=8, guidance_scale=7.5, device=str(gpu_device), weight_function=cases.weight_fn_inpaint, preloaded_utils=tools,
              return_latents=True, strength=1.0)
    try:
        with _mode("graph"):
            batch = pw.paint_with_words_inpaint_batch(dict(cases.INPAINT_CONTEXT), au, mask, init, cases.AURORA_PROMPT, [81, 82, 83], **kw)
        with _mode("eager"):
            single = pw.paint_with_words_inpaint(color_context=dict(cases.INPAINT_CONTEXT), color_map_image=au, mask_image=mask, init_image=init,
                                                 input_prompt=cases.AURORA_PROMPT, seed=82, **kw)
    finally:
        uninstall_all()
    g = np.load(os.path.join(G, "loop_tiny_inpaint8.npz"))      # the reference's own run of seed 81
    print(f"inpaint batch: [0] vs reference {rel_l2(batch[0:1], g['latents']):.3e}, [1] vs single call {rel_l2(batch[1:2], single):.3e}")
    assert rel_l2(batch[0:1], g["latents"]) <= 2e-2 and rel_l2(batch[1:2], single) <= 1.5e-2


# ---- hipGraph re-use under the reference's call pattern -------------------------------------------------------------------

def test_graphs_survive_fresh_lambdas_and_new_requests(gpu_device):
    """runner.py:104 and gradio_pww.py:43 build a NEW lambda for every call: the captured step graphs must be re-used
    (same code, same constants), re-captured when a constant changes, and a replay with a different prompt and color map
    (static context refreshed in place: K|V cache, weight maps) must equal the folded eager run of that request."""
    import paint_with_words as pw
    tools = cases.build_tools("tiny", dtype=torch.float16, device=gpu_device)
    ex, au = Image.fromarray(cases.load_example_rgb()), Image.fromarray(cases.load_aurora_rgb())
    kw = dict(num_inference_steps=5, guidance_scale=7.5, device=str(gpu_device), preloaded_utils=tools, return_latents=True)
    fresh = lambda c=0.4: (lambda w, sigma, qk: c * w * math.log(1 + sigma) * qk.max())      # noqa: E731  a new function object per call
    try:
        with _mode("graph"):
            pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), color_map_image=ex, input_prompt=cases.RUNNER_PROMPT, seed=1,
                                weight_function=fresh(), **kw)
            sampler = tools[1]._pww_samplers[(id(tools[4]), "graph")]
            graphs = dict(sampler._graphed.graphs)
            assert len(graphs) == 5
            a = pw.paint_with_words(color_context={k: ",".join(v.split(",")[:2]) for k, v in cases.AURORA_SEED_CONTEXT.items()}, color_map_image=au,
                                    input_prompt=cases.AURORA_PROMPT, seed=2, weight_function=fresh(), **kw)
            assert all(sampler._graphed.graphs[i] is graphs[i] for i in range(5)), "graphs were re-captured for a fresh lambda"
            pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), color_map_image=ex, input_prompt=cases.RUNNER_PROMPT, seed=1,
                                weight_function=fresh(0.5), **kw)
            assert sampler._graphed.graphs[0] is not graphs[0], "a changed constant must re-capture (it is baked into kernel arguments)"
        with _mode("folded"):
            b = pw.paint_with_words(color_context={k: ",".join(v.split(",")[:2]) for k, v in cases.AURORA_SEED_CONTEXT.items()}, color_map_image=au,
                                    input_prompt=cases.AURORA_PROMPT, seed=2, weight_function=fresh(), **kw)
    finally:
        uninstall_all()
    d = rel_l2(a, b)
    print(f"graph replay on a new prompt + color map vs folded eager: rel-L2 {d:.3e}")
    assert d <= 2e-2


# ---- the pipeline classes with the reference's argument names --------------------------------------------------------------

def test_pipeline_call_arguments(gpu_device):
    """`image=` is the img2img input and `eta` its strength (reference :646, :735); positional order is (prompt,
    color_map_image, color_context); height / width size the latent; callback / output_type / return_dict work."""
    import paint_with_words as pw
    vae, unet, text, tok, sch = cases.build_tools("tiny", dtype=torch.float16, device=gpu_device)
    ex = Image.fromarray(cases.load_example_rgb())
    init = Image.fromarray(cases.synthetic_init_image(512, 5))
    seen = []
    try:
        pipe = pw.PaintWithWord_StableDiffusionPipeline(vae, text, tok, unet, sch)
        txt = pipe(cases.RUNNER_PROMPT, ex, dict(cases.RUNNER_CONTEXT), cases.weight_fn_runner, num_inference_steps=4, output_type="np",
                   callback=lambda i, t, lat: seen.append((i, int(t), tuple(lat.shape))), callback_steps=2)
        assert isinstance(txt.images, np.ndarray) and txt.images.shape == (1, 512, 512, 3) and txt.nsfw_content_detected is False
        assert [s[0] for s in seen] == [0, 2] and seen[0][2] == (1, 4, 64, 64)
        i2i = pipe(cases.RUNNER_PROMPT, ex, dict(cases.RUNNER_CONTEXT), cases.weight_fn_runner, num_inference_steps=4, image=init, eta=0.5,
                   return_dict=False)
        assert isinstance(i2i, tuple) and isinstance(i2i[0][0], Image.Image)
        assert np.abs(np.asarray(i2i[0][0], np.int32) - (txt.images[0] * 255).round().astype(np.int32)).mean() > 1.0   # img2img is a different image
        small = pipe(cases.RUNNER_PROMPT, ex, dict(cases.RUNNER_CONTEXT), cases.weight_fn_runner, height=256, width=384, num_inference_steps=2)
        assert small.images[0].size == (384, 256)        # latent size from height / width; the maps fall back to _ORIG (:96-101)
        with pytest.raises(ValueError):
            pipe(cases.RUNNER_PROMPT, ex, dict(cases.RUNNER_CONTEXT), height=250, width=256)
    finally:
        uninstall_all()


# ---- full-size configurations ------------------------------------------------------------------------------------------

def test_sd15_plms30_bf16_graph_final_latent(gpu_device):
    """THE benchmarked configuration (BASELINE configs[1]): full-size SD1.5 stand-in, bf16, 30 PLMS steps (31 UNet
    evaluations), CFG 7.5, hipGraph mode, through the public API -- final latent vs the oracle's fp32 PLMS loop
    (tests/golden/loop_sd15_example_plms30_oracle.npz, generated by oracle/make_golden.py plms). Bar of BASELINE.md
    section 4 for bf16: rel-L2 <= 5e-2 and <= 1.5x the drift of the unfused half-precision torch path on this GPU."""
    import paint_with_words as pw
    g = np.load(os.path.join(G, "loop_sd15_example_plms30_oracle.npz"))
    tools = cases.build_tools("sd15", dtype=torch.bfloat16, device=gpu_device, scheduler="plms")
    kw = dict(color_map_image=Image.fromarray(cases.load_example_rgb()), input_prompt=cases.RUNNER_PROMPT, num_inference_steps=30,
              guidance_scale=7.5, seed=0, device=str(gpu_device), weight_function=cases.weight_fn_runner, preloaded_utils=tools, return_latents=True)
    try:
        with _mode("graph"):
            lat = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), **kw)
        from pww_hip import sampler as S
        orig_install = S.install
        S.install = install_unfused
        tools[1].__dict__.pop("_pww_samplers", None)
        try:
            with _mode("eager"):
                base = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), **kw)
        finally:
            S.install = orig_install
    finally:
        uninstall_all()
    d, d0 = rel_l2(lat, g["latents"]), rel_l2(base, g["latents"])
    print(f"SD1.5 PLMS-30 bf16 graph: rel-L2 vs oracle {d:.3e}; unfused torch ops on this GPU {d0:.3e}")
    assert d <= 5e-2 and d <= 1.5 * d0 + 2e-3


def test_sd15_inpaint_batch8_forward(gpu_device):
    """BASELINE config 4 shapes: full-size SD1.5-inpainting stand-in, 8 images folded with their unconditional rows
    (16 rows, fused statistic per image), one UNet forward, vs the reference's forward of images 0 and 5."""
    from pww_hip.conditioning import _encode_text_color_inputs
    from pww_hip.sampler import _fold_context
    import pww_hip
    g = np.load(os.path.join(G, "fwd_sd15_inpaint.npz"))
    for dtype, bar in ((torch.float16, 1e-2), (torch.bfloat16, 5e-2)):
        vae, unet, text, tok, sch = cases.build_tools("sd15_inpaint", dtype=dtype, device=gpu_device)
        pww_hip.install(unet)
        try:
            _, _, cond, uncond = _encode_text_color_inputs(text, tok, gpu_device, cases.load_aurora_rgb(), dict(cases.INPAINT_CONTEXT),
                                                           cases.AURORA_PROMPT, "", dtype=dtype)
            sch.set_timesteps(30)
            i = int(g["step_index"])
            t, sigma = sch.timesteps[i], sch.sigmas[i]
            x8 = torch.from_numpy(g["x"]).to(gpu_device)
            x = torch.cat([sch.scale_model_input(x8[:, :4], t), x8[:, 4:]], dim=1)
            folded = _fold_context(cond, uncond, 8, gpu_device)
            folded.update({"SIGMA": sigma, "WEIGHT_FUNCTION": cases.weight_fn_inpaint})
            with torch.no_grad():
                out = unet(torch.cat([x, x]).to(dtype), t, encoder_hidden_states=folded).sample.float().cpu()
        finally:
            uninstall_all()
        for j in (0, 5):
            dc, du = rel_l2(out[j], g[f"eps_cond_{j}"][0]), rel_l2(out[8 + j], g[f"eps_uncond_{j}"][0])
            gap = rel_l2(g[f"eps_cond_{j}"], g[f"eps_uncond_{j}"])
            print(f"sd15_inpaint batch-8 forward {dtype} image {j}: cond {dc:.3e} uncond {du:.3e} (cond/uncond gap {gap:.3e})")
            assert dc <= bar and du <= bar


def test_sd21_768_forward(gpu_device):
    """BASELINE config 5 shapes: full-size SD2.1 stand-in at 768x768 (N = 9216 tokens, head dim 64, 5/10/20/20 heads,
    linear projections), 12-region grid, 0.4 w log(1+sigma^2) qk.std(), one folded forward vs the reference's."""
    from pww_hip.conditioning import _encode_text_color_inputs
    from pww_hip.sampler import _fold_context
    import pww_hip
    g = np.load(os.path.join(G, "fwd_sd21_grid768.npz"))
    grid, gctx, gprompt = cases.grid_case(seeds=True)
    for dtype, bar in ((torch.float16, 1e-2), (torch.bfloat16, 5e-2)):
        vae, unet, text, tok, sch = cases.build_tools("sd21", dtype=dtype, device=gpu_device)
        pww_hip.install(unet)
        try:
            _, _, cond, uncond = _encode_text_color_inputs(text, tok, gpu_device, grid, dict(gctx), gprompt, "", dtype=dtype)
            sch.set_timesteps(30)
            i = int(g["step_index"])
            t, sigma = sch.timesteps[i], sch.sigmas[i]
            x = sch.scale_model_input(torch.from_numpy(g["x"]).to(gpu_device), t)
            folded = _fold_context(cond, uncond, 1, gpu_device)
            folded.update({"SIGMA": sigma, "WEIGHT_FUNCTION": cases.weight_fn_std})
            with torch.no_grad():
                out = unet(torch.cat([x, x]).to(dtype), t, encoder_hidden_states=folded).sample.float().cpu()
        finally:
            uninstall_all()
        dc, du = rel_l2(out[0:1], g["eps_cond"]), rel_l2(out[1:2], g["eps_uncond"])
        print(f"sd21 768x768 forward {dtype}: cond {dc:.3e} uncond {du:.3e} (cond/uncond gap {rel_l2(g['eps_cond'], g['eps_uncond']):.3e})")
        assert dc <= bar and du <= bar

"""Round-4 GPU tests of the hand-off-free cross-attention path and of the hygiene items (ADVICE):
  * pww_qproj_stat: Q = X W^T vs torch's GEMM, the folded partials vs pww_qk_reduce on the Q it wrote, gates, ragged shapes;
  * pww_cross_attn_fwd_parts vs the two-step path, and the product (inj_forward) with PWW_QPROJ_STAT on vs off;
  * hipGraph mode: a graph captured under a bias-free weight function is never replayed for one that wants the bias (ADVICE high);
  * CROSS_ATTENTION_WEIGHT_ORIG is built lazily, also through two requests of a captured graph;
  * one mask-build launch for the four maps == four launches, bit for bit;
  * N > 1 plumbing on hardware: bench.py under torch.distributed.run with the nccl backend (one rank), and image i of a 2-rank
    one-device run == image i of the 1-rank run (results do not depend on the GPU count).
"""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

import pww_cases as cases
from gpu_util import TOL, uninstall_all, rel_l2

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


QPROJ_SHAPES = [
    # name, B, N, Cin, heads, D, M, shared prompt
    ("sd15_n4096", 2, 4096, 320, 8, 40, 77, False),
    ("sd15_n1024", 2, 1024, 640, 8, 80, 77, False),
    ("sd15_n256", 2, 256, 1280, 8, 160, 77, False),
    ("sd15_n64", 2, 64, 1280, 8, 160, 77, True),
    ("sd15_n4096_b16", 16, 4096, 320, 8, 40, 77, False),
    ("sd21_n2304_b8", 8, 2304, 640, 10, 64, 77, False),
    ("ragged_n1000_m50", 3, 1000, 320, 8, 40, 50, True),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("name,B,N,Cin,H,D,M,shared", QPROJ_SHAPES)
def test_qproj_stat_matches_gemm_and_qk_reduce(gpu_device, dtype, name, B, N, Cin, H, D, M, shared):
    """Q vs fp64 x @ w.T of the same rounded inputs: one rounding to the storage type (2^-9 bf16 / 2^-12 fp16 relative) + fp32
    accumulation. Folded partials vs pww_qk_reduce on the Q the kernel wrote: max / min to 1e-6 of the largest score, mean and
    sum of squares to 1e-6 ("within 1 ulp-of-fp32")."""
    from pww_hip import ops
    C = H * D
    g = torch.Generator().manual_seed(hash(name) % 1000)
    x = torch.randn(B, N, Cin, generator=g).to(dtype)
    w = (torch.randn(C, Cin, generator=g) / math.sqrt(Cin)).to(dtype)
    k = torch.randn(1 if shared else B, M, C, generator=g).to(dtype)
    gate = torch.ones(B)
    gate[B - 1] = 0.0
    assert ops.qproj_parts(x.to(gpu_device), w.to(gpu_device), k.to(gpu_device), H) > 0
    q, parts = ops.qproj_stat(x.to(gpu_device), w.to(gpu_device), k.to(gpu_device), H, ops.STAT_ALL, gate=gate.to(gpu_device))
    ref = x.double() @ w.double().t()
    ulp = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8
    err = (q.cpu().double() - ref).abs()
    assert bool((err <= ulp * ref.abs() + 2e-5).all()), (name, err.max().item())
    stats = ops.qk_stats(q, k.to(gpu_device), H).cpu()
    folded = ops.fold_parts(parts[: B - 1]).cpu()
    cnt = H * N * M
    mag = stats[: B - 1, :2].abs().max().item()
    sd = ((stats[: B - 1, 3] - stats[: B - 1, 2] ** 2 / cnt) / (cnt - 1)).clamp_min(0).sqrt()
    e_ext = (folded[:, :2] - stats[: B - 1, :2]).abs().max().item() / mag
    e_mean = ((folded[:, 2] - stats[: B - 1, 2]).abs() / cnt / sd).max().item()
    e_sq = ((folded[:, 3] - stats[: B - 1, 3]).abs() / stats[: B - 1, 3]).max().item()
    print(f"qproj {name} {dtype}: parts/image {parts.shape[1]}, Q max err {err.max().item():.2e}; extremes {e_ext:.1e}, mean {e_mean:.1e}, sumsq {e_sq:.1e}")
    assert e_ext <= 1e-6 and e_mean <= 1e-6 and e_sq <= 1e-6
    # only the fields a statistic is made of are formed when the caller names it: max alone gives the same maximum
    q2, p2 = ops.qproj_stat(x.to(gpu_device), w.to(gpu_device), k.to(gpu_device), H, ops.STAT_MAX)
    assert torch.equal(q2, q) and torch.equal(p2[: B - 1, :, 0], parts[: B - 1, :, 0])


def test_qproj_unsupported_shapes_say_so(gpu_device):
    from pww_hip import ops
    from pww_hip._lib import PwwHipError
    x = torch.randn(1, 64, 96, device=gpu_device, dtype=torch.float16)        # Cin = 96: not a multiple of 64
    w = torch.randn(320, 96, device=gpu_device, dtype=torch.float16)
    k = torch.randn(1, 77, 320, device=gpu_device, dtype=torch.float16)
    assert ops.qproj_parts(x, w, k, 8) == 0
    with pytest.raises(PwwHipError):
        ops.qproj_stat(x, w, k, 8, ops.STAT_MAX)
    w2 = torch.randn(200, 320, device=gpu_device, dtype=torch.float16)        # C = 200: no whole-head tile
    assert ops.qproj_parts(torch.randn(1, 64, 320, device=gpu_device, dtype=torch.float16), w2, torch.randn(1, 77, 200, device=gpu_device, dtype=torch.float16), 5) == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", ["sd15_n4096", "sd15_n1024", "sd15_n256", "sd15_n64", "sd21_n576"])
def test_product_path_with_and_without_the_gemm_epilogue_statistic(gpu_device, experiments_lib, shape, dtype, monkeypatch):
    """inj_forward through the default path (to_q GEMM with the statistic in its epilogue + pass-2-only attention) against the
    round-3 launch (PWW_QPROJ_STAT=0: stock to_q GEMM + statistic and hand-off inside the attention kernel) on the reference's
    attention cases, every shipped weight function: the two differ by the roundings of two different GEMM kernels (<= a few
    storage-type ulps of max|O|) -- and both keep the golden bar (test_attention_gpu.py)."""
    import pww_hip
    import pww_hip.attention as A
    case = cases.make_attention_case(shape)
    mod = case["attn_cross"].to(gpu_device, dtype)
    hidden = torch.cat([case["hidden"], case["hidden"].flip(1)]).to(gpu_device, dtype)       # 2 images: per-image statistics
    N = case["N"]
    for wname, wf in cases.WEIGHT_FUNCTIONS.items():
        ctx = {"CONTEXT_TENSOR": case["ctx"].to(gpu_device, dtype), f"CROSS_ATTENTION_WEIGHT_{N}": case["w"].to(gpu_device),
               "SIGMA": torch.tensor(7.84), "WEIGHT_FUNCTION": wf, "_PWW_ROW_GATE": torch.tensor([1.0, 1.0], device=gpu_device)}
        monkeypatch.setattr(A, "QPROJ_STAT", "all")
        a = pww_hip.inj_forward(mod, hidden, dict(ctx)).float()
        monkeypatch.setattr(A, "QPROJ_STAT", "0")
        monkeypatch.setattr(A, "FUSED_CROSS", True)          # round 3: stock to_q + statistic and hand-off inside the attention launch
        b = pww_hip.inj_forward(mod, hidden, dict(ctx)).float()
        monkeypatch.setattr(A, "FUSED_CROSS", False)         # round 5: stock to_q + pww_qk_parts + pass-2-only attention (what the C = 1280 layers take)
        c = pww_hip.inj_forward(mod, hidden, dict(ctx)).float()
        scale = b.abs().max().item()
        d = (a - b).abs().max().item()
        d5 = (c - b).abs().max().item()
        print(f"{shape} {dtype} {wname}: GEMM-epilogue route vs round-3 launch: max diff {d:.3e}, qk_parts route vs round-3 launch: {d5:.3e} of max|out| {scale:.3f}")
        assert torch.isfinite(a).all() and d <= TOL[dtype] * scale
        assert torch.isfinite(c).all() and d5 <= TOL[dtype] * scale
        assert (a[0] - a[1]).abs().max().item() > 10 * d or wname == "none"       # the two images really are different rows


def test_stale_graph_is_not_replayed_for_a_function_that_wants_the_bias(gpu_device):
    """request A with `lambda w, s, qk: 0` captures a graph whose cross-attention sites carry NO bias kernel;
    request B (same geometry) with the runner's weight function must not replay it. Every site now registers its class and
    CoeffSlots.update() reports a class change as "re-capture". Also the reverse order and a sigma-thresholded function that
    changes class in the middle of a request."""
    import paint_with_words as pw
    tools = cases.build_tools("tiny", dtype=torch.float16, device=gpu_device)
    kw = dict(color_map_image=Image.fromarray(cases.load_example_rgb()), input_prompt=cases.RUNNER_PROMPT, num_inference_steps=5,
              guidance_scale=7.5, seed=2, device=str(gpu_device), preloaded_utils=tools, return_latents=True)
    zero = lambda w, sigma, qk: 0                                                                    # noqa: E731
    thresholded = lambda w, sigma, qk: 0.4 * w * math.log(1 + sigma) * qk.max() if sigma > 3 else 0     # noqa: E731  (5 LMS steps: sigma = 14.6, 4.7, 1.9, 0.7, 0.03)
    try:
        with _mode("graph"):
            z = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), weight_function=zero, **kw)
            sampler = tools[1]._pww_samplers[(id(tools[4]), "graph")]
            slots = sampler._static_folded["_PWW_COEFF_SLOTS"]
            assert len(slots.sites) >= 3 and all(s["kind"] == slots.NO_BIAS for s in slots.sites)
            n0 = sampler._graphed.captures
            r = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), weight_function=cases.weight_fn_runner, **kw)
            assert sampler._graphed.captures == n0 + 1
            z2 = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), weight_function=zero, **kw)      # and back
            assert sampler._graphed.captures == n0 + 2
            n1 = sampler._graphed.captures
            t = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), weight_function=thresholded, **kw)
            assert sampler._graphed.captures >= n1 + 2       # zero -> biased at the first step, biased -> bias-free at the third
        with _mode("folded"):
            z_ref = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), weight_function=zero, **kw)
            r_ref = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), weight_function=cases.weight_fn_runner, **kw)
            t_ref = pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), weight_function=thresholded, **kw)
    finally:
        uninstall_all()
    gap = rel_l2(r_ref, z_ref)
    print(f"zero fn {rel_l2(z, z_ref):.2e} / runner fn after it {rel_l2(r, r_ref):.2e} / zero again {rel_l2(z2, z_ref):.2e} / thresholded {rel_l2(t, t_ref):.2e}; bias effect {gap:.2e}")
    assert gap > 5e-2
    for got, ref in ((z, z_ref), (r, r_ref), (z2, z_ref), (t, t_ref)):
        assert rel_l2(got, ref) <= 2e-2
    print(f"thresholded vs runner {rel_l2(t_ref, r_ref):.2e}, vs zero {rel_l2(t_ref, z_ref):.2e}")
    assert rel_l2(t_ref, z_ref) > 5e-2                                       # its first two steps carry the bias


def test_orig_map_is_lazy_and_survives_graph_requests(gpu_device):
    """CROSS_ATTENTION_WEIGHT_ORIG (80.7 MB at 512 x 512) is read only on inj_forward's KeyError path (:95-101): a request whose
    layers all find their per-resolution map never builds it; a request that needs it (pipeline class with height / width != the
    color map) builds it on first access -- also under hipGraph mode across two requests with DIFFERENT color maps."""
    import paint_with_words as pw
    from pww_hip.conditioning import _encode_text_color_inputs, PwWContext
    vae, unet, text, tok, sch = cases.build_tools("tiny", dtype=torch.float16, device=gpu_device)
    ex = cases.load_example_rgb()
    try:
        _, _, cond, uncond = _encode_text_color_inputs(text, tok, gpu_device, ex, dict(cases.RUNNER_CONTEXT), cases.RUNNER_PROMPT, "")
        assert isinstance(cond, PwWContext) and cond.pending("CROSS_ATTENTION_WEIGHT_ORIG") and "CROSS_ATTENTION_WEIGHT_ORIG" in cond
        assert "CROSS_ATTENTION_WEIGHT_ORIG" not in dict(cond)
        with _mode("graph"):
            pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), color_map_image=Image.fromarray(ex), input_prompt=cases.RUNNER_PROMPT,
                                num_inference_steps=2, device=str(gpu_device), preloaded_utils=(vae, unet, text, tok, sch), return_latents=True)
            sampler = unet._pww_samplers[(id(sch), "graph")]
            assert sampler._static_folded.pending("CROSS_ATTENTION_WEIGHT_ORIG")          # 512 x 512: nobody asked
        orig = cond["CROSS_ATTENTION_WEIGHT_ORIG"]                                          # first access builds it (ratio-1 mask kernel)
        assert orig.shape == (512, 512, 77) and not cond.pending("CROSS_ATTENTION_WEIGHT_ORIG")
        # height = width = 384 with 512 x 512 color maps: every layer takes the fallback; second request: the map mirrored left-right
        pipe = pw.PaintWithWord_StableDiffusionPipeline(vae, text, tok, unet, sch)
        kw = dict(prompt=cases.RUNNER_PROMPT, color_context=dict(cases.RUNNER_CONTEXT), weight_function=cases.weight_fn_runner, height=384, width=384,
                  num_inference_steps=3, seed=5, output_type="np")
        outs = {}
        for mode in ("graph", "folded"):
            with _mode(mode):
                for tag, rgb in (("a", ex), ("b", np.ascontiguousarray(ex[:, ::-1]))):
                    outs[mode, tag] = pipe(color_map_image=Image.fromarray(rgb), **kw).images
        sampler = unet._pww_samplers[(id(pipe.scheduler), "graph")]        # (the class replaces the scheduler it is given, :533-538)
        assert not sampler._static_folded.pending("CROSS_ATTENTION_WEIGHT_ORIG") and sampler._graphed.captures == 1
    finally:
        uninstall_all()
    for tag in ("a", "b"):
        d = float(np.abs(outs["graph", tag] - outs["folded", tag]).mean())
        assert d <= 2e-2, (tag, d)
    assert float(np.abs(outs["graph", "a"] - outs["graph", "b"]).mean()) > 3 * max(float(np.abs(outs["graph", t] - outs["folded", t]).mean()) for t in ("a", "b"))


def test_one_mask_launch_equals_four(gpu_device):
    """pww_mask_build / pww_mask_build_f32_levels form the 8 / 16 / 32 / 64 maps in ONE launch: bit-identical to a launch per ratio."""
    from pww_hip import ops
    from pww_hip.conditioning import _parse_regions, _column_lists
    from sd_standin import HashTokenizer
    tok = HashTokenizer()
    for rgb_np in (cases.load_example_rgb(), cases.load_example_rgb()[:500, :500].copy(), cases.grid_case()[0]):
        ctx = dict(cases.RUNNER_CONTEXT)
        table = _parse_regions(ctx, tok)
        ids = tok([cases.RUNNER_PROMPT], padding="max_length", max_length=77, truncation=True, return_tensors="pt")["input_ids"][0].tolist()
        cols = _column_lists(table, ids)
        rgb = torch.from_numpy(np.ascontiguousarray(rgb_np)).to(gpu_device)
        regions = [(c[0], c[1], c[2], s) for (_, c, s) in table]
        multi = ops.mask_build(rgb, regions, cols, (8, 16, 32, 64))
        masks = torch.stack([(rgb == torch.tensor(c, dtype=torch.uint8, device=gpu_device)).all(dim=-1).float() * s for (_, c, s) in table])
        multi_f = ops.mask_build_f32(masks, cols, (8, 16, 32, 64))
        for r in (8, 16, 32, 64):
            single = ops.mask_build(rgb, regions, cols, (r,))[r]
            assert torch.equal(multi[r], single) and torch.equal(multi_f[r], single), r


# ---- N > 1 on hardware -----------------------------------------------------------------------------------

def _run_bench(args, env_extra=None, launcher=None, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PWW_BENCH_VERBOSE="0", PWW_MIOPEN_FIND="0")
    env.update(env_extra or {})
    cmd = ([sys.executable] + (launcher or [])) + [os.path.join(cases.REPO, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=cases.REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_under_torchrun_with_rccl(gpu_device):
    """The driver's multi-GPU launch, with one rank: python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 with the nccl
    (= RCCL) backend -- RCCL init, the bucketed weight broadcast, the request broadcast, the barrier and the max-over-ranks
    all-reduce execute on the MI355X in every round's GPU tests (only the 8-GPU node itself is the driver's)."""
    launcher = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29613"]
    line = _run_bench(["--gpus", "1", "--config", "3", "--denoise-steps", "4", "--steps", "1", "--warmup", "1", "--no-roofline-pass",
                       "--no-reference-ops", "--cpu-steps", "0"], launcher=launcher)
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["config"]["weight_broadcast"]["broadcast_bytes"] > 1.5e9
    assert line["config"]["backend"] == "nccl" and line["config"]["weight_broadcast_s"] is not None


def test_results_do_not_depend_on_the_rank_count(gpu_device, tmp_path):
    """image i of the global batch is the same image whatever the number of ranks (seeds by global index, CPU generator:
    paint_with_words.py:446 / gradio_pww.py:24-45). bench.py --dump-latents with 1 rank (4 images) and with 2 ranks sharing this
    GPU (gloo, 2 images each): the same 4 final latents up to the batch-size dependence of the stock conv / GEMM kernels. Since round 5 both
    runs take the DEFAULT cross-attention route (no launch needs its workgroups resident at once: two ranks may share a device), and the
    2-rank line carries every rank's seconds / images per second (`config.per_rank`)."""
    args = ["--config", "3", "--denoise-steps", "3", "--steps", "1", "--warmup", "0", "--no-roofline-pass", "--no-reference-ops", "--cpu-steps", "0",
            "--tiny"]
    one = _run_bench(["--gpus", "1", "--batch", "4", "--dump-latents", str(tmp_path / "r1")] + args)
    two = _run_bench(["--gpus", "2", "--batch", "2", "--dump-latents", str(tmp_path / "r2")] + args, env_extra={"PWW_DIST_ONE_DEVICE": "1"})
    assert one["config"]["images_per_step"] == two["config"]["images_per_step"] == 4
    assert len(two["config"]["per_rank"]["seconds"]) == 2 and len(one["config"]["per_rank"]["images_per_s"]) == 1
    a = np.load(str(tmp_path / "r1") + "_rank0.npy")
    b = np.concatenate([np.load(str(tmp_path / "r2") + "_rank%d.npy" % r) for r in (0, 1)])
    assert a.shape == b.shape == (4, 4, 64, 64)
    for i in range(4):
        d = rel_l2(b[i], a[i])
        print(f"image {i}: 2 ranks vs 1 rank rel-L2 {d:.3e}")
        assert d <= 2e-2
    assert rel_l2(a[0], a[1]) > 0.1

"""Round-6 GPU tests.
  * the configuration bench.py TIMES (hipGraph + channels_last + MIOpen find mode) carries a final-latent check of its
    own -- bench.py's `parity` object -- and it holds for configs 2, 3 (fp16) and 4 against the committed fixtures of the reference's loop;
  * item 7a: with two ranks rank 0 warms up first and rank 1 adopts its MIOpen user db;
  * a statistic-free weight function over a context of more than 128 tokens in hipGraph mode (device coefficient word).
"""
import json
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import pww_cases as cases
from oracle import pww_oracle as O

pytestmark = pytest.mark.gpu


def _run_bench(args, env_extra=None, timeout=1500):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PWW_BENCH_VERBOSE="0")
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(cases.REPO, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=cases.REPO)
    assert out.returncode == 0, out.stderr[-3000:]
    return json.loads(out.stdout)            # stdout carries the line and nothing else


# drift of the unfused half-precision torch path on the same GPU: the calibrated bar of every loop test is 1.5 x that + 2e-3
UNFUSED_DRIFT = {2: 8.3e-3, 3: 1.2e-3, 4: 1.11e-2}


@pytest.mark.parametrize("config", [2, 3, 4])
def test_bench_parity_in_the_timed_configuration(gpu_device, config):
    """bench.py --config {2, 4} exactly as the driver times it -- hipGraph mode, UNet in channels_last, MIOpen find mode ON (the test suite's
    other loop tests run immediate mode and NCHW) -- one warm-up step (global step 0 = the fixture's seeds) and one timed step: the line's
    `parity` holds the rel-L2 of the warm-up step's final latents against the reference-loop fixture, inside BASELINE.md's bar AND inside
    the calibrated bar of the NCHW loop tests (1.5 x the unfused torch path's drift + 2e-3)."""
    line = _run_bench(["--config", str(config), "--steps", "1", "--warmup", "1", "--no-roofline-pass", "--no-reference-ops", "--cpu-steps", "0"],
                      env_extra={"PWW_MIOPEN_FIND": "1"})
    par = line["parity"]
    print("bench.py --config %d parity: %s" % (config, par))
    assert "MIOpen find mode" in line["config"]["stock_op_settings"] and "channels_last" in line["config"]["stock_op_settings"]
    assert line["config"]["mode"] == "graph" and line["config"]["hipgraph_captures"] == 1
    assert par is not None and par["ok"] and par["bar"] == (1e-2 if config == 3 else 5e-2)      # (config 3 is the fp16 workload: BASELINE.md section 4)
    assert par["rel_l2"] <= 1.5 * UNFUSED_DRIFT[config] + 2e-3
    assert sorted(par["per_image"]) == (["0", "5"] if config == 4 else ["0"])


def test_bench_fails_above_the_parity_bar(gpu_device):
    """The run FAILS -- non-zero exit, no json line -- when the final latents leave the bar (here: a bar no half-precision run can meet)."""
    env = dict(os.environ, PWW_BENCH_VERBOSE="0", PWW_MIOPEN_FIND="0")
    out = subprocess.run([sys.executable, os.path.join(cases.REPO, "bench.py"), "--steps", "1", "--warmup", "1", "--no-roofline-pass", "--no-reference-ops",
                          "--cpu-steps", "0", "--parity-bar", "1e-6"], capture_output=True, text=True, timeout=900, env=env, cwd=cases.REPO)
    assert out.returncode != 0 and out.stdout.strip() == ""
    assert "above the 1e-06 bar" in out.stderr


def test_two_ranks_stage_their_warmup(gpu_device, tmp_path):
    """bench.py --gpus 2 (two ranks sharing this GPU over gloo, 1/8-width model, MIOpen find mode on): rank 0 warms up first, rank 1 adopts
    its MIOpen user db before its own first convolution; the line carries every rank's warm-up seconds and the files adopted."""
    line = _run_bench(["--gpus", "2", "--config", "3", "--batch", "1", "--denoise-steps", "2", "--steps", "1", "--warmup", "1", "--no-roofline-pass",
                       "--no-reference-ops", "--cpu-steps", "0", "--tiny"],
                      env_extra={"PWW_DIST_ONE_DEVICE": "1", "PWW_MIOPEN_FIND": "1", "PWW_MIOPEN_DB_BASE": str(tmp_path / "miopen")})
    db = line["config"]["miopen_db"]
    print("staged warm-up:", db, line["config"]["warmup_s_per_rank"])
    assert db["staged_warmup"] is True and db["files_adopted_from_rank0"][0] == 0 and db["files_adopted_from_rank0"][1] >= 1
    assert len(line["config"]["warmup_s_per_rank"]) == 2 and len(line["config"]["warmup_s_per_rank"][0]) == 1
    assert line["parity"] is None                   # (--tiny is not the fixture's workload)
    r0, r1 = (set(os.listdir(tmp_path / "miopen" / ("pww_rank%d" % r))) for r in (0, 1))
    assert {f for f in r0 if not f.endswith(".lock")} <= r1


# ---- stat = (None, STAT_NONE, c) with M > 128 ------------------------------------------------------------------

@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_stat_free_weight_function_long_context(gpu_device, dtype):
    """A statistic-free weight function (0.4 w log(1 + sigma)) over a 154-token context with the hipGraph mode's device coefficient word:
    attention.py hands ops.attention stat = (None, STAT_NONE, c) without scratch or partials. Round 5's dispatch sent that to the
    pass-2-only launch (<= 128 keys) and raised; it belongs on the general launch (no key limit)."""
    from pww_hip import ops
    B, N, H, D, M = 2, 256, 8, 40, 154
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, N, H * D, generator=g).to(dtype)
    k = torch.randn(1, M, H * D, generator=g).to(dtype)
    v = torch.randn(1, M, H * D, generator=g).to(dtype)
    w = (torch.rand(N, M, generator=g) < 0.2).float() * torch.rand(N, M, generator=g)
    gate = torch.tensor([1.0, 0.0])
    c = 0.4 * math.log(1 + 7.84)
    dev = gpu_device
    coeff_dev = torch.tensor([c], dtype=torch.float32, device=dev)
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), H, D ** -0.5, bias=w.to(dev), bias_coeff=gate.to(dev), stat=(None, ops.STAT_NONE, 123.0),
                        coeff_dev=coeff_dev).float().cpu()
    out_scalar = ops.attention(q.to(dev), k.to(dev), v.to(dev), H, D ** -0.5, bias=w.to(dev), bias_coeff=gate.to(dev), stat=(None, ops.STAT_NONE, c)).float().cpu()
    qh, kh, vh = (O.split_heads(t.double(), H) for t in (q, k.expand(B, -1, -1), v.expand(B, -1, -1)))
    logits = torch.matmul(qh, kh.transpose(-1, -2)).view(B, H, N, M) + (c * gate.double())[:, None, None, None] * w.double()[None, None]
    ref = O.merge_heads(torch.matmul((logits * D ** -0.5).softmax(-1).view(B * H, N, M), vh), H)
    tol = 2e-3 if dtype == torch.float16 else 1.6e-2
    for name, o in (("device word", out), ("scalar", out_scalar)):
        err = (o.double() - ref).abs().max().item() / ref.abs().max().item()
        print(f"stat-free weight function, M = {M}, {dtype}, coefficient from the {name}: max err / max|O| = {err:.3e}")
        assert err <= tol
    # with partials the pass-2-only launch still says what it takes
    with pytest.raises(ops.PwwHipError):
        ops.attention(q.to(dev), k.to(dev), v.to(dev), H, D ** -0.5, bias=w.to(dev), stat=(None, ops.STAT_MAX, c),
                      parts=torch.zeros(B, 4, 4, dtype=torch.float64, device=dev))


# ---- VERDICT round 5 item 8 / "missing" 4: the GPU reference's half-rounded statistic vs the fp64 statistic here ------------------------

@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", ["sd15_n4096", "sd15_n256"])
def test_half_rounded_statistic_delta(gpu_device, shape, dtype):
    """On a GPU the reference runs inj_forward under fp16 autocast (paint_with_words.py:60): `qk` is the HALF output of a half GEMM, so its
    `qk.max()` (:87, runner.py:104) is the true maximum rounded to half precision -- one half-ulp (2^-11 fp16 / 2^-8 bf16, relative) away
    from the fp64 statistic this package forms over fp32-accumulated scores (the stated target is the fp32 CPU path: closer to that). The
    size of the difference, measured: the statistic itself, and what it does to the attention output through c = 0.4 w log(1 + sigma) max
    (the same kernel, once with each statistic). DESIGN section 2 quotes these numbers."""
    from pww_hip import ops
    case = cases.make_attention_case(shape)
    N, C, H = case["N"], case["C"], case["H"]
    mod = case["attn_cross"].to(gpu_device, dtype)
    hidden, ctx = case["hidden"].to(gpu_device, dtype), case["ctx"].to(gpu_device, dtype)
    q, k, v = mod.to_q(hidden), mod.to_k(ctx), mod.to_v(ctx)
    w = case["w"].to(gpu_device)
    stats = ops.qk_stats(q, k, H)                                              # fp64 { max, min, sum, sum of squares } over fp32-accumulated scores
    half_scores = torch.matmul(O.split_heads(q, H), O.split_heads(k, H).transpose(-1, -2))      # what the reference's weight function sees on a GPU
    half_max = half_scores.max().double()
    rel = abs(half_max.item() - stats[0, 0].item()) / abs(stats[0, 0].item())
    ulp_half = 2.0 ** -11 if dtype == torch.float16 else 2.0 ** -8
    c = 0.4 * math.log(1 + 7.84)
    exact = ops.attention(q, k, v, H, mod.scale, bias=w, stat=(stats, ops.STAT_MAX, c)).float()
    rounded_stats = stats.clone()
    rounded_stats[0, 0] = half_max
    rounded = ops.attention(q, k, v, H, mod.scale, bias=w, stat=(rounded_stats, ops.STAT_MAX, c)).float()
    d_out = (exact - rounded).abs().max().item() / exact.abs().max().item()
    d_bias = c * float(w.max()) * abs(half_max.item() - stats[0, 0].item())      # largest change of a raw logit
    print(f"half-rounded qk.max() {shape} {dtype}: fp64 max {stats[0, 0].item():.4f}, half max {half_max.item():.4f} (relative {rel:.2e}, half-ulp {ulp_half:.2e}); "
          f"largest logit change {d_bias:.3e}; output change {d_out:.2e} of max|O| (per-call bar {2e-3 if dtype == torch.float16 else 1.6e-2:.1e})")
    assert rel <= 1.01 * ulp_half
    assert d_out <= (2e-3 if dtype == torch.float16 else 1.6e-2)


# ---- the fp16 d = 40 launch on hot logits ------------------------------------------------------------------------

def _path_counts(fn, device):
    import ctypes
    from pww_hip import _lib
    lib = _lib.load()
    counts = torch.zeros(4, dtype=torch.int32, device=device)
    lib.pww_debug_path_counts(ctypes.c_void_p(counts.data_ptr()))
    try:
        out = fn()
        torch.cuda.synchronize()
    finally:
        lib.pww_debug_path_counts(None)
    c = counts.tolist()
    return out, {"fast": c[0], "lazy": c[1], "raw": c[3], "exact": c[2]}


@pytest.mark.parametrize("std,expect", [(0.5, "fast"), (4.0, "lazy"), (5.0, "lazy")])
def test_fp16_hot_rows_leave_the_range_free_mode_not_the_fast_path(gpu_device, std, expect):
    """fp16, N = 4096, d = 40 (the dominant launch of config 3): on cold logits (scaled-logit std 0.5: the random-init UNet's regime) every
    workgroup stays range-free; at std 4 / 5 (what trained layers produce) the workgroups follow the running maximum lazily and NONE
    recomputes its rows on the exact path (round 5: a third / nearly all of them did, after a complete fast pass) -- inside the per-call
    bar against fp64 on the same rounded inputs. pww_debug_path_counts reports the workgroups per path."""
    from pww_hip import ops
    B, N, H, D = 2, 4096, 8, 40
    g = torch.Generator().manual_seed(7)
    gain = math.sqrt(std)
    q = (torch.randn(B, N, H * D, generator=g) * gain).to(torch.float16)
    k = (torch.randn(B, N, H * D, generator=g) * gain).to(torch.float16)
    v = torch.randn(B, N, H * D, generator=g).to(torch.float16)
    qd, kd, vd = q.to(gpu_device), k.to(gpu_device), v.to(gpu_device)
    out, paths = _path_counts(lambda: ops.attention(qd, kd, vd, H, D ** -0.5), gpu_device)
    total = sum(paths.values())
    rows = torch.arange(0, N, 97)
    qh = q[0, rows].double().view(len(rows), H, D).transpose(0, 1)
    kh, vh = (t[0].double().view(N, H, D).transpose(0, 1) for t in (k, v))
    logits = torch.matmul(qh, kh.transpose(-1, -2)) * D ** -0.5
    ref = torch.matmul(logits.softmax(-1), vh).transpose(0, 1).reshape(len(rows), H * D)
    err = (out[0, rows].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"fp16 d=40 scaled-logit std {std}: row maxima {logits.max(-1).values.mean():.1f} (max {logits.max():.1f}) natural units; workgroups {paths}; max err / max|O| = {err:.2e}")
    # (cold: a stray row -- 1 of 65536 had a first-stage sum below 8 when this was written -- may take its workgroup to the lazy path: +2 % on that one)
    on_path = paths["fast"] if expect == "fast" else paths["lazy"] + paths["raw"]      # (hot: lazy reference, with the folded or -- near the guard -- the exact scale)
    assert total == B * H * (N // 256) and paths["exact"] == 0 and on_path >= (0.98 if expect == "fast" else 1.0) * total
    assert err <= 2e-3
    # the same inputs in bf16 never leave the range-free fast path (8 exponent bits)
    _, pb = _path_counts(lambda: ops.attention(qd.bfloat16(), kd.bfloat16(), vd.bfloat16(), H, D ** -0.5), gpu_device)
    assert pb["fast"] == sum(pb.values()) > 0


def test_fp16_rows_past_the_magnitude_guard_take_the_exact_path(gpu_device):
    """Scaled-logit std 8 (row maxima of 29 natural units and more: past the 33-unit guard for many rows) in the 4-WAVE form (one image of
    2048 rows; it has no registers for the exact-scale loop): those workgroups recompute on exact_rows -- counted as such -- and the result
    stays inside the per-call bar."""
    from pww_hip import ops
    B, N, H, D = 1, 2048, 8, 40
    g = torch.Generator().manual_seed(3)
    gain = math.sqrt(8.0)
    q = (torch.randn(B, N, H * D, generator=g) * gain).to(torch.float16)
    k = (torch.randn(B, N, H * D, generator=g) * gain).to(torch.float16)
    v = torch.randn(B, N, H * D, generator=g).to(torch.float16)
    out, paths = _path_counts(lambda: ops.attention(q.to(gpu_device), k.to(gpu_device), v.to(gpu_device), H, D ** -0.5), gpu_device)
    rows = torch.arange(0, N, 53)
    qh = q[0, rows].double().view(len(rows), H, D).transpose(0, 1)
    kh, vh = (t[0].double().view(N, H, D).transpose(0, 1) for t in (k, v))
    ref = torch.matmul((torch.matmul(qh, kh.transpose(-1, -2)) * D ** -0.5).softmax(-1), vh).transpose(0, 1).reshape(len(rows), H * D)
    err = (out[0, rows].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"fp16 d=40 scaled-logit std 8: workgroups {paths}; max err / max|O| = {err:.2e}")
    assert paths["exact"] > 0 and torch.isfinite(out).all() and err <= 2e-3


@pytest.mark.parametrize("std,B,N", [(6.0, 2, 4096), (7.0, 2, 4096), (8.0, 2, 4096), (6.0, 3, 4000), (8.0, 3, 4000)])
def test_fp16_rows_past_the_guard_continue_on_the_exact_scale(gpu_device, std, B, N):
    """fp16, 8-wave workgroups (B = 2, N = 4096), scaled-logit std 6 / 7: rows whose logits pass the magnitude guard (33 natural units) no longer
    send their workgroup through a second pass -- the workgroup continues with the unscaled Q and P = exp2(x c1) from the stage where the
    reference comes close to the limit (`raw` workgroups), inside the per-call bar; a FIRST key stage that is already past the limit (std 7 / 8:
    most workgroups) starts over in that mode: no workgroup of the 8-wave form takes exact_rows any more."""
    from pww_hip import ops
    H, D = 8, 40           # (N = 4000: a ragged last key stage -- 32 keys -- and a ragged last query block, in the exact-scale mode)
    g = torch.Generator().manual_seed(7)
    gain = math.sqrt(std)
    q = (torch.randn(B, N, H * D, generator=g) * gain).to(torch.float16)
    k = (torch.randn(B, N, H * D, generator=g) * gain).to(torch.float16)
    v = torch.randn(B, N, H * D, generator=g).to(torch.float16)
    out, paths = _path_counts(lambda: ops.attention(q.to(gpu_device), k.to(gpu_device), v.to(gpu_device), H, D ** -0.5), gpu_device)
    rows = torch.cat([torch.arange(0, N, 41), torch.tensor([N - 1])])
    err = 0.0
    for b in range(B):
        qh = q[b, rows].double().view(len(rows), H, D).transpose(0, 1)
        kh, vh = (t[b].double().view(N, H, D).transpose(0, 1) for t in (k, v))
        ref = torch.matmul((torch.matmul(qh, kh.transpose(-1, -2)) * D ** -0.5).softmax(-1), vh).transpose(0, 1).reshape(len(rows), H * D)
        err = max(err, (out[b, rows].double().cpu() - ref).abs().max().item() / ref.abs().max().item())
    print(f"fp16 d=40 scaled-logit std {std}: workgroups {paths}; max err / max|O| = {err:.2e}")
    assert sum(paths.values()) == B * H * ((N + 255) // 256) and paths["raw"] > 0 and paths["fast"] == 0
    assert paths["exact"] == 0
    assert torch.isfinite(out).all() and err <= 2e-3


# ---- workgroup orders that keep adjacent heads on one XCD (d = 40: 80-byte slices of 128-byte lines) --------------------------------------

def test_head_group_workgroup_orders_are_bit_identical_to_round5s(gpu_device):
    """Round 6 deals the workgroups of the d = 40 self-attention launches in groups of two adjacent heads per XCD and the multi-block
    cross-attention launch with all heads of an (image, query chunk) unit on one XCD (a read request moves a whole 128-byte line: 12 line
    fills per 640-byte row became 8 and 5, profiles/r06_sector_sharing.md). Both are permutations of the grid: every output must be
    BIT-identical to round 5's order (PWW_DEBUG=attn_head_pairs=0,cross_head_major=0; own processes: the library reads its knobs once) and
    inside the per-call bars -- self-attention at 2 / 3 / 4 / 16 rows (even and odd pair counts, ragged N, hot logits), d = 80 / 160, and the
    batched cross route with partials, the 32-column bound and the gated-images hint at 16 / 8 / 6 / 5 rows and SD2.1's 9216 tokens x 5 heads
    of 64. The batched cross launches run on the small kernel's several-blocks-per-workgroup form since round 6 (34 us against the general
    kernel's 37 at 16 rows): the same tile code in the same order -- bit-identical to the general kernel too (PWW_DEBUG=cross_lean_multi=0)."""
    script = os.path.join(cases.REPO, "tools", "diag_wg_order.py")
    runs = {}
    for name, knobs in (("round 6", ""), ("round 5", "attn_head_pairs=0,cross_head_major=0"), ("groups of four", "attn_head_pairs=4"),
                        ("general cross kernel", "cross_lean_multi=0"), ("round 5, general cross kernel", "cross_lean_multi=0,cross_head_major=0")):
        out = subprocess.run(["timeout", "600", sys.executable, script], capture_output=True, text=True, env=dict(os.environ, PWW_DEBUG=knobs))
        assert out.returncode == 0, (name, out.stdout[-2000:], out.stderr[-2000:])
        runs[name] = [l for l in out.stdout.splitlines() if l.startswith("CASE")]
        assert len(runs[name]) == 24 and not [l for l in runs[name] if "FAIL" in l]
    print("\n".join(runs["round 6"]))
    assert runs["round 6"] == runs["round 5"] == runs["groups of four"] == runs["general cross kernel"] == runs["round 5, general cross kernel"]

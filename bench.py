#!/usr/bin/env python
"""bench.py -- SD1.5 + Paint-with-Words images/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}]

One "step" = one pass of the hot path over one batch: every rank generates `batch` images through the PUBLIC batched
entry point (paint_with_words_batch / paint_with_words_inpaint_batch: mask build + conditioning + the denoise loop
with the fused PwW attention, CFG 7.5) up to the final latent (the quantity parity is checked on; the VAE decode sits
outside the path and outside the timed region, like in the reference's own it/s numbers).

Workloads (numbered as there; `--config 2` is the default and the one `metric` is quoted on):
  2  SD1.5 512x512 bf16, 30 PLMS steps, 5-region example_input.png (runner.py:9-20), 0.4 w log(1+sigma) qk.max(), 1 image / GPU
  3  SD1.5 512x512 fp16, 50 LMS steps, 8 vertical stripes, 8 images / GPU (64 over 8 GPUs), seeds 0..63
  4  SD1.5-inpainting (9 input channels) 512x512 bf16, 30 LMS steps, aurora_1.png + 4 regions + moon_mask.png,
     0.15 w log(1+sigma) qk.max() (runner_inpaint.py:87), 8 images / GPU, seeds 81+i
  5  SD2.1 768x768 bf16 (head dim 64), 30 LMS steps, 12-region grid with per-region seeds, 0.4 w log(1+sigma^2) qk.std(),
     4 images / GPU (32 over 8 GPUs)
All UNets are the random-init stand-ins of the real topologies (seed 1234), built on rank 0 and broadcast over RCCL.

`--gpus N` with N > 1 and no launcher in the environment re-executes itself under torch.distributed.run (one rank per
GPU, 127.0.0.1 rendezvous); under the driver's own torchrun it uses the ranks it is given.

Prints ONE JSON line (rank 0) with the driver's contract fields plus
  "roofline"     dominant kernel (self-attention at the finest resolution) TFLOP/s vs the dense MFMA peak, timed live in
                 an instrumented (eager) pass of the same workload: every launch stamps its own start / end device
                 timestamps into a HIP event pair on the launch stream (pww_profile_arm -> hipExtLaunchKernelGGL), i.e.
                 the kernel duration rocprofv3 reports, without host latency or dispatch gaps; the 40-launch back-to-back
                 hipGraph replay number (event interval / 40) is kept beside it; `traffic` (HBM-side bytes per launch) and
                 `mfma_busy` are measured LIVE by three rocprofv3 --pmc passes over the native harness's launch of the same
                 shape (~4 s; the committed constant, labelled as such, when rocprofv3 is absent or with --no-live-counters);
  "counters_cross_route"  the same two counters for the C = 1280 cross-attention pair (pww_qk_parts + the small kernel);
  "attention_path"  per UNet forward: the dominant launches, and every other pww launch class summed (kernel-only in the
                 workload, and back to back);
  "kernels"      the same pass's table for EVERY pww launch class: average duration, algorithmic TFLOP/s and GB/s,
                 bounding roofline and fraction; plus a hot-logit run of the dominant shape;
  "cpu_baseline" the CPU oracle (port of the reference path) timed on this box's host cores on a bounded sample
                 (rank 0, N=1 only): torch thread counts 8 / 16 / 32 / 64 / 128 are swept (median of three UNet forwards each) and the
                 BEST one is used for the timed denoise step(s) -- the 256-vCPU bench box is slower oversubscribed --, with the
                 AST-loaded reference's own timing from the build box beside it (`host`: gpu_box_port / build_box_reference);
  "parity"       rel-L2 of THIS run's final latents of global step 0 (its first warm-up step: the timed configuration itself -- hipGraph,
                 channels_last, MIOpen find mode) against the committed fixture of the workload under tests/golden/ (final latents of the
                 reference's own loop, fp32 CPU; config 2: the oracle's PLMS loop), with the bar (bf16 5e-2 / fp16 1e-2: BASELINE.md section 4);
                 the run FAILS above the bar. null when the run is not the fixture's workload (overridden dtype / steps / guidance, --tiny);
  "config"       the workload, plus block_norms_calls (fused / declined calls of the block plug; the run fails on a hit rate below
                 1.0), per_rank (every rank's seconds and images/s), shards, warmup_s (N > 1: warmup_s_per_rank -- rank 0 warms up first and
                 the other ranks adopt its MIOpen user db, `miopen_db`), the weight / request broadcast seconds.
stdout carries that line and nothing else; the log and whatever the path prints go to stderr.

Test infrastructure used as bench infrastructure (deliberately, so that bench and parity tests see the same inputs):
`tests/pww_cases.py` (workload definitions: color maps, color_contexts, prompts, weight functions, stand-in builders) and
`tests/gpu_util.py` (install_unfused: the reference's op sequence as unfused torch ops, the "reference ops on this GPU" bar);
`oracle/pww_oracle.py` only inside the `cpu_baseline` leg.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "paint-with-words-sd_amd")
for p in (PKG, REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

os.environ.setdefault("PYTORCH_MIOPEN_SUGGEST_NHWC", "1")    # lets MIOpen take NHWC tensors as they are (see --memory-format)

import numpy as np  # noqa: E402
import torch  # noqa: E402

_T0 = time.time()


def log(*a):
    if os.environ.get("PWW_BENCH_VERBOSE", "1") == "1":
        print("[bench %7.1fs]" % (time.time() - _T0), *a, file=sys.stderr, flush=True)


MFMA_PEAK_TFLOPS = 2500.0     # dense bf16/f16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

# ---- workloads ------------------------------------------------------------------------------------------------------

CONFIGS = {
    2: dict(model="sd15", size=512, dtype="bf16", scheduler="plms", denoise_steps=30, batch=1, wf="runner", kind="txt2img",
            name="BASELINE configs[1]: SD1.5 UNet topology random-init (seed 1234), 512x512, 5-region example_input.png mask"),
    3: dict(model="sd15", size=512, dtype="fp16", scheduler="lms", denoise_steps=50, batch=8, wf="runner", kind="txt2img",
            name="BASELINE configs[2]: SD1.5 UNet topology random-init (seed 1234), 512x512, 8 vertical stripes (strengths 0.2+0.2i)"),
    4: dict(model="sd15_inpaint", size=512, dtype="bf16", scheduler="lms", denoise_steps=30, batch=8, wf="inpaint", kind="inpaint",
            name="BASELINE configs[3]: SD1.5-inpainting topology (9 input channels) random-init, 512x512, aurora_1.png 4 regions + moon_mask.png, strength 1.0"),
    5: dict(model="sd21", size=768, dtype="bf16", scheduler="lms", denoise_steps=30, batch=4, wf="std", kind="txt2img",
            name="BASELINE configs[4]: SD2.1 UNet topology (head dim 64, linear projections) random-init, 768x768, 12-region grid with per-region seeds"),
}


# Final-latent parity of the benchmarked configuration: the fixture of each BASELINE workload under tests/golden/
# (final latents of the reference's own loop on the same request and seeds, fp32 CPU; config 2: the oracle's PLMS loop -- the reference cannot
# run PLMS) and which images of global step 0 it holds. The bars are BASELINE.md section 4's: rel-L2 <= 5e-2 (bf16) / 1e-2 (fp16).
PARITY_FIXTURES = {
    2: ("loop_sd15_example_plms30_oracle.npz", {0: "latents"}),
    3: ("loop_sd15_stripes8_lms50.npz", {0: "latents_0"}),          # (image 5 of the fixture has a rotated map: the bench's request is shared)
    4: ("loop_sd15_inpaint_lms30.npz", {0: "latents_81", 5: "latents_86"}),
    5: ("loop_sd21_grid768_lms30.npz", {0: "latents_0"}),
}
PARITY_BARS = {"bf16": 5e-2, "fp16": 1e-2}


def parity_check(args, cfg, lat0, lo):
    """rel-L2 of this rank's final latents of GLOBAL step 0 (seeds base + global index, the fixture's) against the committed fixture, in the
    EXACT mode the timed steps run in (hipGraph / channels_last / MIOpen find as selected). `lo` = global index of lat0[0]. None when the run
    is not the fixture's workload (overridden dtype / scheduler / step count / guidance, 1/8-width model) or holds none of its images."""
    base = CONFIGS[args.config]
    if args.tiny or lat0 is None or abs(args.guidance - 7.5) > 1e-9 or any(cfg[k] != base[k] for k in ("dtype", "scheduler", "denoise_steps")):
        return None
    name, images = PARITY_FIXTURES[args.config]
    path = os.path.join(REPO, "tests", "golden", name)
    if not os.path.isfile(path):
        return None
    g = np.load(path)
    got = lat0.detach().float().cpu().numpy()
    per_image = {}
    for gi, key in images.items():
        if lo <= gi < lo + got.shape[0]:
            ref = g[key].astype(np.float64)
            per_image[str(gi)] = float(np.linalg.norm(got[gi - lo:gi - lo + 1].astype(np.float64) - ref) / np.linalg.norm(ref))
    if not per_image:
        return None
    worst = max(per_image.values())
    bar = args.parity_bar if getattr(args, "parity_bar", None) is not None else PARITY_BARS[cfg["dtype"]]
    return {"rel_l2": round(worst, 6), "bar": bar, "ok": bool(worst <= bar), "fixture": "tests/golden/" + name, "per_image": {k: round(v, 6) for k, v in per_image.items()},
            "reference": "oracle PLMS loop (the reference cannot run PLMS), fp32 CPU" if args.config == 2 else "the reference's own loop, fp32 CPU (oracle/make_golden.py)",
            "mode": "the timed configuration itself: global step 0 (the first warm-up step, or the first timed step with --warmup 0)"}


def weight_functions():
    import pww_cases as cases
    return {"runner": cases.weight_fn_runner, "std": cases.weight_fn_std, "inpaint": cases.weight_fn_inpaint, "default": cases.weight_fn_default}


def make_request(cfg_id):
    """Rank 0: the request of a workload as plain arrays / python values (broadcast to the other ranks)."""
    import pww_cases as cases
    if cfg_id == 2:
        return {"rgb": cases.load_example_rgb(), "context": dict(cases.RUNNER_CONTEXT), "prompt": cases.RUNNER_PROMPT}
    if cfg_id == 3:
        img, ctx, prompt = cases.stripes_case(8, 512)
        return {"rgb": img, "context": ctx, "prompt": prompt}
    if cfg_id == 4:
        return {"rgb": cases.load_aurora_rgb(), "context": dict(cases.INPAINT_CONTEXT), "prompt": cases.AURORA_PROMPT,
                "mask": np.array(cases.load_moon_mask().convert("L")), "init": cases.synthetic_init_image(512, 81)}
    if cfg_id == 5:
        img, ctx, prompt = cases.grid_case(3, 4, 768, 768, seeds=True)
        return {"rgb": img, "context": ctx, "prompt": prompt}
    raise ValueError("unknown config %r" % cfg_id)


def build_tools(device, dtype, scheduler_name, model, tiny=False):
    """Stand-in UNet of the workload's topology built on rank 0 from seed 1234 and broadcast over RCCL (gloo on a CPU
    dry run); small stand-ins for text encoder / tokenizer / VAE (outside the hot path)."""
    import sd_standin as S
    from pww_hip import dist as pdist
    cfg = {"sd15": S.SD15_CONFIG, "sd15_inpaint": S.SD15_INPAINT_CONFIG, "sd21": S.SD21_CONFIG}[model]
    if tiny:      # dry run on a box without GPUs: same block structure at 1/8 width
        cfg = dict(S.TINY_SD2_CONFIG if model == "sd21" else S.TINY_CONFIG, in_channels=cfg["in_channels"])
    t0 = time.time()
    timing = {}
    unet, nbytes = pdist.build_and_broadcast(lambda: S.build_unet(cfg, seed=1234, dtype=dtype, device="cpu", qk_gain=2.0),
                                             lambda: S.UNet2DConditionModel(**cfg), device, dtype, src=0, timing=timing)
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize()
    text = S.TinyTextEncoder(cfg["cross_attention_dim"], seed=1235).to(device=device, dtype=dtype)
    vae = S.TinyVAE(4, seed=1236).to(device=device, dtype=dtype)
    sched = (S.PLMSScheduler() if scheduler_name == "plms" else
             S.LMSDiscreteScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000))
    info = {"build_and_broadcast_s": round(time.time() - t0, 2), "rank0_build_s": timing.get("build_s"), "weight_broadcast_s": timing.get("broadcast_s"),
            "broadcast_bytes": int(nbytes), "bucket_bytes": pdist.BROADCAST_BUCKET_BYTES}
    return (vae, unet, text, S.HashTokenizer(), sched), info


# ---- instrumentation ------------------------------------------------------------------------------------------------

class EventTimer:
    """HIP-event timing of pww kernel launches on the stream they are launched on (instrumented pass only).
    Attention launches: kernel-only (the dispatch stamps its own start / end timestamps into an event pair of the library,
    pww_profile_arm) -- exact for every launch of the pass, whatever the host does around it; each class is also replayed
    back to back from a hipGraph (warm caches, no host in the loop) as a second view. The streaming helpers (mask build,
    CFG combine) are bracketed by an ordinary event pair in situ, which includes host launch latency: an upper bound."""

    def __init__(self):
        self.pairs = {}
        self.sample = {}
        self.slots = {}       # kernel-only timing slots of the library (pww_profile_arm), per launch class

    def _timed(self, key, fn, a, kw, kernel_only=False):
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if kernel_only:       # the attention launch inside fn stamps its own start / end device timestamps into this slot
            from pww_hip import _lib
            slot = _lib.load().pww_profile_arm()
            if slot >= 0:
                self.slots.setdefault(key, []).append(slot)
        e0.record(s)
        out = fn(*a, **kw)
        e1.record(s)
        self.pairs.setdefault(key, []).append((e0, e1))
        self.sample.setdefault(key, (fn, a, kw))
        return out

    def kernel_us(self, key):
        """Mean kernel-only duration (us) of the launches of one class in the instrumented pass, or None."""
        import ctypes
        from pww_hip import _lib
        lib, ts = _lib.load(), []
        for slot in self.slots.get(key, []):
            us = ctypes.c_float(0.0)
            if lib.pww_profile_elapsed_us(slot, ctypes.byref(us)) == 0:
                ts.append(us.value)
        return sum(ts) / len(ts) if ts else None

    def wrap_attention(self, fn):
        def wrapped(q, k, v, heads, scale, bias=None, **kw):
            fused = kw.get("stat") is not None and kw["stat"][0] is None
            kind = ("cross+stat" if fused else "cross") if bias is not None else ("self" if k.shape[1] == q.shape[1] else "cross-nobias")
            key = (kind, q.shape[0], q.shape[1], k.shape[1], q.shape[2] // heads, heads, k.shape[0])
            return self._timed(key, fn, (q, k, v, heads, scale), dict(bias=bias, **kw), kernel_only=True)
        return wrapped

    def wrap_qproj(self, fn):
        """the to_q GEMM with the score statistic in its epilogue (pww_qproj_stat): kernel-only like the attention launches"""
        def wrapped(x, weight, k, heads, kind, gate=None):
            key = ("qproj+stat", x.shape[0], x.shape[1], k.shape[1], weight.shape[0] // heads, heads, x.shape[2])
            return self._timed(key, fn, (x, weight, k, heads, kind), dict(gate=gate), kernel_only=True)
        return wrapped

    def wrap_qkparts(self, fn):
        """the statistic's partials over a finished Q (pww_qk_parts: the layers whose to_q stays the stock GEMM): kernel-only"""
        def wrapped(q, k, heads, kind, gate=None, gated=0):
            key = ("qk_parts", q.shape[0], q.shape[1], k.shape[1], q.shape[2] // heads, heads, k.shape[0])
            return self._timed(key, fn, (q, k, heads, kind), dict(gate=gate, gated=gated), kernel_only=True)
        return wrapped

    def wrap_stats(self, fn):
        def wrapped(q, k, heads):
            key = ("qk_reduce", q.shape[0], q.shape[1], k.shape[1], q.shape[2] // heads, heads, k.shape[0])
            return self._timed(key, fn, (q, k, heads), {})
        return wrapped

    def wrap_stream(self, kind, fn, nbytes_of):
        """HBM-streaming helpers (mask build, CFG combine): key carries the algorithmic byte count."""
        def wrapped(*a, **kw):
            return self._timed((kind, int(nbytes_of(*a, **kw)), 0, 0, 0, 0, 0), fn, a, kw, kernel_only=True)
        return wrapped

    def _replay_us(self, key, reps=40):
        """Average duration of `reps` back-to-back launches of one captured call, replayed from a hipGraph (so the
        queue never runs dry: host launch latency, ~18 us per eager call, stays out of the number)."""
        fn, a, kw = self.sample[key]
        return replay_us(lambda: fn(*a, **kw), reps)

    def table(self, elem_bytes):
        """Per launch class: launches in the pass, back-to-back replay duration, algorithmic FLOPs / bytes
         and the fraction of the bounding roofline."""
        torch.cuda.synchronize()
        rows = []
        for key, pairs in self.pairs.items():
            kind, B, N, M, D, Hh, Bk = key
            if kind in ("mask_build", "cfg_combine"):   # streaming helpers: kernel-only like the attention launches (round 5); the event bracket around the
                us_bracket = sum(a.elapsed_time(b) for a, b in pairs) * 1e3 / len(pairs)      # eager call (host launch latency included) beside it
                k_us = self.kernel_us(key)
                us = k_us if k_us is not None else us_bracket
                rows.append({"kernel": kind + (" (kernel-only, in the workload)" if k_us is not None else " (event bracket around the eager call)"), "launches": len(pairs),
                             "avg_us": round(us, 2), "avg_us_event_bracket_eager": round(us_bracket, 2), "algorithmic_bytes": B,
                             "gbs": round(B / us / 1e3, 1), "bound": "hbm", "frac": round(B / us / 1e3 / HBM_PEAK_GBS, 4)})
                continue
            us = self._replay_us(key)
            k_us = self.kernel_us(key)      # the launches of the workload pass themselves, kernel-only (what rocprofv3 reports for them)
            row = kernel_row(kind, k_us if k_us is not None else us, B, N, M, D, Hh, Bk, elem_bytes, len(pairs))
            row["avg_us_back_to_back"] = round(us, 2)     # 40-launch hipGraph replay, event interval / 40 (warm caches, incl. dispatch gaps)
            rows.append(row)
        rows.sort(key=lambda r: -r["avg_us"] * r["launches"])
        return rows   # (mask_build = the four per-resolution launches of one request)

    def mean_us(self, pred):
        torch.cuda.synchronize()
        sel = [(k, v) for k, v in self.pairs.items() if pred(k)]
        ts = [a.elapsed_time(b) * 1e3 for _, v in sel for a, b in v]
        return (sum(ts) / len(ts), len(ts), sel[0][0][1]) if ts else (None, 0, 0)


def replay_us(call, reps=40):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            call()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def kernel_row(kind, us, B, N, M, D, Hh, Bk, elem_bytes, launches, label=None):
    C = Hh * D
    names = {"qk_reduce": "qk_reduce (ticket init + reduce)", "cross+stat": "cross (attention with the statistic: folded partials, or formed in the launch)",
             "qproj+stat": "to_q GEMM + score-statistic partials (pww_qproj_stat)", "qk_parts": "score-statistic partials over the finished Q (pww_qk_parts)"}
    if kind == "qproj+stat":      # (Bk carries Cin here) algorithmic: the GEMM alone -- the score blocks of the epilogue are not counted
        flops = 2.0 * B * N * C * Bk
        nbytes = elem_bytes * (B * N * Bk + B * N * C + C * Bk)
    elif kind in ("qk_reduce", "qk_parts"):
        flops = 2.0 * B * Hh * N * M * D
        nbytes = elem_bytes * (B * N * C + Bk * M * C)
    else:
        flops = 4.0 * B * Hh * N * M * D
        nbytes = elem_bytes * (2 * B * N * C + 2 * Bk * M * C) + (N * M * 4 if kind.startswith("cross") and kind != "cross-nobias" else 0)
    tf, gbs = flops / us / 1e6, nbytes / us / 1e3
    bound = "mfma" if flops / nbytes > MFMA_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS else "hbm"
    return {"kernel": label or names.get(kind, kind), "B": B, "N": N, "M": M, "D": D, "launches": launches, "avg_us": round(us, 2),
            "tflops": round(tf, 1), "gbs": round(gbs, 1), "bound": bound,
            "frac": round(tf / MFMA_PEAK_TFLOPS if bound == "mfma" else gbs / HBM_PEAK_GBS, 4)}


def hot_logit_row(device, dtype, B, N, D, heads, std=4.0):
    """The dominant shape on HOT logits (scaled-logit std `std`, row maxima of 3.5 std and more in natural units: what trained SD layers
    produce, unlike the random-init UNet whose scaled logits stay below 1): this row prices what the folded-reference kernel does about them
    -- bf16: nothing (8 exponent bits); fp16: a workgroup whose first key stage shows a hot row leaves the range-free mode and follows the
    running maximum lazily, rows past the magnitude guard take the exact path. `paths` = workgroups of ONE launch per path (the kernel's
    debug counters, pww_debug_path_counts): range-free fast path / lazy reference / lazy reference with the exact scale / exact recomputation."""
    import ctypes
    from pww_hip import ops, _lib
    g = torch.Generator(device="cpu").manual_seed(7)
    gain = math.sqrt(std)        # q, k ~ N(0, gain^2): q.k / sqrt(D) has std gain^2
    q = (torch.randn(B, N, heads * D, generator=g) * gain).to(device=device, dtype=dtype)
    k = (torch.randn(B, N, heads * D, generator=g) * gain).to(device=device, dtype=dtype)
    v = torch.randn(B, N, heads * D, generator=g).to(device=device, dtype=dtype)
    us = replay_us(lambda: ops.attention(q, k, v, heads, D ** -0.5))
    row = kernel_row("self", us, B, N, N, D, heads, B, 2, 0, label="self (hot logits: scaled-logit std %g, synthetic q/k)" % std)
    lib = _lib.load()
    counts = torch.zeros(4, dtype=torch.int32, device=device)
    lib.pww_debug_path_counts(ctypes.c_void_p(counts.data_ptr()))
    try:
        ops.attention(q, k, v, heads, D ** -0.5)
        torch.cuda.synchronize()
    finally:
        lib.pww_debug_path_counts(None)
    c = counts.tolist()
    row["paths"] = {"fast": c[0], "lazy": c[1], "lazy_exact_scale": c[3], "exact": c[2]}
    return row


# gfx950: FETCH_SIZE = TCC_EA0_RDREQ x 64 bytes, but a read request moves a whole 128-byte line (MI355X_MICROARCH.md, HBM section: "double
# it before comparing with a byte count" for 16-byte-per-lane loads -- every load of these kernels; re-measured on this access pattern in
# round 6: tools/ubench_linefill.cpp, profiles/r06_linefill.txt -- a 32-, 64- or 80-byte touch of a line is ONE request and costs the time
# of 128 bytes). WRITE_SIZE is exact (32- and 64-byte requests, counted as such). Rounds 2 - 5 used 1.0 (calibrated on a kernel whose rows
# were sliced the same way: the calibration could not see it) and under-reported the read side by half.
FETCH_LINE_FACTOR = 2.0
TRAFFIC_CORRECTION = ("hbm bytes = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE tallies a read request at 64 bytes and a request moves a whole "
                      "128-byte line (MI355X_MICROARCH.md HBM section; profiles/r06_linefill.txt); rounds 2 - 5 reported FETCH_SIZE + WRITE_SIZE")


def measured_traffic(n_tok, d, b_rows, dtype, live=False):
    """HBM bytes per launch of the dominant kernel (2 x FETCH_SIZE + WRITE_SIZE: FETCH_LINE_FACTOR above; separate rocprofv3 --pmc passes as
    MI355X_MICROARCH.md prescribes). Default: the committed PMC pass of the SHIPPED kernel and dtype for this shape
    (profiles/r06_traffic.json: this round's passes, corrected); `live` = run the two passes now over the same launch through the native
    harness. None if the shape is not covered."""
    if live:
        return live_traffic(n_tok, d, b_rows, dtype)
    for name in ("r06_traffic.json",):
        path = os.path.join(REPO, "profiles", name)
        if not os.path.isfile(path):
            continue
        recs = json.load(open(path))
        recs = recs.get("records", [recs]) if isinstance(recs, dict) else recs
        for rec in recs:
            if (rec.get("N"), rec.get("D"), rec.get("B"), rec.get("dtype")) == (n_tok, d, b_rows, dtype) and rec.get("kind", "self") == "self":
                return int(rec["hbm_bytes_per_launch"])
    return None


def measured_mfma_busy(n_tok, d, b_rows, dtype):
    """Matrix-pipe busy fraction of the dominant kernel from the committed PMC passes (profiles/r04_pmc.json: SQ_VALU_MFMA_BUSY_CYCLES over
    kernel cycles x 1024 SIMDs, the shipped kernel on this shape), or None."""
    path = os.path.join(REPO, "profiles", "r04_pmc.json")
    if os.path.isfile(path):
        for rec in json.load(open(path)).get("records", []):
            if (rec.get("N"), rec.get("D"), rec.get("B"), rec.get("dtype")) == (n_tok, d, b_rows, dtype) and rec.get("kind") == "self":
                return rec.get("mfma_busy")
    return None


# Issue-bound ceiling of the d = 40 / d = 64 self-attention loops: per 64-key tile and wave the loop issues 448
# matrix-pipe cycles (14 MFMAs x 32) and ~500 VALU issue cycles (32 v_exp_f32 at 8.5 cycles, 16 v_cvt_pk, the fma / max chains:
# profiles/r01_valu_ubench.md), and on one SIMD the two ADD UP (profiles/r02_attn_pingpong.md: t(no exp) + t(no MFMA) = t(full)):
# the matrix pipe can be busy at most 448 / (448 + 500) = 0.47 of the time, of which 40 / 64 (d = 40 padded to 48 + 64 MFMA rows: 0.71
# averaged over the score and PV MFMAs) is algorithmic work: 0.47 x 0.71 = 0.34 of the dense peak at the nominal clock. d = 64 has no
# padding: 0.47.
ATTAINABLE = {40: 0.34, 64: 0.47}


HARNESS_CASES = {(4096, 40, 2, "bf16"): "sd15_self_n4096_d40_bf16_b2", (4096, 40, 2, "fp16"): "sd15_self_n4096_d40_f16_b2",
                 (4096, 40, 16, "fp16"): "sd15_self_n4096_d40_f16_b16", (4096, 40, 16, "bf16"): "sd15_self_n4096_d40_bf16_b16",
                 (9216, 64, 4, "bf16"): "sd21_self_n9216_d64_b4", (9216, 64, 8, "bf16"): "sd21_self_n9216_d64_b8"}


CROSS16_COUNTER_CASE = ("qproj_sd15_n4096_b16", "cross N=4096 d=40, 16 rows (8 gated in): pww_qproj_stat + pww_cross_attn_fwd_parts (the batched C = 320 layers' route)")
CROSS_COUNTER_CASE = ("qproj_sd15_n256_b2", "cross N=256 d=160, 2 rows: pww_qk_parts + pww_cross_attn_fwd_parts (the C = 1280 layers' route)")


def live_counters(case, kernel_keys, flags=()):
    """rocprofv3 --pmc passes over tests/native/attn_check --only <case> (separate passes with --kernel-trace only, as MI355X_MICROARCH.md
    prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass; SQ_VALU_MFMA_BUSY_CYCLES with GRBM_GUI_ACTIVE for the kernel's cycles): per
    kernel whose name contains one of `kernel_keys`: {"hbm_bytes": (2 x FETCH_SIZE + WRITE_SIZE) KiB -> bytes per dispatch (FETCH_LINE_FACTOR), "mfma_busy":
    SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), "kernel_cycles", "dispatches"}. None when rocprofv3 or the harness
    is missing, {} when a pass produced nothing."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = os.path.join(REPO, "tests", "native", "attn_check")
    if not os.path.isfile(exe) or shutil.which("rocprofv3") is None:
        return None
    acc = {}
    for counters in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"):
        out = tempfile.mkdtemp(prefix="pww_pmc_")
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            subprocess.run(["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--", exe] + list(flags) + ["--only", case],
                           cwd="/tmp", env=env, capture_output=True, timeout=300)
            for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    for key in kernel_keys:
                        if key in row["Kernel_Name"]:
                            acc.setdefault(key, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        except (subprocess.TimeoutExpired, OSError):
            pass
        finally:
            shutil.rmtree(out, ignore_errors=True)
    res = {}
    for key, c in acc.items():
        m = {k: sum(v) / len(v) for k, v in c.items()}
        r = {"dispatches": max(len(v) for v in c.values())}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            r["hbm_bytes"] = int((FETCH_LINE_FACTOR * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0)
            r["fetch_size_kib_raw"], r["write_size_kib"] = round(m["FETCH_SIZE"], 1), round(m["WRITE_SIZE"], 1)
        if m.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            r["kernel_cycles"] = int(m["GRBM_GUI_ACTIVE"] / 8.0)
            r["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
        res[key] = r
    return res


def live_traffic(n_tok, d, b_rows, dtype):
    """rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each; they do not fit one pass) over tests/native/attn_check --only <case>:
    mean per dispatch of the attention kernel, KiB -> bytes, the read side x FETCH_LINE_FACTOR. None if rocprofv3 or a harness case for the shape is missing."""
    import csv
    import glob
    import shutil
    import tempfile
    case = HARNESS_CASES.get((n_tok, d, b_rows, dtype))
    exe = os.path.join(REPO, "tests", "native", "attn_check")
    if case is None or not os.path.isfile(exe) or shutil.which("rocprofv3") is None:
        return None
    total = 0.0
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="pww_pmc_")
        env = dict(os.environ, TMPDIR="/tmp")
        subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--", exe, "--only", case],
                       cwd="/tmp", env=env, capture_output=True, timeout=600)
        vals = []
        for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "attn_fwd" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    vals.append(float(row["Counter_Value"]))
        shutil.rmtree(out, ignore_errors=True)
        if not vals:
            return None
        total += sum(vals) / len(vals) * 1024.0 * (FETCH_LINE_FACTOR if counter == "FETCH_SIZE" else 1.0)
    return int(total)


# ---- baselines ------------------------------------------------------------------------------------------------------

def reference_ops_same_gpu(cfg, tools, request, device, dtype, guidance):
    """The reference's op sequence as plain torch ops (materialised scores, half matmuls, fp32 softmax: what its
    inj_forward does under autocast on a GPU) with its call pattern (eager, two batch-1 UNet calls per step) on
    THIS GPU -- separates the fused-kernel / folding / graph gain from the CPU->GPU gain. One image, timed."""
    from pww_hip.conditioning import _encode_text_color_inputs
    from pww_hip.sampler import PwWSampler, initial_latents
    import pww_hip.sampler as S
    from gpu_util import install_unfused, uninstall_all
    vae, unet, text, tok, sched = tools
    rgb, context, prompt = request["rgb"], request["context"], request["prompt"]
    wf = weight_functions()[cfg["wf"]]
    orig_install = S.install
    S.install = install_unfused
    try:
        sampler = PwWSampler(unet, sched, "eager")
        times = []
        for it in range(2):     # first pass warms the unfused kernels up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, _, cond, uncond = _encode_text_color_inputs(text, tok, device, rgb, dict(context), prompt, "", dtype=dtype)
            sched.set_timesteps(cfg["denoise_steps"])
            lat = initial_latents(0, unet.in_channels, rgb.shape[0], rgb.shape[1], batch_seeds=[0]).to(device) * sched.init_noise_sigma
            sampler.sample(cond, uncond, lat, sched.timesteps, guidance, wf)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
    finally:
        S.install = orig_install
        uninstall_all()
        import pww_hip
        pww_hip.install(unet)
    return {"value": round(1.0 / times[-1], 4), "unit": "images/s", "kind": "unfused torch ops, eager, 2 batch-1 UNet calls/step (the reference's GPU path)",
            "s_per_image": round(times[-1], 3)}


def cpu_baseline(args, cfg, request):
    """The oracle (CPU port of the reference path, fp32) on the host cores: a bounded sample of the same workload --
    `--cpu-steps` denoise steps (2 UNet forwards each) of the full-size loop at the BEST torch thread count of a sweep
    (median of three conditional UNet forwards per candidate: 8 / 16 / 32 / 64 / 128 threads, capped by the box)."""
    from oracle import pww_oracle as O
    import pww_cases as cases
    log("cpu baseline: building fp32 UNet")
    vae, unet, text, tok, sch = cases.build_tools(cfg["model"], dtype=torch.float32, device="cpu", scheduler="lms", qk_gain=2.0)
    wf = weight_functions()[cfg["wf"]]
    n_denoise_steps = cfg["denoise_steps"]
    ncpu = os.cpu_count() or 1
    threads0 = torch.get_num_threads()
    O.install_oracle_attention(unet)
    sweep = {}
    try:
        seeds, regions, cond, uncond = O.encode_text_color_inputs(text, tok, request["rgb"], dict(request["context"]), request["prompt"], "")
        latents = O.initial_latents(0, 4, request["rgb"].shape[0], request["rgb"].shape[1])
        sch.set_timesteps(n_denoise_steps)
        latents = latents * sch.init_noise_sigma
        t, sigma = sch.timesteps[0], sch.sigmas[0]
        x = sch.scale_model_input(latents, t)
        cond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": wf})
        cands = sorted({n for n in (8, 16, 32, 64, 128) if n <= ncpu} | ({ncpu} if ncpu < 8 else set()))
        torch.set_num_threads(cands[len(cands) // 2])
        unet(x, t, encoder_hidden_states=cond)                       # one-time costs (primitive creation, page faults) stay out of the sweep
        for n in cands:
            torch.set_num_threads(n)
            ts = []
            for _ in range(3):        # three forwards per candidate, the median counts (one forward moved the pick -- and the figure by 20 % -- from run to run)
                t0 = time.perf_counter()
                unet(x, t, encoder_hidden_states=cond)
                ts.append(time.perf_counter() - t0)
            sweep[n] = round(sorted(ts)[1], 3)
            log("cpu baseline sweep: %d threads %.2f s per conditional UNet forward (median of %s)" % (n, sweep[n], ["%.2f" % v for v in ts]))
        best = min(sweep, key=sweep.get)
        torch.set_num_threads(best)
        times = []
        for i, t in enumerate(sch.timesteps[: args.cpu_steps]):
            t0 = time.perf_counter()
            sigma = sch.sigmas[i]
            x = sch.scale_model_input(latents, t)
            cond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": wf})
            ec = unet(x, t, encoder_hidden_states=cond).sample
            uncond.update({"SIGMA": sigma, "WEIGHT_FUNCTION": lambda w, sigma, qk: 0.0})
            eu = unet(x, t, encoder_hidden_states=uncond).sample
            latents = sch.step(O.cfg_combine(ec, eu, 7.5), t, latents).prev_sample
            times.append(time.perf_counter() - t0)
            log("cpu baseline step", i, "%.2f s at %d threads" % (times[-1], best))
    finally:
        torch.set_num_threads(threads0)
        from sd_standin import CrossAttention
        if "__call__" in CrossAttention.__dict__:
            del CrossAttention.__call__
    per_step = float(np.mean(times))
    unet_evals = n_denoise_steps + (1 if cfg["scheduler"] == "plms" else 0)
    out = {"value": round(1.0 / (per_step * unet_evals), 6), "unit": "images/s", "cores": best,
           "kind": "port", "host": "gpu_box_port",       # measured NOW on this box's host cores (the other leg below is a different machine)
           "sample": "%d of %d denoise steps (2 fp32 UNet forwards each, oracle attention) of the same %dx%d workload at the best of the swept "
                     "torch thread counts (%d of %d logical CPUs), %.2f s/step, extrapolated to %d steps"
                     % (len(times), unet_evals, cfg["size"], cfg["size"], best, ncpu, per_step, unet_evals),
           "thread_sweep_s_per_unet_forward": {str(k): v for k, v in sweep.items()}, "logical_cpus": ncpu}
    ref_path = os.path.join(REPO, "tests", "golden", "ref_cpu_timing.json")
    if os.path.isfile(ref_path):     # the AST-loaded, unmodified reference timed on the BUILD box (oracle/make_golden.py reftime):
        rec = json.load(open(ref_path))   # /root/reference does not exist on the GPU box, so this figure travels as a fixture
        out["reference_on_build_box"] = {
            "value": rec["images_per_s_30_steps"], "unit": "images/s", "cores": rec["threads"], "kind": "reference",
            "host": "build_box_reference",     # NOT this machine: a fixture recorded where /root/reference exists
            "sample": rec["what"] + ": %.1f s wall, %.2f s per UNet forward, %.1f s inside the reference's inj_forward; extrapolated to 30 steps"
                      % (rec["wall_s"], rec["s_per_unet_forward"], rec["s_inside_inj_forward"])}
    return out


# ---- launcher -------------------------------------------------------------------------------------------------------

def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def respawn_under_launcher(n):
    """`python bench.py --gpus N` outside a launcher: run this very command as N ranks of ONE node."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PWW_BENCH_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("no launcher in the environment: spawning %d ranks:" % n, " ".join(cmd))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (one step = one batch of images per GPU); default 4 (config 2) or 1")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.md section 2 workload number")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default: the workload's)")
    ap.add_argument("--global-batch", type=int, default=None, help="images per step over ALL ranks (default: --batch x ranks); need not divide by the rank count: "
                    "the contiguous split gives the first (global batch mod ranks) ranks one image more, a rank without images idles")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16"])
    ap.add_argument("--mode", default="graph", choices=["eager", "folded", "graph"])
    ap.add_argument("--scheduler", default=None, choices=["plms", "lms"])
    ap.add_argument("--denoise-steps", type=int, default=None)
    ap.add_argument("--guidance", type=float, default=7.5)
    ap.add_argument("--cpu-steps", type=int, default=1, help="denoise steps timed for the CPU baseline after the thread sweep (0 = skip)")
    ap.add_argument("--live-traffic", action="store_true", help="measure roofline.traffic NOW: two extra rocprofv3 --pmc passes (FETCH_SIZE, "
                    "WRITE_SIZE) over the dominant launch through tests/native/attn_check (default: the committed PMC pass of this kernel under profiles/)")
    ap.add_argument("--no-live-counters", action="store_true", help="skip the rocprofv3 --pmc passes (HBM-side bytes and matrix-pipe busy of the dominant launch and of "
                    "the C = 1280 cross-attention route, ~1 min through tests/native/attn_check); the line then carries the labelled constants of the committed pass")
    ap.add_argument("--no-roofline-pass", action="store_true")
    ap.add_argument("--no-reference-ops", action="store_true", help="skip the unfused-torch-ops-on-this-GPU pass")
    ap.add_argument("--memory-format", default="auto", choices=["auto", "nchw", "channels_last"],
                    help="memory format of the UNet (a stock PyTorch setting: MIOpen's bf16/fp16 convolutions are NHWC kernels, NCHW pays a "
                         "transpose either side). auto = channels_last for the SD1.5 topologies (measured +4.6 %% at batch 1, +1..3 %% at "
                         "batch 8), NCHW for SD2.1 at 768x768 (channels_last measured -6 %% there)")
    ap.add_argument("--no-fused-norm", action="store_true", help="A/B: the UNet blocks' GroupNorm (+ addend, + SiLU) as stock PyTorch ops instead of pww_group_norm_fwd")
    ap.add_argument("--tiny", action="store_true", help="1/8-width stand-in of the workload's topology (tests of the launcher / sharding plumbing on a GPU; the line says so)")
    ap.add_argument("--parity-bar", type=float, default=None, help="rel-L2 bar of the final-latent parity check (default: BASELINE.md section 4's 5e-2 for bf16, "
                    "1e-2 for fp16); the run fails above it")
    ap.add_argument("--dump-latents", default=None, metavar="PREFIX", help="save this rank's final latents of the last timed step to PREFIX_rank<r>.npy")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: spawn / rendezvous (gloo) / broadcast a 1/8-width model and the request, print the line with value null")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_launcher(args.gpus))
    # stdout carries the ONE JSON line and nothing else: whatever the path prints on its way (the reference's own "Use region based
    # seeding" line of the region-seeded configs, paint_with_words.py:449) goes to stderr with the bench log
    # -- at the file-descriptor level: gloo and RCCL print their banners from C++ straight to fd 1
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    sys.stdout = sys.stderr

    cfg = dict(CONFIGS[args.config])
    for k_arg, k_cfg in (("batch", "batch"), ("dtype", "dtype"), ("scheduler", "scheduler"), ("denoise_steps", "denoise_steps")):
        if getattr(args, k_arg) is not None:
            cfg[k_cfg] = getattr(args, k_arg)
    if args.steps is None:
        args.steps = 4 if args.config == 2 else 1

    from pww_hip import dist as pdist, ops
    import pww_hip
    import importlib
    pw_api = importlib.import_module("paint_with_words.paint_with_words")   # (the package re-exports a FUNCTION of the same name)
    from paint_with_words import paint_with_words_batch, paint_with_words_inpaint_batch
    from PIL import Image

    if os.environ.get("NCCL_DEBUG", "").upper() in ("VERSION", ""):
        os.environ["NCCL_DEBUG"] = "WARN"     # the pool exports NCCL_DEBUG=VERSION: RCCL would print a banner on stdout
    on_gpu = torch.cuda.is_available() and not args.dry_run
    if not on_gpu and not args.dry_run:
        raise SystemExit("bench.py needs a HIP device (torch.cuda.is_available() is False); `--dry-run` exercises the launcher / "
                         "rendezvous / broadcast path on a box without GPUs")
    rank, world, local = pdist.init_from_env("cuda" if on_gpu else "cpu")
    assert world == args.gpus, "launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    device = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    dtype = (torch.bfloat16 if cfg["dtype"] == "bf16" else torch.float16) if on_gpu else torch.float32
    if on_gpu:
        torch.cuda.set_device(local)
        pww_hip.enable_miopen_find()      # MIOpen find mode for the UNet's stock convolutions (PWW_MIOPEN_FIND=0: PyTorch's default)
        pww_hip.load_library()
    pw_api.DEFAULT_MODE = args.mode
    if args.no_fused_norm:
        pww_hip.blocks.FUSED_NORM = False

    log("cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads(), "rank", rank, "world", world, "config", args.config)
    tools, build_info = build_tools(device, dtype, cfg["scheduler"], cfg["model"], tiny=not on_gpu or args.tiny)
    log("tools built", build_info)
    vae, unet, text, tok, sched = tools
    channels_last = args.memory_format == "channels_last" or (args.memory_format == "auto" and cfg["model"] != "sd21")
    if channels_last:
        unet.to(memory_format=torch.channels_last)

    # request: rank 0 owns the color map (and, for inpainting, the mask and the init image); every rank builds its own
    # weight maps with the HIP mask kernel
    t0 = time.time()
    request = pdist.broadcast_request(make_request(args.config) if rank == 0 else None, device, src=0)
    req_bcast_s = time.time() - t0
    wf = weight_functions()[cfg["wf"]]
    color_map = Image.fromarray(request["rgb"])
    n_global = args.global_batch if args.global_batch is not None else cfg["batch"] * world
    if n_global < 1:
        raise SystemExit("--global-batch must be at least 1")
    base_seed = 81 if cfg["kind"] == "inpaint" else 0
    shards = [pdist.shard_range(n_global, r, world) for r in range(world)]
    my_lo, my_hi = shards[rank]

    def one_step(step_idx):
        """mask build + conditioning + full denoise loop for this rank's images of global step `step_idx`, through the
        public batched entry points."""
        seeds = pdist.image_seeds(base_seed + step_idx * n_global, n_global, rank, world)
        if not seeds:             # more ranks than images (--global-batch): this rank idles through the step
            return None
        common = dict(num_inference_steps=cfg["denoise_steps"], guidance_scale=args.guidance, device=str(device), weight_function=wf,
                      preloaded_utils=tools, return_latents=True)
        if cfg["kind"] == "inpaint":
            return paint_with_words_inpaint_batch(dict(request["context"]), color_map, Image.fromarray(request["mask"]),
                                                  Image.fromarray(request["init"]), request["prompt"], seeds, strength=1.0, **common)
        return paint_with_words_batch(dict(request["context"]), color_map, request["prompt"], seeds, **common)

    n_unet_evals = cfg["denoise_steps"] + (1 if cfg["scheduler"] == "plms" else 0)
    result = {
        "metric": "512x512 images/sec (30 steps, CFG) SD1.5+PwW", "value": None, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
        "config": {"workload": "%s, %d %s steps (%d UNet evaluations x {cond,uncond}), CFG %.1f, weight function '%s', batch %d per GPU, "
                               "final latent (VAE decode excluded)" % (cfg["name"], cfg["denoise_steps"], cfg["scheduler"].upper(), n_unet_evals,
                                                                        args.guidance, cfg["wf"], my_hi - my_lo),
                   "baseline_config": args.config, "mode": args.mode, "images_per_step": n_global, "shards": [list(sh) for sh in shards],
                   "stock_op_settings": "MIOpen find mode%s" % (", UNet in channels_last memory format" if channels_last else ""),
                   "parallelism": "image-sharded x%d, no data-path collective" % world, "backend": pdist.backend_name(),
                   "block_norms": "pww_group_norm_fwd (GroupNorm + time-embedding addend + SiLU of the ResnetBlock2D / Transformer2DModel blocks)" if pww_hip.blocks.FUSED_NORM else "stock PyTorch ops",
                   "weight_broadcast": build_info, "weight_broadcast_s": build_info.get("weight_broadcast_s"), "request_broadcast_s": round(req_bcast_s, 4)},
    }
    if args.config != 2 or cfg["denoise_steps"] != 30:     # (an overridden step count must not carry the headline's "30 steps" label)
        result["metric"] = "%dx%d images/sec (%d steps, CFG) %s+PwW" % (cfg["size"], cfg["size"], cfg["denoise_steps"],
                                                                       {"sd15": "SD1.5", "sd15_inpaint": "SD1.5-inpainting", "sd21": "SD2.1"}[cfg["model"]])

    if args.dry_run:
        pdist.barrier(device)
        result.update({"dry_run": True, "dtype": "fp32", "data": "synthetic (1/8-width model, nothing timed)"})
        # what every rank would generate in global step 0, gathered from the ranks themselves: image counts and first seeds of a split that need
        # not be even (--global-batch 8 over 3 ranks: 3 / 3 / 2)
        mine = pdist.image_seeds(base_seed, n_global, rank, world)
        result["config"]["per_rank_images"] = [int(v) for v in pdist.all_ranks(len(mine), device)]
        result["config"]["per_rank_first_seed"] = [int(v) for v in pdist.all_ranks(mine[0] if mine else -1, device)]
        if rank == 0:
            print(json.dumps(result), file=json_out, flush=True)
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    def fail_together(message):
        """Raise on EVERY rank if any rank has a message (a rank that raises alone leaves the others blocked in the next collective)."""
        bad = pdist.max_over_ranks(1.0 if message else 0.0, device)
        if bad:
            raise SystemExit("bench.py: " + (message or "another rank failed its check (see its log)"))

    warm_s = []
    lat0 = None        # this rank's final latents of GLOBAL step 0: what the parity fixtures hold
    pww_hip.blocks.reset_stats()
    # N > 1: rank 0 warms up FIRST (MIOpen's find search for every convolution shape of the UNet: ~25 s), then every other rank copies rank 0's
    # MIOpen user db into its own (one db directory per local rank, pww_hip.dist.init_from_env) and warms up against the recorded answers
    # instead of repeating the search N times side by side. One node: the ranks share a file system.
    staged = world > 1 and args.warmup > 0 and on_gpu and os.environ.get("PWW_MIOPEN_FIND", "1") != "0"
    db_files = 0
    if on_gpu and os.environ.get("PWW_FIND_PREWARM", "1") == "1" and os.environ.get("PWW_MIOPEN_FIND", "1") != "0" and args.warmup > 0 and not args.tiny:
        # One eager pass in MIOpen's immediate mode BEFORE the warm-up pass that carries the find search: clocks, allocator and code caches are
        # warm when the solvers are timed. The search times every candidate once, and with a cold GPU its choices moved the batched configs from
        # process to process on ONE box (config 4: 10.5 / 8.7 / 10.4 / 9.1 / 8.7 images/s; with the pre-pass 10.0 / 10.0 / 9.6 / 9.8; config 5:
        # 2.88 / 2.31 / 2.99 against 2.79 / 3.08 / 2.88; configs 2 and 3: no difference -- profiles/r06_miopen_find_prepass.txt). Untimed, 0.3 - 1.4 s;
        # PWW_FIND_PREWARM=0 skips it.
        prev_bm, prev_mode = torch.backends.cudnn.benchmark, pw_api.DEFAULT_MODE
        torch.backends.cudnn.benchmark, pw_api.DEFAULT_MODE = False, "eager"
        try:
            one_step(0)
            torch.cuda.synchronize()
        finally:
            torch.backends.cudnn.benchmark, pw_api.DEFAULT_MODE = prev_bm, prev_mode
        log("immediate-mode pre-pass done")
    for phase in ((0, 1) if staged else (None,)):
        mine = phase is None or (phase == 0) == (rank == 0)
        if mine:
            if phase == 1:
                db_files = pdist.adopt_miopen_db(0)
            for w in range(args.warmup):
                t0 = time.perf_counter()
                out = one_step(w)
                torch.cuda.synchronize()
                warm_s.append(round(time.perf_counter() - t0, 2))
                log("warmup step", w, "done in %.2f s" % warm_s[-1])
                if w == 0:
                    lat0 = out
        if phase is not None:
            pdist.barrier(device)
    # every rank's warm-up seconds (the first holds MIOpen's solver search -- or, on ranks > 0, the look-ups in rank 0's db -- and the ONE
    # hipGraph capture of the geometry)
    result["config"]["warmup_s"] = warm_s
    if world > 1:
        result["config"]["warmup_s_per_rank"] = [[round(v, 2) for v in row] for row in zip(*[pdist.all_ranks(v, device) for v in warm_s])] if warm_s else []
        result["config"]["miopen_db"] = {"staged_warmup": bool(staged), "files_adopted_from_rank0": [int(v) for v in pdist.all_ranks(db_files, device)],
                                         "user_db_path": os.environ.get("MIOPEN_USER_DB_PATH")}
    # what the block plug did with the calls of the warm-up passes (in graph mode: the discovery pass and the capture the timed steps replay).
    # A call the kernels declined ran the stock op: the line must say so, and the default configuration must not have any.
    bst = pww_hip.blocks.stats()
    result["config"]["block_norms_calls"] = bst
    verdict = None
    if pww_hip.blocks.FUSED_NORM and args.warmup > 0 and not args.tiny and my_hi > my_lo:
        if bst["hit_rate"] is None:
            verdict = "the block plug saw no calls in the warm-up (rank %d): is it installed?" % rank
        elif bst["hit_rate"] != 1.0:
            verdict = "the block plug handed calls to the stock ops (rank %d): %s" % (rank, bst)
    fail_together(verdict)
    if args.mode == "graph":
        smp = getattr(unet, "_pww_samplers", {}).get((id(sched), "graph"))
        if smp is not None and smp._graphed is not None:
            result["config"]["hipgraph_captures"] = smp._graphed.captures
    pdist.barrier(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mono0 = time.monotonic_ns()
    lat = None
    for s in range(args.steps):
        lat = one_step(args.warmup + s)
        if args.warmup == 0 and s == 0:
            lat0 = lat
    pdist.barrier(device)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    elapsed = pdist.max_over_ranks(elapsed_local, device)
    per_rank_s = pdist.all_ranks(elapsed_local, device)
    log("timed region CLOCK_MONOTONIC ns %d %d" % (mono0, time.monotonic_ns()))      # (tools/rocpd_stats.py --window: the kernels of the timed steps only)
    for smp in getattr(unet, "_pww_samplers", {}).values():
        smp.check_errors()            # fused hand-off time-outs of any timed request (raises; the requests are complete: synchronised above)
    verdict = None if lat is None or bool(torch.isfinite(lat).all()) else "non-finite latents on rank %d" % rank
    # final-latent parity of THIS run's configuration against the committed fixture of the workload (rank 0 holds global images 0..: the fixture's)
    par = parity_check(args, cfg, lat0, my_lo) if rank == 0 else None
    if par is not None:
        result["parity"] = par
        log("parity of the timed configuration: rel-L2 %.3e (bar %.0e) vs %s" % (par["rel_l2"], par["bar"], par["fixture"]))
        if not par["ok"] and verdict is None:
            verdict = "final latents of global step 0 are %.3e (rel-L2) away from %s: above the %.0e bar" % (par["rel_l2"], par["fixture"], par["bar"])
    elif rank == 0:
        result["parity"] = None       # (not the fixture's workload: overridden dtype / scheduler / steps / guidance, or --tiny)
    fail_together(verdict)
    if args.dump_latents and lat is not None:
        np.save("%s_rank%d.npy" % (args.dump_latents, rank), lat.float().cpu().numpy())
    if args.tiny:
        result["data"] = "synthetic (1/8-width model: plumbing test, not a measurement)"
    log("timed region done: %.3f s for %d steps" % (elapsed, args.steps))
    images = args.steps * n_global
    result["value"] = round(images / elapsed, 4)
    result["ms_per_step"] = round(elapsed / args.steps * 1e3, 2)
    # per-rank view of the same timed region (the line's value is the job's: total images / the slowest rank's time): a rank that lags --
    # a slower box slot, a MIOpen search that did not finish in the warm-up -- shows here instead of hiding behind the max
    lo_hi = shards
    result["config"]["per_rank"] = {"seconds": [round(t, 4) for t in per_rank_s],
                                    "images_per_s": [round(args.steps * (hi - lo) / t, 4) for (lo, hi), t in zip(lo_hi, per_rank_s)]}

    if rank == 0 and not args.no_roofline_pass:
        # instrumented pass (same workload, folded mode so single launches can be bracketed by HIP events
        # on the launch stream): dominant kernel = self-attention at the finest resolution
        timer = EventTimer()
        orig, orig_stats, orig_mask, orig_cfg, orig_qproj, orig_qkparts = ops.attention, ops.qk_stats, ops.mask_build, ops.cfg_combine, ops.qproj_stat, ops.qk_parts
        ops.attention, ops.qk_stats, ops.qproj_stat = timer.wrap_attention(orig), timer.wrap_stats(orig_stats), timer.wrap_qproj(orig_qproj)
        ops.qk_parts = timer.wrap_qkparts(orig_qkparts)
        # K4: reads the RGB map once per resolution, writes the [N_r, 77] fp32 maps; CFG combine: 2 half reads + 1 fp32 write
        ops.mask_build = timer.wrap_stream("mask_build", orig_mask, lambda rgb, regions, cols, ratios=(8, 16, 32, 64):
                                           sum(rgb.numel() + (-(-rgb.shape[0] // r)) * (-(-rgb.shape[1] // r)) * len(cols) * 4 for r in ratios))
        ops.cfg_combine = timer.wrap_stream("cfg_combine", orig_cfg, lambda c, u, g: c.numel() * (2 * c.element_size() + 4))
        pw_api.DEFAULT_MODE = "folded"
        # which path the folded-reference self-attention kernel's workgroups take ON THE WORKLOAD's own q / k (the kernel's debug counters,
        # summed over every d = 40 launch of the pass): range-free / lazy reference / lazy reference on the exact scale / exact recomputation
        import ctypes as _ct
        from pww_hip import _lib as _pl
        path_counts = torch.zeros(4, dtype=torch.int32, device=device)
        _pl.load().pww_debug_path_counts(_ct.c_void_p(path_counts.data_ptr()))
        try:
            one_step(0)
            torch.cuda.synchronize()
        finally:
            _pl.load().pww_debug_path_counts(None)
            ops.attention, ops.qk_stats, ops.mask_build, ops.cfg_combine, ops.qproj_stat, ops.qk_parts = orig, orig_stats, orig_mask, orig_cfg, orig_qproj, orig_qkparts
            pw_api.DEFAULT_MODE = args.mode
        pc = path_counts.tolist()
        workload_paths = {"fast": pc[0], "lazy": pc[1], "lazy_exact_scale": pc[3], "exact": pc[2]}
        H, W = request["rgb"].shape[:2]
        n_dom = (H // 8) * (W // 8)
        us_situ, n_launch, b_rows = timer.mean_us(lambda k: k[0] == "self" and k[2] == n_dom)
        result["kernels"] = timer.table(2)
        dom = [r for r in result["kernels"] if r["kernel"] == "self" and r.get("N") == n_dom]
        # the attention path per UNet forward (everything outside the dominant launch <= 400 us): sum over the launch classes of
        # kernel-only time x launches of the instrumented pass / its UNet forwards (one forward = the 5 dominant launches)
        if dom and dom[0]["launches"]:
            fw = dom[0]["launches"] / 5.0
            attn_rows = [r for r in result["kernels"] if "N" in r]
            tot = lambda rows, f: round(sum(r[f] * r["launches"] for r in rows if r.get(f) is not None) / fw, 1)      # noqa: E731
            others = [r for r in attn_rows if r is not dom[0]]
            result["attention_path"] = {"unet_forwards_in_pass": fw, "dominant_us_per_forward": tot(dom[:1], "avg_us"), "others_us_per_forward": tot(others, "avg_us"),
                                        "others_us_per_forward_back_to_back": tot(others, "avg_us_back_to_back"),
                                        "method": "sum over the pww launch classes of the instrumented eager pass: kernel-only duration x launches / UNet forwards (to_q GEMMs of "
                                                  "the stock library are not pww launches and not counted; pww_qproj_stat, which contains its GEMM, is)"}
        us_b2b = dom[0]["avg_us_back_to_back"] if dom else None     # hipGraph replay of 40 back-to-back launches, event interval / 40 (incl. the dispatch gaps)
        # the roofline number: the kernel's own start -> end device timestamps (HIP events stamped by the dispatch itself,
        # pww_profile_arm) averaged over every launch of the dominant class in the workload pass above -- what rocprofv3
        # reports for those launches
        us = dom[0]["avg_us"] if dom else None
        log("roofline pass done", us, us_b2b, n_launch)
        if us:
            kdom = [k for k in timer.pairs if k[0] == "self" and k[2] == n_dom][0]
            heads, d = kdom[5], kdom[4]
            flops = 4.0 * b_rows * heads * n_dom * n_dom * d      # algorithmic: QK^T + PV
            ach = flops / (us * 1e-6) / 1e12
            for hot_std in (4.0, 6.0):
                result["kernels"].append(hot_logit_row(device, dtype, b_rows, n_dom, d, heads, std=hot_std))
            result["roofline"] = {"bound": "mfma", "kernel": "self-attention N=%d d=%d (%s, B=%d rows folded)" % (n_dom, d, cfg["dtype"], b_rows),
                                  "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                                  "attainable": ATTAINABLE.get(d), "attainable_model": "issue-bound: MFMA and VALU issue cycles of a SIMD add up (448 vs ~500 per "
                                  "64-key tile and wave), times the algorithmic share of the padded MFMA rows; see DESIGN section 4",
                                  "mfma_busy": measured_mfma_busy(n_dom, d, b_rows, cfg["dtype"]), "mfma_busy_source": "CONSTANT: committed PMC pass (profiles/r04_pmc.json): "
                                  "SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), back-to-back launches of this shape",
                                  "traffic": measured_traffic(n_dom, d, b_rows, cfg["dtype"], live=args.live_traffic),
                                  "traffic_source": "live rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes" if args.live_traffic else "CONSTANT from the committed PMC pass of the shipped kernel "
                                  "(profiles/r06_traffic.json), not measured in this run",
                                  "traffic_correction": TRAFFIC_CORRECTION,
                                  "algorithmic_bytes": 2 * (2 * b_rows * n_dom * heads * d) * 2,
                                  "avg_us": round(us, 2), "avg_us_method": "kernel-only HIP event timestamps (hipExtLaunchKernelGGL start/stop events) of every launch of this "
                                  "class in an eager pass of the same workload, on the launch stream",
                                  "avg_us_back_to_back_graph_replay": round(us_b2b, 2), "avg_us_event_bracket_eager": round(us_situ, 2),
                                  "launches": n_launch, "flops_per_launch": flops,
                                  "workload_paths": workload_paths, "workload_paths_note": "workgroups of the d = 40 folded-reference launches of the instrumented pass per path "
                                  "(pww_debug_path_counts) on the workload's own q / k"}
        # counters measured NOW: a kernel change that doubles the traffic shows in the driver's own line
        if us and world == 1 and not args.no_live_counters and not args.live_traffic and not args.tiny:      # (N > 1: the labelled constant; the scaling runs need no second profiler process next to seven other ranks)
            t_pmc = time.perf_counter()
            case = HARNESS_CASES.get((n_dom, d, b_rows, cfg["dtype"]))
            live = live_counters(case, ["attn_fwd"]) if case else None
            lv = (live or {}).get("attn_fwd")
            if lv and "hbm_bytes" in lv:
                result["roofline"]["traffic"] = lv["hbm_bytes"]
                result["roofline"]["traffic_source"] = "LIVE: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this run over tests/native/attn_check --only %s (%d dispatches)" % (case, lv["dispatches"])
            if lv and "mfma_busy" in lv:
                result["roofline"]["mfma_busy"] = lv["mfma_busy"]
                result["roofline"]["mfma_busy_source"] = "LIVE: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), same passes"
            if lv and "fetch_size_kib_raw" in lv:
                result["roofline"]["traffic_counters_raw"] = {"FETCH_SIZE_KiB": lv["fetch_size_kib_raw"], "WRITE_SIZE_KiB": lv["write_size_kib"]}
            cross16 = None
            cross = live_counters(CROSS_COUNTER_CASE[0], ["cross_lean_kernel", "qk_parts_kernel", "cross_fused_kernel"], flags=("--product-only",))
            if cross:
                # algorithmic bytes of the two launches at 2 rows, N = 256, C = 1280, M = 77: attention 2 (2 B N C + 2 B M C) + the cond row's
                # [N, 32] bias span; partials: Q of the conditional row + K
                alg = {"cross_lean_kernel": 2 * (2 * 2 * 256 * 1280 + 2 * 2 * 77 * 1280) + 256 * 32 * 4, "qk_parts_kernel": 2 * (256 * 1280 + 77 * 1280)}
                for k, r in cross.items():
                    if k in alg and "hbm_bytes" in r:
                        r["algorithmic_bytes"] = alg[k]
                        r["traffic_over_algorithmic"] = round(r["hbm_bytes"] / alg[k], 2)
                result["counters_cross_route"] = {"case": CROSS_COUNTER_CASE[1], "kernels": cross, "traffic_correction": TRAFFIC_CORRECTION,
                                                  "source": "LIVE: rocprofv3 --pmc passes of this run over tests/native/attn_check --product-only --only %s" % CROSS_COUNTER_CASE[0]}
            # the batched route (configs 3 / 4: 16 folded rows, N = 4096, C = 320): the small kernel walking several query blocks per workgroup
            # (VERDICT round 5 item 5). Algorithmic bytes: Q + O (2 x 2 B N C) + K / V (2 x 2 B M C) + the shared map's 32-column span (N x 32 x 4)
            cross16 = live_counters(CROSS16_COUNTER_CASE[0], ["cross_lean_kernel", "cross_fused_kernel"], flags=("--product-only",))
            if cross16:
                alg16 = 2 * 2 * 16 * 4096 * 320 + 2 * 2 * 16 * 77 * 320 + 4096 * 32 * 4
                for k, r in cross16.items():
                    if "hbm_bytes" in r:
                        r["algorithmic_bytes"] = alg16
                        r["traffic_over_algorithmic"] = round(r["hbm_bytes"] / alg16, 2)
                result["counters_cross_route_16rows"] = {"case": CROSS16_COUNTER_CASE[1], "kernels": cross16, "traffic_correction": TRAFFIC_CORRECTION,
                                                         "source": "LIVE: rocprofv3 --pmc passes of this run over tests/native/attn_check --product-only --only %s" % CROSS16_COUNTER_CASE[0]}
            log("live counters done in %.1f s" % (time.perf_counter() - t_pmc), lv, cross, cross16)
        from pww_hip import _lib as _pww_lib
        _pww_lib.load().pww_profile_reset()
    if rank == 0 and world == 1 and not args.no_reference_ops and cfg["kind"] == "txt2img":
        result["reference_ops_same_gpu"] = reference_ops_same_gpu(cfg, tools, request, device, dtype, args.guidance)
        log("reference-ops pass done", result["reference_ops_same_gpu"])
    if rank == 0 and world == 1 and args.cpu_steps > 0 and cfg["kind"] == "txt2img":
        result["cpu_baseline"] = cpu_baseline(args, cfg, request)
    if rank == 0:
        print(json.dumps(result), file=json_out, flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Workgroup orders of round 6 (groups of adjacent heads on one XCD: pww_attn_core.h wg_to_pair_block, pww_cross_kernel.h head_major) are
PERMUTATIONS of the launch's workgroups: the arithmetic of every workgroup is untouched, so the results must be bit-identical to round 5's
order (PWW_DEBUG=attn_head_pairs=0,cross_head_major=0; the library reads its knobs once per process: run this script once per setting and
compare the lines). So is the batched cross-attention launch on the small kernel's several-blocks-per-workgroup form against the general
kernel (PWW_DEBUG=cross_lean_multi=0): same tile code, same order of operations. One line per case: sha256 of the output, max error against fp64 on sampled rows.
    python tools/diag_wg_order.py            (on a GPU box; tests/test_round6_gpu.py runs it twice)"""
import hashlib
import math
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "paint-with-words-sd_amd"))
from pww_hip import ops  # noqa: E402


def sha(t):
    return hashlib.sha256(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:16]


def self_case(dev, dtype, B, N, H, D, seed, std=1.0):
    g = torch.Generator().manual_seed(seed)
    gain = math.sqrt(std)
    q = (torch.randn(B, N, H * D, generator=g) * gain).to(dtype)
    k = (torch.randn(B, N, H * D, generator=g) * gain).to(dtype)
    v = torch.randn(B, N, H * D, generator=g).to(dtype)
    out = ops.attention(q.to(dev), k.to(dev), v.to(dev), H, D ** -0.5)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, N, max(1, N // 37)), torch.tensor([N - 1])])
    err = 0.0
    for b in (0, B - 1):
        qh = q[b, rows].double().view(len(rows), H, D).transpose(0, 1)
        kh, vh = (t[b].double().view(N, H, D).transpose(0, 1) for t in (k, v))
        ref = torch.matmul((torch.matmul(qh, kh.transpose(-1, -2)) * D ** -0.5).softmax(-1), vh).transpose(0, 1).reshape(len(rows), H * D)
        err = max(err, (out[b, rows].double().cpu() - ref).abs().max().item() / ref.abs().max().item())
    return sha(out), err


def cross_case(dev, dtype, B, N, H, D, M, gated, seed):
    """The batched C = 320 route: to_q with the statistic's partials (pww_qproj_stat), then the pass-2-only launch (several query blocks per
    workgroup at 16 rows: cross_fused_kernel) with the shared [N, M] map, its 32-column bound and the gated-images hint."""
    C = H * D
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, C, generator=g).to(dtype)
    w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dtype)
    k = torch.randn(B, M, C, generator=g).to(dtype)
    v = torch.randn(B, M, C, generator=g).to(dtype)
    bias = torch.zeros(N, M)
    bias[:, :32] = (torch.rand(N, 32, generator=g) < 0.3).float() * torch.rand(N, 32, generator=g) * 1.5
    gate = torch.tensor([1.0] * gated + [0.0] * (B - gated))
    xd, wd, kd, vd, bd, gd = (t.to(dev) for t in (x, w, k, v, bias, gate))
    q, parts = ops.qproj_stat(xd, wd, kd, H, ops.STAT_MAX, gate=gd)
    out = ops.attention(q, kd, vd, H, D ** -0.5, bias=bd, bias_coeff=gd, stat=(None, ops.STAT_MAX, 0.37), parts=parts, bias_cols=32, gated=gated)
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, N, max(1, N // 29)), torch.tensor([N - 1])])
    err = 0.0
    for b in (0, gated - 1, B - 1):
        qf = q[b].double().cpu()
        qh = qf.view(N, H, D).transpose(0, 1)
        kh, vh = (t[b].double().view(M, H, D).transpose(0, 1) for t in (k, v))
        s = torch.matmul(qh, kh.transpose(-1, -2))
        c = 0.37 * s.max().item() * gate[b].item()
        logits = (s[:, rows] + c * bias.double()[rows][None]) * D ** -0.5
        ref = torch.matmul(logits.softmax(-1), vh).transpose(0, 1).reshape(len(rows), C)
        err = max(err, (out[b, rows].double().cpu() - ref).abs().max().item() / ref.abs().max().item())
    return sha(out), err


def main():
    dev = torch.device("cuda:0")
    bars = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}
    fail = 0
    cases = [("self d=40 B=2 N=4096", lambda dt: self_case(dev, dt, 2, 4096, 8, 40, 1)),
             ("self d=40 B=2 N=1000 (ragged)", lambda dt: self_case(dev, dt, 2, 1000, 8, 40, 2)),
             ("self d=40 B=16 N=1024", lambda dt: self_case(dev, dt, 16, 1024, 8, 40, 3)),
             ("self d=40 B=3 N=2048 (24 pairs: no head groups)", lambda dt: self_case(dev, dt, 3, 2048, 8, 40, 4)),
             ("self d=40 B=4 N=2048 hot (std 6)", lambda dt: self_case(dev, dt, 4, 2048, 8, 40, 5, std=6.0)),
             ("self d=80 B=2 N=1024", lambda dt: self_case(dev, dt, 2, 1024, 8, 80, 6)),
             ("self d=160 B=2 N=256", lambda dt: self_case(dev, dt, 2, 256, 8, 160, 7)),
             ("cross d=40 B=16 N=4096 M=77 gated 8", lambda dt: cross_case(dev, dt, 16, 4096, 8, 40, 77, 8, 8)),
             ("cross d=40 B=8 N=4096 M=77 gated 4", lambda dt: cross_case(dev, dt, 8, 4096, 8, 40, 77, 4, 9)),
             ("cross d=40 B=6 N=4000 M=77 gated 3 (ragged)", lambda dt: cross_case(dev, dt, 6, 4000, 8, 40, 77, 3, 10)),
             ("cross d=40 B=5 N=4096 M=77 gated 2", lambda dt: cross_case(dev, dt, 5, 4096, 8, 40, 77, 2, 11)),
             ("cross d=64 B=8 N=9216 M=77 gated 4 (SD2.1)", lambda dt: cross_case(dev, dt, 8, 9216, 5, 64, 77, 4, 12))]
    for name, fn in cases:
        for dt in (torch.float16, torch.bfloat16):
            h, err = fn(dt)
            ok = err <= bars[dt]
            fail += not ok
            print("CASE %-50s %-9s sha %s err %.2e %s" % (name, str(dt).split(".")[1], h, err, "ok" if ok else "FAIL"), flush=True)
    sys.exit(1 if fail else 0)


if __name__ == "__main__":
    main()

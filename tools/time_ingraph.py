#!/usr/bin/env python
"""Why is every small pww launch 15 - 35 % slower inside the UNet's hipGraph than replayed back to back?
One captured graph per variant, 40 repetitions, event interval / 40 (us); the partner kernel's own time is measured the same way and subtracted.
  b2b          the attention launch alone, 40x (inputs never change: warm in every L2)
  +producer    q = F.linear(x, w) before every launch (the real situation: q was just written by another kernel, on other CUs / XCDs)
  +rewrite_q   q.copy_(q0) before every launch (q freshly written, but by a trivial kernel)
  +rewrite_kv  k / v rewritten instead of q
  +other_gemm  an unrelated GEMM of the producer's size before every launch (nothing the attention reads is touched: what is left is the switch
               between two different kernels -- instruction cache, kernel arguments, clocks)
  +tiny        an unrelated 1-element fill before every launch
Usage: python tools/time_ingraph.py [out.md]"""
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


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
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    from pww_hip import ops
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    rows = []
    for name, N, C, H, self_attn in (("self N=1024 d=80", 1024, 640, 8, True), ("self N=256 d=160", 256, 1280, 8, True), ("self N=64 d=160", 64, 1280, 8, True),
                                      ("cross N=4096 d=40 (parts)", 4096, 320, 8, False), ("cross N=256 d=160 (parts)", 256, 1280, 8, False)):
        B, D = 2, C // H
        g = torch.Generator().manual_seed(3)
        x = torch.randn(B, N, C, generator=g).to(dev, dtype)
        w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev, dtype)
        x2 = torch.randn(B, N, C, generator=g).to(dev, dtype)
        w2 = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev, dtype)
        out2 = torch.empty(B, N, C, device=dev, dtype=dtype)
        q = torch.empty(B, N, C, device=dev, dtype=dtype)
        q0 = F.linear(x, w)
        q.copy_(q0)
        M = N if self_attn else 77
        k = torch.randn(B, M, C, generator=g).to(dev, dtype)
        v = torch.randn(B, M, C, generator=g).to(dev, dtype)
        k0, v0 = k.clone(), v.clone()
        one = torch.zeros(1, device=dev)
        scale = D ** -0.5
        if self_attn:
            attn = lambda: ops.attention(q, k, v, H, scale)      # noqa: E731
        else:
            bias = ((torch.rand(N, 77, generator=g) < 0.3).float() * 1.5)
            bias[:, 32:] = 0
            bias = bias.to(dev)
            gate = torch.tensor([1.0, 0.0], device=dev)
            _, parts = ops.qproj_stat(x, w, k, H, ops.STAT_MAX, gate=gate)
            attn = lambda: ops.attention(q, k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, ops.STAT_MAX, 0.37), parts=parts, bias_cols=32, gated=1)   # noqa: E731
        t_attn = replay_us(attn)
        partners = {
            "+producer (q = x @ w)": lambda: torch.matmul(x, w.t(), out=q),
            "+rewrite_q (copy)": lambda: q.copy_(q0),
            "+rewrite_kv (copy)": lambda: (k.copy_(k0), v.copy_(v0)),
            "+other_gemm (untouched operands)": lambda: torch.matmul(x2, w2.t(), out=out2),
            "+tiny (1-element fill)": lambda: one.fill_(1.0),
        }
        row = [name, t_attn]
        for pname, pf in partners.items():
            t_p = replay_us(pf)
            t_pair = replay_us(lambda: (pf(), attn()))
            row.append((pname, t_p, t_pair, t_pair - t_p))
        rows.append(row)
        print("%-28s b2b %6.2f us |" % (name, t_attn), " | ".join("%s: partner %.2f, pair %.2f -> attention %.2f (%+.0f %%)" % (p[0], p[1], p[2], p[3], (p[3] / t_attn - 1) * 100)
                                                                 for p in row[2:]), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write("# Small attention launches: back to back vs behind another kernel (hipGraph replay of 40, us; attention time = pair - partner alone)\n\n")
            f.write("| launch (B = 2, bf16) | back to back | " + " | ".join(p[0] for p in rows[0][2:]) + " |\n|---|---|" + "---|" * len(rows[0][2:]) + "\n")
            for r in rows:
                f.write("| %s | %.2f | " % (r[0], r[1]) + " | ".join("%.2f (%+.0f %%)" % (p[3], (p[3] / r[1] - 1) * 100) for p in r[2:]) + " |\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""A/B of the `to_out` epilogue (reference paint_with_words.py:118-123 + the residual add of the
transformer block that follows it):  y = attn_out @ W^T + b + residual  as

  A  F.linear(x, W, b) + residual                      what the product runs: hipBLASLt GEMM with bias epilogue, then ONE elementwise add
  B  torch.addmm(residual, x, W^T) + b                  residual through beta = 1; the bias becomes the elementwise launch (no launch saved)
  C  torch.addmm(residual, [x | 1 0..], [W | b 0..]^T)  bias folded in as an extra K column (K = C + 8): ONE launch, the form measured and
                                                        removed in round 2 (2.683 vs 2.701 images/s)
  D  torch.baddbmm / addmm with bias pre-added to the residual (residual + b computed once per block input is NOT available: the residual
     changes every layer) -- not a candidate, listed for completeness only.

Each variant is replayed 50x back to back from a hipGraph (event interval / 50), for the four to_out shapes of the SD1.5 UNet at 2 and 16
folded rows, bf16. Prints a markdown table (committed as profiles/r03_to_out_epilogue.md)."""
import sys

import torch
import torch.nn.functional as F


def replay_us(call, reps=50):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            call()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return min(ts)


def main():
    dev, dt = torch.device("cuda:0"), torch.bfloat16
    torch.manual_seed(0)
    print("| rows B | tokens N | channels C | A: linear+bias, add (2 launches) | B: addmm(beta=1), +bias (2 launches) | C: addmm, bias as K column (1 launch) | C / A |")
    print("|---|---|---|---|---|---|---|")
    tot = {2: [0.0, 0.0, 0.0], 16: [0.0, 0.0, 0.0]}
    for B in (2, 16):
        for N, C, per_fwd in ((4096, 320, 5), (1024, 640, 5), (256, 1280, 5), (64, 1280, 1)):
            x = torch.randn(B, N, C, device=dev, dtype=dt)
            res = torch.randn(B, N, C, device=dev, dtype=dt)
            W = torch.randn(C, C, device=dev, dtype=dt) * C ** -0.5
            b = torch.randn(C, device=dev, dtype=dt)
            xp = torch.zeros(B, N, C + 8, device=dev, dtype=dt)
            xp[..., :C] = x
            xp[..., C] = 1.0
            Wp = torch.zeros(C, C + 8, device=dev, dtype=dt)
            Wp[:, :C] = W
            Wp[:, C] = b
            Wt, Wpt = W.t(), Wp.t()
            x2, r2, xp2 = x.view(-1, C), res.view(-1, C), xp.view(-1, C + 8)
            a = replay_us(lambda: F.linear(x, W, b) + res)
            bb = replay_us(lambda: torch.addmm(r2, x2, Wt) + b)
            c = replay_us(lambda: torch.addmm(r2, xp2, Wpt))
            ya, yc = F.linear(x, W, b) + res, torch.addmm(r2, xp2, Wpt).view(B, N, C)
            assert (ya.float() - yc.float()).abs().max().item() <= 0.15 * ya.float().abs().max().item()
            for i, v in enumerate((a, bb, c)):
                tot[B][i] += v * per_fwd * 2       # attn1 + attn2 per block
            print("| %d | %d | %d | %.2f | %.2f | %.2f | %.2f |" % (B, N, C, a, bb, c, c / a))
    for B in (2, 16):
        print("| %d | all 32 to_out calls of one UNet forward | | **%.0f us** | %.0f us | **%.0f us** | %.2f |" % (B, tot[B][0], tot[B][1], tot[B][2], tot[B][2] / tot[B][0]))


if __name__ == "__main__":
    main()

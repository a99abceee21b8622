#!/usr/bin/env python
"""What a small launch costs when its operands are NOT where the previous replay left them ("every small launch pays
+25 - 35 % inside the replayed graph over its back-to-back time"). Back-to-back replays of ONE launch over ONE set of buffers find every
operand in the L2 of the XCD that read it 10 us ago; in the UNet the operands were written by another kernel (other XCDs' L2s -> Infinity
Cache) or not touched since the previous forward (weights -> HBM). Here the same captured launch is replayed over ROTATING buffer sets:
    warm      1 set                      every operand L2-resident (the "back to back" column of bench.py)
    l2-cold   48 sets (> 8 x 4 MB L2)    operands come from the Infinity Cache (MALL)
    hbm-cold  sets worth > 512 MB        operands come from HBM
for pww_qk_parts, the pass-2-only cross-attention and the N = 1024 self-attention at the headline's 2 folded rows. us per launch (hipGraph of
one pass over the sets, best of 5)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def timed_graph(calls):
    for c in calls[:2]:
        c()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for c in calls:
            c()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(calls))
    return best


def main():
    from pww_hip import ops
    dev = torch.device("cuda:0")
    dt = torch.bfloat16
    B, M, H = 2, 77, 8
    gate = torch.tensor([1.0, 0.0], device=dev)
    rows = []
    for name, N, C in (("N=256 C=1280", 256, 1280), ("N=1024 C=640", 1024, 640), ("N=4096 C=320", 4096, 320)):
        per_set = (2 * B * N * C + 2 * B * M * C) * 2 + N * M * 4
        for label, nset in (("warm", 1), ("l2-cold", max(48, int(40e6 // per_set))), ("hbm-cold", max(64, int(600e6 // per_set)))):
            nset = min(nset, 400)
            sets = []
            for i in range(nset):
                g = torch.Generator(device="cpu").manual_seed(i)
                q = torch.randn(B, N, C, generator=g).to(dev, dt)
                k = torch.randn(B, M, C, generator=g).to(dev, dt)
                v = torch.randn(B, M, C, generator=g).to(dev, dt)
                w = torch.rand(N, M, generator=g).to(dev)
                sets.append((q, k, v, w))
            reps = max(40, nset)
            parts = [ops.qk_parts(s[0], s[1], H, ops.STAT_MAX, gate=gate, gated=1) for s in sets]
            t_parts = timed_graph([(lambda s=sets[i % nset]: ops.qk_parts(s[0], s[1], H, ops.STAT_MAX, gate=gate, gated=1)) for i in range(reps)])
            t_cross = timed_graph([(lambda s=sets[i % nset], p=parts[i % nset]: ops.attention(s[0], s[1], s[2], H, (C // H) ** -0.5, bias=s[3], bias_coeff=gate,
                                                                                               stat=(None, ops.STAT_MAX, 0.37), parts=p, bias_cols=32, gated=1)) for i in range(reps)])
            t_self = timed_graph([(lambda s=sets[i % nset]: ops.attention(s[0], s[0], s[0], H, (C // H) ** -0.5)) for i in range(reps)]) if N <= 1024 else float("nan")
            rows.append("| %s | %s (%d sets, %.0f MB) | %.2f | %.2f | %.2f |" % (name, label, nset, nset * per_set / 1e6, t_parts, t_cross, t_self))
            print(rows[-1], flush=True)
            del sets, parts
            torch.cuda.empty_cache()
    print()
    print("| layer (2 rows, bf16) | operands | pww_qk_parts | cross-attention (pass 2 only) | self-attention (q = k = v buffers) |")
    print("|---|---|---|---|---|")
    print("\n".join(rows))


if __name__ == "__main__":
    main()

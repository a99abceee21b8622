#!/usr/bin/env python
"""Per-shape timing of the two cross-attention routes on this GPU (profiles/r04_qproj.md):
  round 3:  to_q through the stock GEMM (hipBLASLt, F.linear)  +  pww_cross_attn_fwd_fused_ex (statistic + hand-off in the attention launch)
  round 4:  pww_qproj_stat (to_q with the statistic's partials in its epilogue)  +  pww_cross_attn_fwd_parts (folds them, pass 2 only)
Every launch class is captured once and replayed 40x back to back from a hipGraph (event interval / 40: includes the dispatch gaps,
no host in the loop). Usage: python tools/time_qproj.py [out.md]"""
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
    shapes = [("SD1.5 N=4096 C=320 d=40", 4096, 320, 8, 40), ("SD1.5 N=1024 C=640 d=80", 1024, 640, 8, 80), ("SD1.5 N=256 C=1280 d=160", 256, 1280, 8, 160),
              ("SD1.5 N=64 C=1280 d=160", 64, 1280, 8, 160), ("SD2.1 N=9216 C=320 d=64", 9216, 320, 5, 64), ("SD2.1 N=2304 C=640 d=64", 2304, 640, 10, 64),
              ("SD2.1 N=576 C=1280 d=64", 576, 1280, 20, 64)]
    rows = []
    for dtype in (torch.bfloat16, torch.float16):
        for name, N, C, H, D in shapes:
            for B in ((2, 16) if "SD1.5" in name else (2, 8)):
                if dtype == torch.float16 and B == 2:
                    continue
                g = torch.Generator().manual_seed(1)
                x = torch.randn(B, N, C, generator=g).to(dev, dtype)
                w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev, dtype)
                k = torch.randn(B, 77, C, generator=g).to(dev, dtype)
                v = torch.randn(B, 77, C, generator=g).to(dev, dtype)
                bias = ((torch.rand(N, 77, generator=g) < 0.3).float() * torch.rand(N, 77, generator=g) * 1.5)
                bias[:, 32:] = 0
                bias = bias.to(dev)
                gate = torch.cat([torch.ones(B // 2), torch.zeros(B - B // 2)]).to(dev)
                scale = D ** -0.5
                q = F.linear(x, w)
                scratch = ops.FusedScratch()
                t_lin = replay_us(lambda: F.linear(x, w))
                t_fused = replay_us(lambda: ops.attention(q, k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, ops.STAT_MAX, 0.37), scratch=scratch,
                                                          bias_cols=32, gated=B // 2))
                q2, parts = ops.qproj_stat(x, w, k, H, ops.STAT_MAX, gate=gate)
                t_qp = replay_us(lambda: ops.qproj_stat(x, w, k, H, ops.STAT_MAX, gate=gate))
                t_parts = replay_us(lambda: ops.attention(q2, k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, ops.STAT_MAX, 0.37), parts=parts,
                                                          bias_cols=32, gated=B // 2))
                t_old = replay_us(lambda: ops.attention(F.linear(x, w), k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, ops.STAT_MAX, 0.37), scratch=scratch,
                                                        bias_cols=32, gated=B // 2))

                def new_route():
                    qq, pp = ops.qproj_stat(x, w, k, H, ops.STAT_MAX, gate=gate)
                    return ops.attention(qq, k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, ops.STAT_MAX, 0.37), parts=pp, bias_cols=32, gated=B // 2)
                t_new = replay_us(new_route)
                o_old = ops.attention(q, k, v, H, scale, bias=bias, bias_coeff=gate, stat=(None, ops.STAT_MAX, 0.37), scratch=scratch, bias_cols=32, gated=B // 2)
                o_new = new_route()
                diff = (o_old.float() - o_new.float()).abs().max().item() / o_old.float().abs().max().item()
                gemm_bytes = 2 * (B * N * C * 2 + C * C)
                attn_bytes = 2 * (2 * B * N * C + 2 * B * 77 * C) + N * 32 * 4
                rows.append((name, str(dtype).split(".")[-1], B, parts.shape[1], t_lin, t_qp, t_fused, t_parts, t_old, t_new, diff, gemm_bytes / t_qp / 1e3, attn_bytes / t_parts / 1e3))
                print("%-26s %-8s B=%-2d parts %-3d | to_q stock %6.2f  qproj_stat %6.2f | fused attn %6.2f  parts attn %6.2f | route r3 %6.2f  r4 %6.2f us | diff %.1e"
                      % rows[-1][:11], flush=True)
    out = sys.argv[1] if len(sys.argv) > 1 else None
    if out:
        with open(out, "w") as f:
            f.write("# Cross-attention routes per shape (hipGraph replay of 40 launches, event interval / 40, us; MI355X)\n\n")
            f.write("| shape | dtype | rows B | partials / image | to_q, stock GEMM | pww_qproj_stat | round-3 fused attention | pww_cross_attn_fwd_parts | route r3: GEMM + fused | route r4: qproj_stat + parts | max diff / max\\|O\\| | qproj GB/s (algorithmic) | parts attention GB/s |\n")
            f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
            for r in rows:
                f.write("| %s | %s | %d | %d | %.2f | %.2f | %.2f | %.2f | %.2f | **%.2f** | %.1e | %.0f | %.0f |\n" % r)


if __name__ == "__main__":
    main()

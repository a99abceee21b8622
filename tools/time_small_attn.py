#!/usr/bin/env python
"""The attention-path launches of one SD1.5 UNet forward OUTSIDE the dominant self-attention: every launch class of
the headline workload (2 folded rows: conditional + unconditional, bf16) replayed 40x back to back from a hipGraph (event interval / 40, us).
Library knobs are read once per process (PWW_DEBUG): run the script once per variant, e.g.

    python tools/time_small_attn.py                                   # the shipped configuration
    PWW_DEBUG=cross_lean=0 python tools/time_small_attn.py            # pass-2-only launches on the general kernel (round 4)
    PWW_DEBUG=attn_ksplit_nw=4 python tools/time_small_attn.py self   # only the self-attention rows

Rows: self-attention N = 1024 / 256 / 64; per cross-attention layer class the three routes
    r3   stock to_q GEMM + pww_cross_attn_fwd_fused (statistic + hand-off in the attention launch)
    r4   pww_qproj_stat (to_q with the statistic in its epilogue) + pww_cross_attn_fwd_parts
    r5   stock to_q GEMM + pww_qk_parts + pww_cross_attn_fwd_parts
and their single launches. Output: one markdown table (stdout, or the file named by --out)."""
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
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


def main():
    from pww_hip import ops
    what = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = None
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
        what = [a for a in what if a != out]
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    B = 2
    lines = ["# small attention launches, PWW_DEBUG=%s (hipGraph replay of 40 launches, best of 5, us)" % os.environ.get("PWW_DEBUG", ""), ""]
    if not what or "self" in what:
        lines += ["| self-attention (B = %d rows) | us | TFLOP/s |" % B, "|---|---|---|"]
        for N, C, H in ((4096, 320, 8), (1024, 640, 8), (256, 1280, 8), (64, 1280, 8)):
            D = C // H
            g = torch.Generator().manual_seed(5)
            qkv = torch.randn(B, N, 3 * C, generator=g).to(dev, dtype)
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            t = replay_us(lambda: ops.attention(q, k, v, H, D ** -0.5))
            lines.append("| N=%d d=%d | %.2f | %.0f |" % (N, D, t, 4.0 * B * H * N * N * D / t / 1e6))
            print(lines[-1], flush=True)
        lines.append("")
    if not what or "cross" in what:
        lines += ["| cross-attention layer (B = %d rows, gate [1, 0]) | stock to_q | qproj_stat | qk_parts | parts attention | fused (r3) attention | route r3 | route r4 | route r5 | r5 vs r3 max diff |" % B,
                  "|---|---|---|---|---|---|---|---|---|---|"]
        for N, C, H in ((4096, 320, 8), (1024, 640, 8), (256, 1280, 8), (64, 1280, 8)):
            D = C // H
            g = torch.Generator().manual_seed(1)
            x = torch.randn(B, N, C, generator=g).to(dev, dtype)
            w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev, dtype)
            k = torch.randn(B, 77, C, generator=g).to(dev, dtype)
            v = torch.randn(B, 77, C, generator=g).to(dev, dtype)
            bias = ((torch.rand(N, 77, generator=g) < 0.3).float() * torch.rand(N, 77, generator=g) * 1.5)
            bias[:, 32:] = 0
            bias = bias.to(dev)
            gate = torch.tensor([1.0, 0.0]).to(dev)
            scale = D ** -0.5
            kw = dict(bias=bias, bias_coeff=gate, bias_cols=32, gated=1)
            q = F.linear(x, w)
            scratch = ops.FusedScratch()
            t_lin = replay_us(lambda: F.linear(x, w))
            t_fused = replay_us(lambda: ops.attention(q, k, v, H, scale, stat=(None, ops.STAT_MAX, 0.37), scratch=scratch, **kw))
            t_r3 = replay_us(lambda: ops.attention(F.linear(x, w), k, v, H, scale, stat=(None, ops.STAT_MAX, 0.37), scratch=scratch, **kw))
            have_qp = ops.qproj_parts(x, w, k, H) > 0
            t_qp = t_r4 = float("nan")
            if have_qp:
                t_qp = replay_us(lambda: ops.qproj_stat(x, w, k, H, ops.STAT_MAX, gate=gate))

                def r4():
                    qq, pp = ops.qproj_stat(x, w, k, H, ops.STAT_MAX, gate=gate)
                    return ops.attention(qq, k, v, H, scale, stat=(None, ops.STAT_MAX, 0.37), parts=pp, **kw)
                t_r4 = replay_us(r4)
            parts = ops.qk_parts(q, k, H, ops.STAT_MAX, gate=gate, gated=1)
            t_qkp = replay_us(lambda: ops.qk_parts(q, k, H, ops.STAT_MAX, gate=gate, gated=1))
            t_parts = replay_us(lambda: ops.attention(q, k, v, H, scale, stat=(None, ops.STAT_MAX, 0.37), parts=parts, **kw))

            def r5():
                qq = F.linear(x, w)
                pp = ops.qk_parts(qq, k, H, ops.STAT_MAX, gate=gate, gated=1)
                return ops.attention(qq, k, v, H, scale, stat=(None, ops.STAT_MAX, 0.37), parts=pp, **kw)
            t_r5 = replay_us(r5)
            o3 = ops.attention(q, k, v, H, scale, stat=(None, ops.STAT_MAX, 0.37), scratch=scratch, **kw).float()
            o5 = r5().float()
            diff = (o3 - o5).abs().max().item() / o3.abs().max().item()
            lines.append("| N=%d C=%d d=%d | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f | **%.2f** | %.1e |" % (N, C, D, t_lin, t_qp, t_qkp, t_parts, t_fused, t_r3, t_r4, t_r5, diff))
            print(lines[-1], flush=True)
        lines.append("")
    if not what or "toout" in what:
        # f-1: what an in-kernel `to_out` epilogue could save at most = the stock GEMM's launch (it cannot run faster inside another kernel) minus
        # nothing: bias rides the GEMM's epilogue, the residual add rides pww_add_layer_norm
        lines += ["| to_out (stock nn.Linear with bias, B = %d rows) | GEMM + bias epilogue | + residual add launch (not issued by the product: fused into pww_add_layer_norm) |" % B, "|---|---|---|"]
        for N, C in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
            g = torch.Generator().manual_seed(2)
            o = torch.randn(B, N, C, generator=g).to(dev, dtype)
            r = torch.randn(B, N, C, generator=g).to(dev, dtype)
            w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev, dtype)
            bias = torch.randn(C, generator=g).to(dev, dtype)
            t1 = replay_us(lambda: F.linear(o, w, bias))
            t2 = replay_us(lambda: F.linear(o, w, bias) + r)
            lines.append("| N=%d C=%d | %.2f | %.2f |" % (N, C, t1, t2))
            print(lines[-1], flush=True)
        lines.append("")
    if not what or "outproj" in what:
        # f-1 measured: cross-attention + to_out in ONE launch (pww_cross_attn_fwd_parts_out: heads sequential in a workgroup of 128 rows) against
        # the two-launch route (pass-2-only attention + the stock GEMM with its bias epilogue), per number of folded rows
        lines += ["| C = 320 cross-attention + to_out (N = 4096, 8 x 40, M = 77, gate = first half) | rows | attention | stock to_out | both back to back | attention_out (one launch) | max diff / max |",
                  "|---|---|---|---|---|---|---|"]
        N, C, H = 4096, 320, 8
        D = C // H
        for B in (2, 4, 8, 16, 32):
            g = torch.Generator().manual_seed(3)
            q = (torch.randn(B, N, C, generator=g) * 0.6).to(dev, dtype)
            k = torch.randn(B, 77, C, generator=g).to(dev, dtype)
            v = torch.randn(B, 77, C, generator=g).to(dev, dtype)
            w = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev, dtype)
            wb = torch.randn(C, generator=g).to(dev, dtype)
            bias = ((torch.rand(N, 77, generator=g) < 0.3).float() * torch.rand(N, 77, generator=g) * 1.5)
            bias[:, 32:] = 0
            bias = bias.to(dev)
            gate = torch.tensor([1.0] * (B // 2) + [0.0] * (B - B // 2)).to(dev)
            parts = ops.qk_parts(q, k, H, ops.STAT_MAX, gate=gate, gated=B // 2)
            kw = dict(bias_coeff=gate, stat=(None, ops.STAT_MAX, 0.37), parts=parts, bias_cols=32, gated=B // 2)
            t_a = replay_us(lambda: ops.attention(q, k, v, H, D ** -0.5, bias=bias, **kw))
            o = ops.attention(q, k, v, H, D ** -0.5, bias=bias, **kw)
            t_l = replay_us(lambda: F.linear(o, w, wb))
            t_2 = replay_us(lambda: F.linear(ops.attention(q, k, v, H, D ** -0.5, bias=bias, **kw), w, wb))
            t_1 = replay_us(lambda: ops.attention_out(q, k, v, H, D ** -0.5, bias, w, wb, **kw))
            a, b2 = ops.attention_out(q, k, v, H, D ** -0.5, bias, w, wb, **kw).float(), F.linear(o, w, wb).float()
            lines.append("| | %d | %.2f | %.2f | %.2f | **%.2f** | %.1e |" % (B, t_a, t_l, t_2, t_1, (a - b2).abs().max().item() / b2.abs().max().item()))
            print(lines[-1], flush=True)
        lines.append("")
    text = "\n".join(lines) + "\n"
    if out:
        with open(out, "a") as f:
            f.write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The d = 40 self-attention launch on hot logits: time, path counts and error against fp64 for scaled-logit
std 1 / 4 / 5 / 6 / 7 / 8 at 2 and 16 folded rows, fp16 (and bf16 as the control). Library knobs are read once per process:

    python tools/diag_hot_f16.py                                   # the shipped configuration
    PWW_DEBUG=attn_hot_sum=0 python tools/diag_hot_f16.py          # round 5: range-free until the end, whole-workgroup exact redo
    PWW_DEBUG=attn_hot_sum=-1 python tools/diag_hot_f16.py         # lazy reference from the second stage on, always
    PWW_DEBUG=attn_fold_limit_f16=44 python tools/diag_hot_f16.py  # magnitude guard of the folded-reference kernel at 44 exp2 units
Prints one markdown row per (dtype, rows, std)."""
import ctypes
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)

import torch  # noqa: E402


def replay_us(call, reps=20):
    for _ in range(2):
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
    from pww_hip import ops, _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    N, D, H = 4096, 40, 8
    stds = [float(a) for a in sys.argv[1:]] or [1.0, 4.0, 5.0, 6.0, 7.0, 8.0]
    print("| PWW_DEBUG=%s | rows | std | row max (nat) mean / max | us | fast / lazy / lazy-exact-scale / exact workgroups | max err / max|O| |" % os.environ.get("PWW_DEBUG", ""))
    print("|---|---|---|---|---|---|---|")
    for dtype in (torch.float16, torch.bfloat16):
        for B in (2, 16):
            for std in stds:
                g = torch.Generator().manual_seed(7)
                gain = math.sqrt(std)
                q = (torch.randn(B, N, H * D, generator=g) * gain).to(dtype)
                k = (torch.randn(B, N, H * D, generator=g) * gain).to(dtype)
                v = torch.randn(B, N, H * D, generator=g).to(dtype)
                qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
                us = replay_us(lambda: ops.attention(qd, kd, vd, H, D ** -0.5))
                counts = torch.zeros(4, dtype=torch.int32, device=dev)
                lib.pww_debug_path_counts(ctypes.c_void_p(counts.data_ptr()))
                out = ops.attention(qd, kd, vd, H, D ** -0.5).float().cpu()
                torch.cuda.synchronize()
                lib.pww_debug_path_counts(None)
                c = counts.tolist()
                # fp64 on sampled rows of image 0 (and the last image), all heads
                rows = torch.arange(0, N, 61)
                err, mx_mean, mx_max = 0.0, 0.0, 0.0
                for b in sorted({0, B - 1}):
                    qh = q[b, rows].double().view(len(rows), H, D).transpose(0, 1)
                    kh = k[b].double().view(N, H, D).transpose(0, 1)
                    vh = v[b].double().view(N, H, D).transpose(0, 1)
                    logits = torch.matmul(qh, kh.transpose(-1, -2)) * D ** -0.5
                    ref = torch.matmul(logits.softmax(-1), vh).transpose(0, 1).reshape(len(rows), H * D)
                    err = max(err, (out[b, rows].double() - ref).abs().max().item() / ref.abs().max().item())
                    mx_mean, mx_max = logits.max(-1).values.mean().item(), max(mx_max, logits.max().item())
                print("| %s | %d | %g | %.1f / %.1f | %.1f | %d / %d / %d / %d | %.2e |" % (str(dtype).replace("torch.", ""), B, std, mx_mean, mx_max, us, c[0], c[1], c[3], c[2], err), flush=True)


if __name__ == "__main__":
    main()

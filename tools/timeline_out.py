#!/usr/bin/env python
"""Phase time stamps of pww_cross_attn_fwd_parts_out (pww_debug_timeline): per stamp slot, microseconds since the earliest kernel-entry
stamp of the workgroup, averaged over the query blocks of one image. Stamps: 0 entry, 6 prologue loads issued, 3 partials folded, 1 head 0 staged,
2 phase 1 (all heads) done, 5 phase 2 (projection) done, 4 outputs stored.   python tools/timeline_out.py [rows ...]"""
import ctypes
import math
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "paint-with-words-sd_amd"))
import torch  # noqa: E402


def main():
    from pww_hip import ops, _lib
    lib = _lib.load()
    lib.pww_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    dev, dtype = torch.device("cuda:0"), torch.bfloat16
    N, C, H = 4096, 320, 8
    D = C // H
    for B in [int(a) for a in sys.argv[1:]] or [2, 8]:
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
        for _ in range(3):
            ops.attention_out(q, k, v, H, D ** -0.5, bias, w, wb, **kw)
        nwg = B * (N // 128)
        buf = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        lib.pww_debug_timeline(ctypes.c_void_p(buf.data_ptr()), buf.numel() * 8)
        ops.attention_out(q, k, v, H, D ** -0.5, bias, w, wb, **kw)
        torch.cuda.synchronize()
        lib.pww_debug_timeline(None, 0)
        t = buf.cpu().reshape(nwg, 8).double()[: N // 128]      # (the stamp index is blockIdx.x: one entry per query block, whichever image wrote last)
        t0 = t[:, 0].min()
        us = (t - t0) / 100.0
        names = {0: "entry", 6: "loads issued", 3: "partials folded", 1: "head 0 staged", 2: "phase 1 done", 5: "phase 2 done", 4: "stored"}
        print("TIMELINE attention_out rows=%d workgroups=%d: " % (B, nwg) + ", ".join("%s %.2f" % (names[s], (us[:, s] - us[:, 0]).mean()) for s in (0, 6, 3, 1, 2, 5, 4)), flush=True)


if __name__ == "__main__":
    main()

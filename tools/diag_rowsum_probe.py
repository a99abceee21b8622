#!/usr/bin/env python
"""Which value does a self-attention kernel divide by? V[key][c] = c + 1 for every key: the exact output is c + 1; a kernel that normalises
by sum(P * V[.][x]) instead of sum(P) returns (c + 1) / (x + 1)."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
from pww_hip import ops  # noqa: E402
dev = torch.device("cuda:0")
for D, N, H in ((152, 128, 2), (88, 600, 4), (120, 256, 8), (144, 128, 2)):
    C = H * D
    g = torch.Generator().manual_seed(1)
    q = (torch.randn(1, N, C, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    k = (torch.randn(1, N, C, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    v = (torch.arange(D).float() + 1).repeat(H)[None, None, :].expand(1, N, C).contiguous().to(torch.bfloat16).to(dev)
    out = ops.attention(q, k, v, H, D ** -0.5).float()
    r = out[0, 0, :D] / v[0, 0, :D].float()
    print("D=%d N=%d: out/(c+1) over the channels of head 0, row 0: min %.4f max %.4f -> divides by channel value %.2f; row 5: %.4f" % (D, N, r.min(), r.max(), 1.0 / r.mean(), (out[0, 5, :D] / v[0, 5, :D].float()).mean()))

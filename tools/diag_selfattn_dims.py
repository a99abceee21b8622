#!/usr/bin/env python
"""Self-attention through the library vs fp64 on the same rounded inputs for a sweep of head dims / token counts (which kernel a shape takes
depends on both): max|err| / max|O| per case. PWW_DEBUG selects kernel variants (e.g. attn_ksplit1=0)."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
from pww_hip import ops  # noqa: E402

dev = torch.device("cuda:0")
print("PWW_DEBUG=%s" % os.environ.get("PWW_DEBUG", ""))
for dtype in (torch.float16, torch.bfloat16):
    for D in (72, 80, 88, 96, 104, 112, 120, 128, 136, 144, 152, 160):
        for N, H, B in ((128, 2, 1), (256, 8, 2), (600, 4, 1), (1024, 8, 1)):
            C = H * D
            g = torch.Generator().manual_seed(D + N)
            qkv = (torch.randn(B, N, 3 * C, generator=g) * 0.7).to(dtype).to(dev)
            q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
            out = ops.attention(q, k, v, H, D ** -0.5)
            qh = q.double().reshape(B, N, H, D).permute(0, 2, 1, 3)
            kh = k.double().reshape(B, N, H, D).permute(0, 2, 1, 3)
            vh = v.double().reshape(B, N, H, D).permute(0, 2, 1, 3)
            ref = (((qh @ kh.transpose(-1, -2)) * D ** -0.5).softmax(-1) @ vh).permute(0, 2, 1, 3).reshape(B, N, C)
            err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
            tol = 2e-3 if dtype == torch.float16 else 1.6e-2
            print("%s D=%3d N=%4d H=%d B=%d err %.2e %s" % (str(dtype)[6:], D, N, H, B, err, "" if err <= tol else "<<<<<< FAIL"))


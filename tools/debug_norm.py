import sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from pww_hip import ops
import test_norm_gpu as T
for dtype in (torch.bfloat16, torch.float16):
  for cl in (False, True):
    for shape in T.SHAPES:
        B, C, H, W = shape
        g = torch.Generator(device="cpu").manual_seed(B * 1000 + C + H)
        x = (torch.randn(shape, generator=g) * 1.7 + 0.4 * torch.randn(1, C, 1, 1, generator=g)).to("cuda:0", dtype)
        if cl: x = x.contiguous(memory_format=torch.channels_last)
        w = (1.0 + 0.3 * torch.randn(C, generator=g)).to("cuda:0", dtype)
        b = (0.2 * torch.randn(C, generator=g)).to("cuda:0", dtype)
        add = (0.8 * torch.randn(B, C, generator=g)).to("cuda:0", dtype)
        for use_add, act in ((False, None), (False, "silu"), (True, "silu"), (True, None)):
            y = ops.group_norm(x, 32, w, b, 1e-5, add=add if use_add else None, act=act)
            ref = T._reference(x, add if use_add else None, w, b, 32, 1e-5, act, dtype)
            h = x if not use_add else x + add[:, :, None, None]
            stock = F.group_norm(h, 32, w, b, 1e-5); stock = F.silu(stock) if act == "silu" else stock
            n1, r1 = T._close(y, ref, dtype); n2, r2 = T._close(y, stock, dtype, 2); n3, r3 = T._close(stock, ref, dtype)
            print(dtype, cl, shape, use_add, act, "vs ref bad %d rel %.2e | vs stock bad %d rel %.2e | stock vs ref bad %d rel %.2e | nan %d" % (n1, r1, n2, r2, n3, r3, int(torch.isnan(y.float()).sum())))

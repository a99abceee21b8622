"""Diagnostic: bitwise repeatability of (a) the HIP ops alone, (b) one UNet forward with unfused torch attention,
(c) one UNet forward with the HIP attention."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import pww_hip
from pww_hip import ops
import pww_cases as cases
from gpu_util import install_unfused, uninstall_all
dev = torch.device("cuda:0"); dt = torch.float16
g = torch.Generator().manual_seed(0)
for dt in (torch.float16, torch.bfloat16):
  for (B, H, N, M, D) in ((2, 8, 4096, 4096, 40), (1, 8, 4096, 4096, 40), (2, 8, 4096, 77, 40), (2, 4, 4096, 4096, 8), (2, 4, 1024, 77, 16)):
      q = torch.randn(B, N, H * D, generator=g).to(dev, dt); k = torch.randn(B, M, H * D, generator=g).to(dev, dt); v = torch.randn(B, M, H * D, generator=g).to(dev, dt)
      bias = torch.rand(N, M, generator=g).to(dev) if M == 77 else None
      outs = [ops.attention(q, k, v, H, D ** -0.5, bias=bias).clone() for _ in range(4)]
      st = [ops.qk_stats(q, k, H).clone() for _ in range(4)]
      line = "attention %s bitwise equal: %s | stats equal: %s" % ((B, H, N, M, D), all(torch.equal(outs[0], o) for o in outs[1:]), all(torch.equal(st[0], s) for s in st[1:]))
      if bias is not None:      # statistic formed in the attention launch (pww_cross_attn_fwd_fused): hand-off order must not leak into the result
          scratch = ops.FusedScratch()
          fo = [ops.attention(q, k, v, H, D ** -0.5, bias=bias, stat=(None, ops.STAT_STD, 0.3), scratch=scratch).clone() for _ in range(6)]
          line += " | fused statistic launch equal: %s" % all(torch.equal(fo[0], o) for o in fo[1:])
      print(line, flush=True)
dt = torch.float16
for name, inst in (("unfused-torch", install_unfused), ("hip", pww_hip.install)):
    vae, unet, text, tok, sch = cases.build_tools("tiny", dtype=dt, device=dev)
    inst(unet)
    x = torch.randn(2, 4, 64, 64, generator=g).to(dev, dt); ctx = torch.randn(2, 77, 64, generator=g).to(dev, dt)
    with torch.no_grad():
        ys = [unet(x, torch.tensor(500.0), encoder_hidden_states=ctx).sample.clone() for _ in range(4)]
    print("UNet forward", name, "bitwise equal:", all(torch.equal(ys[0], y) for y in ys[1:]), "max diff", max(float((ys[0] - y).abs().max()) for y in ys[1:]), flush=True)
    uninstall_all()

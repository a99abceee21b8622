"""Diagnostic: UNet forward time (B=2 rows, bf16) under memory-format / MIOpen NHWC settings."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
import pww_hip
from sd_standin import build_unet, SD15_CONFIG
dev = torch.device("cuda:0"); dtype = torch.bfloat16
mode = sys.argv[1]
unet = build_unet(SD15_CONFIG, seed=1234, dtype=torch.float32, device="cpu").to(dev, dtype)
pww_hip.install(unet)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
x = torch.randn(B, 4, 64, 64, device=dev, dtype=dtype); ctx = torch.randn(B, 77, 768, device=dev, dtype=dtype); t = torch.tensor(500.0, device=dev)
if mode == "cl":
    unet = unet.to(memory_format=torch.channels_last); x = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    for i in range(3): unet(x, t, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): y = unet(x, t, encoder_hidden_states=ctx).sample
    for i in range(3): g.replay()
    torch.cuda.synchronize(); t0 = time.time()
    for i in range(20): g.replay()
    torch.cuda.synchronize()
    print("mode=%s NHWC_env=%s B=%d: %.2f ms per forward (graph replay), out mean %.5f" % (mode, os.environ.get("PYTORCH_MIOPEN_SUGGEST_NHWC"), B, (time.time() - t0) * 50, float(y.float().abs().mean())), flush=True)

"""Diagnostic: where does time go in the first SD1.5-size UNet forwards on the GPU (not a test)."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
T0 = time.time()
def log(*a):
    print("[%7.1fs]" % (time.time() - T0), *a, flush=True)
import torch
log("torch imported", torch.__version__)
import pww_hip
from sd_standin import build_unet, SD15_CONFIG
dev = torch.device("cuda:0")
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float16
unet = build_unet(SD15_CONFIG, seed=1234, dtype=torch.float32, device="cpu")
log("unet built on cpu")
unet = unet.to(dev, dtype)
torch.cuda.synchronize(); log("unet on gpu", dtype)
pww_hip.install(unet)
x = torch.randn(2, 4, 64, 64, device=dev, dtype=dtype)
ctx = torch.randn(2, 77, 768, device=dev, dtype=dtype)
t = torch.tensor(500.0, device=dev)
with torch.no_grad():
    for i in range(4):
        t0 = time.time(); y = unet(x, t, encoder_hidden_states=ctx).sample; torch.cuda.synchronize()
        log("eager forward %d: %.3f s" % (i, time.time() - t0), float(y.float().abs().mean()))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        unet(x, t, encoder_hidden_states=ctx)
    torch.cuda.current_stream().wait_stream(s)
    t0 = time.time()
    with torch.cuda.graph(g):
        y = unet(x, t, encoder_hidden_states=ctx).sample
    torch.cuda.synchronize(); log("capture: %.3f s" % (time.time() - t0))
    for i in range(3):
        t0 = time.time(); g.replay(); torch.cuda.synchronize(); log("replay %d: %.2f ms" % (i, (time.time() - t0) * 1e3))
    t0 = time.time()
    for i in range(20): g.replay()
    torch.cuda.synchronize(); log("20 replays: %.2f ms each" % ((time.time() - t0) * 1e3 / 20))
    t0 = time.time()
    for i in range(10): unet(x, t, encoder_hidden_states=ctx)
    torch.cuda.synchronize(); log("10 eager: %.2f ms each" % ((time.time() - t0) * 1e3 / 10))
    # channels_last variant
    unet_cl = unet.to(memory_format=torch.channels_last)
    xcl = x.contiguous(memory_format=torch.channels_last)
    for i in range(3):
        t0 = time.time(); unet_cl(xcl, t, encoder_hidden_states=ctx); torch.cuda.synchronize(); log("channels_last eager %d: %.3f s" % (i, time.time() - t0))
    t0 = time.time()
    for i in range(10): unet_cl(xcl, t, encoder_hidden_states=ctx)
    torch.cuda.synchronize(); log("10 eager channels_last: %.2f ms each" % ((time.time() - t0) * 1e3 / 10))
log("done")

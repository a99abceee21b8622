"""Shader clock / power while the dominant self-attention launch runs back to back (run on the GPU box).
Queues a few seconds of launches of the bench's dominant shape, samples `rocm-smi` meanwhile, then does the same for idle."""
import re
import subprocess
import sys
import time

import torch

sys.path.insert(0, "paint-with-words-sd_amd")
from pww_hip import ops  # noqa: E402


def sample(tag, n=6, dt=0.25):
    for i in range(n):
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.findall(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
        pw = re.findall(r"Power \(W\): ([\d.]+)", out)
        print(tag, "sample", i, "sclk MHz", sclk[:1], "power W", pw[:1], flush=True)
        time.sleep(dt)


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    q = torch.randn(2, 4096, 320, device=dev, dtype=torch.bfloat16)
    k = torch.randn(2, 4096, 320, device=dev, dtype=torch.bfloat16)
    v = torch.randn(2, 4096, 320, device=dev, dtype=torch.bfloat16)
    for _ in range(10):
        ops.attention(q, k, v, 8, 40 ** -0.5)
    torch.cuda.synchronize()
    sample("idle")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200):
            ops.attention(q, k, v, 8, 40 ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(250):          # ~3 s of back-to-back launches, queued asynchronously
        g.replay()
    e1.record()
    sample("busy")
    torch.cuda.synchronize()
    print("back-to-back average: %.2f us per launch over %d launches" % (e0.elapsed_time(e1) * 1e3 / (250 * 200), 250 * 200))


if __name__ == "__main__":
    main()

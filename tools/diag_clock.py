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
    run("random N(0,1) inputs", zeros=False)
    time.sleep(1.0)
    run("all-zero inputs (same instruction stream, nothing toggles in the operands)", zeros=True)


def run(label, zeros):
    print("====", label, flush=True)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    mk = (lambda: torch.zeros(2, 4096, 320, device=dev, dtype=torch.bfloat16)) if zeros else \
         (lambda: torch.randn(2, 4096, 320, device=dev, dtype=torch.bfloat16))
    q, k, v = mk(), mk(), mk()
    for _ in range(10):
        ops.attention(q, k, v, 8, 40 ** -0.5)
    torch.cuda.synchronize()
    if not zeros:
        sample("idle", n=3)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(200):
            ops.attention(q, k, v, 8, 40 ** -0.5)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(250):          # ~3 s of back-to-back launches, queued asynchronously
        g.replay()
    e1.record()
    sample("busy", n=4)
    torch.cuda.synchronize()
    print("back-to-back average: %.2f us per launch over %d launches" % (e0.elapsed_time(e1) * 1e3 / (250 * 200), 250 * 200))


if __name__ == "__main__":
    main()

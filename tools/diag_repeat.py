"""Diagnostic: repeated identical requests in graph mode must give identical latents."""
import os, sys, importlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "paint-with-words-sd_amd"), REPO, os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch, numpy as np
from PIL import Image
import pww_cases as cases
import paint_with_words as pw
pww_mod = importlib.import_module("paint_with_words.paint_with_words")
dev = "cuda:0"
img = Image.fromarray(cases.load_example_rgb())
def rel(a, b): return float((a.float() - b.float()).norm() / b.float().norm())
for mode in ("graph", "folded", "eager"):
    pww_mod.DEFAULT_MODE = mode
    tools = cases.build_tools("tiny", dtype=torch.float16, device=dev)
    outs = []
    for i in range(3):
        outs.append(pw.paint_with_words(color_context=dict(cases.RUNNER_CONTEXT), color_map_image=img, input_prompt=cases.RUNNER_PROMPT,
                                        num_inference_steps=3, guidance_scale=7.5, seed=0, device=dev,
                                        weight_function=cases.weight_fn_runner, preloaded_utils=tools, return_latents=True).clone())
    print(mode, "call2 vs call1 %.3e  call3 vs call1 %.3e  call3 vs call2 %.3e" % (rel(outs[1], outs[0]), rel(outs[2], outs[0]), rel(outs[2], outs[1])), flush=True)
    if mode == "graph": g = outs
    else: print("   ", mode, "vs graph call1 %.3e, vs graph call2 %.3e" % (rel(outs[0], g[0]), rel(outs[0], g[1])))

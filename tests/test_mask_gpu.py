"""HIP mask-build path (pww_mask_build through pww_hip.conditioning) vs the oracle (bit-exact: both
use the plain fp32 bilinear formula) and vs the reference goldens (<= 1e-6)."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import pww_cases as cases
from oracle import pww_oracle as O
from sd_standin import HashTokenizer, TinyTextEncoder
from test_oracle_golden import _mask_cases, _dense, MASK_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", MASK_CASES)
def test_weight_maps(gpu_device, name):
    from pww_hip import conditioning as C
    img, ctx, prompt = _mask_cases()[name]
    g = np.load(os.path.join(cases.GOLDEN, f"masks_{name}.npz"))
    tok, enc = HashTokenizer(), TinyTextEncoder(64).to(gpu_device)
    _, _, cond, uncond = C._encode_text_color_inputs(enc, tok, gpu_device, Image.fromarray(img), dict(ctx), prompt, "")
    # oracle on the CPU
    ctx2, seeds, sigmas = O.extract_seed_and_sigma(dict(ctx))
    regions, W, H = O.separate_regions(img, ctx2, tok)
    for k, s in sigmas.items():
        regions[k] = (regions[k][0], O.gaussian_blur(regions[k][1], s))
    ids = g["token_ids"].tolist()
    for r in (8, 16, 32, 64):
        key = f"CROSS_ATTENTION_WEIGHT_{O.always_round(H / r) * O.always_round(W / r)}"
        got = cond[key].cpu().numpy()
        assert uncond[key] == 0
        want = O.tokens_img_attention_weight(regions, ids, r)
        if name != "blur":
            assert np.array_equal(got, want), (name, r, np.abs(got - want).max())     # bit-exact vs oracle
            assert np.abs(got - _dense(g, r)).max() <= 1e-6                            # vs the reference
        else:
            assert np.abs(got - want).max() <= 2e-5 and np.abs(got - _dense(g, r)).max() <= 2e-5
    orig = cond["CROSS_ATTENTION_WEIGHT_ORIG"].cpu().numpy()
    assert list(orig.shape) == g["orig_shape"].tolist()
    np.testing.assert_allclose(orig.sum(axis=(0, 1), dtype=np.float64), g["orig_colsum"], rtol=1e-5)
    if name == "example":
        assert np.array_equal(orig, O.tokens_img_attention_weight(regions, ids, 1, original_shape=True))


def test_empty_context_and_missing_phrase(gpu_device, capsys):
    from pww_hip import conditioning as C
    tok, enc = HashTokenizer(), TinyTextEncoder(64).to(gpu_device)
    img = Image.fromarray(cases.load_example_rgb())
    _, _, cond, _ = C._encode_text_color_inputs(enc, tok, gpu_device, img, {}, "a photo", "")
    assert float(cond["CROSS_ATTENTION_WEIGHT_4096"].abs().sum()) == 0.0
    _, _, cond, _ = C._encode_text_color_inputs(enc, tok, gpu_device, img, {(0, 0, 0): "zebra,1.0", (1, 2, 3): "dog,1.0"},
                                                "a photo of a dog", "")
    out = capsys.readouterr().out
    assert "not found in text" in out and "not a single color" in out      # reference warnings :234, :271
    assert float(cond["CROSS_ATTENTION_WEIGHT_4096"].abs().sum()) == 0.0


def test_cfg_combine_bit_exact(gpu_device):
    from pww_hip import ops
    g = torch.Generator().manual_seed(0)
    for dtype in (torch.float16, torch.bfloat16):
        c, u = torch.randn(2, 4, 64, 64, generator=g).to(dtype), torch.randn(2, 4, 64, 64, generator=g).to(dtype)
        want = O.cfg_combine(c.float(), u.float(), 7.5)
        got = ops.cfg_combine(c.to(gpu_device), u.to(gpu_device), 7.5).cpu()
        assert torch.equal(got, want)

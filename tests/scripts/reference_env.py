"""Run a reference-style runner script (runner.py / runner_inpaint.py of cloneofsimo/paint-with-words-sd) against THIS
package, in an interpreter where `diffusers`, `transformers` and `dotenv` -- absent from the offline image -- are
stand-in modules whose `from_pretrained` loaders return the seeded random-init stand-ins of sd_standin (1/8-width UNet).
TEST INFRASTRUCTURE: the product never imports this.

    python tests/scripts/reference_env.py <script.py>        (cwd = a directory holding the script's `contents/`)
"""
import os
import runpy
import sys
import types

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "paint-with-words-sd_amd"))


def install_stand_in_modules(inpaint=False):
    import torch
    import sd_standin as S

    cfg = dict(S.TINY_CONFIG, in_channels=9) if inpaint else S.TINY_CONFIG

    def loader(build):
        class _Loader:
            @classmethod
            def from_pretrained(cls, path, subfolder=None, torch_dtype=None, **kwargs):
                return build(torch_dtype or torch.float32)
        return _Loader

    diffusers = types.ModuleType("diffusers")
    diffusers.AutoencoderKL = loader(lambda dt: S.TinyVAE(4, seed=1236).to(dt))
    diffusers.UNet2DConditionModel = loader(lambda dt: S.build_unet(cfg, seed=1234, dtype=dt, device="cpu", qk_gain=2.0))
    diffusers.LMSDiscreteScheduler = S.LMSDiscreteScheduler
    transformers = types.ModuleType("transformers")
    transformers.CLIPTextModel = loader(lambda dt: S.TinyTextEncoder(cfg["cross_attention_dim"], seed=1235))
    transformers.CLIPTokenizer = loader(lambda dt: S.HashTokenizer())
    dotenv = types.ModuleType("dotenv")
    dotenv.load_dotenv = lambda *a, **k: False
    sys.modules.update(diffusers=diffusers, transformers=transformers, dotenv=dotenv)


if __name__ == "__main__":
    script = sys.argv[1]
    install_stand_in_modules(inpaint="inpaint" in os.path.basename(script))
    sys.argv = [script] + sys.argv[2:]
    runpy.run_path(script, run_name="__main__")

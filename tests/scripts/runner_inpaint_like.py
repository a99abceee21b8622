"""What a user of the reference's inpainting entry script does (its runner_inpaint.py:40-92): import the package by the
reference's names, run paint_with_words_inpaint WITHOUT pre-loaded modules (so pww_load_tools runs) and save the PIL result --
then the branch that script keeps behind `use_pipeline`: PaintWithWord_StableDiffusionInpaintPipeline called with the
reference's keyword list (paint_with_words_inpaint.py:340-575) plus a callback. Run through tests/scripts/reference_env.py."""
import json
import math

import dotenv
from PIL import Image

from paint_with_words import paint_with_words_inpaint, PaintWithWord_StableDiffusionInpaintPipeline, pww_load_tools
import torch

SETTINGS = {
    "color_context": {
        (7, 9, 182): "aurora,0.5",
        (136, 178, 92): "full moon,1.5",
        (51, 193, 217): "mountains,0.4",
        (61, 163, 35): "a half-frozen lake,0.3",
        (89, 102, 255): "boat,2.0",
    },
    "color_map_img_path": "contents/aurora_1.png",
    "input_prompt": "A digital painting of a half-frozen lake near mountains under a full moon and aurora. A boat is in the middle of the lake. Highly detailed.",
    "img_path": "contents/init.png",
    "mask_path": "contents/moon_mask.png",
}

if __name__ == "__main__":
    dotenv.load_dotenv()
    color_map_image = Image.open(SETTINGS["color_map_img_path"]).convert("RGB")
    init_image = Image.open(SETTINGS["img_path"]).convert("RGB")
    mask_image = Image.open(SETTINGS["mask_path"])
    wf = lambda w, sigma, qk: 0.15 * w * math.log(1 + sigma) * qk.max()   # noqa: E731  (runner_inpaint.py:87)

    img = paint_with_words_inpaint(color_context=dict(SETTINGS["color_context"]), color_map_image=color_map_image, init_image=init_image,
                                   mask_image=mask_image, input_prompt=SETTINGS["input_prompt"], num_inference_steps=20, guidance_scale=7.5,
                                   device="cuda:0", seed=81, weight_function=wf, strength=1.0)
    img.save("contents/output_inpaint_function_api.png")

    tools = pww_load_tools("cuda:0", hf_model_path="runwayml/stable-diffusion-inpainting")
    pipe = PaintWithWord_StableDiffusionInpaintPipeline(vae=tools[0], text_encoder=tools[2], tokenizer=tools[3], unet=tools[1], scheduler=tools[4],
                                                        safety_checker=None, feature_extractor=None).to("cuda")
    generator = torch.Generator(device="cuda")
    generator.manual_seed(81)
    calls = []
    out = pipe(prompt=SETTINGS["input_prompt"], image=init_image, color_context=dict(SETTINGS["color_context"]), color_map_image=color_map_image,
               mask_image=mask_image.resize(init_image.size, Image.NEAREST), num_inference_steps=20, guidance_scale=7.5, seed=81, weight_function=wf,
               eta=1.0, generator=generator, height=init_image.size[1], width=init_image.size[0],
               callback=lambda i, t, latents: calls.append((int(i), float(t), tuple(latents.shape), bool(torch.isfinite(latents).all()))),
               callback_steps=3)
    out.images[0].save("contents/output_inpaint_pipeline.png")
    json.dump(calls, open("contents/callback_calls.json", "w"))
    try:
        pipe(prompt=SETTINGS["input_prompt"], image=init_image, color_context=dict(SETTINGS["color_context"]), color_map_image=color_map_image,
             mask_image=mask_image.resize(init_image.size, Image.NEAREST), num_inference_steps=2, height=256, width=256)
        json.dump({"raised": False}, open("contents/size_mismatch.json", "w"))
    except ValueError as e:
        json.dump({"raised": True, "message": str(e)}, open("contents/size_mismatch.json", "w"))

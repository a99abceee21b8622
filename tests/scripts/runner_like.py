"""What a user of the reference does (its runner.py): import the package by the reference's names, paint one image
from a color map + color_context + prompt WITHOUT pre-loaded modules (so pww_load_tools runs), save the PIL result.
Then the same request through the pipeline class and the batched entry point. Run through tests/scripts/reference_env.py."""
import math

import dotenv
from PIL import Image

from paint_with_words import paint_with_words, paint_with_words_batch, PaintWithWord_StableDiffusionPipeline, pww_load_tools

REQUEST = {
    "color_context": {(0, 0, 0): "cat,1.0", (255, 255, 255): "dog,1.0", (13, 255, 0): "tree,1.5", (90, 206, 255): "sky,0.2",
                      (74, 18, 1): "ground,0.2"},
    "map": "contents/example_input.png",
    "prompt": "realistic photo of a dog, cat, tree, with beautiful sky, on sandy ground",
}

if __name__ == "__main__":
    dotenv.load_dotenv()
    color_map = Image.open(REQUEST["map"]).convert("RGB")
    image = paint_with_words(color_context=dict(REQUEST["color_context"]), color_map_image=color_map, input_prompt=REQUEST["prompt"],
                             num_inference_steps=30, guidance_scale=7.5, device="cuda:0",
                             weight_function=lambda w, sigma, qk: 0.4 * w * math.log(1 + sigma) * qk.max())
    image.save("contents/output_function_api.png")

    tools = pww_load_tools("cuda:0", hf_model_path="CompVis/stable-diffusion-v1-4")
    pipe = PaintWithWord_StableDiffusionPipeline(vae=tools[0], text_encoder=tools[2], tokenizer=tools[3], unet=tools[1], scheduler=tools[4],
                                                 safety_checker=None, feature_extractor=None).to("cuda:0")
    out = pipe(color_context=dict(REQUEST["color_context"]), color_map_image=color_map, prompt=REQUEST["prompt"], num_inference_steps=30,
               guidance_scale=7.5, weight_function=lambda w, sigma, qk: 0.4 * w * math.log(1 + sigma) * qk.max()).images[0]
    out.save("contents/output_pipeline.png")

    several = paint_with_words_batch(dict(REQUEST["color_context"]), color_map, REQUEST["prompt"], seeds=[0, 1, 2], num_inference_steps=30,
                                     guidance_scale=7.5, device="cuda:0", preloaded_utils=tools,
                                     weight_function=lambda w, sigma, qk: 0.4 * w * math.log(1 + sigma) * qk.max())
    for i, im in enumerate(several):
        im.save("contents/output_batch_%d.png" % i)

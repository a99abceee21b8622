"""fig_from_settings (reference paint_with_words/utils.py:10-85): side-by-side figure of the color
map and the generated image with the region legend. Presentation only -- outside the hot path."""
from PIL import Image, ImageDraw


def fig_from_settings(settings, output_img=None, legend_width=256, row=28):
    cmap = Image.open(settings["color_map_img_path"]).convert("RGB") if isinstance(settings.get("color_map_img_path"), str) \
        else settings["color_map_image"]
    out = output_img if output_img is not None else Image.open(settings["output_img_path"]).convert("RGB")
    w, h = cmap.size
    fig = Image.new("RGB", (w * 2 + legend_width, h), (255, 255, 255))
    fig.paste(cmap, (0, 0))
    fig.paste(out.resize((w, h)), (w, 0))
    draw = ImageDraw.Draw(fig)
    for i, (color, text) in enumerate(settings["color_context"].items()):
        if isinstance(color, str):
            color = tuple(int(color[j:j + 2], 16) for j in (1, 3, 5))
        y = 8 + i * row
        draw.rectangle([2 * w + 8, y, 2 * w + 8 + row - 8, y + row - 8], fill=tuple(color), outline=(0, 0, 0))
        draw.text((2 * w + 8 + row, y + 4), text, fill=(0, 0, 0))
    draw.text((2 * w + 8, h - 40), settings.get("input_prompt", "")[:60], fill=(0, 0, 0))
    return fig

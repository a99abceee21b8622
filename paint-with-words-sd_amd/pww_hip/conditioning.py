"""Conditioning builder: color map + color_context + prompt -> the two encoder_hidden_states dicts of
the PwW protocol (reference paint_with_words/paint_with_words.py:207-388), with the per-resolution
token weight maps produced ON DEVICE by pww_mask_build instead of the reference's Python loops over
F.interpolate (:247-276).

Host side (strings, token matching, the region table) stays Python, as in the reference; function
names follow the reference's so the call sites read the same.
"""
import numpy as np
import torch
import torch.nn.functional as F  # noqa: F401  (region-seed masks: _get_binary_mask)

import os

from . import ops

COMPACT_BIAS = os.environ.get("PWW_COMPACT_BIAS", "0") == "1"   # see prepare_conditioning: private compact weight maps, opt-in


class PwWContext(dict):
    """The encoder_hidden_states dict of the PwW protocol (:370-386) with entries that are BUILT ON FIRST ACCESS.

    `CROSS_ATTENTION_WEIGHT_ORIG` (:343-345) is 80.7 MB at 512 x 512 and is only ever read on inj_forward's KeyError path
    (:95-101: a layer whose token count has no per-resolution map -- image sizes that are not multiples of 64, a latent sized
    differently from the color map). The reference builds it for every request; here `context["CROSS_ATTENTION_WEIGHT_ORIG"]`
    runs the mask kernel at ratio 1 the first time somebody asks (dict.__missing__), so requests that never take the fallback
    never pay for it, and code that indexes the dict the way the reference does sees no difference. `in`, `.get()` and `.copy()`
    know about the pending entries; `dict(ctx)` / `.items()` see only what has been built."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self._thunks = {}

    def set_lazy(self, key, thunk):
        self._thunks[key] = thunk
        return self

    def pending(self, key):
        return key in self._thunks and not dict.__contains__(self, key)

    def __missing__(self, key):
        thunk = self._thunks.get(key)
        if thunk is None:
            raise KeyError(key)
        value = thunk()
        self[key] = value
        return value

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._thunks

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def copy(self):
        c = PwWContext(self)
        c._thunks = dict(self._thunks)
        return c


def always_round(x):
    """:18-26 (round half up for the non-negative sizes it is applied to)."""
    intx = int(x)
    if intx % 2 == 0:
        return intx if x < intx + 0.5 else intx + 1
    return round(x)


def _extract_seed_and_sigma_from_context(color_context, ignore_seed=-1):
    """:279-297: "text,strength[,seed[,sigma]]" -> strip the tail; mutates `color_context` like the
    reference does (:296). Returns (color_context, {ordinal: seed}, {ordinal: sigma})."""
    extra_seeds, extra_sigmas = {}, {}
    for i, (k, ctx) in enumerate(color_context.items()):
        parts = ctx.split(",")
        if len(parts) > 2:
            try:
                seed = int(parts[-2])
                sigma = float(parts[-1])
                parts = parts[:-2]
                extra_sigmas[i] = sigma
            except ValueError:
                seed = int(parts[-1])
                parts = parts[:-1]
            if seed != ignore_seed:
                extra_seeds[i] = seed
        color_context[k] = ",".join(parts)
    return color_context, extra_seeds, extra_sigmas


def _parse_regions(color_context, tokenizer):
    """Host half of _image_context_seperator (:218-230): [(token_ids, (r, g, b), strength)]."""
    table = []
    for color, v in color_context.items():
        fields = v.split(",")
        strength = float(fields[-1])
        text = ",".join(fields[:-1])
        ids = tokenizer(text, max_length=tokenizer.model_max_length, truncation=True)["input_ids"][1:-1]
        if isinstance(color, str):
            color = (int(color[1:3], 16), int(color[3:5], 16), int(color[5:7], 16))
        table.append((list(ids), tuple(int(c) for c in color), strength))
    return table


def _column_lists(table, token_lis, ratio_tag="8"):
    """For every prompt position the region ordinals accumulated into it, in the reference's order
    (:257-268); warns like :270-271 when a phrase does not occur in the prompt."""
    cols = [[] for _ in token_lis]
    for r, (ids, _, _) in enumerate(table):
        L = len(ids)
        found = False
        for idx in range(len(token_lis)):
            if token_lis[idx: idx + L] == ids:
                found = True
                for c in range(idx, min(idx + L, len(token_lis))):
                    cols[c].append(r)
        if not found:
            print(f"Warning ratio {ratio_tag} : tokens {ids} not found in text")
    return cols


def gaussian_blur_mask(mask, sigma, ksize=39):
    """_blur_image_mask (:307-312): torchvision GaussianBlur(39x39, sigma) semantics on the device mask
    (pww_gauss_blur: two 1-D passes, reflect padding, fp64 accumulation)."""
    return ops.gauss_blur(mask, sigma, ksize)


def build_weight_maps(color_map_rgb, table, token_lis, device, extra_sigmas=None, with_orig=False):
    """RGB map -> {N_8: [N, T], ...} fp32 device tensors, keyed like :370-377 -- ONE launch for the four resolutions -- plus
    "ORIG_THUNK": a callable that builds the [H, W, T] ratio-1 map on demand (with_orig=True: built now, key "ORIG").
    color_map_rgb: uint8 numpy / tensor [H, W, 3]."""
    rgb = torch.as_tensor(np.ascontiguousarray(color_map_rgb), dtype=torch.uint8).to(device)
    H, W = rgb.shape[:2]
    cols = _column_lists(table, token_lis)
    regions = [(c[0], c[1], c[2], s) for (_, c, s) in table]
    ratios = (8, 16, 32, 64)
    blurred = {}
    if extra_sigmas:
        # blurred regions need float masks (:338-340): build them on device, blur (pww_gauss_blur), then accumulate
        print("Use extra sigma to smooth mask", extra_sigmas)
        masks = []
        for r, (_, c, s) in enumerate(table):
            m = (rgb == torch.tensor(c, dtype=torch.uint8, device=device)).all(dim=-1).float() * s
            if r in extra_sigmas:
                m = blurred[r] = gaussian_blur_mask(m, extra_sigmas[r])
            masks.append(m)
        masks = torch.stack(masks)
        outs = ops.mask_build_f32(masks, cols, ratios)
        orig_thunk = lambda: ops.mask_build_f32(masks, cols, (1,))[1].reshape(H, W, len(token_lis))      # noqa: E731
    else:
        outs = ops.mask_build(rgb, regions, cols, ratios)
        orig_thunk = lambda: ops.mask_build(rgb, regions, cols, (1,))[1].reshape(H, W, len(token_lis))    # noqa: E731
    maps = {}
    for r in ratios:
        maps[always_round(H / r) * always_round(W / r)] = outs[r]
    maps["ORIG_THUNK"] = orig_thunk
    if with_orig:
        maps["ORIG"] = orig_thunk()
    maps["_BLURRED"] = blurred      # region ordinal -> blurred float mask (region seeding thresholds these, :300-304)
    maps["_COLS"] = [c for c, lst in enumerate(cols) if lst]     # prompt positions that carry any region weight (:257-268)
    return maps


def _warn_missing_colors(color_map_rgb, table):
    """:233-234."""
    img = np.asarray(color_map_rgb)
    for _, color, _ in table:
        if not (img == np.array(color, dtype=img.dtype)).all(axis=-1).any():
            print(f"Warning : not a single color {color} not found in image")


def _encode_text_color_inputs(text_encoder, tokenizer, device, color_map_image, color_context, input_prompt,
                              unconditional_input_prompt, dtype=None, use_sigma=True):
    """:315-388 with the weight maps built by the HIP mask kernel. Returns
    (extra_seeds, seperated_word_contexts, encoder_hidden_states, uncond_encoder_hidden_states);
    `seperated_word_contexts` is (region table [(token_ids, (r,g,b), strength)], rgb, {ordinal: blurred mask}) (the
    reference returns full-resolution float masks here; the only consumer, region seeding :451, gets what it
    needs from the table + color map, plus the blurred masks of the regions that carry a sigma)."""
    text_input = tokenizer([input_prompt], padding="max_length", max_length=tokenizer.model_max_length,
                           truncation=True, return_tensors="pt")
    color_context, extra_seeds, extra_sigmas = _extract_seed_and_sigma_from_context(color_context)
    if not use_sigma:      # the pipeline classes parse the sigma tail and drop it (reference :574): no blur there
        extra_sigmas = {}
    if color_map_image is None:
        # the reference's _image_context_seperator(None, ...) (:239-243): one dummy region over a 512 x 512 all-zero map --
        # plain Stable Diffusion (the pipeline class's default call, `pipe(prompt)`)
        rgb, height, width, table = None, 512, 512, []
    else:
        rgb = np.array(color_map_image.convert("RGB")) if hasattr(color_map_image, "convert") else np.asarray(color_map_image)
        height, width = rgb.shape[:2]
        table = _parse_regions(color_context, tokenizer)
    token_lis = text_input["input_ids"][0].tolist()
    keys = [always_round(height / r) * always_round(width / r) for r in (8, 16, 32, 64)]
    if table:
        _warn_missing_colors(rgb, table)
        maps = build_weight_maps(rgb, table, token_lis, device, extra_sigmas)
        blurred = maps.pop("_BLURRED")
        nz_cols = maps.pop("_COLS")
    else:   # empty color_context (:242-243): all-zero maps
        maps = {k: torch.zeros((k, len(token_lis)), dtype=torch.float32, device=device) for k in keys}
        maps["ORIG_THUNK"] = lambda: torch.zeros((height, width, len(token_lis)), dtype=torch.float32, device=device)
        blurred = {}
        nz_cols = None

    cond_embeddings = text_encoder(text_input.input_ids.to(device))[0]
    uncond_input = tokenizer([unconditional_input_prompt], padding="max_length",
                             max_length=text_input.input_ids.shape[-1], return_tensors="pt")
    uncond_embeddings = text_encoder(uncond_input.input_ids.to(device))[0]
    if dtype is not None:
        cond_embeddings, uncond_embeddings = cond_embeddings.to(dtype), uncond_embeddings.to(dtype)

    # CROSS_ATTENTION_WEIGHT_ORIG (:343-345, :372) is built when somebody indexes it (PwWContext): only inj_forward's KeyError path does
    encoder_hidden_states = PwWContext({"CONTEXT_TENSOR": cond_embeddings}).set_lazy("CROSS_ATTENTION_WEIGHT_ORIG", maps["ORIG_THUNK"])
    uncond_encoder_hidden_states = {"CONTEXT_TENSOR": uncond_embeddings, "CROSS_ATTENTION_WEIGHT_ORIG": 0}
    for k in keys:
        encoder_hidden_states[f"CROSS_ATTENTION_WEIGHT_{k}"] = maps[k]
        uncond_encoder_hidden_states[f"CROSS_ATTENTION_WEIGHT_{k}"] = 0
    if nz_cols is not None:
        # Private, optional hints for the fused kernel (SURVEY.md 8b "may add private keys, must not require them"): only the
        # prompt positions covered by a region phrase are non-zero in ANY of the maps (5 - 17 of 77 for the shipped examples) --
        # the column bound, and the compact [N, R] + col_idx form of every per-resolution map.
        from .attention import BIAS_COLS, COMPACT_W, COMPACT_IDX
        # (both are kernel-launch geometry, i.e. part of a captured hipGraph's identity: rounded up -- the bound to 16 columns,
        # the compact width to 8 slots with unused ones marked -1 -- so that similar prompts replay the same graph)
        encoder_hidden_states[BIAS_COLS] = ((max(nz_cols) + 16) // 16 * 16) if nz_cols else 16
        # The compact form is OPT-IN (PWW_COMPACT_BIAS=1 / conditioning.COMPACT_BIAS): measured on MI355X it is 0.5 - 6 us SLOWER per
        # launch than the dense LDS tile bounded by BIAS_COLS in every UNet shape (profiles/r03_cross_timeline.md: the launch is bound
        # by the statistic hand-off and the score passes, not by the bias bytes), and building it costs 5 extra kernels per request.
        if COMPACT_BIAS and 1 <= len(nz_cols) <= ops.COMPACT_MAX_R:
            R = (len(nz_cols) + 7) // 8 * 8
            idx = torch.tensor(nz_cols, dtype=torch.int64, device=device)
            pad = torch.full((R - len(nz_cols),), -1, dtype=torch.int32, device=device)
            encoder_hidden_states[COMPACT_IDX] = torch.cat([idx.to(torch.int32), pad])
            for k in keys:
                wc = maps[k].new_zeros((maps[k].shape[0], R))
                wc[:, :len(nz_cols)] = maps[k].index_select(1, idx)
                encoder_hidden_states[COMPACT_W + str(k)] = wc
    return extra_seeds, (table, rgb, blurred), encoder_hidden_states, uncond_encoder_hidden_states


def _get_binary_mask(region_info, extra_seeds, dtype, size):
    """:300-304: per seeded region, (mask > 0) bilinearly resized (align_corners=False) to `size`. The reference
    thresholds the masks AFTER _blur_image_mask has replaced the blurred ones in place (:338-340), so a region given
    as "text,strength,seed,sigma" seeds the dilated area its blur reaches."""
    table, rgb = region_info[0], region_info[1]
    blurred = region_info[2] if len(region_info) > 2 else {}
    img = torch.as_tensor(rgb)
    out = []
    for k in extra_seeds.keys():
        _, color, strength = table[k]
        if k in blurred:
            m = (blurred[k].cpu() > 0).to(dtype)
        else:
            m = ((img == torch.tensor(color, dtype=img.dtype)).all(dim=-1).float() * strength > 0).to(dtype)
        out.append(F.interpolate(m[None, None], size=size, mode="bilinear"))
    return out

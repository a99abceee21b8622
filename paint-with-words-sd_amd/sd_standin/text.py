"""Tokenizer / text-encoder stand-ins.

The reference uses the CLIP tokenizer and text model of the SD checkpoint
(paint_with_words/paint_with_words.py:170-171); neither vocabulary nor weights exist offline. The
hot path only needs (a) token ids framed BOS ... EOS and padded to 77 so that region phrases can be
matched against the prompt (:222-227, :251-260) and (b) a [1, 77, ctx_dim] embedding (:360-368).
"""
import re
import zlib
from types import SimpleNamespace

import torch
import torch.nn as nn

BOS, EOS = 49406, 49407


class _Encoding(dict):
    """Mapping with attribute access, like transformers' BatchEncoding (:251 uses ["input_ids"], :360 .input_ids)."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class HashTokenizer:
    """Deterministic word-level tokenizer: lower-cased words and punctuation marks hash to ids in
    [1000, 49000); BOS=49406, EOS=49407, padding with EOS (the CLIP conventions)."""
    model_max_length = 77

    _pat = re.compile(r"[a-z0-9]+|[^\sa-z0-9]", re.IGNORECASE)

    def _ids(self, text):
        return [1000 + zlib.crc32(w.lower().encode("utf-8")) % 48000 for w in self._pat.findall(text)]

    def __call__(self, text, padding=False, max_length=None, truncation=False, return_tensors=None):
        batched = isinstance(text, (list, tuple))
        texts = list(text) if batched else [text]
        max_length = max_length or self.model_max_length
        out = []
        for t in texts:
            ids = self._ids(t)
            if truncation or True:
                ids = ids[: max_length - 2]
            ids = [BOS] + ids + [EOS]
            if padding == "max_length":
                ids = ids + [EOS] * (max_length - len(ids))
            out.append(ids)
        if return_tensors == "pt":
            return _Encoding(input_ids=torch.tensor(out, dtype=torch.long))
        return _Encoding(input_ids=out if batched else out[0])


class TinyTextEncoder(nn.Module):
    """Random-init embedding + position + LayerNorm text encoder: returns ([B, 77, ctx_dim],)."""

    def __init__(self, ctx_dim=768, vocab=49408, max_len=77, seed=1235):
        super().__init__()
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        try:
            self.tok = nn.Embedding(vocab, ctx_dim)
            self.pos = nn.Parameter(torch.randn(max_len, ctx_dim) * 0.1)
            self.norm = nn.LayerNorm(ctx_dim)
        finally:
            torch.random.set_rng_state(state)
        self.requires_grad_(False)

    @property
    def dtype(self):
        return self.pos.dtype

    def forward(self, input_ids):
        x = self.tok(input_ids) + self.pos[: input_ids.shape[1]]
        return (self.norm(x),)


def build_clip_text_encoder(ctx_dim=768, seed=1235, device="cpu", dtype=torch.float32):
    """Random-init transformers CLIPTextModel with SD1.x (ctx_dim 768) / SD2.x (1024) sizes; falls back to
    TinyTextEncoder if transformers cannot build it offline."""
    try:
        from transformers import CLIPTextConfig, CLIPTextModel
        layers, heads, inter = (12, 12, 3072) if ctx_dim == 768 else (23, 16, 4096)
        cfg = CLIPTextConfig(vocab_size=49408, hidden_size=ctx_dim, intermediate_size=inter, num_hidden_layers=layers,
                             num_attention_heads=heads, max_position_embeddings=77)
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        try:
            enc = CLIPTextModel(cfg)
        finally:
            torch.random.set_rng_state(state)
        return enc.to(device=device, dtype=dtype).eval().requires_grad_(False)
    except Exception:
        return TinyTextEncoder(ctx_dim, seed=seed).to(device=device, dtype=dtype).eval()

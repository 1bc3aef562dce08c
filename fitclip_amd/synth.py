"""Seeded synthetic weights / clips / captions for the encode-and-score path.

There is no network in the build or bench environment (no CLIP checkpoint, no BPE vocabulary, no WebVid), so every
tensor is generated from a counter-based integer hash.  The generator is plain numpy integer arithmetic followed by ONE
correctly-rounded float32 multiply, so the container that writes the golden fixtures and the GPU box that replays them
produce bit-identical tensors without shipping the 598 MB of weights.

Shapes and init scales follow the reference:
  * dims: /root/reference/config/encoder/clip_from_scratch_vit_b_16.yaml:5-16
  * init: /root/reference/aligner/encoder/slip.py:438-452 (attn D^-1/2, proj D^-1/2 (2L)^-1/2, fc (2D)^-1/2, ...)
  * batch layout {"video": f32[B,F,3,H,W], "text": {"input_ids": int64[B,77]}}:
    /root/reference/aligner/data/video_dataset.py:102-112
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, asdict
from typing import Dict, Tuple

import numpy as np

def _fmix32(x: np.ndarray) -> np.ndarray:
    """murmur3 finaliser on uint32 lanes (array arithmetic wraps modulo 2**32)."""
    x = x ^ (x >> np.uint32(16))
    x = x * np.uint32(0x85EBCA6B)
    x = x ^ (x >> np.uint32(13))
    x = x * np.uint32(0xC2B2AE35)
    x = x ^ (x >> np.uint32(16))
    return x


def hash_u32(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """32-bit hash of (seed, stream, idx).  `idx` is any integer array with values < 2**32.  Returns uint32."""
    idx = np.asarray(idx).astype(np.uint32)
    key = np.uint32((seed * 0x9E3779B1 + stream * 0x7F4A7C15 + 0x165667B1) & 0xFFFFFFFF)
    x = idx * np.uint32(0x9E3779B1) + key
    x = _fmix32(x)
    x = _fmix32(x ^ np.uint32((stream * 0x27D4EB2F + seed) & 0xFFFFFFFF))
    return x


_IH_STD = float(np.sqrt(4.0 * (65536.0 ** 2 - 1.0) / 12.0))  # std of a sum of four uniform 16-bit integers


def hash_normal(seed: int, stream: int, n: int, std: float = 1.0, offset: int = 0) -> np.ndarray:
    """`n` float32 samples, approximately N(0, std^2) (Irwin-Hall of four 16-bit fields, |x| <= 3.46 std).

    Exact integer sum -> exact int->float32 -> one IEEE float32 multiply: bit-reproducible everywhere.
    """
    out = np.empty(n, dtype=np.float32)
    scale = np.float32(std / _IH_STD)
    step = 1 << 22
    for s in range(0, n, step):
        e = min(n, s + step)
        idx = np.arange(offset + s, offset + e, dtype=np.uint64).astype(np.uint32)
        h1 = hash_u32(seed, 2 * stream, idx)
        h2 = hash_u32(seed, 2 * stream + 1, idx)
        tot = (h1 & np.uint32(0xFFFF)) + (h1 >> np.uint32(16)) + (h2 & np.uint32(0xFFFF)) + (h2 >> np.uint32(16))
        out[s:e] = (tot.astype(np.int32) - np.int32(131070)).astype(np.float32) * scale
    return out


def hash_uniform_int(seed: int, stream: int, n: int, lo: int, hi: int) -> np.ndarray:
    """`n` int64 samples in [lo, hi] (inclusive)."""
    h = hash_u32(seed, stream, np.arange(n, dtype=np.uint64))
    return (lo + (h % np.uint32(hi - lo + 1)).astype(np.int64)).astype(np.int64)


@dataclass(frozen=True)
class ClipDims:
    """Architecture of the dual encoder (defaults = CLIP ViT-B/16)."""
    embed_dim: int = 512
    image_resolution: int = 224
    vision_layers: int = 12
    vision_width: int = 768
    vision_patch_size: int = 16
    context_length: int = 77
    vocab_size: int = 49408
    transformer_width: int = 512
    transformer_heads: int = 8
    transformer_layers: int = 12

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64  # OpenAI CLIP: heads = width // 64

    @property
    def grid(self) -> int:
        return self.image_resolution // self.vision_patch_size

    @property
    def vision_tokens(self) -> int:
        return self.grid * self.grid + 1

    def to_dict(self) -> Dict[str, int]:
        return asdict(self)


VIT_B_16 = ClipDims()
# A structurally identical miniature (same head size 64, same patch 16) used by the fast parity tests.
TINY = ClipDims(embed_dim=128, image_resolution=64, vision_layers=2, vision_width=256, vision_patch_size=16,
                context_length=16, vocab_size=1024, transformer_width=128, transformer_heads=2, transformer_layers=2)

SOT_OFFSET = 2  # SOT = vocab-2, EOT = vocab-1 (49406 / 49407 for the CLIP BPE vocabulary)


def parameter_shapes(d: ClipDims, with_logit_scale: bool = False) -> "OrderedDict[str, Tuple[int, ...]]":
    """OpenAI-CLIP parameter names -> shapes, in `named_parameters()` order (301 tensors for ViT-B/16)."""
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["positional_embedding"] = (d.context_length, d.transformer_width)
    s["text_projection"] = (d.transformer_width, d.embed_dim)
    if with_logit_scale:
        s["logit_scale"] = ()
    vw, p = d.vision_width, d.vision_patch_size
    s["visual.class_embedding"] = (vw,)
    s["visual.positional_embedding"] = (d.vision_tokens, vw)
    s["visual.proj"] = (vw, d.embed_dim)
    s["visual.conv1.weight"] = (vw, 3, p, p)
    s["visual.ln_pre.weight"] = (vw,)
    s["visual.ln_pre.bias"] = (vw,)

    def blocks(prefix: str, width: int, layers: int) -> None:
        for i in range(layers):
            b = f"{prefix}.resblocks.{i}"
            s[f"{b}.attn.in_proj_weight"] = (3 * width, width)
            s[f"{b}.attn.in_proj_bias"] = (3 * width,)
            s[f"{b}.attn.out_proj.weight"] = (width, width)
            s[f"{b}.attn.out_proj.bias"] = (width,)
            s[f"{b}.ln_1.weight"] = (width,)
            s[f"{b}.ln_1.bias"] = (width,)
            s[f"{b}.mlp.c_fc.weight"] = (4 * width, width)
            s[f"{b}.mlp.c_fc.bias"] = (4 * width,)
            s[f"{b}.mlp.c_proj.weight"] = (width, 4 * width)
            s[f"{b}.mlp.c_proj.bias"] = (width,)
            s[f"{b}.ln_2.weight"] = (width,)
            s[f"{b}.ln_2.bias"] = (width,)

    blocks("visual.transformer", vw, d.vision_layers)
    s["visual.ln_post.weight"] = (vw,)
    s["visual.ln_post.bias"] = (vw,)
    blocks("transformer", d.transformer_width, d.transformer_layers)
    s["token_embedding.weight"] = (d.vocab_size, d.transformer_width)
    s["ln_final.weight"] = (d.transformer_width,)
    s["ln_final.bias"] = (d.transformer_width,)
    return s


def _init_std(name: str, d: ClipDims) -> Tuple[float, float]:
    """(std, mean) of the synthetic init for parameter `name` (slip.py:438-452 scheme; biases / LN perturbed so
    that every bias and affine path is exercised by the parity tests)."""
    visual = name.startswith("visual.")
    width = d.vision_width if visual else d.transformer_width
    layers = d.vision_layers if visual else d.transformer_layers
    if name.endswith("in_proj_weight"):
        return width ** -0.5, 0.0
    if name.endswith("out_proj.weight") or name.endswith("c_proj.weight"):
        return (width ** -0.5) * ((2 * layers) ** -0.5), 0.0
    if name.endswith("c_fc.weight"):
        return (2 * width) ** -0.5, 0.0
    if name.endswith("token_embedding.weight"):
        return 0.02, 0.0
    if name.endswith("positional_embedding"):
        return 0.01, 0.0
    if name == "visual.class_embedding":
        return width ** -0.5, 0.0
    if name in ("visual.proj", "text_projection"):
        return width ** -0.5, 0.0
    if name == "visual.conv1.weight":
        # 4x the 1/sqrt(fan_in) scale: keeps the image content (not the shared CLS/pos terms) dominant so that
        # embeddings of different synthetic clips are well separated ("measured caveat").
        return 4.0 * (3 * d.vision_patch_size ** 2) ** -0.5, 0.0
    if ".ln_" in name or name.startswith("ln_final"):
        return (0.1, 1.0) if name.endswith("weight") else (0.05, 0.0)
    if name.endswith("bias"):
        return 0.02, 0.0
    raise KeyError(name)


def make_state_dict(d: ClipDims = VIT_B_16, seed: int = 42, variant: int = 0) -> "OrderedDict[str, np.ndarray]":
    """Seeded float32 state dict with OpenAI-CLIP names.  `variant` selects an independent stream family."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, (name, shape) in enumerate(parameter_shapes(d).items()):
        std, mean = _init_std(name, d)
        n = int(np.prod(shape)) if shape else 1
        v = hash_normal(seed, 1000 * (variant + 1) + k, n, std)
        if mean:
            v = v + np.float32(mean)
        out[name] = v.reshape(shape)
    return out


def perturbed_state_dict(base: "OrderedDict[str, np.ndarray]", d: ClipDims, seed: int, rel: float = 0.05
                         ) -> "OrderedDict[str, np.ndarray]":
    """A "student": base + rel * std(param) * noise.  Used as model2 of the WiSE ensemble (BASELINE config 3)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, (name, w) in enumerate(base.items()):
        std, _ = _init_std(name, d)
        noise = hash_normal(seed, 500000 + k, w.size, rel * std).reshape(w.shape)
        out[name] = (w + noise).astype(np.float32)
    return out


def make_video(n_clips: int, n_frames: int, d: ClipDims = VIT_B_16, seed: int = 42, first_clip: int = 0
               ) -> np.ndarray:
    """float32 [n_clips, n_frames, 3, H, W] in the post-normalisation pixel range of the CLIP eval transform.

    Each clip is a clip-specific low-frequency pattern (shared by its frames) plus per-frame white noise, so the
    frames of one clip resemble each other and different clips do not.  Clip `i` only depends on `first_clip + i`,
    which lets every rank of a sharded run generate exactly its own slice.
    """
    H = d.image_resolution
    g = 8  # low-frequency grid
    out = np.empty((n_clips, n_frames, 3, H, H), dtype=np.float32)
    rep = H // g
    for i in range(n_clips):
        cid = first_clip + i
        low = hash_normal(seed, 7, 3 * g * g, 1.0, offset=cid * 3 * g * g).reshape(3, g, g)
        base = np.repeat(np.repeat(low, rep, axis=1), rep, axis=2)
        noise = hash_normal(seed, 8, n_frames * 3 * H * H, 0.5, offset=cid * n_frames * 3 * H * H)
        out[i] = np.clip(base[None] + noise.reshape(n_frames, 3, H, H), -2.5, 2.5)
    return out


def make_text(n_texts: int, d: ClipDims = VIT_B_16, seed: int = 42, first_text: int = 0,
              all_random: bool = False) -> np.ndarray:
    """int64 [n_texts, context_length] token ids shaped like `clip.tokenize(truncate=True)` output:
    SOT, l random tokens, EOT (the max id, so `argmax` finds it), zero padding.

    `all_random` fills every position with random ids < EOT (no EOT at all), exercising the first-max tie
    semantics of `ids.argmax(-1)` (slip.py:478).
    """
    L = d.context_length
    sot, eot = d.vocab_size - SOT_OFFSET, d.vocab_size - 1
    idx = np.arange(first_text * L, (first_text + n_texts) * L, dtype=np.uint64)
    toks = (hash_u32(seed, 11, idx) % np.uint32(sot)).astype(np.int64).reshape(n_texts, L)
    if all_random:
        return toks
    lens = (hash_u32(seed, 12, np.arange(first_text, first_text + n_texts, dtype=np.uint64))
            % np.uint32(max(1, L - 5))).astype(np.int64) + 3  # 3 .. L-3 content tokens
    lens = np.minimum(lens, L - 2)
    pos = np.arange(L)[None, :]
    ids = np.where(pos <= lens[:, None], toks, 0)
    ids[:, 0] = sot
    ids[np.arange(n_texts), lens + 1] = eot
    return ids.astype(np.int64)

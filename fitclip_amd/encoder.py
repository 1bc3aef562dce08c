"""`ClipVideoTextEncoder`: the reference's CLIP encoder plugin (`aligner/encoder/clip_video_text_encoder.py:68-146`)
on the HIP path.  Constructor `(model, num_frames=4)`, `encode_video` / `encode_text` / `forward(video, text)`, the
tokenizer / transform / frame-sampler factories and `should_pad_batch` have the reference's meaning, so it can be the
`_target_` of `config/encoder/clip.yaml` and the `model1` / `model2` of `config/encoder/wise.yaml`.
"""
from __future__ import annotations

import os
import zlib
from typing import Iterable, Iterator, Mapping, Optional

import torch
import torch.nn.functional as F

from . import ops
from .clip_model import CLIP
from .samplers import FrameSampler, RandomFromUniformIntervalsFrameSampler, UniformFrameSampler
from .plugin_api import (TYPE_TEXT_INPUT, TYPE_TOKENIZER, TYPE_TRANSFORM, TYPE_VIDEO_INPUT, VideoTextEncoder,
                         float_standard_denormalize)

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)  # clip_video_text_encoder.py:72
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


_OVERLAP_TEXT = os.environ.get("FITCLIP_OVERLAP_TEXT", "1") not in ("0", "")  # A/B switch of the two-stream forward


class HashTokenizer:
    """Stand-in for `clip.tokenize(texts, truncate=True)` (clip_video_text_encoder.py:64-65).

    The CLIP BPE vocabulary file is not available offline, so words are mapped to ids in
    [1, SOT) with a CRC; the FRAMING is the real one: SOT, tokens, EOT (the largest id, which `encode_text` locates
    with argmax), zero padding, truncation that keeps EOT last.
    """

    def __init__(self, context_length: int, vocab_size: int) -> None:
        self.context_length, self.vocab_size = context_length, vocab_size

    def __call__(self, texts: Iterable[str]) -> Mapping[str, torch.Tensor]:
        sot, eot = self.vocab_size - 2, self.vocab_size - 1
        texts = [texts] if isinstance(texts, str) else list(texts)
        ids = torch.zeros((len(texts), self.context_length), dtype=torch.long)
        for i, text in enumerate(texts):
            toks = [sot] + [1 + zlib.crc32(w.encode()) % (sot - 1) for w in text.lower().split()] + [eot]
            if len(toks) > self.context_length:
                toks = toks[:self.context_length]
                toks[-1] = eot
            ids[i, :len(toks)] = torch.tensor(toks)
        return {"input_ids": ids}


class ClipVideoTextEncoder(VideoTextEncoder):
    def __init__(self, model: CLIP, num_frames: int = 4, bpe_path: Optional[str] = None) -> None:
        super().__init__()
        self.model = model
        self.num_frames = num_frames
        self.bpe_path = bpe_path  # local bpe_simple_vocab_16e6.txt.gz; None -> HashTokenizer (framing only)
        self.mean, self.std = CLIP_MEAN, CLIP_STD
        self.overlap_text = _OVERLAP_TEXT  # two-stream forward (see `forward`); plain attribute, can be switched off
        self._side_stream = None  # second HIP stream of the two-stream forward (created on first use)
        self._tokenizer = None    # ClipBpeTokenizer over `bpe_path`, built on first use (one native handle per encoder)
        # Same as the reference (:75-77): the CLIP temperature is unused, drop the parameter so it is not in
        # `named_parameters()` (WiSE) nor in the optimiser.
        if hasattr(self.model, "logit_scale"):
            delattr(self.model, "logit_scale")

    def encode_video(self, video: TYPE_VIDEO_INPUT) -> torch.Tensor:
        """f32 [B, F, 3, H, W] -> f32 [B, E]: every frame through the visual tower, unit-normalise each frame
        embedding, average over the F frames (not re-normalised) - reference :80-89."""
        batch_size = video.shape[0]
        images = video.reshape(-1, *video.shape[2:])
        frame_features = self.model.encode_image(images)
        frames = images.shape[0] // batch_size if batch_size else 1
        return ops.pool_normalize(frame_features, batch_size, frames)

    def encode_video_uint8(self, video: torch.Tensor) -> torch.Tensor:
        """uint8 [B, F, H, W, 3] straight from the decoder -> f32 [B, E]: the eval transform runs on the device
        (`fc_preprocess_u8`), 4x fewer input bytes than float frames and no per-sample CPU transform."""
        b, f = video.shape[:2]
        frames = video.reshape(b * f, *video.shape[2:]).to(self.model._device()).contiguous()
        images = ops.preprocess_u8(frames, self.model.visual.input_resolution, self.mean, self.std)
        return ops.pool_normalize(self.model.encode_image(images), b, f)

    def encode_text(self, text: TYPE_TEXT_INPUT) -> torch.Tensor:
        """{"input_ids": int [B, 77]} -> unit-norm f32 [B, E] - reference :92-94."""
        return ops.l2_normalize(self.model.encode_text(text["input_ids"]))

    def forward(self, video: TYPE_VIDEO_INPUT, text: TYPE_TEXT_INPUT):  # noqa: signature of VideoTextEncoder.forward
        """(encode_video(video), encode_text(text)) - reference video_text_encoder.py:21-22.

        The text tower (2 % of the FLOPs, GEMMs of 150-600 tiles that leave CUs idle) is enqueued on a second HIP stream
        BEFORE the visual tower is enqueued on the caller's stream, so its kernels fill the shadows of the visual
        tower's memory-bound kernels (LayerNorm, attention): -1.4 % step time, identical results.  The caller's stream
        waits for the side stream before returning, so the outputs behave like any other tensor of that stream.
        FITCLIP_OVERLAP_TEXT=0 restores the sequential order."""
        if not self.overlap_text or not video.is_cuda:
            return self.encode_video(video), self.encode_text(text)
        self.model._ensure_ready()  # weight packing (first call / after a weight update) happens on the caller's stream
        main = torch.cuda.current_stream()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=video.device)
        side = self._side_stream
        side.wait_stream(main)
        with torch.cuda.stream(side):
            encoded_text = self.encode_text(text)
        encoded_video = self.encode_video(video)
        main.wait_stream(side)
        encoded_text.record_stream(main)
        return encoded_video, encoded_text

    def get_tokenizer(self) -> TYPE_TOKENIZER:
        if self.bpe_path:
            if self._tokenizer is None:
                from .bpe import ClipBpeTokenizer
                self._tokenizer = ClipBpeTokenizer(self.bpe_path, self.model.context_length)
            return self._tokenizer
        return HashTokenizer(self.model.context_length, self.model.vocab_size)

    def decode_text(self, text: TYPE_TEXT_INPUT) -> Iterator[str]:
        """Reference :100-103: `clip._tokenizer.decode(ids)` per instance - EVERY id of the row goes through the BPE decoder,
        so the string carries `<|startoftext|>`, `<|endoftext|>` and one decoded pad token per padded position, exactly as
        the reference's prediction dumps do.  Accepts the batch mapping (`{"input_ids": [n, L]}`) or an iterable of
        per-instance mappings (what the reference's loop indexes).  Without a vocabulary file (`bpe_path=None`: the
        HashTokenizer is one-way) the ids are printed as `<id>` placeholders, padding dropped."""
        rows = text["input_ids"] if isinstance(text, Mapping) else (instance["input_ids"] for instance in text)
        tokenizer = self.get_tokenizer() if self.bpe_path else None
        for ids in rows:
            ids = [int(t) for t in ids]
            yield tokenizer.decode(ids) if tokenizer is not None else " ".join(f"<{t}>" for t in ids if t != 0)

    def get_train_frame_sampler(self) -> FrameSampler:
        return RandomFromUniformIntervalsFrameSampler(self.num_frames)

    def get_eval_frame_sampler(self) -> FrameSampler:
        return UniformFrameSampler(self.num_frames)

    def _normalize(self, v: torch.Tensor) -> torch.Tensor:
        mean = torch.as_tensor(self.mean, dtype=v.dtype, device=v.device).view(-1, 1, 1)
        std = torch.as_tensor(self.std, dtype=v.dtype, device=v.device).view(-1, 1, 1)
        return (v - mean) / std

    def get_eval_transform(self, dtype: torch.dtype) -> TYPE_TRANSFORM:
        """uint8 [F, H, W, C] -> `dtype` [F, C, R, R]: BHWC->BCHW, /255, bicubic resize of the shorter side to R,
        centre crop, CLIP mean/std (reference :125-133; data-side preprocessing, not part of the timed path)."""
        size = self.model.visual.input_resolution

        def transform(v: torch.Tensor) -> torch.Tensor:
            v = v.permute(0, 3, 1, 2)
            v = v.to(dtype) / 255 if not v.is_floating_point() else v.to(dtype)
            h, w = v.shape[-2:]
            nh, nw = (size, int(size * w / h)) if h <= w else (int(size * h / w), size)  # torchvision Resize(int)
            v = F.interpolate(v, size=(nh, nw), mode="bicubic", align_corners=False)  # torchvision Resize on tensors: no antialias
            top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))  # torchvision CenterCrop
            return self._normalize(v[..., top:top + size, left:left + size])

        return transform

    def get_train_transform(self, dtype: torch.dtype) -> TYPE_TRANSFORM:
        """uint8 [F, H, W, C] -> `dtype` [F, C, R, R], the reference's training augmentation (:113-122): BHWC->BCHW, /255,
        `RandomResizedCropWithRandomInterpolation(R, scale=(0.5, 1.0))` (aligner/transforms.py:56-61 over torchvision's
        RandomResizedCrop: ONE crop box per clip - area fraction uniform in [0.5, 1], log-uniform aspect in [3/4, 4/3], ten
        tries then the centre fallback - resized with bilinear or bicubic interpolation chosen at random, antialias off as
        torchvision does for tensors), `RandomHorizontalFlip` (p = 0.5, the whole clip), CLIP mean/std.  Data-side
        preprocessing of the training step, drawn from torch's global RNG (seeded by the trainer, as in the reference)."""
        import math
        size = self.model.visual.input_resolution

        def crop_box(h: int, w: int):
            area = h * w
            log_lo, log_hi = math.log(3.0 / 4.0), math.log(4.0 / 3.0)
            for _ in range(10):
                target = area * float(torch.empty(1).uniform_(0.5, 1.0))
                aspect = math.exp(float(torch.empty(1).uniform_(log_lo, log_hi)))
                cw, ch = int(round(math.sqrt(target * aspect))), int(round(math.sqrt(target / aspect)))
                if 0 < cw <= w and 0 < ch <= h:
                    return int(torch.randint(0, h - ch + 1, (1,))), int(torch.randint(0, w - cw + 1, (1,))), ch, cw
            ratio = w / h                                            # fallback: the largest centred box inside the bounds
            if ratio < 3.0 / 4.0:
                cw, ch = w, int(round(w / (3.0 / 4.0)))
            elif ratio > 4.0 / 3.0:
                ch, cw = h, int(round(h * (4.0 / 3.0)))
            else:
                cw, ch = w, h
            return (h - ch) // 2, (w - cw) // 2, ch, cw

        def transform(v: torch.Tensor) -> torch.Tensor:
            v = v.permute(0, 3, 1, 2)
            v = v.to(dtype) / 255 if not v.is_floating_point() else v.to(dtype)
            top, left, ch, cw = crop_box(v.shape[-2], v.shape[-1])
            mode = "bilinear" if int(torch.randint(0, 2, (1,))) == 0 else "bicubic"
            v = F.interpolate(v[..., top:top + ch, left:left + cw], size=(size, size), mode=mode, align_corners=False)
            if float(torch.rand(1)) < 0.5:
                v = v.flip(-1)
            return self._normalize(v)

        return transform

    @property
    def should_pad_batch(self) -> bool:
        return True

    def to_bchw(self, t: torch.Tensor) -> torch.Tensor:
        return t

    def denormalize_video_tensor(self, video: TYPE_VIDEO_INPUT) -> torch.Tensor:
        return float_standard_denormalize(video, mean=self.mean, std=self.std)

"""The encoder plugin contract of the reference, restated (interface only).

Mirrors `aligner/encoder/video_encoder.py:14-63` and `aligner/encoder/video_text_encoder.py:15-31`: the Lightning
modules call `encoder(video=..., text=...)`, the data modules call the tokenizer / transform / frame-sampler factories
and `should_pad_batch`.  Same method names, argument meaning and error behaviour (`NotImplementedError` for abstract
methods), so an encoder written against the reference ABC and one written against this one are interchangeable.
"""
from __future__ import annotations

from abc import abstractmethod
from typing import Callable, Iterable, Iterator, Mapping, Optional, Tuple

import torch
from torch import nn

from .samplers import FrameSampler

TYPE_VIDEO_INPUT = torch.Tensor
TYPE_TRANSFORM = Callable[[torch.Tensor], torch.Tensor]
TYPE_TEXT_INPUT = Mapping[str, torch.Tensor]
TYPE_OUTPUT = Tuple[torch.Tensor, torch.Tensor]
TYPE_TOKENIZER = Callable[[Iterable[str]], Mapping[str, torch.Tensor]]


class VideoEncoder(nn.Module):
    @abstractmethod
    def encode_video(self, video: TYPE_VIDEO_INPUT) -> torch.Tensor:
        raise NotImplementedError

    def forward(self, video: TYPE_VIDEO_INPUT) -> torch.Tensor:
        return self.encode_video(video)

    @abstractmethod
    def get_train_frame_sampler(self) -> FrameSampler:
        raise NotImplementedError

    @abstractmethod
    def get_eval_frame_sampler(self) -> FrameSampler:
        raise NotImplementedError

    @abstractmethod
    def get_train_transform(self, dtype: torch.dtype) -> TYPE_TRANSFORM:
        raise NotImplementedError

    @abstractmethod
    def get_eval_transform(self, dtype: torch.dtype) -> TYPE_TRANSFORM:
        raise NotImplementedError

    @property
    def should_pad_batch(self) -> bool:
        raise NotImplementedError

    @abstractmethod
    def to_bchw(self, t: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    @abstractmethod
    def denormalize_video_tensor(self, video: TYPE_VIDEO_INPUT) -> torch.Tensor:
        """Converts a transformed video tensor into an unsigned 8-bit integer tensor in the range 0-255."""
        raise NotImplementedError


class VideoTextEncoder(VideoEncoder):
    @abstractmethod
    def encode_text(self, text: TYPE_TEXT_INPUT) -> torch.Tensor:
        raise NotImplementedError

    def forward(self, video: TYPE_VIDEO_INPUT, text: TYPE_TEXT_INPUT) -> TYPE_OUTPUT:  # noqa
        return self.encode_video(video), self.encode_text(text)

    @abstractmethod
    def get_tokenizer(self) -> TYPE_TOKENIZER:
        raise NotImplementedError

    @abstractmethod
    def decode_text(self, text: TYPE_TEXT_INPUT) -> Iterator[str]:
        """Decodes a batch of texts."""
        raise NotImplementedError


def float_standard_denormalize(video: TYPE_VIDEO_INPUT, mean: Optional[Tuple[float, float, float]] = None,
                               std: Optional[Tuple[float, float, float]] = None) -> torch.Tensor:
    """Undo a per-channel (x - mean) / std normalisation and quantise to uint8 0-255.

    Like the reference helper (`video_encoder.py:55-63`) it scales and shifts `video` IN PLACE before the cast."""
    def per_channel(values: Tuple[float, float, float]) -> torch.Tensor:
        return torch.as_tensor(values, dtype=video.dtype, device=video.device).reshape(-1, 1, 1)

    if std is not None:
        video.mul_(per_channel(std))
    if mean is not None:
        video.add_(per_channel(mean))
    return video.mul(255).to(torch.uint8)

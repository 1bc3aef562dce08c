"""Frame-index samplers handed to the data side by the encoder (reference `aligner/data/frame_sampler.py:12-41`).

Only the two samplers `ClipVideoTextEncoder` returns are provided: midpoints of `max_frames` equal intervals for
evaluation, one random frame per interval for training.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Sequence

import torch


class FrameSampler(ABC):
    """Returns the frame indices to seek for the given clip start and end frame indices."""

    @abstractmethod
    def __call__(self, start_frame: int, end_frame: int, fps: float) -> Sequence[int]:
        raise NotImplementedError


def _interval_ticks(max_frames: int, start_frame: int, end_frame: int) -> torch.Tensor:
    num_frames = min(max_frames, end_frame - start_frame + 1)
    return torch.linspace(start=start_frame, end=end_frame, steps=num_frames + 1, dtype=torch.int)


class UniformFrameSampler(FrameSampler):
    def __init__(self, max_frames: int) -> None:
        self.max_frames = max_frames

    def __call__(self, start_frame: int, end_frame: int, fps: float) -> Sequence[int]:
        ticks = _interval_ticks(self.max_frames, start_frame, end_frame)
        return [int(torch.round((lo + hi) / 2)) for lo, hi in zip(ticks[:-1], ticks[1:])]


class RandomFromUniformIntervalsFrameSampler(FrameSampler):
    def __init__(self, max_frames: int) -> None:
        self.max_frames = max_frames

    def __call__(self, start_frame: int, end_frame: int, fps: float) -> Sequence[int]:
        ticks = _interval_ticks(self.max_frames, start_frame, end_frame)
        return [int(torch.randint(int(lo), int(hi) + 1, size=())) for lo, hi in zip(ticks[:-1], ticks[1:])]

"""CPU restatement of the CLIP eval transform.  TEST INFRASTRUCTURE ONLY (see oracle/clip_oracle.py).

What `ClipVideoTextEncoder.get_eval_transform` of the reference builds (aligner/encoder/clip_video_text_encoder.py:125-133
with aligner/transforms.py:13-17 and aligner/data/video_dataset.py:85-91):

    ConvertBHWCtoBCHW -> ConvertImageDtype(float) [uint8 / 255] -> Resize(size, BICUBIC) -> CenterCrop(size)
    -> Normalize(mean, std)

torchvision is not installed here, so its published tensor semantics are restated in float64 numpy, independently of the
product's own torch expression (`fitclip_amd.encoder.ClipVideoTextEncoder.get_eval_transform`) and of the HIP kernel
(`fc_preprocess_u8`), both of which are tested against this file:
  * `Resize(int)`: the SHORTER side becomes `size`, the other `int(size * long / short)` (truncation);
  * bicubic on tensors = `torch.nn.functional.interpolate(mode="bicubic", align_corners=False)`, no antialias:
    source coordinate `(dst + 0.5) * in / out - 0.5`, the cubic convolution kernel with A = -0.75 on the four
    neighbours `floor(x) - 1 .. floor(x) + 2`, neighbour indices clamped to the image (border replication);
  * `CenterCrop(size)`: top = int(round((h - size) / 2.0)), left likewise (Python banker's rounding);
  * `Normalize`: (x - mean[c]) / std[c].
"""
from __future__ import annotations

from typing import Sequence

import numpy as np


def _cubic(t: np.ndarray, a: float = -0.75) -> np.ndarray:
    """The four tap weights for fractional offset t (taps at -1, 0, +1, +2)."""
    def near(x):   # |x| <= 1
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1.0

    def far(x):    # 1 < |x| < 2
        return ((a * x - 5.0 * a) * x + 8.0 * a) * x - 4.0 * a

    return np.stack([far(t + 1.0), near(t), near(1.0 - t), far(2.0 - t)], axis=-1)


def _resize_axis(img: np.ndarray, out_len: int, axis: int) -> np.ndarray:
    in_len = img.shape[axis]
    src = (np.arange(out_len, dtype=np.float64) + 0.5) * (in_len / out_len) - 0.5
    base = np.floor(src)
    w = _cubic(src - base)                                           # [out_len, 4]
    idx = np.clip(base[:, None].astype(np.int64) + np.arange(-1, 3)[None, :], 0, in_len - 1)
    gathered = np.take(img, idx, axis=axis)                          # axis -> (out_len, 4)
    shape = [1] * gathered.ndim
    shape[axis], shape[axis + 1] = out_len, 4
    return (gathered * w.reshape(shape)).sum(axis=axis + 1)


def eval_transform(frames_u8: np.ndarray, size: int, mean: Sequence[float], std: Sequence[float]) -> np.ndarray:
    """uint8 [F, H, W, 3] -> float64 [F, 3, size, size]."""
    x = frames_u8.astype(np.float64).transpose(0, 3, 1, 2) / 255.0
    h, w = x.shape[-2:]
    nh, nw = (size, int(size * w / h)) if h <= w else (int(size * h / w), size)
    x = _resize_axis(_resize_axis(x, nh, axis=2), nw, axis=3)
    top, left = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))
    x = x[..., top:top + size, left:left + size]
    m = np.asarray(mean, dtype=np.float64).reshape(1, 3, 1, 1)
    s = np.asarray(std, dtype=np.float64).reshape(1, 3, 1, 1)
    return (x - m) / s

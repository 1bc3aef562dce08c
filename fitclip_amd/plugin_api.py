"""The encoder plugin contract of the reference, as a table.

The reference wires its Lightning modules and data modules to an encoder through two small abstract classes
(`aligner/encoder/video_encoder.py:14-52`, `aligner/encoder/video_text_encoder.py:15-31`).  What matters for a drop-in
is only WHICH members are called by whom and what they mean, so the contract is written down once, as data
(`VIDEO_CONTRACT`, `TEXT_CONTRACT`: member -> meaning and caller), and the two base classes are generated from it:
every contract member raises `NotImplementedError` until a subclass defines it (the reference's behaviour for an
abstract member that was not overridden), `forward` dispatches to the `encode_*` members, and `missing_members()`
tells a subclass author what is still open.  An encoder written against the reference classes satisfies these, and
vice versa.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import torch
from torch import nn

TYPE_VIDEO_INPUT = torch.Tensor                                     # f32 [B, F, 3, H, W], already transformed
TYPE_TEXT_INPUT = Mapping[str, torch.Tensor]                        # {"input_ids": int [B, context_length]}
TYPE_OUTPUT = Tuple[torch.Tensor, torch.Tensor]                     # (video f32 [B, E], text f32 [B, E])
TYPE_TRANSFORM = Callable[[torch.Tensor], torch.Tensor]
TYPE_TOKENIZER = Callable[[Iterable[str]], Mapping[str, torch.Tensor]]

# member -> (meaning, caller in the reference)
VIDEO_CONTRACT: Dict[str, Tuple[str, str]] = {
    "encode_video": ("video f32 [B, F, 3, H, W] -> one embedding per clip, f32 [B, E]",
                     "video_text_module.py:41 via forward; video_text_classification.py:58"),
    "get_train_frame_sampler": ("FrameSampler used for training clips", "data/video_data_module.py:40-55"),
    "get_eval_frame_sampler": ("FrameSampler used for evaluation clips", "data/video_data_module.py:40-55"),
    "get_train_transform": ("dtype -> callable turning uint8 frames [F, H, W, 3] into the model input (training)",
                            "data/video_data_module.py:40-55"),
    "get_eval_transform": ("dtype -> callable turning uint8 frames [F, H, W, 3] into the model input (evaluation)",
                           "data/video_data_module.py:40-55"),
    "to_bchw": ("a model-input tensor rearranged to [B, C, H, W]", "visualisation scripts"),
    "denormalize_video_tensor": ("a transformed video tensor back to uint8 0-255", "visualisation scripts"),
}
TEXT_CONTRACT: Dict[str, Tuple[str, str]] = {
    "encode_text": ("{'input_ids': int [B, L]} -> f32 [B, E]",
                    "video_text_module.py:41 via forward; video_text_classification.py:52,89,116"),
    "get_tokenizer": ("callable: iterable of str -> {'input_ids': ...}",
                      "data/video_data_module.py:75-78; teacher_student.py:88,112"),
    "decode_text": ("a batch of token ids back to an iterator of strings", "prediction dumps"),
}


def _unimplemented(name: str, meaning: str, caller: str):
    def member(self, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not implement {name}(): {meaning}")

    member.__name__ = member.__qualname__ = name
    member.__doc__ = f"{meaning}.  Called from {caller}."
    member._contract_placeholder = True
    return member


def _with_contract(contract: Mapping[str, Tuple[str, str]]):
    def decorate(cls):
        for name, (meaning, caller) in contract.items():
            if name not in cls.__dict__:
                setattr(cls, name, _unimplemented(name, meaning, caller))
        return cls

    return decorate


@_with_contract(VIDEO_CONTRACT)
class VideoEncoder(nn.Module):
    """Video-only half of the contract; calling the module encodes the video."""

    def forward(self, video: TYPE_VIDEO_INPUT) -> torch.Tensor:
        return self.encode_video(video)

    @property
    def should_pad_batch(self) -> bool:
        """Whether the data module pads every clip of a batch to the same number of frames (data/video_data_module.py)."""
        raise NotImplementedError(f"{type(self).__name__} does not define should_pad_batch")

    @classmethod
    def missing_members(cls) -> List[str]:
        """Contract members this class still inherits as placeholders."""
        names: Sequence[str] = [*VIDEO_CONTRACT, *(TEXT_CONTRACT if issubclass(cls, VideoTextEncoder) else ())]
        return [n for n in names if getattr(getattr(cls, n), "_contract_placeholder", False)]


@_with_contract(TEXT_CONTRACT)
class VideoTextEncoder(VideoEncoder):
    """Both towers; calling the module returns `(encode_video(video), encode_text(text))`."""

    def forward(self, video: TYPE_VIDEO_INPUT, text: TYPE_TEXT_INPUT) -> TYPE_OUTPUT:  # noqa: the signature widens on purpose
        return self.encode_video(video), self.encode_text(text)


def float_standard_denormalize(video: TYPE_VIDEO_INPUT, mean: Optional[Tuple[float, float, float]] = None,
                               std: Optional[Tuple[float, float, float]] = None) -> torch.Tensor:
    """Undoes a per-channel `(x - mean) / std` and quantises to uint8 0-255.

    As with the reference helper (`video_encoder.py:55-63`) the scaling and shifting happen IN PLACE on `video`."""
    for values, apply in ((std, video.mul_), (mean, video.add_)):
        if values is not None:
            apply(torch.as_tensor(values, dtype=video.dtype, device=video.device).reshape(-1, 1, 1))
    return video.mul(255).to(torch.uint8)

"""`command=evaluate` semantics without Lightning: the three hooks of the reference's retrieval module
(`aligner/text_video_retrieval.py:40-98` on top of `aligner/video_text_module.py:25-76`) as plain methods.

    module = TextVideoRetrievalModule(encoder, init_temperature=0.015)         # config/trainer.yaml:17-20
    for batch in loader:  out = module.validation_step(batch); module.validation_step_end(out)
    metrics = module.validation_epoch_end()   # {"loss/val", "r1", "r5", "r10", "mr"}

All tensor arithmetic (encoders, score matrices, losses, ranks) runs in the HIP library.
"""
from __future__ import annotations

import math
from typing import Any, Dict, List, MutableMapping, Optional, Tuple

import torch

from . import distributed as D
from . import ops
from .plugin_api import TYPE_OUTPUT, VideoTextEncoder

TYPE_INPUT = MutableMapping[str, Any]


class VideoTextModule:
    def __init__(self, encoder: VideoTextEncoder, init_temperature: float = 0.05,
                 min_temperature: float = 0.001) -> None:
        self.encoder = encoder
        self.logit_scale = -math.log(init_temperature)          # video_text_module.py:32
        self.max_logit_scale = -math.log(min_temperature)

    def forward(self, batch: TYPE_INPUT) -> TYPE_OUTPUT:
        batch.pop("video_id", None)                              # video_text_module.py:40-41
        return self.encoder(**batch)

    __call__ = forward

    def step_scores(self, encoded_video: torch.Tensor, encoded_text: torch.Tensor) -> torch.Tensor:
        """logit_scale.exp() * V @ T^T (video_text_module.py:62-63)."""
        scale = math.exp(min(self.logit_scale, self.max_logit_scale))
        return ops.similarity(encoded_video, encoded_text, alpha=scale)

    def state_dict(self) -> "Dict[str, torch.Tensor]":
        """Keys of the reference module's Lightning checkpoint: `logit_scale`, `encoder.*` (+ `teacher.*`)."""
        from .checkpoint import module_state_dict
        return module_state_dict(self)

    def load_state_dict(self, state_dict: "Dict[str, torch.Tensor]", strict: bool = True):
        """A module without a teacher ignores `teacher*` keys (text_video_retrieval.py:101-131)."""
        from .checkpoint import load_module_state_dict
        return load_module_state_dict(self, state_dict, strict=strict)

    def predict_step(self, batch: TYPE_INPUT) -> Dict[str, Any]:
        video_ids = batch.get("video_id")
        encoded_video, encoded_text = self(dict(batch))
        return {"encoded_videos": encoded_video, "encoded_texts": encoded_text, "video_ids": video_ids}

    def check_range(self) -> None:
        """precision "fp32x3": raises FC_ERANGE if any encoder call of this module so far met a value its fp16 planes cannot hold
        (`CLIP.check_range`; one host synchronisation).  Every path that hands embeddings out - `validation_epoch_end`, the
        classification module's, `command=predict` before it saves, the training step with a split-mode teacher - calls it
        first.  A no-op for the other precisions and for encoders without the method."""
        for enc in (self.encoder, getattr(self, "teacher", None)):
            model = getattr(enc, "model", None)
            if hasattr(model, "check_range"):
                model.check_range()


class TextVideoRetrievalModule(VideoTextModule):
    """`gather_batches` chooses what `loss/val` means on more than one rank (the retrieval metrics do not depend on it):
      * False (default): every rank computes the NCE of ITS eval batches (B rows) and the epoch mean is all-reduced - the one
        exchange step the path needs is the embedding all-gather at the epoch's end;
      * True: the reference's semantics (text_video_retrieval.py:44-58): every step all-gathers the batch's embeddings over
        the ranks first and the NCE is taken over the world x B gathered rows (a larger contrastive batch: a larger loss).
        One collective per eval batch; every rank must run the same number of steps (the reference's DDP sampler pads the
        shards to that; `python -m fitclip_amd gather_batches=true` feeds empty batches where a shard ends early).
    On ONE rank the two are the same number."""

    def __init__(self, encoder: VideoTextEncoder, init_temperature: float = 0.05, min_temperature: float = 0.001,
                 n_total: Optional[int] = None, gather_batches: bool = False) -> None:
        super().__init__(encoder, init_temperature, min_temperature)
        self.n_total = n_total
        self.gather_batches = bool(gather_batches)
        self._outputs: List[TYPE_OUTPUT] = []
        self._losses: List[Tuple[torch.Tensor, int]] = []

    def validation_step(self, batch: TYPE_INPUT) -> TYPE_OUTPUT:
        return self(batch)

    def validation_step_end(self, output: TYPE_OUTPUT) -> TYPE_OUTPUT:
        """Per-batch `loss/val` = NCE(exp(logit_scale) V T^T) on the batch (text_video_retrieval.py:44-58).  The
        reference gathers the batch across DDP ranks first; with exact clip shards every rank logs its own batches
        and the epoch-level mean is all-reduced instead."""
        encoded_video, encoded_text = output
        if self.gather_batches and D.collectives_active():
            # `all_gather(self, output)` of the reference: the rows of every rank's batch, in rank order (ragged last batches:
            # the sizes are exchanged first - this is the per-batch collective the default mode exists to avoid)
            counts = D.all_gather_counts(len(encoded_video), encoded_video.device)
            if sum(counts):
                gathered_video, gathered_text = D.all_gather_many((encoded_video, encoded_text), counts)
                self._losses.append((ops.nce_loss(self.step_scores(gathered_video, gathered_text)), sum(counts)))
        elif len(encoded_video):
            loss = ops.nce_loss(self.step_scores(encoded_video, encoded_text))
            # (the loss stays on the device until the epoch ends, as Lightning's `self.log` keeps it: a `float()` here would
            # drain the stream after every 32-clip batch)
            self._losses.append((loss, len(encoded_video)))
        if len(encoded_video):
            self._outputs.append(output)   # the LOCAL rows: the epoch-end scoring gathers once
        return output

    def validation_epoch_end(self) -> Dict[str, float]:
        """cat all batches, scores = T @ V^T, target = arange, R@1/5/10 + median rank
        (text_video_retrieval.py:67-83); the scores only ever exist tile by tile inside `fc_similarity_ranks`."""
        self.check_range()   # precision fp32x3: an activation beyond fp16's range since the weights were packed is an error
        encoded_videos = torch.cat([o[0] for o in self._outputs])
        encoded_texts = torch.cat([o[1] for o in self._outputs])
        rank, world_size = D.world()
        n_total = self.n_total if self.n_total is not None else len(encoded_videos) * world_size
        # (ranks straight from the scoring GEMM's epilogue: no [n_local, n_total] score matrix - 268 MB at 8192 x 8192)
        metrics = D.sharded_retrieval(encoded_videos.contiguous(), encoded_texts.contiguous(), n_total,
                                      similarity_ranks=lambda t, v, off: ops.similarity_ranks(t, v, off))
        losses = torch.stack([torch.as_tensor(l, device=encoded_videos.device).reshape(()) for l, _ in self._losses]).double()
        sizes = torch.tensor([float(b) for _, b in self._losses], dtype=torch.float64, device=encoded_videos.device)
        num = torch.stack([(losses * sizes).sum(), sizes.sum()])
        if D.collectives_active():
            torch.distributed.all_reduce(num)
        metrics["loss/val"] = float(num[0] / num[1])
        self._outputs, self._losses = [], []
        return metrics


class TeacherStudentModule(VideoTextModule):
    """Forward + loss value of the distillation module (`aligner/teacher_student.py:93-96,142-173`); no backward."""

    def __init__(self, encoder: VideoTextEncoder, teacher: VideoTextEncoder, init_temperature: float = 0.05,
                 min_temperature: float = 0.001) -> None:
        super().__init__(encoder, init_temperature, min_temperature)
        self.teacher = teacher
        self.teacher_student_logit_scale = self.logit_scale       # teacher_student.py:68-69

    def step(self, batch: TYPE_INPUT) -> Tuple[TYPE_OUTPUT, TYPE_OUTPUT]:
        return (self.encoder(video=batch["video_student"], text=batch["text_student"]),
                self.teacher(video=batch["video_teacher"], text=batch["text_teacher"]))

    def dataset_step_end(self, output: Tuple[TYPE_OUTPUT, TYPE_OUTPUT], labeled: bool) -> torch.Tensor:
        (video, text), (teacher_video, teacher_text) = output
        rank, world_size = D.world()
        if D.collectives_active():
            counts = [len(video)] * world_size  # training batches are equal-sized per rank (DDP)
            video, text, teacher_video, teacher_text = D.all_gather_many(
                (video, text, teacher_video, teacher_text), counts)  # one collective for the four embeddings
        scores = self.step_scores(video, text)
        if labeled:
            return ops.nce_loss(scores)
        ts_scale = math.exp(self.teacher_student_logit_scale)
        teacher_scores = ops.similarity(teacher_video, teacher_text, alpha=ts_scale)
        return ops.teacher_student_nce_loss(scores, teacher_scores) * ts_scale ** 2

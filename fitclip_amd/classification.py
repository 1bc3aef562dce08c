"""Zero-shot video classification on the HIP path.

The reference module (`aligner/video_text_classification.py:29-140`) encodes every `template.format(label)` prompt once,
averages the (unit-norm) prompt embeddings over the templates of a label (:88-90), scores a batch of videos with
`encode_video(video) @ encoded_labels.T` (:107-108) and reports Accuracy@1, Accuracy@5 and the median rank of the true
label (:62-63).  Same constructor arguments and metric names here, without Lightning.
"""
from __future__ import annotations

from typing import Any, Dict, Iterable, List, Mapping, Optional

import numpy as np
import torch

from . import ops
from .distributed import metrics_from_ranks
from .plugin_api import VideoTextEncoder
from .retrieval import VideoTextModule

LABEL_BATCH_SIZE = 32  # video_text_classification.py:73


class VideoTextClassificationModule(VideoTextModule):
    def __init__(self, encoder: VideoTextEncoder, labels: Iterable[str], templates: Optional[Iterable[str]] = None,
                 **kwargs: Any) -> None:
        super().__init__(encoder, **kwargs)
        labels = list(labels)
        self.label_count = len(labels)
        if templates:
            templates = list(templates)
            self.template_count = len(templates)
            labels = [template.format(label) for label in labels for template in templates]
        else:
            self.template_count = 1
        device = next(encoder.parameters()).device
        self.tokenized_labels = {k: v.to(device) for k, v in encoder.get_tokenizer()(labels).items()}
        self.encoded_labels: Optional[torch.Tensor] = None
        self._ranks: List[torch.Tensor] = []

    def on_start(self) -> None:
        """Encode the label prompts in batches of 32 and average over templates (reference `_on_start`, :67-96)."""
        n = len(next(iter(self.tokenized_labels.values())))
        encoded = [self.encoder.encode_text({k: v[s:s + LABEL_BATCH_SIZE] for k, v in self.tokenized_labels.items()})
                   for s in range(0, n, LABEL_BATCH_SIZE)]
        self.encoded_labels = ops.group_mean(torch.cat(encoded), self.template_count)

    def forward(self, video: torch.Tensor) -> torch.Tensor:  # noqa
        if self.encoded_labels is None:
            self.on_start()
        return ops.similarity(self.encoder.encode_video(video), self.encoded_labels)

    __call__ = forward

    def validation_step(self, batch: Mapping[str, Any]) -> torch.Tensor:
        scores = self(batch["video"])
        label_id = batch["target"][1]
        self._ranks.append(ops.ranks_of(scores, torch.as_tensor(label_id)))
        return scores

    def predict_step(self, batch: Mapping[str, Any]) -> Dict[str, Any]:
        scores = self(batch["video"])
        best = torch.from_numpy(scores.cpu().numpy().argmax(axis=-1))  # host-side report of the winning label
        self.check_range()   # (the `.cpu()` above has drained the stream: the flag of this batch is in)
        return {"predictions": best, "labels": batch["target"][1], "video_ids": batch.get("video_id")}

    def validation_epoch_end(self) -> Dict[str, float]:
        self.check_range()   # precision fp32x3: no accuracy is reported from embeddings whose fp16 planes overflowed
        ranks = torch.cat(self._ranks).cpu().numpy() if self._ranks else np.zeros(0, dtype=np.int64)
        self._ranks = []
        m = metrics_from_ranks(ranks)
        return {"a1": m["r1"], "a5": m["r5"], "mr": m["mr"]}

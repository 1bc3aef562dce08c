"""Multi-GPU glue of the encode-and-score path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" on CPU for the tests).

The path shards by clip: every rank encodes a contiguous block of clips and their captions, then ONE exchange step -
an all-gather of the [N/world, 512] fp32 embeddings (2 MiB per rank per tensor at N = 8192: latency-bound, done once
after the whole shard is encoded, never per batch) - after which each rank scores its own text rows against all
videos and the integer hit counts are all-reduced.  This is what the reference does through
`util/tensor_utils.all_gather` (:48-66) + torchmetrics' `dist_reduce_fx="cat"` (`aligner/metrics.py:13`), minus the
padding / duplication of its DDP sampler (the shards here are exact).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collectives_active() -> bool:
    """True when the exchange steps have to run: more than one rank - or a ONE-rank process group under
    FITCLIP_FORCE_COLLECTIVES=1, the single-GPU rehearsal of the RCCL path (every collective then goes through the
    library with the production tensors, shapes and streams, although there is nobody to exchange with)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("FITCLIP_FORCE_COLLECTIVES") == "1"


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, exact partition of range(n): the first n % world ranks hold one extra item."""
    base, extra = divmod(n, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_counts(n: int, world_size: int) -> List[int]:
    return [shard_bounds(n, world_size, r)[1] - shard_bounds(n, world_size, r)[0] for r in range(world_size)]


def all_gather_rows(local: torch.Tensor, counts: Sequence[int]) -> torch.Tensor:
    """Concatenation over ranks of `local` ([counts[rank], ...]), in rank order - the flattened
    `[world, B, ...] -> [world * B, ...]` view of the reference wrapper (tensor_utils.py:58-60), for ragged shards."""
    rank, world_size = world()
    if not collectives_active():
        return local
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    biggest = max(counts)
    padded = local
    if local.shape[0] != biggest:
        padded = local.new_zeros((biggest, *local.shape[1:]))
        padded[:local.shape[0]] = local
    if local.is_cuda and dist.get_backend() == "gloo":  # CPU rehearsal of the multi-rank path: stage through the host
        host = padded.contiguous().cpu()
        host_out = host.new_empty((world_size * biggest, *local.shape[1:]))
        dist.all_gather_into_tensor(host_out, host)
        gathered = host_out.to(local.device)
    else:
        gathered = local.new_empty((world_size * biggest, *local.shape[1:]))
        dist.all_gather_into_tensor(gathered, padded.contiguous())
    if all(c == biggest for c in counts):
        return gathered
    return torch.cat([gathered[r * biggest: r * biggest + c] for r, c in enumerate(counts)])


def all_gather_counts(n_local: int, device) -> List[int]:
    """Row counts of every rank's local batch (one tiny collective + a host read: only the per-batch gather of
    `TextVideoRetrievalModule(gather_batches=True)` needs it - shards and training batches know their sizes)."""
    rank, world_size = world()
    if not collectives_active():
        return [n_local]
    on_host = dist.get_backend() == "gloo"
    mine = torch.tensor([n_local], dtype=torch.int64, device="cpu" if on_host else device)
    out = torch.empty(world_size, dtype=torch.int64, device=mine.device)
    dist.all_gather_into_tensor(out, mine)
    return [int(c) for c in out.tolist()]


def all_gather_many(tensors: Sequence[torch.Tensor], counts: Sequence[int]) -> List[torch.Tensor]:
    """`all_gather_rows` of several [n_local, d_i] tensors in ONE collective (they are packed side by side into one
    [n_local, sum d_i] buffer): the (student video, student text, teacher video, teacher text) tuple the reference's
    wrapper walks tensor by tensor (tensor_utils.py:48-66, teacher_student.py:143)."""
    if not collectives_active():
        return list(tensors)
    widths = [int(np.prod(t.shape[1:])) for t in tensors]
    # (an EMPTY local batch - a shard that ended early in a per-batch gather - still takes part with its [0, width] rows)
    gathered = all_gather_rows(torch.cat([t.reshape(t.shape[0], w) for t, w in zip(tensors, widths)], dim=1), counts)
    return [g.contiguous() for g in gathered.split(widths, dim=1)]


def all_reduce_sum_(t: torch.Tensor, async_op: bool = False):
    """In-place sum over the ranks (gradient exchange of the training step: RCCL all-reduce of a slice of the flat
    gradient buffer).  Returns the work handle when `async_op` (None if there is nothing to wait for)."""
    if not collectives_active() or t.numel() == 0:
        return None
    if t.is_cuda and dist.get_backend() == "gloo":  # CPU rehearsal of the multi-rank path: stage through the host
        host = t.cpu()
        dist.all_reduce(host)
        t.copy_(host)
        return None
    return dist.all_reduce(t, async_op=async_op) if async_op else dist.all_reduce(t)


def metrics_from_ranks(ranks: np.ndarray) -> Dict[str, float]:
    """R@1/5/10 = fraction of ranks < k (torchmetrics Recall(top_k), text_video_retrieval.py:21); MedianRank =
    lower-middle median + 1 (torch.median semantics, aligner/metrics.py:33-36)."""
    ranks = np.asarray(ranks).astype(np.int64)
    if ranks.size == 0:
        return {"r1": float("nan"), "r5": float("nan"), "r10": float("nan"), "mr": float("nan")}
    out = {f"r{k}": float((ranks < k).mean()) for k in (1, 5, 10)}
    out["mr"] = float(np.sort(ranks)[(ranks.size - 1) // 2] + 1)
    return out


def sharded_retrieval(local_videos: torch.Tensor, local_texts: torch.Tensor, n_total: int,
                      similarity: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
                      ranks_of: Optional[Callable[[torch.Tensor, int], torch.Tensor]] = None,
                      similarity_ranks: Optional[Callable[[torch.Tensor, torch.Tensor, int], torch.Tensor]] = None
                      ) -> Dict[str, float]:
    """Epoch-end scoring (text_video_retrieval.py:67-83) over embeddings sharded by clip.

    `similarity_ranks(T_local, V_all, offset)` -> rank of column `offset + i` in row i of T_local @ V_all^T, computed WITHOUT the
    [n_local, n_total] score matrix (production: `ops.similarity_ranks`, the comparison runs in the scoring GEMM's epilogue);
    or the two-step form `similarity(T_local, V_all)` -> scores, `ranks_of(scores, offset)` -> ranks (the CPU tests inject the
    oracle's functions there).
    """
    rank, world_size = world()
    counts = shard_counts(n_total, world_size)
    start, _ = shard_bounds(n_total, world_size, rank)
    all_videos = all_gather_rows(local_videos, counts)
    if similarity_ranks is not None:
        local_ranks = similarity_ranks(local_texts, all_videos, start).to(torch.int32)
    else:
        local_ranks = ranks_of(similarity(local_texts, all_videos), start).to(torch.int32)
    all_ranks = all_gather_rows(local_ranks, counts)
    return metrics_from_ranks(all_ranks.cpu().numpy())

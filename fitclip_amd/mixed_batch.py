"""Mixed-batch composition of the teacher-student training step: which clips of which source go into every batch.

The caller side of `TeacherStudentTrainer.training_step` (its `batch["dataset"]` keys): the reference trains on batches
with a FIXED composition - `train_sequence_sizes` items of every source, in source order (`config/data/mixed_batch_*.yaml`:
8 labeled + 8 unlabeled) - drawn by `RoundRobinMultiSourceSampler` in `max_size_cycle` mode over the concatenated
datasets, grouped by a `BatchSampler(drop_last=True)` and, when distributed, dealt to the ranks batch by batch
(`aligner/data/data_module_group.py:105-166`, `aligner/data/multi_source_sampler.py:14-104`).

Host logic only (integer index streams); pinned by the reference's own known-answer lists
(`aligner/tests/data/multi_source_sampler_test.py:18-33`, kept as data in `tests/golden/multi_source_sampler.json`).
"""
from __future__ import annotations

import bisect
import itertools
import sys
from typing import Any, Dict, Iterable, Iterator, List, Mapping, Sequence, Tuple, Union

from torch.utils.data import Sampler

MODES = ("min_size", "max_size_cycle")


class CycleSampler:
    """`length` items of `source`, starting over (a fresh `iter(source)`: a random sub-sampler reshuffles) whenever it
    runs out (multi_source_sampler.py:14-37)."""

    def __init__(self, source: Iterable[int], length: int = sys.maxsize) -> None:
        self.data_source, self.length = source, length

    def __len__(self) -> int:
        return self.length

    def __iter__(self) -> Iterator[int]:
        produced = 0
        while produced < self.length:
            before = produced
            for item in self.data_source:
                yield item
                produced += 1
                if produced >= self.length:
                    return
            if produced == before:  # an empty source would spin forever
                return


class RoundRobinMultiSourceSampler(Sampler):
    """Indices into the CONCATENATION of the sources: `sequence_sizes[0]` indices of source 0, then `sequence_sizes[1]`
    of source 1, ..., round after round; source i's indices are offset by the sizes of the sources before it.

    `min_size`: the stream ends with the first source that cannot fill its sequence (what it still had is emitted).
    `max_size_cycle`: the source with the most whole rounds (the first one among equals) paces the stream the same way;
    all the others start over when they run out (multi_source_sampler.py:40-104).
    """

    def __init__(self, sub_samplers: Iterable[Iterable[int]], sequence_sizes: Union[int, Iterable[int]] = 1,
                 mode: str = "min_size") -> None:
        self.sources = list(sub_samplers)
        self.sequence_sizes = ([int(sequence_sizes)] * len(self.sources) if isinstance(sequence_sizes, int)
                               else [int(s) for s in sequence_sizes])
        if mode not in MODES:
            raise ValueError(f"mode must be one of {MODES}, got {mode!r}")
        if len(self.sources) != len(self.sequence_sizes) or not self.sources:
            raise ValueError("one sequence size per sub-sampler")
        self.source_sizes = [len(s) for s in self.sources]  # noqa: the sub-samplers need `len`
        if min(self.source_sizes) <= 0 or min(self.sequence_sizes) <= 0:
            raise ValueError("sub-samplers and sequence sizes must be non-empty / positive")
        self.mode = mode
        rounds = [n // s for n, s in zip(self.source_sizes, self.sequence_sizes)]
        # the source that ends the stream: most whole rounds when the others cycle, fewest when nothing cycles
        self.pacer = rounds.index(max(rounds)) if mode == "max_size_cycle" else rounds.index(min(rounds))
        self.offsets = [0, *itertools.accumulate(self.source_sizes)][:-1]

    def _streams(self) -> List[Iterator[int]]:
        if self.mode == "min_size":
            return [iter(s) for s in self.sources]
        return [iter(s) if i == self.pacer else iter(CycleSampler(s)) for i, s in enumerate(self.sources)]

    def __iter__(self) -> Iterator[int]:
        streams = self._streams()
        while True:
            for stream, size, offset in zip(streams, self.sequence_sizes, self.offsets):
                taken = 0
                for local in itertools.islice(stream, size):
                    yield offset + local
                    taken += 1
                if taken < size:
                    return

    def __len__(self) -> int:
        n, s = self.source_sizes[self.pacer], self.sequence_sizes[self.pacer]
        whole = n // s
        # sources before the pacer also emit in the final, partial round
        return (sum(self.sequence_sizes[:self.pacer]) * (whole + 1) + sum(self.sequence_sizes[self.pacer + 1:]) * whole + n)


class MixedBatchSampler:
    """Batches of `sum(sequence_sizes)` concatenated indices with the same per-source composition, incomplete batches
    dropped, and for `world > 1` dealt out batch by batch: rank r takes batches r, r + world, ...; the list is first
    padded with its own head to a multiple of `world` so that every rank runs the same number of steps (the
    torchvision `DistributedSampler` the reference wraps its batch sampler in, data_module_group.py:141-153).

    `keys` names the sources (`ConcatDatasetWithDatasetKey`, data_module_group.py:81-95); iteration yields
    `(indices, dataset_keys)`; `locate(i)` turns a concatenated index into `(key, index inside its source)`.
    """

    def __init__(self, sub_samplers: Union[Sequence[Iterable[int]], Mapping[str, Iterable[int]]],
                 sequence_sizes: Union[int, Sequence[int], Mapping[str, int]] = 1, mode: str = "max_size_cycle",
                 rank: int = 0, world: int = 1) -> None:
        if isinstance(sub_samplers, Mapping):
            self.keys: List[Any] = list(sub_samplers)
            sources = [sub_samplers[k] for k in self.keys]
        else:
            sources = list(sub_samplers)
            self.keys = list(range(len(sources)))
        if isinstance(sequence_sizes, Mapping):
            sequence_sizes = [sequence_sizes[k] for k in self.keys]
        self.sampler = RoundRobinMultiSourceSampler(sources, sequence_sizes, mode)
        self.batch_size = sum(self.sampler.sequence_sizes)
        if not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside world {world}")
        self.rank, self.world = rank, world
        self._ends = list(itertools.accumulate(self.sampler.source_sizes))

    def locate(self, index: int) -> Tuple[Any, int]:
        which = bisect.bisect_right(self._ends, index)
        return self.keys[which], index - self.sampler.offsets[which]

    def __len__(self) -> int:
        batches = len(self.sampler) // self.batch_size
        return -(-batches // self.world)

    def __iter__(self) -> Iterator[Tuple[List[int], List[Any]]]:
        stream = iter(self.sampler)
        batches: List[List[int]] = []
        while True:
            batch = list(itertools.islice(stream, self.batch_size))
            if len(batch) < self.batch_size:
                break
            batches.append(batch)
        if self.world > 1 and batches:
            padded = -(-len(batches) // self.world) * self.world
            batches += [batches[i % len(batches)] for i in range(padded - len(batches))]
            batches = batches[self.rank::self.world]
        for batch in batches:
            yield batch, [self.locate(i)[0] for i in batch]


def composition(keys: Sequence[Any]) -> Dict[Any, int]:
    """Items per source of one batch, in first-seen order (the `dataset` column `training_step` groups by)."""
    out: Dict[Any, int] = {}
    for k in keys:
        out[k] = out.get(k, 0) + 1
    return out

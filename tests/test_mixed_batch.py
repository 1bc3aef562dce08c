"""Mixed-batch composition (fitclip_amd/mixed_batch.py) against the reference's own known-answer lists
(aligner/tests/data/multi_source_sampler_test.py, kept as data in tests/golden/multi_source_sampler.json) and its
batch / rank dealing rules (aligner/data/data_module_group.py:105-166)."""
import json
import string
from pathlib import Path

import pytest
from torch.utils.data import ConcatDataset, DataLoader, RandomSampler, SequentialSampler

from fitclip_amd.mixed_batch import CycleSampler, MixedBatchSampler, RoundRobinMultiSourceSampler, composition

GOLDEN = json.loads((Path(__file__).parent / "golden" / "multi_source_sampler.json").read_text())


def _create_sample_data_loader(mode):
    dataset1 = string.ascii_lowercase
    dataset2 = range(10)
    dataset = ConcatDataset([dataset1, dataset2])  # noqa
    sampler = RoundRobinMultiSourceSampler([SequentialSampler(dataset1), SequentialSampler(dataset2)],
                                           sequence_sizes=[4, 3], mode=mode)
    return DataLoader(dataset, sampler=sampler, batch_size=None)


@pytest.mark.parametrize("mode", ["min_size", "max_size_cycle"])
def test_multi_source_sampler_known_answers(mode):
    data_loader = _create_sample_data_loader(mode)
    expected_list = GOLDEN[mode]
    assert len(data_loader) == len(expected_list)
    assert list(data_loader) == expected_list


@pytest.mark.parametrize("sizes,seq", [((26, 10), (4, 3)), ((10, 26), (3, 4)), ((12, 9, 7), (4, 3, 2)), ((8, 8), (8, 8)),
                                       ((5, 50), (1, 7)), ((7, 7, 7), 2)])
@pytest.mark.parametrize("mode", ["min_size", "max_size_cycle"])
def test_length_matches_the_stream(sizes, seq, mode):
    sampler = RoundRobinMultiSourceSampler([range(n) for n in sizes], seq, mode)
    stream = list(sampler)
    assert len(stream) == len(sampler)
    ends = [sum(sizes[:i + 1]) for i in range(len(sizes))]
    assert all(0 <= i < ends[-1] for i in stream)
    if mode == "min_size":  # nothing repeats
        assert len(set(stream)) == len(stream)
    else:  # the pacing source is seen exactly once, completely
        p = sampler.pacer
        mine = [i for i in stream if ends[p] - sizes[p] <= i < ends[p]]
        assert sorted(mine) == list(range(ends[p] - sizes[p], ends[p]))


def test_cycle_sampler_restarts_and_stops():
    assert list(CycleSampler(range(3), length=8)) == [0, 1, 2, 0, 1, 2, 0, 1]
    assert list(CycleSampler(range(3), length=0)) == []
    assert list(CycleSampler([], length=5)) == []
    shuffled = list(CycleSampler(RandomSampler(range(50)), length=100))
    assert sorted(shuffled[:50]) == sorted(shuffled[50:]) == list(range(50))  # every pass is a full permutation


def test_argument_checks():
    with pytest.raises(ValueError):
        RoundRobinMultiSourceSampler([range(3), range(0)], 1)
    with pytest.raises(ValueError):
        RoundRobinMultiSourceSampler([range(3), range(4)], [1, 0])
    with pytest.raises(ValueError):
        RoundRobinMultiSourceSampler([range(3), range(4)], [1])
    with pytest.raises(ValueError):
        RoundRobinMultiSourceSampler([range(3)], 1, mode="max_size")
    with pytest.raises(ValueError):
        MixedBatchSampler([range(4)], 2, rank=2, world=2)


def test_mixed_batches_have_a_fixed_composition():
    # config/data/mixed_batch_webvid_4_5k_rest.yaml: 8 labeled + 8 unlabeled per batch; the small labeled set cycles
    batches = MixedBatchSampler({"labeled": range(20), "unlabeled": range(100)}, {"labeled": 8, "unlabeled": 8})
    seen = list(batches)
    assert len(seen) == len(batches) == 100 // 8  # the last, incomplete batch is dropped
    for indices, keys in seen:
        assert keys == ["labeled"] * 8 + ["unlabeled"] * 8 and composition(keys) == {"labeled": 8, "unlabeled": 8}
        assert all(i < 20 for i in indices[:8]) and all(20 <= i < 120 for i in indices[8:])
    labeled = [i for indices, _ in seen for i in indices[:8]]
    assert labeled[:20] == list(range(20)) and labeled[20:24] == [0, 1, 2, 3]  # second pass over the labeled clips
    assert batches.locate(19) == ("labeled", 19) and batches.locate(20) == ("unlabeled", 0)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_ranks_take_every_world_th_batch(world):
    sources, seq = [range(9), range(70)], [2, 5]
    everything = [b for b, _ in MixedBatchSampler(sources, seq)]
    per_rank = [[b for b, _ in MixedBatchSampler(sources, seq, rank=r, world=world)] for r in range(world)]
    steps = -(-len(everything) // world)
    assert all(len(p) == steps == len(MixedBatchSampler(sources, seq, rank=r, world=world)) for r, p in enumerate(per_rank))
    dealt = [per_rank[i % world][i // world] for i in range(steps * world)]
    assert dealt[:len(everything)] == everything
    assert dealt[len(everything):] == everything[:steps * world - len(everything)]  # padded with the head of the list

"""CPU (gloo, world_size 2): the multi-GPU glue - exact clip shards, one all-gather of the embeddings, row-block
scoring, gathered ranks - gives the same metrics as the single-process oracle, including ragged shards.  The scorers
injected here are the oracle's (tests may use it); in production they are the HIP operators."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fitclip_amd import distributed as D
from oracle import clip_oracle as O


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _planted(n: int, dim: int = 64, seed: int = 0):
    g = torch.Generator().manual_seed(seed)
    v = torch.nn.functional.normalize(torch.randn(n, dim, generator=g), dim=-1)
    t = torch.nn.functional.normalize(v + 0.35 * torch.randn(n, dim, generator=g), dim=-1)
    return v, t


def _worker(rank: int, world: int, port: int, n: int, out_path: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v, t = _planted(n)
        s, e = D.shard_bounds(n, world, rank)
        metrics = D.sharded_retrieval(v[s:e].contiguous(), t[s:e].contiguous(), n,
                                      similarity=lambda a, b: O.retrieval_scores(a, b),
                                      ranks_of=lambda sc, off: O.ranks_of_target(sc, torch.arange(sc.shape[0]) + off))
        gathered = D.all_gather_rows(v[s:e].contiguous(), D.shard_counts(n, world))
        assert torch.equal(gathered, v)
        if rank == 0:
            np.save(out_path, np.array([metrics[k] for k in ("r1", "r5", "r10", "mr")]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 37])  # even and ragged shards
def test_sharded_retrieval_matches_single_process(tmp_path, n):
    out = str(tmp_path / "m.npy")
    mp.spawn(_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    got = np.load(out)
    v, t = _planted(n)
    ref = O.retrieval_metrics(O.retrieval_scores(t, v))
    assert got.tolist() == pytest.approx([ref["r1"], ref["r5"], ref["r10"], ref["mr"]])
    assert 0.1 < ref["r1"] < 1.0  # the planted task is not vacuous


def test_shard_bounds_are_exact_and_contiguous():
    for n in (0, 1, 7, 8, 8192, 8193):
        for w in (1, 2, 3, 8):
            bounds = [D.shard_bounds(n, w, r) for r in range(w)]
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            assert all(bounds[i][1] == bounds[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in bounds]
            assert max(sizes) - min(sizes) <= 1 and sizes == D.shard_counts(n, w)


def test_metrics_from_ranks_semantics():
    m = D.metrics_from_ranks(np.array([0, 0, 3, 3]))
    assert m == {"r1": 0.5, "r5": 1.0, "r10": 1.0, "mr": 1.0}  # lower-middle median + 1
    m = D.metrics_from_ranks(np.array([9, 10, 4]))
    assert m["r10"] == pytest.approx(2 / 3) and m["r5"] == pytest.approx(1 / 3) and m["mr"] == 10.0


def test_driver_config_parsing():
    from fitclip_amd.__main__ import instantiate, load_encoder_config, parse_overrides
    cfg = parse_overrides(["command=evaluate", "encoder=wise", "n_clips=8", "weight_for_2=0.5", "precision=fp32"])
    node = load_encoder_config("wise", cfg, device="cuda:0")
    assert node["_target_"] == "fitclip_amd.wise.wise" and node["weight_for_2"] == 0.5
    assert node["model2"]["model"] == {"_target_": "fitclip_amd.clip_model.load_clip_model",
                                       "name": "synthetic-student:42", "precision": "fp32", "device": "cuda:0"}
    assert instantiate({"_target_": "collections.OrderedDict", "a": 1}) == {"a": 1}
    with pytest.raises(SystemExit):
        parse_overrides(["bogus=1"])


class _OracleOps:
    """Stands in for `fitclip_amd.ops` inside the CPU workers below (tests may use the oracle; the product cannot)."""

    @staticmethod
    def similarity(a, b, alpha=1.0):
        return alpha * (a @ b.T)

    nce_loss = staticmethod(O.nce_loss)
    teacher_student_nce_loss = staticmethod(O.teacher_student_nce_loss)


def _ts_worker(rank: int, world: int, port: int, n: int, out_path: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fitclip_amd.retrieval as R
        R.ops = _OracleOps()
        sv, st = _planted(n, seed=1)
        tv, tt = _planted(n, seed=2)
        per = n // world
        sl = slice(rank * per, (rank + 1) * per)
        module = R.TeacherStudentModule(encoder=None, teacher=None, init_temperature=0.05)
        out = ((sv[sl].contiguous(), st[sl].contiguous()), (tv[sl].contiguous(), tt[sl].contiguous()))
        got = [float(module.dataset_step_end(out, labeled=True)), float(module.dataset_step_end(out, labeled=False))]
        many = D.all_gather_many([sv[sl].contiguous(), tt[sl, :7].contiguous()], [per] * world)
        assert torch.equal(many[0], sv) and torch.equal(many[1], tt[:, :7])
        np.save(out_path + f".{rank}.npy", np.array(got))
    finally:
        dist.destroy_process_group()


def test_teacher_student_step_end_gathers_across_ranks(tmp_path):
    """`TeacherStudentModule.dataset_step_end` on 2 ranks (each holding half of the batch, ONE collective for the four
    embedding matrices) returns on every rank the losses of the whole batch (teacher_student.py:142-173)."""
    n, out = 32, str(tmp_path / "ts")
    mp.spawn(_ts_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    sv, st = _planted(n, seed=1)
    tv, tt = _planted(n, seed=2)
    s, t = O.step_scores(sv, st, 0.05), O.step_scores(tv, tt, 0.05)
    want = [float(O.nce_loss(s)), float(O.teacher_student_nce_loss(s, t) * (1 / 0.05) ** 2)]
    for rank in range(2):
        assert np.load(out + f".{rank}.npy").tolist() == pytest.approx(want, rel=1e-5)


class _ToyEncoder:
    """Encoder stand-in for the CPU workers: the "embeddings" of a batch are the planted rows themselves."""

    def __call__(self, video, text):
        return video, text


def _loss_val_worker(rank: int, world: int, port: int, sizes, gather: bool, out_path: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fitclip_amd.retrieval as R

        class _Ops(_OracleOps):
            similarity_ranks = staticmethod(lambda t, v, off: O.ranks_of_target(O.retrieval_scores(t, v), torch.arange(t.shape[0]) + off))

        R.ops = _Ops()
        n = sum(sum(per_rank) for per_rank in sizes)
        v, t = _planted(n, seed=3)
        # rank r owns a contiguous shard, cut into the batch sizes given for it (an entry 0 = a step with an empty local batch)
        start = sum(sum(sizes[r]) for r in range(rank))
        module = R.TextVideoRetrievalModule(_ToyEncoder(), init_temperature=0.015, n_total=n, gather_batches=gather)
        for b in sizes[rank]:
            module.validation_step_end(module.validation_step({"video": v[start:start + b].contiguous(), "text": t[start:start + b].contiguous()}))
            start += b
        metrics = module.validation_epoch_end()
        np.save(out_path + f".{rank}.npy", np.array([metrics[k] for k in ("loss/val", "r1", "r5", "r10", "mr")]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [((8, 8), (8, 8)), ((8, 8, 1), (8, 8, 0)), ((8, 5), (8, 4))])
def test_loss_val_semantics_on_two_ranks(tmp_path, sizes):
    """What `loss/val` means at N > 1.  gather_batches=True is the REFERENCE's number (text_video_retrieval.py:44-58: every
    step all-gathers the batch over the ranks, NCE over the world x B rows, logged with batch_size = the gathered size);
    the default logs every rank's own batches and all-reduces the weighted mean.  Both give the same retrieval metrics; the
    losses differ (a larger contrastive batch has a larger NCE), and each must equal the oracle's value for ITS definition.
    Ragged cases: a last batch only one rank has (the other feeds an empty one), last batches of different sizes."""
    n = sum(sum(r) for r in sizes)
    v, t = _planted(n, seed=3)
    ref = O.retrieval_metrics(O.retrieval_scores(t, v))
    offs = [0, sum(sizes[0])]
    # the reference's number: step i gathers rank 0's batch i and rank 1's batch i, in rank order
    num = den = 0.0
    pos = list(offs)
    for b0, b1 in zip(*sizes):
        rows = list(range(pos[0], pos[0] + b0)) + list(range(pos[1], pos[1] + b1))
        pos[0] += b0
        pos[1] += b1
        num += float(O.nce_loss(O.step_scores(v[rows], t[rows], 0.015))) * len(rows)
        den += len(rows)
    want_gathered = num / den
    # the default's number: every local batch on its own, weighted by its size
    num = den = 0.0
    for r in range(2):
        p = offs[r]
        for b in sizes[r]:
            if b:
                num += float(O.nce_loss(O.step_scores(v[p:p + b], t[p:p + b], 0.015))) * b
                den += b
            p += b
    want_local = num / den
    assert want_gathered > want_local + 1e-3   # the two definitions are different numbers on this task
    for gather, want in ((True, want_gathered), (False, want_local)):
        out = str(tmp_path / f"lv{int(gather)}")
        mp.spawn(_loss_val_worker, args=(2, _free_port(), sizes, gather, out), nprocs=2, join=True)
        for rank in range(2):
            got = np.load(out + f".{rank}.npy")
            assert got[0] == pytest.approx(want, rel=1e-5), (gather, rank)
            assert got[1:].tolist() == pytest.approx([ref["r1"], ref["r5"], ref["r10"], ref["mr"]])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a torch.distributed.run environment starts two rank processes by itself
    (fresh children, 127.0.0.1 rendezvous) and rank 0 prints ONE JSON line.  `--dry-run`: no GPU work on this box."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    bench = Path(__file__).resolve().parent.parent / "bench.py"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, str(bench), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    doc = json.loads(lines[0])
    assert len(lines) == 1 and (doc["dry_run"], doc["n_gpus"], doc["sum_of_ranks"]) == (True, 2, 1.0)
    assert doc["shards [rank, first clip, clips, encoder calls]"] == [[0, 0, 256, 1], [1, 256, 256, 1]]   # c2: 256 clips per rank
    one = subprocess.run([sys.executable, str(bench), "--dry-run"], capture_output=True, text=True, timeout=300, env=env)
    assert json.loads(one.stdout.strip().splitlines()[-1])["n_gpus"] == 1


# ------------------------------------------------------------------------------------------------------------------ eight ranks
# The 1 -> 8 GPU curve is measured by the driver with no builder in the loop: everything rank-dependent on the path - shard
# arithmetic, the padded ragged gather, the rank gather behind the median, the packed four-tensor gather of the distillation step,
# both `loss/val` semantics, bench.py's launcher and shard plans - has run at world size 8 here, on gloo (tensor_utils.py:48-66,
# metrics.py:13 of the reference).

def test_sharded_retrieval_on_eight_ranks_with_a_ragged_total(tmp_path):
    n = 8192 + 3   # BASELINE configs[3]'s size plus a remainder: shards of 1025, 1025, 1025, 1024, ...
    out = str(tmp_path / "m8.npy")
    mp.spawn(_worker, args=(8, _free_port(), n, out), nprocs=8, join=True)
    got = np.load(out)
    v, t = _planted(n)
    ref = O.retrieval_metrics(O.retrieval_scores(t, v))
    assert got.tolist() == pytest.approx([ref["r1"], ref["r5"], ref["r10"], ref["mr"]])
    assert D.shard_counts(n, 8) == [1025, 1025, 1025, 1024, 1024, 1024, 1024, 1024]


def _ts8_worker(rank: int, world: int, port: int, n: int, out_path: str) -> None:
    _ts_worker(rank, world, port, n, out_path)


def test_teacher_student_step_end_on_eight_ranks(tmp_path):
    """The packed all-gather of the four embedding matrices (`all_gather_many`) and the full-batch NCE / KD losses with every rank
    holding an eighth of BASELINE configs[4]'s 512 clips."""
    n, out = 512, str(tmp_path / "ts8")
    mp.spawn(_ts8_worker, args=(8, _free_port(), n, out), nprocs=8, join=True)
    sv, st = _planted(n, seed=1)
    tv, tt = _planted(n, seed=2)
    s, t = O.step_scores(sv, st, 0.05), O.step_scores(tv, tt, 0.05)
    want = [float(O.nce_loss(s)), float(O.teacher_student_nce_loss(s, t) * (1 / 0.05) ** 2)]
    for rank in range(8):
        assert np.load(out + f".{rank}.npy").tolist() == pytest.approx(want, rel=1e-5)


def test_loss_val_semantics_on_eight_ranks(tmp_path):
    """Both meanings of `loss/val` at world size 8 over the exact shards of 103 clips (13 x 7 + 12): a ragged last step - one row
    on ranks 0-6, nothing left on rank 7, which feeds an empty batch into the step's collective."""
    sizes = tuple((6, 6, 1) if r < 7 else (6, 6, 0) for r in range(8))
    n = sum(sum(r) for r in sizes)
    assert [sum(r) for r in sizes] == D.shard_counts(n, 8)
    v, t = _planted(n, seed=3)
    ref = O.retrieval_metrics(O.retrieval_scores(t, v))
    offs = [sum(sum(sizes[q]) for q in range(r)) for r in range(8)]
    num = den = 0.0
    pos = list(offs)
    for step in range(3):   # the reference's number: step i gathers every rank's batch i, in rank order
        rows = []
        for r in range(8):
            rows += list(range(pos[r], pos[r] + sizes[r][step]))
            pos[r] += sizes[r][step]
        num += float(O.nce_loss(O.step_scores(v[rows], t[rows], 0.015))) * len(rows)
        den += len(rows)
    want_gathered = num / den
    num = den = 0.0
    for r in range(8):      # the default's number: every local batch on its own, weighted by its size
        p = offs[r]
        for b in sizes[r]:
            if b:
                num += float(O.nce_loss(O.step_scores(v[p:p + b], t[p:p + b], 0.015))) * b
                den += b
            p += b
    want_local = num / den
    assert want_gathered > want_local + 1e-3
    for gather, want in ((True, want_gathered), (False, want_local)):
        out = str(tmp_path / f"lv8{int(gather)}")
        mp.spawn(_loss_val_worker, args=(8, _free_port(), sizes, gather, out), nprocs=8, join=True)
        for rank in range(8):
            got = np.load(out + f".{rank}.npy")
            assert got[0] == pytest.approx(want, rel=1e-5), (gather, rank)
            assert got[1:].tolist() == pytest.approx([ref["r1"], ref["r5"], ref["r10"], ref["mr"]])


@pytest.mark.parametrize("argv,want", [
    (["--config", "c4"], {"metric": "video-text pairs/sec (16-frame 224^2, 77-tok)", "frames": 16, "scaling": "strong", "n_total": 8192,
                          "shards": [[r, 1024 * r, 1024, 8] for r in range(8)]}),
    (["--config", "c4", "--total-clips", "8195", "--eval-batch", "500"],
     {"metric": "video-text pairs/sec (16-frame 224^2, 77-tok)", "frames": 16, "scaling": "strong", "n_total": 8195,
      "shards": [[0, 0, 1025, 3], [1, 1025, 1025, 3], [2, 2050, 1025, 3]] + [[r, 3075 + 1024 * (r - 3), 1024, 3] for r in range(3, 8)]}),
    (["--config", "c5"], {"metric": "video-text pairs/sec through the KD training step (8-frame 224^2, 77-tok)", "frames": 8,
                          "scaling": "strong", "n_total": 512, "shards": [[r, 64 * r, 64, 1] for r in range(8)]}),
    (["--config", "c3", "--frames", "2"], {"metric": "video-text pairs/sec through command=evaluate, encoder=wise (2-frame 224^2, 77-tok)",
                                          "frames": 2, "scaling": "strong", "n_total": 4096, "shards": [[r, 512 * r, 512, 16] for r in range(8)]}),
    ([], {"metric": "video-text pairs/sec (8-frame 224^2, 77-tok)", "frames": 8, "scaling": "weak", "n_total": 2048,
          "shards": [[r, 256 * r, 256, 1] for r in range(8)]}),
])
def test_bench_plans_eight_ranks(argv, want):
    """`python bench.py --gpus 8 <config> --dry-run`: eight rank processes (bench.py's own launcher, gloo), every rank derives its
    shard with the code the GPU run uses, rank 0 prints what the N-rank line would say about the workload: the metric string
    carries the configuration's frame count (c4 is a 16-frame workload), the shards are exact and contiguous."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    bench = Path(__file__).resolve().parent.parent / "bench.py"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, str(bench), "--gpus", "8", "--dry-run", *argv], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    doc = json.loads(lines[0])
    assert (doc["n_gpus"], doc["sum_of_ranks"]) == (8, 28.0)
    got = {"metric": doc["metric"], "frames": doc["frames"], "scaling": doc["scaling"], "n_total": doc["n_total"],
           "shards": doc["shards [rank, first clip, clips, encoder calls]"]}
    assert got == want


def test_bench_describes_the_arithmetic_from_the_librarys_records():
    """`fp32_split_mode.dtype` is built from the attention records of the instrumented step (fc_prof_record.epilogue = the
    fc_attention precision code that ran, .tile = a split pass followed), not from a string constant."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("bench_module", Path(__file__).resolve().parent.parent / "bench.py")
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    text = {"kind": 1, "epilogue": 0, "tile": 0, "K": 77}
    fused = [{"kind": 1, "epilogue": 6, "tile": 0, "K": 197}, text, {"kind": 0, "epilogue": 10, "tile": 3, "K": 768}]
    assert bench.attention_description(fused, 197) == "three fp16 MFMA products per fp32 product"
    from fitclip_amd import synth
    gemms = [{"kind": 0, "precision": 2, "epilogue": 10, "N": 3072, "K": 3 * 768}, {"kind": 0, "precision": 2, "epilogue": 3, "N": 768, "K": 3 * 768},
             {"kind": 0, "precision": 0, "epilogue": 1, "N": 2048, "K": 512}]          # (the text tower of a small call: fp32 kernels)
    assert bench.gemm_description(gemms, synth.VIT_B_16, 2, 3) == "the visual tower's block GEMMs, its patch embedding"
    gemms[2] = {"kind": 0, "precision": 2, "epilogue": 10, "N": 2048, "K": 3 * 512}  # (a call of >= 2048 token rows)
    assert bench.gemm_description(gemms, synth.VIT_B_16, 2, 3).endswith("its patch embedding, the text tower's block GEMMs")
    assert bench.gemm_description(gemms[2:], synth.VIT_B_16, 1, 6) == "no GEMM recorded"
    dtype = bench.SPLIT_MODES["fp32x3"][5].format(gemms=bench.gemm_description(gemms, synth.VIT_B_16, 2, 3),
                                                  attention=bench.attention_description(fused, 197))
    assert dtype.count("three fp16 MFMA products per fp32 product") == 2 and "six bf16" not in dtype
    fallback = [{"kind": 1, "epilogue": 0, "tile": 1, "K": 257}, text]
    assert bench.attention_description(fallback, 257) == "fp32-input MFMA + a split pass over its fp32 output"
    assert bench.attention_description([text], 197) == "none recorded"
    assert bench.metric_name("c2", 8) == "video-text pairs/sec (8-frame 224^2, 77-tok)"   # BASELINE.json's string at the default
    assert "16-frame" in bench.metric_name("c4", 16) and "4-frame" in bench.metric_name("c3", 4)

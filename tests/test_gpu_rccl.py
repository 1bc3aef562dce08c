"""The exchange steps of the path on the RCCL library itself.  A one-GPU box cannot hold two RCCL ranks (the library
refuses two ranks on one device), so this is the part of SURVEY 8(a10)/(e) that CAN run here: a one-rank "nccl" process
group with FITCLIP_FORCE_COLLECTIVES=1, through which evaluate, the training step and `bench.py` issue every collective
of the multi-GPU path on device tensors.  The multi-rank semantics (shard order, ragged shards, gradient sums) are
covered by the gloo tests (tests/test_distributed_cpu.py, test_gpu_training.py::test_two_rank_*)."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _launch(script_args, extra_env=None, timeout=900):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"HSA_ENABLE_IPC_MODE_LEGACY": "0", "FITCLIP_FORCE_COLLECTIVES": "1", **(extra_env or {})})
    # fresh children: the rank process is started by torch.distributed.run before anything in it has touched the GPU
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), *script_args], capture_output=True, text=True,
                         timeout=timeout, env=env, cwd=str(ROOT))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    return json.loads([line for line in run.stdout.splitlines() if line.startswith("{")][-1])


def test_every_collective_of_the_path_runs_on_rccl():
    out = _launch([str(ROOT / "tests" / "workers" / "rccl_one_rank.py")])
    assert out["backend"] == "nccl" and out["world"] == 1
    for key in ("gather_f32", "gather_i32", "gather_many", "all_reduce_async", "broadcast_barrier"):
        assert out[key] is True, key
    assert out["evaluate"]["equal"], out["evaluate"]
    assert out["evaluate"]["gather_batches_equal"] and out["gather_counts"] == [5], out["evaluate"]
    # every sum of a training step has a fixed order (no float atomics): with or without the collectives, two runs of the
    # same two steps end bitwise equal
    assert out["train"]["losses_rccl"] == out["train"]["losses_plain"], out["train"]
    assert out["train"]["max_param_delta"] == 0.0


def test_bench_runs_its_rccl_path_on_one_rank():
    """`bench.py` under a one-rank torch.distributed.run with forced collectives: process group on the device,
    broadcast of the planted tensors, all-gathers inside the timed steps, barrier + MAX all-reduce around them."""
    out = _launch([str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--clips", "16", "--frames", "2",
                   "--no-cpu-baseline", "--no-bf16-mode", "--no-split-mode", "--no-train-leg"])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["collectives"].startswith("nccl process group"), out["config"]
    assert 0.0 <= out["retrieval"]["r1"] <= 1.0 and out["retrieval"]["n"] == 16


def test_bench_line_describes_the_configuration_that_ran():
    """The claims record follows the configuration, not string constants: the metric carries the frame count of the run, the
    three-product leg names the arithmetic its kernels used (from the library's own records of the instrumented step: the fused
    three-fp16-product attention at ViT-B/16's 197 tokens), and `--config c5` has the forward-only number of BASELINE configs[4]."""
    run = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "1", "--clips", "16", "--frames", "2",
                          "--no-cpu-baseline", "--no-bf16-mode", "--no-train-leg", "--truth-clips", "0"], capture_output=True, text=True,
                         timeout=900, cwd=str(ROOT))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    out = json.loads([line for line in run.stdout.splitlines() if line.startswith("{")][-1])
    assert out["metric"] == "video-text pairs/sec (2-frame 224^2, 77-tok)" and "x 2 frames" in out["config"]["workload"]
    split = out["fp32_split_mode"]
    assert split["precision"] == "fp32x3" and split["dtype"].count("three fp16 MFMA products per fp32 product") == 2, split["dtype"]
    assert "six bf16" not in split["dtype"]
    c5 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "c5", "--steps", "1", "--warmup", "1", "--total-clips", "8",
                         "--frames", "2"], capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert c5.returncode == 0, (c5.stdout[-2000:], c5.stderr[-4000:])
    doc = json.loads([line for line in c5.stdout.splitlines() if line.startswith("{")][-1])
    assert doc["metric"] == "video-text pairs/sec through the KD training step (2-frame 224^2, 77-tok)" and doc["config"]["frames"] == 2
    fwd = doc["kd_forward"]
    assert fwd["value"] > doc["value"] > 0 and len(fwd["losses"]) == 2 and all(l == l for l in fwd["losses"])

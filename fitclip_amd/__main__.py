"""`python -m fitclip_amd command=evaluate encoder=clip_vit_b_16 data=synthetic ...`
`python -m fitclip_amd command=train encoder=teacher_student_vit_b_16 steps=10 n_labeled=8 n_unlabeled=8 ...`
`python -m fitclip_amd command=predict encoder=clip_vit_b_16 n_clips=64 output_path=predictions.pt`

The `command=evaluate` path of the reference CLI (`aligner/__main__.py:27-69`, `aligner/cli.py:81-150`) without Hydra /
Lightning: `key=value` overrides, `_target_` instantiation of the encoder config group, seed 42
(config/trainer.yaml:41), batches of `eval_batch_size` = 32 (`aligner/data/video_data_module.py:32`) shaped
{"video", "text": {"input_ids"}, "video_id"}, `TextVideoRetrievalModule(init_temperature=0.015)`
(config/trainer.yaml:17-20), metric names `loss/val`, `r1`, `r5`, `r10`, `mr`.  One JSON line on stdout.

Multi-GPU: `gpus=N` starts N rank processes (one per GPU, RCCL) itself, or launch with
`python -m torch.distributed.run --nproc-per-node N -m fitclip_amd ...`; every rank evaluates its own contiguous shard
of the clips (exact, no padding) and the embeddings are all-gathered once before scoring.  `backend=gloo` rehearses the
multi-rank path on a single GPU (all ranks on device 0, collectives staged through the host).
"""
from __future__ import annotations

import importlib
import json
import os
import socket
import subprocess
import sys

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before anything initialises HIP (RCCL needs dmabuf IPC here)
from pathlib import Path
from typing import Any, Dict, Mapping

import numpy as np
import torch
import yaml

from . import distributed as D
from . import synth

CONFIG_DIR = Path(__file__).resolve().parent / "config"
DEFAULTS: Dict[str, Any] = {
    "command": "evaluate", "encoder": "clip_vit_b_16", "data": "synthetic", "seed": 42, "n_clips": 64,
    "num_frames": 4, "eval_batch_size": 32, "init_temperature": 0.015, "precision": None, "weight_for_2": None,
    "gpus": 1, "backend": "nccl", "bpe_path": None,
    # evaluate on several ranks: loss/val over the per-step GATHERED batch, as the reference logs it (text_video_retrieval.py:
    # 44-58), instead of every rank's own batches (retrieval.TextVideoRetrievalModule); the retrieval metrics are the same
    "gather_batches": False,
    # command=train (config/teacher_student_trainer.yaml + config/data/mixed_batch_*.yaml): per-rank batch composition
    "steps": 10, "n_labeled": 8, "n_unlabeled": 8, "lr": 3e-6, "fit_temperature": False, "output_path": "predictions.pt",
    "repeat_batch": False,  # train: every step sees the first batch again (overfitting smoke test)
    "labeled_clips": 32, "unlabeled_clips": 256,  # train: sizes of the two synthetic sources the mixed batches draw from
    "teacher_on_labeled": False,  # train: also run the teacher on the labeled rows (its output there is never read)
    # precision=fp32x3: every encoder call waits for its own fp16-range flag (one host synchronisation per call) instead of
    # leaving it to the check that evaluate / predict / train make before they use or save the embeddings
    "strict_range": False,
}


def parse_overrides(argv) -> Dict[str, Any]:
    cfg = dict(DEFAULTS)
    for arg in argv:
        if "=" not in arg:
            raise SystemExit(f"expected key=value, got {arg!r}")
        key, value = arg.split("=", 1)
        if key not in cfg:
            raise SystemExit(f"unknown option {key!r}; known: {sorted(cfg)}")
        cfg[key] = yaml.safe_load(value)
        if isinstance(cfg[key], str):  # YAML 1.1 reads "3e-6" as a string: accept the usual float spellings
            try:
                cfg[key] = float(cfg[key])
            except ValueError:
                pass
    return cfg


def instantiate(node: Any, **overrides: Any) -> Any:
    """Minimal `hydra.utils.instantiate`: recursive `_target_` construction (aligner/cli.py:89)."""
    if isinstance(node, Mapping):
        kwargs = {k: instantiate(v) for k, v in node.items() if k != "_target_"}
        kwargs.update(overrides)
        if "_target_" in node:
            module, _, attr = node["_target_"].rpartition(".")
            return getattr(importlib.import_module(module), attr)(**kwargs)
        return kwargs
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node


def load_encoder_config(name: str, cfg: Mapping[str, Any], device: Any = None) -> Dict[str, Any]:
    # `encoder=<name>` picks config/encoder/<name>.yaml of the package; `encoder=<path>.yaml` a config file of the caller's (Hydra's
    # config search path in one step: a local checkpoint, another precision, ...)
    path = Path(name) if str(name).endswith((".yaml", ".yml")) else CONFIG_DIR / "encoder" / f"{name}.yaml"
    node = yaml.safe_load(path.read_text())

    def patch(n: Any) -> None:
        if isinstance(n, dict):
            if n.get("_target_", "").endswith("load_clip_model"):
                if cfg.get("precision"):
                    n["precision"] = cfg["precision"]
                if cfg.get("strict_range"):
                    n["strict_range"] = True
                if device is not None:
                    n["device"] = str(device)  # weights go straight to the ROCm device (WiSE blends there)
            if "num_frames" in n:
                n["num_frames"] = cfg["num_frames"]
            if "bpe_path" in n and cfg.get("bpe_path"):
                n["bpe_path"] = cfg["bpe_path"]
            for v in n.values():
                patch(v)

    patch(node)
    if cfg.get("weight_for_2") is not None and "weight_for_2" in node:
        node["weight_for_2"] = cfg["weight_for_2"]
    return node


def evaluate(cfg: Mapping[str, Any]) -> Dict[str, float]:
    from .retrieval import TextVideoRetrievalModule
    rank, world = D.world()
    device = _device(cfg)
    torch.cuda.set_device(device)
    torch.manual_seed(cfg["seed"])
    encoder = instantiate(load_encoder_config(cfg["encoder"], cfg, device)).to(device)
    dims = encoder.model.dims
    if cfg["data"] != "synthetic":
        raise SystemExit("only data=synthetic is available offline (no datasets in this environment)")
    n = cfg["n_clips"]
    start, end = D.shard_bounds(n, world, rank)
    gather = bool(cfg.get("gather_batches"))
    module = TextVideoRetrievalModule(encoder, init_temperature=cfg["init_temperature"], n_total=n, gather_batches=gather)
    bs = cfg["eval_batch_size"]
    # gather_batches (the reference's per-batch gathered loss/val): every rank runs the step count of the largest shard - a
    # shard that ends early feeds empty batches, which only take part in the gather (the reference's sampler pads instead)
    steps = -(-max(D.shard_counts(n, world)) // bs) if gather else -(-(end - start) // bs)
    with torch.inference_mode():
        for i in range(steps):
            s = min(end, start + i * bs)
            e = min(end, s + bs)
            if e == s:   # (gather_batches only: this shard has ended, the step still takes part in the gather)
                empty = torch.empty((0, dims.embed_dim), dtype=torch.float32, device=device)
                module.validation_step_end((empty, empty))
                continue
            batch = {"video": torch.from_numpy(synth.make_video(e - s, cfg["num_frames"], dims, cfg["seed"], s)).to(device),
                     "text": {"input_ids": torch.from_numpy(synth.make_text(e - s, dims, cfg["seed"], s)).to(device)},
                     "video_id": [f"clip{i}" for i in range(s, e)]}
            module.validation_step_end(module.validation_step(batch))
        return module.validation_epoch_end()


def train(cfg: Mapping[str, Any]) -> Dict[str, Any]:
    """`command=train` with `encoder=teacher_student_*`: the distillation loop of the reference
    (config/teacher_student_trainer.yaml -> TeacherStudentLightningModule; optimiser torch.optim.AdamW lr 3e-6,
    config/trainer.yaml:21-23; temperatures init 0.015, fit_temperature false, :17-20) over synthetic mixed batches
    (`n_labeled` + `n_unlabeled` clips per rank and step out of `labeled_clips` / `unlabeled_clips`, composed as
    config/data/mixed_batch_*.yaml does: fitclip_amd/mixed_batch.py)."""
    from .training import TeacherStudentTrainer
    rank, world = D.world()
    device = _device(cfg)
    torch.cuda.set_device(device)
    torch.manual_seed(cfg["seed"])
    node = load_encoder_config(cfg["encoder"], {**cfg, "precision": None}, device)
    if not (isinstance(node, dict) and {"teacher", "student"} <= set(node)):
        raise SystemExit("command=train needs an encoder config with `teacher` and `student` (e.g. teacher_student_vit_b_16)")
    teacher, student = instantiate(node["teacher"]).to(device), instantiate(node["student"]).to(device)
    module = TeacherStudentTrainer(student, teacher, init_temperature=cfg["init_temperature"], lr=cfg["lr"],
                                   fit_temperature=bool(cfg["fit_temperature"]),
                                   teacher_on_labeled=bool(cfg["teacher_on_labeled"]))
    dims = student.model.dims
    per = cfg["n_labeled"] + cfg["n_unlabeled"]
    # batches of a fixed composition drawn round-robin from the two sources, the small one cycling, and dealt to the
    # ranks batch by batch (MixedBatchDataModule.train_dataloader, data_module_group.py:124-166); clip ids are the
    # indices into the concatenated sources
    from torch.utils.data import RandomSampler
    from .mixed_batch import MixedBatchSampler
    sizes = {"labeled": max(cfg["labeled_clips"], cfg["n_labeled"]), "unlabeled": max(cfg["unlabeled_clips"], cfg["n_unlabeled"])}
    sequence = {k: n for k, n in (("labeled", cfg["n_labeled"]), ("unlabeled", cfg["n_unlabeled"])) if n > 0}
    generator = torch.Generator().manual_seed(cfg["seed"])  # the same shuffles on every rank
    sampler = MixedBatchSampler({k: RandomSampler(range(sizes[k]), generator=generator) for k in sequence}, sequence,
                                rank=rank, world=world)
    epoch = iter(sampler)
    losses, first = [], None
    for step in range(cfg["steps"]):
        if first is None or not cfg["repeat_batch"]:
            try:
                clip_ids, names = next(epoch)
            except StopIteration:  # next epoch: the sub-samplers reshuffle
                epoch = iter(sampler)
                clip_ids, names = next(epoch)
            first = first or (clip_ids, names)
        if cfg["repeat_batch"]:
            clip_ids, names = first
        video = torch.from_numpy(np.concatenate([synth.make_video(1, cfg["num_frames"], dims, cfg["seed"], i) for i in clip_ids])).to(device)
        ids = torch.from_numpy(np.concatenate([synth.make_text(1, dims, cfg["seed"], i) for i in clip_ids])).to(device)
        losses.append(module.fit_step({"video_student": video, "text_student": {"input_ids": ids}, "video_teacher": video,
                                       "text_teacher": {"input_ids": ids}, "dataset": names}))
    return {"steps": cfg["steps"], "clips_per_step": per * world, "loss/train": losses,
            "loss/train_labeled": module.last_losses.get("labeled"), "loss/train_unlabeled": module.last_losses.get("unlabeled"),
            "temperature/labeled": 1 / __import__("math").exp(module.logit_scale)}


def predict(cfg: Mapping[str, Any]) -> Dict[str, Any]:
    """`command=predict` (aligner/__main__.py:70-90): encodes every clip / caption and saves {"encoded_videos",
    "encoded_texts", "video_ids"} (the `predict_step` mapping, video_text_module.py:84-92) to `output_path`."""
    from .retrieval import VideoTextModule
    device = _device(cfg)
    torch.cuda.set_device(device)
    encoder = instantiate(load_encoder_config(cfg["encoder"], cfg, device)).to(device)
    module, dims, bs = VideoTextModule(encoder, init_temperature=cfg["init_temperature"]), encoder.model.dims, cfg["eval_batch_size"]
    outs = []
    with torch.inference_mode():
        for s in range(0, cfg["n_clips"], bs):
            e = min(cfg["n_clips"], s + bs)
            outs.append(module.predict_step({
                "video": torch.from_numpy(synth.make_video(e - s, cfg["num_frames"], dims, cfg["seed"], s)).to(device),
                "text": {"input_ids": torch.from_numpy(synth.make_text(e - s, dims, cfg["seed"], s)).to(device)},
                "video_id": [f"clip{i}" for i in range(s, e)]}))
    # precision fp32x3: nothing is written before the range flag of the LAST batch has been seen (FC_ERANGE raises here)
    module.check_range()
    merged = {k: torch.cat([o[k] for o in outs]).cpu() if isinstance(outs[0][k], torch.Tensor)
              else [x for o in outs for x in o[k]] for k in outs[0]}
    torch.save(merged, cfg["output_path"])
    return {"output_path": cfg["output_path"], "n": len(merged["video_ids"])}


def _device(cfg: Mapping[str, Any]) -> torch.device:
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # modulo the visible devices: the single-GPU gloo rehearsal (every rank on device 0), and launchers that show each
    # rank only its own GPU
    return torch.device("cuda", local_rank % max(1, torch.cuda.device_count()))


def _self_launch(argv) -> int:
    """gpus=N without a torch.distributed.run environment: N fresh rank processes, started before this process has
    touched the GPU (a process that has initialised HIP is never re-exec'ed)."""
    n = parse_overrides(argv)["gpus"]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), "-m", "fitclip_amd", *argv]
    return subprocess.run(cmd, env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}).returncode


def main(argv=None) -> None:
    argv = sys.argv[1:] if argv is None else list(argv)
    cfg = parse_overrides(argv)
    if cfg["command"] not in ("evaluate", "validate", "train", "predict"):
        raise SystemExit("commands: evaluate (alias validate), predict, train (tune / test are Lightning plumbing)")
    if cfg["gpus"] > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_launch(argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        device = _device(cfg)
        torch.cuda.set_device(device)  # before the process group: RCCL binds the communicator to the current device
        if cfg["backend"] == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(cfg["backend"])
    metrics = {"train": train, "predict": predict}.get(cfg["command"], evaluate)(cfg)
    if D.world()[0] == 0:
        print(json.dumps({"command": cfg["command"], "encoder": cfg["encoder"], "n_clips": cfg["n_clips"],
                          "num_frames": cfg["num_frames"], **metrics}))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Times the KD training step (student fp32 forward with kept activations + frozen teacher forward + loss + backward +
AdamW) at one rank's share of BASELINE configs[4] (64 clips x 8 frames + 64 texts per GPU, ViT-B/16) and prints one JSON
line: ms per step, clip-pairs/s, and the achieved TFLOP/s against the fp32-input MFMA peak.  FLOPs: 3x the student's
forward GEMM/attention FLOPs (forward + dgrad + wgrad) + 1x the frozen teacher's forward when it runs in fp32 (the
reference's precision, the default); `--teacher-precision bf16` is the labelled faster variant, its teacher FLOPs are
not counted.  Run it under rocprofv3 with the program directly after `--`.

    python tools/train_bench.py [--clips 64] [--frames 8] [--steps 3] [--teacher-precision fp32|bf16]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from fitclip_amd import synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402
from fitclip_amd.training import TeacherStudentTrainer  # noqa: E402

GF_PER_FRAME, GF_PER_TEXT = 35.127e9, 5.960e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--lr", type=float, default=3e-6, help="AdamW learning rate (config/trainer.yaml:21-23)")
    ap.add_argument("--teacher-precision", default="fp32", choices=["fp32", "fp32x6", "bf16"])
    ap.add_argument("--teacher-on-labeled", action="store_true",
                    help="run the teacher over the labeled half as well (the reference's literal schedule; its output is never read)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    d = synth.VIT_B_16
    teacher_sd = synth.make_state_dict(d, seed=42)
    student_sd = synth.perturbed_state_dict(teacher_sd, d, seed=5, rel=0.05)
    student = ClipVideoTextEncoder(build_clip(student_sd, precision="fp32", device=dev), num_frames=a.frames)
    teacher = ClipVideoTextEncoder(build_clip(teacher_sd, precision=a.teacher_precision, device=dev), num_frames=a.frames)
    module = TeacherStudentTrainer(student, teacher, init_temperature=0.05, lr=a.lr, teacher_on_labeled=a.teacher_on_labeled)
    g = torch.Generator(device=dev).manual_seed(0)
    video = torch.randn((a.clips, a.frames, 3, 224, 224), generator=g, device=dev).clamp_(-2.5, 2.5)
    ids = torch.from_numpy(synth.make_text(a.clips, d, seed=1)).to(dev)
    batch = {"video_student": video, "text_student": {"input_ids": ids}, "video_teacher": video,
             "text_teacher": {"input_ids": ids}, "dataset": ["labeled"] * (a.clips // 2) + ["unlabeled"] * (a.clips - a.clips // 2)}
    losses = []
    for _ in range(a.warmup):
        losses.append(module.fit_step(batch))
    torch.cuda.synchronize()
    phases = {"forward": 0.0, "loss": 0.0, "backward": 0.0, "optimizer": 0.0}
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t = time.perf_counter()
        out = module.training_step(batch)
        torch.cuda.synchronize(); phases["forward"] += time.perf_counter() - t; t = time.perf_counter()
        loss = module.training_step_end(out)
        losses.append(loss)
        torch.cuda.synchronize(); phases["loss"] += time.perf_counter() - t; t = time.perf_counter()
        module.backward()
        torch.cuda.synchronize(); phases["backward"] += time.perf_counter() - t; t = time.perf_counter()
        module.optimizer_step()
        torch.cuda.synchronize(); phases["optimizer"] += time.perf_counter() - t
    el = time.perf_counter() - t0
    fwd = a.clips * (a.frames * GF_PER_FRAME + GF_PER_TEXT)
    teacher_share = 1.0 if a.teacher_on_labeled else (a.clips - a.clips // 2) / a.clips  # the half whose loss reads the teacher
    flops = (3 + (teacher_share if a.teacher_precision != "bf16" else 0.0)) * fwd
    res = {"metric": "KD training step, one rank's share of BASELINE configs[4]", "clips": a.clips, "frames": a.frames,
           "teacher_precision": a.teacher_precision, "teacher_rows": "all" if a.teacher_on_labeled else "unlabeled half only",
           "ms_per_step": round(el / a.steps * 1e3, 2), "pairs_per_s": round(a.clips * a.steps / el, 2),
           "tflops_fp32_mfma": round(flops * a.steps / el / 1e12, 2),
           "frac_of_157.3": round(flops * a.steps / el / 1e12 / 157.3, 4),
           "note": ("FLOPs = 3 x student forward (forward + dgrad + wgrad) + the fp32 teacher forward over the rows it runs on" if a.teacher_precision != "bf16" else
                    "FLOPs = 3 x student forward; the bf16 teacher forward is inside the step time but not in the FLOP count"),
           "phases_ms": {k: round(v / a.steps * 1e3, 2) for k, v in phases.items()}, "lr": a.lr, "losses (the same batch every step)": [round(x, 6) for x in losses],
           "peak_memory_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

"""Worker of tests/test_gpu_training.py::test_two_rank_training_step_equals_single_rank: one KD training step on the tiny
student with the batch split over the ranks of a gloo group that shares ONE GPU (rehearsal of the RCCL path: the
collectives are staged through the host, everything else is the production code), or the whole batch on one rank.
Writes the flat gradient buffer, the loss and the updated parameters of rank 0 to --out."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

from fitclip_amd import synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402
from fitclip_amd.training import TeacherStudentTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo")
    d = synth.TINY
    teacher_sd = synth.make_state_dict(d, seed=42)
    student_sd = synth.perturbed_state_dict(teacher_sd, d, seed=5, rel=0.3)
    n, f, per = 8, 2, 4                      # 8 clips: 4 labeled + 4 unlabeled; each of 2 ranks holds 2 + 2
    video = torch.from_numpy(synth.make_video(n, f, d, seed=9))
    ids = torch.from_numpy(synth.make_text(n, d, seed=9))
    lab, unl = list(range(0, per)), list(range(per, n))
    if world > 1:
        share = per // world
        rows = lab[rank * share:(rank + 1) * share] + unl[rank * share:(rank + 1) * share]
        names = ["labeled"] * share + ["unlabeled"] * share
    else:
        rows, names = lab + unl, ["labeled"] * per + ["unlabeled"] * per
    rows = torch.tensor(rows)
    student = ClipVideoTextEncoder(build_clip(student_sd, precision="fp32", device="cuda:0"))
    teacher = ClipVideoTextEncoder(build_clip(teacher_sd, precision="fp32", device="cuda:0"))
    module = TeacherStudentTrainer(student, teacher, init_temperature=0.05, lr=1e-3)
    batch = {"video_student": video[rows].cuda(), "text_student": {"input_ids": ids[rows].cuda()},
             "video_teacher": video[rows].cuda(), "text_teacher": {"input_ids": ids[rows].cuda()}, "dataset": names}
    loss = module.training_step_end(module.training_step(batch))
    module.backward()
    grads = module.student.grads.cpu().numpy().copy()
    scale_grads = module.scale_grads.cpu().numpy().copy()
    module.optimizer_step()
    if rank == 0:
        np.savez(a.out, loss=loss, grads=grads, scale_grads=scale_grads, params=module.student.params.cpu().numpy())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

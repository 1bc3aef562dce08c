"""Worker of tests/test_gpu_rccl.py: ONE rank in an RCCL ("nccl") process group with FITCLIP_FORCE_COLLECTIVES=1, so
every exchange step of the path - the embedding / rank all-gathers of evaluate, the loss all-reduce, the 4-tensor
gather and the three gradient all-reduces of the training step - goes through the RCCL library on device tensors
(the branch the gloo rehearsals never take).  With one rank every collective is the identity, so each result must
equal the same computation with the collectives switched off.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from fitclip_amd import distributed as D  # noqa: E402
from fitclip_amd import synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402
from fitclip_amd.retrieval import TextVideoRetrievalModule  # noqa: E402
from fitclip_amd.training import TeacherStudentTrainer  # noqa: E402


def evaluate(sd, d, video, ids, gather_batches=False):
    module = TextVideoRetrievalModule(ClipVideoTextEncoder(build_clip(sd, precision="fp32", device="cuda:0")),
                                      init_temperature=0.05, n_total=len(video), gather_batches=gather_batches)
    with torch.inference_mode():
        for s in range(0, len(video), 8):
            module.validation_step_end(module.validation_step({"video": video[s:s + 8], "text": {"input_ids": ids[s:s + 8]},
                                                               "video_id": [f"c{i}" for i in range(s, s + 8)]}))
        return module.validation_epoch_end()


def train(teacher_sd, student_sd, video, ids):
    student = ClipVideoTextEncoder(build_clip(student_sd, precision="fp32", device="cuda:0"))
    teacher = ClipVideoTextEncoder(build_clip(teacher_sd, precision="fp32", device="cuda:0"))
    module = TeacherStudentTrainer(student, teacher, init_temperature=0.05, lr=1e-3)
    n = len(video)
    batch = {"video_student": video, "text_student": {"input_ids": ids}, "video_teacher": video,
             "text_teacher": {"input_ids": ids}, "dataset": ["labeled"] * (n // 2) + ["unlabeled"] * (n - n // 2)}
    losses = [module.fit_step(batch) for _ in range(2)]
    return losses, module.student.params.detach().clone()


def main():
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=device)
    assert dist.get_world_size() == 1 and dist.get_backend() == "nccl"
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}

    os.environ["FITCLIP_FORCE_COLLECTIVES"] = "1"
    assert D.collectives_active()
    g = torch.Generator(device="cuda").manual_seed(0)
    emb = torch.randn(37, 512, device="cuda", generator=g)
    ranks = torch.arange(37, device="cuda", dtype=torch.int32)
    out["gather_f32"] = bool(torch.equal(D.all_gather_rows(emb, [37]), emb))
    out["gather_i32"] = bool(torch.equal(D.all_gather_rows(ranks, [37]), ranks))
    four = [torch.randn(37, w, device="cuda", generator=g) for w in (512, 512, 64, 8)]
    out["gather_many"] = all(bool(torch.equal(a, b)) for a, b in zip(D.all_gather_many(four, [37]), four))
    flat = torch.randn(1 << 20, device="cuda", generator=g)
    want = flat.clone()
    handles = [D.all_reduce_sum_(flat[:1000], async_op=True), D.all_reduce_sum_(flat[1000:], async_op=True)]
    for h in handles:
        h.wait()
    out["all_reduce_async"] = bool(torch.equal(flat, want)) and all(h is not None for h in handles)
    t = torch.full((3, 5), 7.0, device="cuda")
    dist.broadcast(t, src=0)
    dist.barrier()
    out["broadcast_barrier"] = bool((t == 7).all())

    d = synth.TINY
    sd = synth.make_state_dict(d, seed=42)
    student_sd = synth.perturbed_state_dict(sd, d, seed=5, rel=0.3)
    video = torch.from_numpy(synth.make_video(24, 2, d, seed=9)).cuda()
    ids = torch.from_numpy(synth.make_text(24, d, seed=9)).cuda()
    with_rccl = evaluate(sd, d, video, ids)
    # the reference's per-step gathered loss/val: the size exchange and the packed two-tensor gather of every eval batch on RCCL
    gathered_rccl = evaluate(sd, d, video, ids, gather_batches=True)
    out["gather_counts"] = D.all_gather_counts(5, device)
    losses_rccl, params_rccl = train(sd, student_sd, video[:8], ids[:8])
    os.environ["FITCLIP_FORCE_COLLECTIVES"] = "0"
    assert not D.collectives_active()
    without = evaluate(sd, d, video, ids)
    losses_plain, params_plain = train(sd, student_sd, video[:8], ids[:8])
    out["evaluate"] = {"rccl": with_rccl, "plain": without, "equal": with_rccl == without,
                       "gather_batches_equal": gathered_rccl == without}  # (one rank: the gathered batch IS the local batch)
    out["train"] = {"losses_rccl": losses_rccl, "losses_plain": losses_plain,
                    "max_param_delta": float((params_rccl - params_plain).abs().max())}
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Headline benchmark: video-text pairs/s of the FitCLIP encode-and-score path on MI355X.

    python bench.py                                  # N = 1, fp32 headline + bf16 secondary mode + CPU baseline
    python bench.py --gpus N --steps K --warmup W    # N > 1 without WORLD_SIZE: starts its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W       # the driver's form: one rank per GPU over RCCL

One step = one pass of the hot path over one batch of synthetic input, inputs already resident in HBM:
  encode_video(256 clips x 8 frames x 3 x 224 x 224 fp32)  +  encode_text(256 x 77 ids)   (CLIP ViT-B/16, random init)
  -> [all-gather of the embeddings over RCCL when N > 1] -> T @ V^T -> rank of every caption's clip (scored tile by tile in
     the GEMM's epilogue: fc_similarity_ranks; no [n, n_total] matrix in memory).
Per-GPU work is fixed as N grows (each rank encodes its own 256 clips): "scaling": "weak"; `value` is whole-job
pairs/s = N * 256 / (max over ranks of the step time).  That is `--config c2` (BASELINE configs[1]), the default and the
driver's line.  The two multi-GPU configurations BASELINE names run through the same command:
  --config c4   configs[3]: 8192 clips x 16 frames + 8192 texts IN TOTAL, exact contiguous clip shards over the N ranks (ragged
                when N does not divide 8192), each rank encodes its shard in eval batches, ONE all-gather of the
                [n_local, 512] embeddings, row-block scoring, int32 ranks gathered: "scaling": "strong"
  --config c5   configs[4]: the distillation training step (teacher + student dual forward, NCE + KD losses, student
                backward, AdamW) on 512 clips x 8 frames IN TOTAL, half labeled on every rank; one packed all-gather of the
                four embedding matrices + three gradient all-reduces per step: "scaling": "strong"
  (--total-clips / --frames / --eval-batch shrink them for rehearsals.)
With more than one rank only the headline leg runs (secondary legs would multiply the collectives of one driver
command); `--all-legs` restores them.  A failing secondary leg is reported under its key, the headline line is still
printed, and the process exits non-zero; after a HIP runtime error no further GPU leg is started.

HEADLINE = the reference's precision: every GEMM / attention product on the fp32-input matrix cores
(`v_mfma_f32_16x16x4_f32`, peak 157.3 TFLOP/s), fp32 everywhere else ("dtype": "fp32"; the reference loads CLIP in
float32, aligner/encoder/clip_video_text_encoder.py:22-25).  The bf16-operand mode (fp32 accumulate / residual stream /
LayerNorm / softmax statistics) is timed afterwards in the same process and reported under "bf16_mode" together with
its Recall deltas; it never is `value`.

The same JSON line carries
  * "roofline": the dominant kernel (the MFMA GEMM instantiation with the largest total time), its average launch
    duration measured with hipEvent pairs recorded by the library on the stream the kernels run on.  The timed region
    itself carries NO instrumentation (`value` pays no profiler tax): the same K steps are repeated right after it with
    event pairs around the four big GEMMs of every block, and that repetition's wall time is reported next to the
    timed one ("instrumented_repeat_ms_per_step");
    achieved = algorithmic FLOPs per launch / that duration; peak = dense MFMA peak of the dtype (MI355X_MICROARCH.md);
    traffic = HBM bytes per launch from the PMC passes of tools/profile_round.sh, accepted only if the pass was made
    with the SAME kernel sources (fingerprint of fitclip_amd/csrc + include), else null.
  * after the timed region, untimed: one fully instrumented step ("time_split", "roofline_all_gemms", and the check of
    which kernel dominates) and K passes of the visual tower alone ("roofline_vit_forward").
  * "retrieval": the towers are random, so `visual.proj`, `visual.ln_post.bias` and `text_projection` are PLANTED before
    the run (same FLOPs): clip embeddings are spread over the sphere (PCA of the CLS features) and 70 % of the captions
    are fitted onto their own clip, 30 % onto a wrong one -> R@1 ~ 0.7 by construction.  Reported: R@k / MedR of the
    device path over the whole batch, and on the CPU sample the device path ("gpu"), the oracle ("ref") and "delta".
  * "cpu_baseline": the CPU oracle (oracle/clip_oracle.py, kind "port") timed on this host's cores over a bounded
    sample of the same workload (rank 0, N = 1 only), next to the parity of the GPU embeddings on that sample.
  * `fp32_split_mode`: the same step with `precision="fp32x3"` - the towers' block GEMMs, the patch embedding and the visual
    attention on the fp16 matrix cores over two-plane split-fp32 operands, three fp16 products per fp32 product
    (fc_config.split_gemm = 2; its `dtype` names what the step's own records show) - with its parity
    against the fp32 path and the oracle.  A labelled secondary mode like `bf16_mode`; never `value`.  `--split6` adds
    `fp32_split6_mode` (precision "fp32x6": three bf16 planes, six bf16 products - the split mode of rounds 2-4).
  * `kd_training_step` (N = 1 only, after the timed region): the distillation training step at one
    rank's share of BASELINE configs[4]; never part of `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from collections import defaultdict

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before anything initialises HIP (RCCL needs dmabuf IPC here)

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GF_PER_FRAME = 35.127e9   # BASELINE.md section 3 (GEMM + attention MACs x 2)
GF_PER_TEXT = 5.960e9
PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}   # dense MFMA peaks (MI355X_MICROARCH.md)
EPI_BIAS, EPI_GELU = 0, 1
EPI_NAMES = {0: "bias", 1: "bias_quickgelu", 2: "bias_residual", 3: "patch_embed", 4: "store_f32"}
EPI_RESID = 2


def host_cores() -> int:
    """CPU threads this process may really use: min(logical CPUs, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()
            if quota != "max":
                n = min(n, max(1, int(quota) // int(period)))
        except (OSError, ValueError):
            pass
    return n


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv) -> int:
    """`--gpus N` (N > 1) without a torch.distributed.run environment: start the N rank processes ourselves, as fresh
    children of a process that has not touched the GPU (never re-exec a process that has initialised HIP)."""
    args = [a for a in argv]
    n = int(args[args.index("--gpus") + 1]) if "--gpus" in args else 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *args]
    return subprocess.run(cmd, env={**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": "0"}).returncode


def synth_video_on_device(n_clips: int, n_frames: int, res: int, seed: int, device) -> torch.Tensor:
    """Clip-specific low-frequency pattern + per-frame noise, clipped to the CLIP-normalised pixel range (the same
    recipe as fitclip_amd.synth.make_video, generated with the device RNG so 1.2 GB never cross PCIe)."""
    g = torch.Generator(device=device).manual_seed(seed)
    low = torch.randn((n_clips, 1, 3, 8, 8), generator=g, device=device)
    base = torch.nn.functional.interpolate(low.view(n_clips, 3, 8, 8), size=(res, res), mode="nearest")
    video = torch.randn((n_clips, n_frames, 3, res, res), generator=g, device=device).mul_(0.5)
    video.add_(base.unsqueeze(1)).clamp_(-2.5, 2.5)
    return video


PLANTED = ("visual.proj", "visual.ln_post.bias", "text_projection")


def plant_retrieval_weights(sd, video, ids, dims, device, wrong_frac=0.3, block=32):
    """Gives the random towers a retrieval task with wide margins and unsaturated recall, by replacing
    three tensors (no FLOP changes): `visual.ln_post.bias` centres the CLS features of this batch, `visual.proj` keeps
    their top principal directions (clips then spread over the sphere instead of sitting at cosine 0.99), and
    `text_projection` is the least-squares map of the pre-projection caption features onto the embedding of the
    caption's own clip - except for `wrong_frac` of the captions, planted on another clip of the same block of `block`
    clips.  Features come from the fp32 HIP path itself (setup, untimed); the fit runs in float64 on the host."""
    from fitclip_amd.clip_model import build_clip
    from fitclip_amd.encoder import ClipVideoTextEncoder
    W, E, TW = dims.vision_width, dims.embed_dim, dims.transformer_width
    model = build_clip(sd, precision="fp32", device=device)
    enc = ClipVideoTextEncoder(model, num_frames=video.shape[1])
    frames = video.view(-1, *video.shape[2:])
    feats = []
    for s in range(0, W, E):  # ln_post(CLS) features through selector projections, E columns at a time
        sel = torch.zeros((W, E), device=device)
        k = min(E, W - s)
        sel[s:s + k, :k] = torch.eye(k, device=device)
        model.visual.proj.data.copy_(sel)
        model.invalidate_weights()
        feats.append(model.encode_image(frames)[:, :k].double().cpu())
    cls = torch.cat(feats, dim=1)
    mu = cls.mean(0)
    _, _, vt = torch.linalg.svd(cls - mu, full_matrices=False)
    proj_v = torch.zeros((W, E), dtype=torch.float64)
    r = min(E, vt.shape[0])                 # fewer frames than embedding dimensions: the remaining columns stay zero
    proj_v[:, :r] = vt[:r].T
    out = {"visual.proj": proj_v.float().contiguous(),
           "visual.ln_post.bias": (model.visual.ln_post.bias.detach().double().cpu() - mu).float()}
    model.visual.proj.data.copy_(out["visual.proj"].to(device))
    model.visual.ln_post.bias.data.copy_(out["visual.ln_post.bias"].to(device))
    sel = torch.zeros((TW, E), device=device)
    sel[:min(TW, E), :min(TW, E)] = torch.eye(min(TW, E), device=device)
    model.text_projection.data.copy_(sel)
    model.invalidate_weights()
    ev = enc.encode_video(video).double().cpu()
    tf = model.encode_text(ids)[:, :min(TW, E)].double().cpu()
    n = ev.shape[0]
    g = torch.Generator().manual_seed(5)
    wrong = torch.rand(n, generator=g) < wrong_frac
    idx = torch.arange(n)
    other = (idx // block) * block + (idx % block + 7) % block
    other = torch.where(other < n, other, idx)
    perm = torch.where(wrong, other, idx)
    sol = torch.linalg.lstsq(tf, ev[perm]).solution  # [min(TW,E), E]
    proj = torch.zeros((TW, E), dtype=torch.float64)
    proj[:sol.shape[0]] = sol
    out["text_projection"] = proj.float().contiguous()
    del enc, model
    return {k: v.numpy() for k, v in out.items()}


def aggregate(recs):
    by_kernel = defaultdict(lambda: [0.0, 0, 0.0])
    other = defaultdict(lambda: [0.0, 0])  # attention / add+LayerNorm launches of the transformer blocks
    for r in recs:
        if r["ms"] <= 0:
            continue
        if r["kind"] != 0:
            agg = other[{1: "attention", 2: "add_layernorm"}.get(r["kind"], "other")]
            agg[0] += r["ms"]
            agg[1] += 1
            continue
        agg = by_kernel[(r["epilogue"], r["N"], r["K"], r["M"], r["tile"])]
        agg[0] += r["ms"]
        agg[1] += 1
        agg[2] += 2.0 * r["M"] * r["N"] * r["K"]
    return by_kernel, other


def load_traffic(precision, shape, epilogue):
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read from
    inside the process): tools/pmc_traffic.py writes them to profiles/; a file measured with other kernel sources than
    the ones in this tree is refused."""
    from fitclip_amd.build import source_fingerprint
    fp = source_fingerprint()
    seen = []
    for name in (f"traffic_r06_{precision}.json", f"traffic_r06_{precision}_c3.json", f"traffic_r05_{precision}.json", f"traffic_r05_{precision}_c3.json", f"traffic_r04_{precision}.json", f"traffic_r04_{precision}_c3.json", f"traffic_r03_{precision}.json",
                 f"traffic_r02_{precision}.json"):
        path = os.path.join(REPO, "profiles", name)
        if not os.path.exists(path):
            continue
        doc = json.load(open(path))
        match = [t for t in doc.get("kernels", [doc]) if t.get("shape") == list(shape) and t.get("precision") == precision
                 and t.get("epilogue") == epilogue]
        if not match:
            seen.append(name)
            continue
        t = match[0]
        if doc.get("source_fingerprint", t.get("source_fingerprint")) != fp:
            return None, (f"{name} is stale: PMC pass made with kernel sources "
                          f"{doc.get('source_fingerprint', t.get('source_fingerprint'))}, tree is {fp}")
        M, N, K = shape
        note = (f"PMC pass ({name}, sources {fp}): fetch {t['fetch_bytes_per_launch'] / 1e6:.0f} MB + write "
                f"{t['write_bytes_per_launch'] / 1e6:.0f} MB per launch; algorithmic "
                f"{t.get('algorithmic_bytes_per_launch', (M * K + N * K + M * N) * (2 if precision == 'bf16' else 4)) / 1e6:.0f} MB")
        return t["hbm_bytes_per_launch"], note
    if seen:
        return None, f"{', '.join(seen)}: no PMC record for {precision} {epilogue} {list(shape)}"
    return None, "no PMC pass committed for this precision"


def training_leg(sd, video, ids, args, dims, device, clips=64, steps=2):
    """Outside the timed region, N = 1 only: the KD training step (aligner/teacher_student.py:99-183) at
    one rank's share of BASELINE configs[4] - 64 clips x 8 frames + 64 captions, half labeled; student fp32 forward with
    kept activations, frozen fp32 teacher forward over the rows whose loss reads it, NCE + KD losses, backward of both
    towers, AdamW.  FLOPs = 3 x the student's forward + the teacher's forward over its rows."""
    from fitclip_amd import synth
    from fitclip_amd.clip_model import build_clip
    from fitclip_amd.encoder import ClipVideoTextEncoder
    from fitclip_amd.training import TeacherStudentTrainer
    n = min(clips, video.shape[0]) // 2 * 2
    frames = video.shape[1]
    torch.cuda.empty_cache()
    student = ClipVideoTextEncoder(build_clip(synth.perturbed_state_dict(sd, dims, seed=5, rel=0.05), precision="fp32",
                                              device=device), num_frames=frames)
    teacher = ClipVideoTextEncoder(build_clip(sd, precision="fp32", device=device), num_frames=frames)
    module = TeacherStudentTrainer(student, teacher, init_temperature=0.05, lr=3e-7)
    batch = {"video_student": video[:n], "text_student": {"input_ids": ids[:n]}, "video_teacher": video[:n],
             "text_teacher": {"input_ids": ids[:n]}, "dataset": ["labeled"] * (n // 2) + ["unlabeled"] * (n // 2)}
    losses = [module.fit_step(batch)]  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses.append(module.fit_step(batch))
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    fwd = n * (frames * 35.127e9 + 5.960e9)
    flops = 3.5 * fwd
    out = {"ms_per_step": round(el * 1e3, 2), "pairs_per_s": round(n / el, 2), "dtype": "fp32",
           "tflops": round(flops / el / 1e12, 2), "frac_of_fp32_mfma_peak": round(flops / el / 1e12 / PEAK_TFLOPS["fp32"], 4),
           "config": f"{n} clips x {frames} frames + {n} captions per GPU, half labeled (one rank's share of BASELINE configs[4])",
           "flops": "3 x student forward (forward, dgrad, wgrad) + fp32 teacher forward over the unlabeled half",
           "weights": "random init (seed 42) teacher, student = teacher perturbed by 5 %; AdamW lr 3e-7 (the reference's "
                      "3e-6 is tuned for pretrained towers and overshoots on random ones), the same batch every step",
           "losses": [round(x, 6) for x in losses], "peak_memory_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    del module, teacher
    torch.cuda.empty_cache()
    # labelled variant: the FROZEN teacher in precision fp32x3 (fp32-accurate embeddings from the fp16 matrix cores; the student and
    # every gradient stay on the fp32-input MFMA path).  Same batch, a fresh student: its first losses must agree with the above.
    try:
        student3 = ClipVideoTextEncoder(build_clip(synth.perturbed_state_dict(sd, dims, seed=5, rel=0.05), precision="fp32",
                                                   device=device), num_frames=frames)
        teacher3 = ClipVideoTextEncoder(build_clip(sd, precision="fp32x3", device=device), num_frames=frames)
        module3 = TeacherStudentTrainer(student3, teacher3, init_temperature=0.05, lr=3e-7)
        l3 = [module3.fit_step(batch)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            l3.append(module3.fit_step(batch))
        torch.cuda.synchronize()
        el3 = (time.perf_counter() - t0) / steps
        teacher3.model.check_range()
        out["with_fp32x3_teacher"] = {"ms_per_step": round(el3 * 1e3, 2), "pairs_per_s": round(n / el3, 2),
                                      "first_loss_abs_delta_vs_fp32_teacher": abs(l3[0] - losses[0]),
                                      "note": "teacher FLOPs run on the fp16 pipe (three products per fp32 product); not counted in `tflops`"}
        del module3, student3, teacher3
    except Exception as exc:  # (a labelled extra: it must not cost the leg)
        out["with_fp32x3_teacher"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
    del student
    torch.cuda.empty_cache()
    return out


class Shards:
    """Which clips this rank holds: `counts[r]` clips on rank r (contiguous, exact - `distributed.shard_counts`), `offset` = the
    global index of this rank's first clip, `eval_batch` = clips per encoder call (None: the whole shard in one call)."""

    def __init__(self, counts, rank, eval_batch=None):
        self.counts, self.rank = list(counts), rank
        self.n_local, self.n_total, self.offset = counts[rank], sum(counts), sum(counts[:rank])
        self.eval_batch = eval_batch if eval_batch and eval_batch < self.n_local else None


def make_step(enc, video, text, shards):
    """One pass of the hot path over this rank's shard: encode (in eval batches when the shard is larger than one), ONE
    all-gather of the video embeddings, the rank's row block of T @ V^T, ranks, all-gather of the int32 ranks."""
    from fitclip_amd import distributed as D
    from fitclip_amd import ops
    ids = text["input_ids"]

    def encode():
        if shards.eval_batch is None:
            return enc(video=video, text=text)
        parts = [enc(video=video[s:s + shards.eval_batch], text={"input_ids": ids[s:s + shards.eval_batch]})
                 for s in range(0, shards.n_local, shards.eval_batch)]
        return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])

    def step():
        ev, et = encode()
        all_v = D.all_gather_rows(ev, shards.counts)
        # the rank of every caption's clip straight from the scoring GEMM's epilogue (fc_similarity_ranks): the [n_local, n_total]
        # matrix - 268 MB for c4 on one GPU - never exists; identical ranks to similarity + ranks (tests/test_gpu_ops.py)
        ranks = ops.similarity_ranks(et.contiguous(), all_v.contiguous(), shards.offset)
        return ev, et, D.all_gather_rows(ranks, shards.counts)

    return step


def timed_steps(step, steps, device, backend):
    """EXACTLY `steps` steps between barrier + synchronize fences; returns (max over ranks of the wall time, last outputs)."""
    import torch.distributed as dist
    from fitclip_amd import distributed as D

    def fence():
        torch.cuda.synchronize()
        if D.collectives_active():
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    fence()
    elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if D.collectives_active():
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    return float(elapsed), out


def run_mode(precision, sd, video, text, args, shards, device, backend, full_detail):
    """Warm-up, the timed K steps (uninstrumented), and - untimed - the same K steps with event pairs around the four big
    block GEMMs, one fully instrumented step and the visual-tower passes, for one precision."""
    from fitclip_amd.clip_model import build_clip
    from fitclip_amd.encoder import ClipVideoTextEncoder

    enc = ClipVideoTextEncoder(build_clip(sd, precision=precision, device=device, chunk_frames=args.chunk_frames,
                                          gemm_tile=args.gemm_tile, prune_last_block=args.prune_last_block),
                               num_frames=args.frames)
    n_local = shards.n_local
    step = make_step(enc, video, text, shards)
    for _ in range(args.warmup):
        step()
    elapsed, (ev, et, all_ranks) = timed_steps(step, args.steps, device, backend)
    # After the timed region: the same K steps once more with hipEvent pairs around the four big GEMMs of every block (the
    # two store epilogues: c_fc+QuickGELU and the bias GEMMs QKV / out_proj / c_proj; c_fc and c_proj are within 2 % of each
    # other, so either can be the dominant one).  An event pair serialises dispatch for a few microseconds; the other ~300
    # launches of a step stay uninstrumented here, so this repetition runs within a fraction of a percent of the timed one
    # (both wall times are reported).  Which kernel dominates is decided by the fully instrumented step that follows.
    enc.model.profile(16384)
    enc.model.profile_select(kind_mask=1, epilogue_mask=(1 << EPI_GELU) | (1 << EPI_BIAS) | (1 << EPI_RESID))
    enc.model.profile_reset()
    repeat_elapsed, _ = timed_steps(step, args.steps, device, backend)
    timed_records = enc.model.profile_records()
    enc.model.profile_select()
    enc.model.profile_reset()
    overlap = enc.overlap_text
    enc.overlap_text = False  # sequential towers: per-kernel durations then describe kernels that own the chip
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    split_elapsed = time.perf_counter() - t1
    enc.overlap_text = overlap
    records = enc.model.profile_records()
    enc.model.profile(0)
    # ViT forward alone: the visual tower + pooling over the same frames, uninstrumented, untimed part
    vit_clips = min(n_local, shards.eval_batch or n_local)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        enc.encode_video(video[:vit_clips])
    torch.cuda.synchronize()
    vit_elapsed = time.perf_counter() - t2

    by_kernel, other_ms = aggregate(records)
    gemm_ms = sum(v[0] for v in by_kernel.values())
    gemm_flops = sum(v[2] for v in by_kernel.values())
    dom_key = max(by_kernel.items(), key=lambda kv: kv[1][0])[0]
    timed_by_kernel, _ = aggregate(timed_records)
    if dom_key in timed_by_kernel:
        ms, cnt, flops = timed_by_kernel[dom_key]
        timing_source, ref_elapsed_ms = ("hipEvent pairs around the four big block GEMMs while the K timed steps are repeated "
                                         "right after the (uninstrumented) timed region"), repeat_elapsed * 1e3
    else:  # the dominant kernel is not the one instrumented in the timed region (other shape)
        ms, cnt, flops = by_kernel[dom_key]
        timing_source, ref_elapsed_ms = "hipEvent pairs in the instrumented extra step", split_elapsed * 1e3
    epi, N, K, M, tile = dom_key
    kname = {1: "gemm_kernel<128x128>", 2: "gemm_kernel<256x256>", 3: "gemm_pipelined_kernel<256x256>",
             8: "gemm_kernel<64x64, 4-stage ring>"}.get(tile, "gemm")
    peak = PEAK_TFLOPS[precision]
    achieved = flops / (ms * 1e-3) / 1e12
    traffic, traffic_note = load_traffic(precision, (M, N, K), EPI_NAMES[epi])
    roofline = {"bound": "mfma", "kernel": f"{kname}<{precision},{EPI_NAMES[epi]}> M={M} N={N} K={K}",
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                "traffic": traffic, "traffic_note": traffic_note, "launches": cnt, "avg_launch_ms": round(ms / cnt, 4),
                "flops_per_launch": flops / cnt, "timing": timing_source,
                "share_of_step_time": round(ms / ref_elapsed_ms, 4)}
    step_flops = max(shards.counts) * (args.frames * GF_PER_FRAME + GF_PER_TEXT)  # the largest shard sets the step time
    all_gemms = {"achieved": round(gemm_flops / (gemm_ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                 "frac": round(gemm_flops / (gemm_ms * 1e-3) / 1e12 / peak, 4),
                 "share_of_step_time": round(gemm_ms / (split_elapsed * 1e3), 4),
                 "timing": "instrumented extra step (every launch carries an event pair)"}
    whole_path = {"achieved": round(step_flops * args.steps / elapsed / 1e12, 2), "unit": "TFLOP/s",
                  "frac": round(step_flops * args.steps / elapsed / 1e12 / peak, 4)}
    vit_tf = vit_clips * args.frames * GF_PER_FRAME * args.steps / vit_elapsed / 1e12
    vit_forward = {"achieved": round(vit_tf, 2), "unit": "TFLOP/s", "frac": round(vit_tf / peak, 4),
                   "ms_per_pass": round(vit_elapsed / args.steps * 1e3, 3),
                   "frames": vit_clips * args.frames, "note": "rank-local encode_video only, after the timed region"}
    out = {"value": round(shards.n_total * args.steps / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "instrumented_repeat_ms_per_step": round(repeat_elapsed / args.steps * 1e3, 3),
           "dtype": precision, "roofline": roofline, "roofline_all_gemms": all_gemms, "roofline_whole_path": whole_path,
           "roofline_vit_forward": vit_forward}
    if full_detail:
        out["time_split"] = {**{k: {"share_of_step_time": round(v[0] / (split_elapsed * 1e3), 4), "launches": v[1],
                                    "avg_launch_ms": round(v[0] / max(1, v[1]), 4)} for k, v in other_ms.items()},
                             "gemm": {"share_of_step_time": all_gemms["share_of_step_time"]},
                             "instrumented_step_ms": round(split_elapsed * 1e3, 3)}
    return out, (ev, et, all_ranks)


SPLIT_MODES = {
    # precision -> (products per fp32 product, profiling precision code of its GEMM records, their epilogues, pipe, kernel, dtype)
    "fp32x3": (3, 2, {6: "bias_f32_out", 10: "bias_quickgelu_x2_out", 8: "bias_residual_f32_out"}, "fp16",
               "gemm_split2_kernel<{rows}x256><two fp16 planes per operand, three MFMA products per fp32 product, {epi}>",
               "fp32 values as two fp16 numbers (x = h1 + 2^-11 h2; weights with a power-of-two scale per tensor); three fp16 MFMA "
               "products per fp32 product, fp32 accumulate ({gemms}); the visual tower's attention products: {attention}; "
               "LayerNorm, softmax arithmetic, residual stream, the text tower's attention and everything not named here in plain fp32"),
    "fp32x6": (6, 1, {6: "bias_f32_out", 7: "bias_quickgelu_x3_out", 8: "bias_residual_f32_out"}, "bf16",
               "gemm_split3_kernel<{rows}x256><three bf16 planes per operand, six MFMA products per fp32 product, {epi}>",
               "fp32 values as three bf16 numbers; six bf16 MFMA products per fp32 product, fp32 accumulate ({gemms}); the visual "
               "tower's attention products: {attention}; LayerNorm, softmax arithmetic, residual stream, the text tower's attention "
               "and everything not named here in plain fp32"),
}


def gemm_description(records, dims, prec_code: int, nprod: int) -> str:
    """Which GEMMs of an instrumented step ran on the mode's plane kernels, from the library's own records (kind 0; .precision = the
    pipe code, .epilogue, .N, .K = products x the fp32 K): the visual tower's block GEMMs, its patch embedding (epilogue 3) and the text
    tower's block GEMMs (fp32x3 runs them that way in calls of >= 2048 token rows: csrc/api.hip kTextX2MinRows)."""
    vw, tw = dims.vision_width, dims.transformer_width
    mine = [r for r in records if r["kind"] == 0 and r["precision"] == prec_code]
    def blocks(w):
        return any(r["epilogue"] in (6, 7, 8, 10) and r["K"] in (nprod * w, nprod * 4 * w) and r["N"] in (3 * w, w, 4 * w) for r in mine)
    parts = []
    if blocks(vw):
        parts.append("the visual tower's block GEMMs")
    if any(r["epilogue"] == 3 for r in mine):
        parts.append("its patch embedding")
    if tw != vw and blocks(tw):
        parts.append("the text tower's block GEMMs")
    return ", ".join(parts) or "no GEMM recorded"


def run_split_mode(sd, video, text, args, shards, device, backend, precision="fp32x3"):
    """Secondary legs `fp32_split_mode` (precision "fp32x3": the block GEMMs on the fp16 matrix cores over two-plane
    operands, THREE fp16 products per fp32 product formed from registers, csrc/gemm_split2.h) and `fp32_split6_mode` (`--split6`;
    precision "fp32x6": three bf16 planes, six bf16 products, csrc/gemm_split3.h) - fp32 accuracy from the pipes that are 16x
    faster than the fp32-input one.  Timed like the headline; the per-kernel figures come from one instrumented extra step."""
    from fitclip_amd.clip_model import build_clip
    from fitclip_amd.encoder import ClipVideoTextEncoder

    nprod, prec_code, epi_names, pipe, kernel_fmt, dtype = SPLIT_MODES[precision]
    enc = ClipVideoTextEncoder(build_clip(sd, precision=precision, device=device, chunk_frames=args.split_chunk_frames),
                               num_frames=args.frames)
    n_local = shards.n_local
    step = make_step(enc, video, text, shards)
    for _ in range(args.warmup):
        step()
    elapsed, (ev, et, all_ranks) = timed_steps(step, args.steps, device, backend)
    enc.model.check_range()   # (fp32x3: FC_ERANGE if an activation left fp16's range; a no-op otherwise)
    enc.model.profile(16384)
    enc.model.profile_reset()
    overlap, enc.overlap_text = enc.overlap_text, False
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    split_elapsed = time.perf_counter() - t1
    enc.overlap_text = overlap
    records = enc.model.profile_records()
    enc.model.profile(0)
    six = [r for r in records if r["kind"] == 0 and r["precision"] == prec_code and r["epilogue"] in epi_names and r["ms"] > 0]
    by = defaultdict(lambda: [0.0, 0])
    for r in six:
        by[(r["epilogue"], r["N"], r["K"], r["M"])][0] += r["ms"]
        by[(r["epilogue"], r["N"], r["K"], r["M"])][1] += 1
    (epi, N, K6, M), (ms, cnt) = max(by.items(), key=lambda kv: kv[1][0])
    # the tile height the launcher picked for the dominant launches (fc_prof_record.tile of a three-product GEMM; the other kernel: 256)
    heights = {r["tile"] for r in six if (r["epilogue"], r["N"], r["K"], r["M"]) == (epi, N, K6, M) and r["tile"] in (128, 192, 256)}
    rows = heights.pop() if len(heights) == 1 else 256
    pipe_flops = 2.0 * M * N * K6  # executed on the matrix pipe: `nprod` products per fp32 product
    epi_name = epi_names[epi]
    traffic, traffic_note = load_traffic(precision, (M, N, K6), epi_name)
    six_ms = sum(r["ms"] for r in six)
    six_flops = sum(2.0 * r["M"] * r["N"] * r["K"] for r in six)
    step_flops = n_local * (args.frames * GF_PER_FRAME + GF_PER_TEXT)
    other = aggregate(records)[1]
    return {
        "precision": precision,
        "value": round(shards.n_total * args.steps / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "dtype": dtype.format(gemms=gemm_description(records, enc.model.dims, prec_code, nprod), attention=attention_description(
            records, (enc.model.dims.image_resolution // enc.model.dims.vision_patch_size) ** 2 + 1)),
        "roofline": {"bound": "mfma", "kernel": kernel_fmt.format(rows=rows, epi=epi_name) + f" M={M} N={N} K={K6 // nprod} (x {nprod} products)",
                     "achieved": round(pipe_flops * cnt / (ms * 1e-3) / 1e12, 1), "peak": PEAK_TFLOPS[pipe], "unit": "TFLOP/s",
                     "frac": round(pipe_flops * cnt / (ms * 1e-3) / 1e12 / PEAK_TFLOPS[pipe], 4),
                     "fp32_equivalent_tflops": round(pipe_flops / nprod * cnt / (ms * 1e-3) / 1e12, 1),
                     "launches": cnt, "avg_launch_ms": round(ms / cnt, 4), "traffic": traffic, "traffic_note": traffic_note,
                     "timing": "hipEvent pairs in the instrumented extra step"},
        "plane_gemms": {"achieved": round(six_flops / (six_ms * 1e-3) / 1e12, 1), "unit": f"TFLOP/s of {pipe} MFMA",
                        "frac": round(six_flops / (six_ms * 1e-3) / 1e12 / PEAK_TFLOPS[pipe], 4),
                        "fp32_equivalent_tflops": round(six_flops / nprod / (six_ms * 1e-3) / 1e12, 1),
                        "share_of_step_time": round(six_ms / (split_elapsed * 1e3), 4)},
        "whole_path_fp32_equivalent_tflops": round(step_flops * args.steps / elapsed / 1e12, 2),
        "time_split": {**{k: {"share_of_step_time": round(v[0] / (split_elapsed * 1e3), 4), "launches": v[1],
                              "avg_launch_ms": round(v[0] / max(1, v[1]), 4)} for k, v in other.items()},
                       "instrumented_step_ms": round(split_elapsed * 1e3, 3)},
    }, (ev, et, all_ranks)


class Legs:
    """Secondary legs must not cost the headline line.  On ONE rank a failing leg is reported under its key, the run goes on
    and the process exits non-zero at the end; after a HIP RUNTIME error (a launch failure or fault, as opposed to a
    Python-level assertion) no further GPU leg is started on the possibly poisoned context.  With more ranks an exception
    propagates: the rank exits non-zero and torch.distributed.run stops its peers (a rank that merely skipped the leg's
    collectives would hang the others)."""

    def __init__(self, result, world):
        self.result, self.world, self.failed, self.gpu_poisoned = result, world, [], False

    @staticmethod
    def _is_hip_runtime_error(e) -> bool:
        text = f"{type(e).__name__}: {e}"
        return ("(-2)" in text and "libfitclip_hip" in text or "fc_" in text and "(-2)" in text or "HIP error" in text
                or "hipError" in text or type(e).__name__ == "AcceleratorError")

    def run(self, key, fn, uses_gpu=True):
        if uses_gpu and self.gpu_poisoned:
            self.result[key] = {"skipped": "an earlier leg ended in a HIP runtime error"}
            return None
        if self.world > 1:
            return fn()
        try:
            return fn()
        except Exception as e:  # noqa: BLE001
            import traceback
            traceback.print_exc(file=sys.stderr)
            self.result[key] = {"error": f"{type(e).__name__}: {e}"}
            self.failed.append(key)
            if self._is_hip_runtime_error(e):
                self.gpu_poisoned = True
            else:
                torch.cuda.empty_cache()
            return None


CONFIG_DEFAULTS = {  # BASELINE.json `configs`: index, clips, frames, scaling
    "c2": {"baseline_index": 1, "frames": 8, "scaling": "weak"},
    "c3": {"baseline_index": 2, "total_clips": 4096, "frames": 4, "scaling": "strong", "eval_batch": 32},
    "c4": {"baseline_index": 3, "total_clips": 8192, "frames": 16, "scaling": "strong", "eval_batch": 128},
    "c5": {"baseline_index": 4, "total_clips": 512, "frames": 8, "scaling": "strong"},
}


def metric_name(config: str, frames: int) -> str:
    """The line's `metric`: BASELINE.json's metric with the frame count of the configuration that RAN (c2 at its default 8 frames is
    BASELINE's string verbatim; c4 is a 16-frame workload, c3 a 4-frame one, and `--frames` changes any of them)."""
    through = {"c2": "", "c4": "", "c3": " through command=evaluate, encoder=wise", "c5": " through the KD training step"}[config]
    return f"video-text pairs/sec{through} ({frames}-frame 224^2, 77-tok)"


def plan_shards(args, world: int):
    """(clips per rank, clips per encoder call) of a run: c2 is weak scaling (`--clips` per rank), the others shard `total_clips`
    exactly and contiguously (`distributed.shard_counts`: ragged when the world does not divide the total)."""
    from fitclip_amd import distributed as D
    conf = CONFIG_DEFAULTS[args.config]
    if args.config == "c2":
        return [args.clips] * world, None
    total = args.total_clips or conf["total_clips"]
    if total < world:
        raise SystemExit(f"--config {args.config}: {total} clips cannot be sharded over {world} ranks")
    return D.shard_counts(total, world), args.eval_batch or conf.get("eval_batch")


ATTENTION_FORMS = {  # fc_prof_record.epilogue of an attention record (include/fitclip_hip.h) -> what the kernel computes with
    0: "fp32-input MFMA", 1: "bf16 MFMA", 3: "fp32-input MFMA (three-plane rows out)", 4: "six bf16 MFMA products per fp32 product",
    5: "six bf16 MFMA products per fp32 product", 6: "three fp16 MFMA products per fp32 product"}


def attention_description(records, tokens: int) -> str:
    """Which arithmetic the visual tower's attention launches (sequences of `tokens` tokens) of an instrumented step ran, from the
    library's own records."""
    forms = sorted({(r["epilogue"], r["tile"]) for r in records if r["kind"] == 1 and r["K"] == tokens})
    return " / ".join(ATTENTION_FORMS.get(code, f"code {code}") + (" + a split pass over its fp32 output" if split_pass else "")
                      for code, split_pass in forms) or "none recorded"


def fill_video(n_clips, frames, res, seed, device, block=256):
    """`synth_video_on_device` in blocks of clips (the generator's temporaries stay small next to a 79 GB shard)."""
    out = torch.empty((n_clips, frames, 3, res, res), dtype=torch.float32, device=device)
    for s in range(0, n_clips, block):
        n = min(block, n_clips - s)
        out[s:s + n] = synth_video_on_device(n, frames, res, seed=seed * 7919 + s, device=device)
    return out


def run_kd_config(sd, dims, args, shards, device, backend):
    """`--config c5` (BASELINE configs[4]): the distillation training step of `TeacherStudentLightningModule`
    (aligner/teacher_student.py:93-183) on `total_clips` clips x `frames` frames sharded over the ranks, half of every rank's
    rows labeled: student forward with kept activations + frozen teacher forward, the packed all-gather of the four embedding
    matrices, NCE + KD losses over the full batch, student backward, three gradient all-reduces, AdamW.  One step = one
    `fit_step`; FLOPs = 3 x the student's forward + the teacher's forward over the unlabeled half."""
    from fitclip_amd import synth
    from fitclip_amd.clip_model import build_clip
    from fitclip_amd.encoder import ClipVideoTextEncoder
    from fitclip_amd.training import TeacherStudentTrainer
    n, frames = shards.n_local, args.frames
    if n < 2:
        raise SystemExit(f"--config c5: rank {shards.rank} holds {n} clip(s); every rank needs a labeled and an unlabeled row")
    video = fill_video(n, frames, dims.image_resolution, seed=1000 + shards.rank, device=device)
    ids = torch.from_numpy(synth.make_text(n, dims, seed=42, first_text=shards.offset)).to(device)
    student = ClipVideoTextEncoder(build_clip(synth.perturbed_state_dict(sd, dims, seed=5, rel=0.05), precision="fp32",
                                              device=device), num_frames=frames)
    teacher = ClipVideoTextEncoder(build_clip(sd, precision="fp32", device=device), num_frames=frames)
    # (activation arenas of at most 512 frames = 60 GB each: a share that is split below keeps several of them)
    module = TeacherStudentTrainer(student, teacher, init_temperature=0.05, lr=3e-7, max_frames_per_pass=512)
    n_lab = n // 2
    batch = {"video_student": video, "text_student": {"input_ids": ids}, "video_teacher": video,
             "text_teacher": {"input_ids": ids}, "dataset": ["labeled"] * n_lab + ["unlabeled"] * (n - n_lab)}
    # BASELINE configs[4] AS WRITTEN - "teacher+student dual forward with KD-loss similarity compute" - is the forward half of the
    # step: both models over the whole local batch (teacher_student.py:93-96), the packed gather of the four embedding matrices,
    # NCE on the labeled rows and KD on the unlabeled ones (teacher_student.py:142-173); no backward, nothing kept.  Its own number:
    from fitclip_amd.retrieval import TeacherStudentModule
    fwd = TeacherStudentModule(student, teacher, init_temperature=0.05)

    def forward_only():
        (sv, st), (tv, tt) = fwd.step(batch)
        labeled = fwd.dataset_step_end(((sv[:n_lab], st[:n_lab]), (tv[:n_lab], tt[:n_lab])), labeled=True)
        unlabeled = fwd.dataset_step_end(((sv[n_lab:], st[n_lab:]), (tv[n_lab:], tt[n_lab:])), labeled=False)
        return labeled, unlabeled

    for _ in range(args.warmup):
        forward_only()
    fwd_elapsed, fwd_losses = timed_steps(forward_only, args.steps, device, backend)
    fwd_flops = 2.0 * max(shards.counts) * (frames * GF_PER_FRAME + GF_PER_TEXT)
    kd_forward = {"value": round(shards.n_total * args.steps / fwd_elapsed, 2), "unit": "pairs/s",
                  "ms_per_step": round(fwd_elapsed / args.steps * 1e3, 3),
                  "workload": "teacher + student forward over the whole batch (no activations kept), one packed all-gather of the four "
                              "embedding matrices, NCE (labeled half) + KD (unlabeled half) loss values - BASELINE configs[4] as written; "
                              "the line's `value` is the full training step (forward + backward + AdamW)",
                  "tflops": round(fwd_flops * args.steps / fwd_elapsed / 1e12, 2),
                  "frac_of_fp32_mfma_peak": round(fwd_flops * args.steps / fwd_elapsed / 1e12 / PEAK_TFLOPS["fp32"], 4),
                  "losses": [round(float(x), 6) for x in fwd_losses]}
    del fwd
    torch.cuda.empty_cache()
    # A rank's share keeps 116 MB of activations per frame (nothing is recomputed): 64 clips x 8 frames = 65 GB fit, the 512 or
    # 256 clips of one or two ranks do not.  The step is then SPLIT (training.py: split_step): as many clips as fit next to one
    # micro-batch keep their activations, the others are forwarded without them and re-forwarded micro-batch by micro-batch in
    # the backward (gradients accumulated; same loss bit for bit).  Sized from the memory that is free now.
    torch.cuda.empty_cache()
    keep, micro = module.plan_split(n, frames, micro_clips=args.micro_clips) if args.keep_clips is None else (args.keep_clips, args.micro_clips)
    if keep < n:
        module.split_step(keep, micro)
    losses = [module.fit_step(batch) for _ in range(args.warmup)]
    elapsed, last = timed_steps(lambda: losses.append(module.fit_step(batch)), args.steps, device, backend)
    n_big = max(shards.counts)
    pair = frames * GF_PER_FRAME + GF_PER_TEXT
    flops = 3.0 * n_big * pair + (n_big - n_big // 2) * pair
    executed = flops + max(0, n_big - keep) * pair if keep < n else flops   # + the second student forward of the clips not kept
    tf = flops * args.steps / elapsed / 1e12
    return {"value": round(shards.n_total * args.steps / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "roofline": {"bound": "mfma", "kernel": "whole training step of the largest shard (forward, dgrad, wgrad, teacher forward)",
                         "achieved": round(tf, 2), "peak": PEAK_TFLOPS["fp32"], "unit": "TFLOP/s",
                         "frac": round(tf / PEAK_TFLOPS["fp32"], 4), "traffic": None,
                         "flops_per_step": flops, "executed_flops_per_step": executed,
                         "executed_frac": round(executed * args.steps / elapsed / 1e12 / PEAK_TFLOPS["fp32"], 4),
                         "timing": "wall time of the timed region, max over ranks"},
            "split_step": ({"kept_clips": keep, "micro_batch_clips": micro, "recomputed_clips": n - keep,
                            "note": "activations of the whole share do not fit: gradient-cache schedule, one more student "
                                    "forward over the clips that were not kept (counted in executed_flops_per_step only)"}
                           if keep < n else None),
            "losses": [round(float(x), 6) for x in losses], "kd_forward": kd_forward,
            "peak_memory_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


def graph_leg(enc, module, video, ids, bs, device, reps=20):
    """The reference-shaped call (one eval batch: both towers, pooling, the batch's score matrix, ranks) captured ONCE into a
    hipGraph and replayed, next to the same call launched eagerly: bitwise equality of every output and ms per call.  What the
    header promises ("all functions may be captured into a hipGraph") exercised on the production path: ~330 kernel launches
    on two streams (the text tower runs beside the visual tower) become one graph launch."""
    from fitclip_amd import ops
    v, t = video[:bs].contiguous(), {"input_ids": ids[:bs].contiguous()}

    def call():
        ev, et = enc(video=v, text=t)
        scores = ops.similarity(et, ev)
        return ev, et, scores, ops.ranks(scores)

    side = torch.cuda.Stream(device=device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            eager = call()  # warm-up on the capture stream: dynamic-LDS attributes, workspaces, side streams exist afterwards
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            captured = call()
        for o in captured:
            o.zero_()  # the replay must rewrite every output
        graph.replay()
        side.synchronize()
        equal = all(torch.equal(a, b) for a, b in zip(eager, captured))

        def timed(fn):
            fn()
            side.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            side.synchronize()
            return (time.perf_counter() - t0) / reps * 1e3
        eager_ms, replay_ms = timed(call), timed(graph.replay)
    torch.cuda.current_stream().wait_stream(side)
    del graph
    return {"call": f"encoder(video[{bs}x{video.shape[1]}], text[{bs}]) + T@V^T + ranks", "eager_ms": round(eager_ms, 3),
            "replay_ms": round(replay_ms, 3), "bitwise_equal_to_eager": bool(equal),
            "note": "one capture on a side stream (torch.cuda.CUDAGraph = hipStreamBeginCapture / hipGraphLaunch), replayed; "
                    "both towers, pooling, scoring and ranks are library launches on the captured streams"}


def run_wise_config(dims, args, shards, device, backend):
    """`--config c3` (BASELINE configs[2]): `encoder=wise` - theta = 0.5 CLIP + 0.5 student, blended on the device by `fc_wise`
    (aligner/wise.py:10-23, config/encoder/wise.yaml:6-9) - evaluated at the WebVid-val shape, `total_clips` clips x 4 frames +
    one caption each, through the evaluate loop of the reference (`TextVideoRetrievalModule`: aligner/text_video_retrieval.py:
    40-83) in eval batches of 32 clips = 128 frames per encoder call (aligner/data/video_data_module.py:32,
    aligner/encoder/clip_video_text_encoder.py:69).  One step = one epoch over this rank's shard: per batch both towers, the
    batch's score matrix and NCE loss; at the end ONE all-gather of the embeddings, T @ V^T, ranks, R@k / MedR."""
    from fitclip_amd import distributed as D
    from fitclip_amd import synth
    from fitclip_amd.__main__ import instantiate, load_encoder_config
    from fitclip_amd.retrieval import TextVideoRetrievalModule
    n, frames, bs = shards.n_local, args.frames, shards.eval_batch or shards.n_local
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc = instantiate(load_encoder_config("wise", {"precision": "fp32", "num_frames": frames, "weight_for_2": 0.5}, device)).to(device)
    enc.num_frames = frames
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    video = fill_video(n, frames, dims.image_resolution, seed=1000 + shards.rank, device=device)
    ids = torch.from_numpy(synth.make_text(n, dims, seed=42, first_text=shards.offset)).to(device)
    module = TextVideoRetrievalModule(enc, init_temperature=0.015, n_total=shards.n_total)

    def epoch(batch=bs):
        with torch.inference_mode():
            for s in range(0, n, batch):
                module.validation_step_end(module.validation_step(
                    {"video": video[s:s + batch], "text": {"input_ids": ids[s:s + batch]}, "video_id": list(range(s, min(n, s + batch)))}))
                if args.sync_batches:
                    torch.cuda.synchronize()   # (profiling aid: bounds how far the host runs ahead of the GPU queue)
            return module.validation_epoch_end()

    for _ in range(max(1, args.warmup)):
        epoch()
    elapsed, metrics = timed_steps(epoch, args.steps, device, backend)
    peak = PEAK_TFLOPS["fp32"]
    flops = max(shards.counts) * (frames * GF_PER_FRAME + GF_PER_TEXT)
    out = {"value": round(shards.n_total * args.steps / elapsed, 2), "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "metrics": metrics, "encoder_build_s": round(build_s, 2),
           "roofline_whole_path": {"achieved": round(flops * args.steps / elapsed / 1e12, 2), "unit": "TFLOP/s",
                                   "frac": round(flops * args.steps / elapsed / 1e12 / peak, 4)}}
    # untimed: one epoch with hipEvent pairs around every GEMM / attention / LayerNorm launch of the towers (text tower on the
    # visual tower's stream, so that a duration describes a kernel that owns the chip)
    enc.model.profile(65536)
    enc.model.profile_reset()
    overlap, enc.overlap_text = enc.overlap_text, False
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    epoch()
    torch.cuda.synchronize()
    inst_s = time.perf_counter() - t1
    enc.overlap_text = overlap
    records = enc.model.profile_records()
    enc.model.profile(0)
    by_kernel, other = aggregate(records)
    (epi, N, K, M, tile), (ms, cnt, kflops) = max(by_kernel.items(), key=lambda kv: kv[1][0])
    gemm_ms, gemm_flops = sum(v[0] for v in by_kernel.values()), sum(v[2] for v in by_kernel.values())
    hp, ht = (None, None)
    if tile == 3:
        from fitclip_amd import ops
        hp, ht = ops.gemm_plan(M, N, K)
    kname = {1: "gemm_kernel<128x128>", 2: "gemm_kernel<256x256>", 3: "gemm_pipelined_kernel<256x256>",
             8: "gemm_kernel<64x64, 4-stage ring>"}.get(tile, "gemm")
    traffic, traffic_note = load_traffic("fp32", (M, N, K), EPI_NAMES[epi])
    out["roofline"] = {"bound": "mfma", "kernel": f"{kname}<fp32,{EPI_NAMES[epi]}> M={M} N={N} K={K}",
                       "achieved": round(kflops / (ms * 1e-3) / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                       "frac": round(kflops / (ms * 1e-3) / 1e12 / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
                       "launches": cnt, "avg_launch_ms": round(ms / cnt, 4), "flops_per_launch": kflops / cnt,
                       "row_cut": None if hp is None else {"head_panels_256": hp, "tail_tile_rows": 64 * ht},
                       "timing": "hipEvent pairs in one instrumented epoch after the timed region",
                       "share_of_step_time": round(ms / (inst_s * 1e3), 4)}
    out["roofline_all_gemms"] = {"achieved": round(gemm_flops / (gemm_ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                                 "frac": round(gemm_flops / (gemm_ms * 1e-3) / 1e12 / peak, 4),
                                 "share_of_step_time": round(gemm_ms / (inst_s * 1e3), 4)}
    out["time_split"] = {**{k: {"share_of_step_time": round(v[0] / (inst_s * 1e3), 4), "launches": v[1],
                                "avg_launch_ms": round(v[0] / max(1, v[1]), 4)} for k, v in other.items()},
                         "gemm": {"share_of_step_time": out["roofline_all_gemms"]["share_of_step_time"]},
                         "per_gemm": [{"epilogue": EPI_NAMES.get(k[0], k[0]), "N": k[1], "K": k[2], "M": k[3], "tile": k[4],
                                       "launches": v[1], "avg_launch_ms": round(v[0] / v[1], 4),
                                       "frac_of_peak": round(v[2] / (v[0] * 1e-3) / 1e12 / peak, 4)}
                                      for k, v in sorted(by_kernel.items(), key=lambda kv: -kv[1][0])[:8]],
                         "instrumented_step_ms": round(inst_s * 1e3, 3)}
    if shards.rank == 0 and len(shards.counts) == 1 and not args.headline_only:
        if n >= 256 and bs != 256:  # the second key: the same epoch in eval batches of 256 clips (1024 frames per call)
            epoch(256)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps):
                epoch(256)
            torch.cuda.synchronize()
            e256 = (time.perf_counter() - t2) / args.steps
            out["eval_batch_256"] = {"value": round(n / e256, 2), "ms_per_step": round(e256 * 1e3, 3),
                                     "frac_of_fp32_mfma_peak": round(flops / e256 / 1e12 / peak, 4)}
        out["hipgraph"] = graph_leg(enc, module, video, ids, min(bs, n), device)
        # secondary, never `value`: the same epoch with `precision="fp32x3"` (fp32 results from three fp16 MFMA products per fp32 product:
        # fc_config.split_gemm = 2) - the reference-shaped call in the mode a user at fp32 tolerance would run
        enc3 = instantiate(load_encoder_config("wise", {"precision": "fp32x3", "num_frames": frames, "weight_for_2": 0.5}, device)).to(device)
        enc3.num_frames = frames
        module3 = TextVideoRetrievalModule(enc3, init_temperature=0.015, n_total=shards.n_total)

        def epoch3():
            with torch.inference_mode():
                for s in range(0, n, bs):
                    module3.validation_step_end(module3.validation_step(
                        {"video": video[s:s + bs], "text": {"input_ids": ids[s:s + bs]}, "video_id": list(range(s, min(n, s + bs)))}))
                return module3.validation_epoch_end()   # (asks the range flag: FC_ERANGE if a value left fp16's range)

        for _ in range(max(1, args.warmup)):
            epoch3()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            metrics3 = epoch3()
        torch.cuda.synchronize()
        e3 = (time.perf_counter() - t3) / args.steps
        k = min(bs, n)
        with torch.inference_mode():
            v1, t1_ = enc(video=video[:k], text={"input_ids": ids[:k]})
            v3, t3_ = enc3(video=video[:k], text={"input_ids": ids[:k]})
        out["fp32_split_mode"] = {
            "precision": "fp32x3", "value": round(n / e3, 2), "ms_per_step": round(e3 * 1e3, 3),
            "speedup_vs_headline": round((elapsed / args.steps) / e3, 3),
            "metrics": metrics3, "metrics_identical_to_fp32_path": all(metrics3[m] == metrics[m] for m in ("r1", "r5", "r10", "mr")),
            "embedding_max_abs_vs_fp32_path": {"video": float((v3 - v1).abs().max()), "text": float((t3_ - t1_).abs().max())},
            "note": "secondary mode, never `value` (see `fp32_split_mode` of the default configuration for its arithmetic)"}
    return out, enc, video, ids


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIG_DEFAULTS),
                    help="c2 = BASELINE configs[1] (default; weak scaling, 256 clips x 8 frames per GPU); c3 = configs[2] (encoder=wise, "
                         "4096 clips x 4 frames through the evaluate loop in eval batches of 32 clips; strong); c4 = configs[3] "
                         "(8192 clips x 16 frames in total, sharded; strong); c5 = configs[4] (512 x 8 teacher+student KD "
                         "training step in total, sharded; strong)")
    ap.add_argument("--total-clips", type=int, default=None, help="c3 / c4 / c5: clips over ALL ranks (default 4096 / 8192 / 512)")
    ap.add_argument("--eval-batch", type=int, default=None, help="c3 / c4: clips per encoder call (default 32 / 128)")
    ap.add_argument("--micro-clips", type=int, default=32, help="c5: clips per micro-batch when a rank's share does not fit (split step)")
    ap.add_argument("--keep-clips", type=int, default=None,
                    help="c5: clips whose activations are kept (default: as many as fit, from the free device memory)")
    ap.add_argument("--headline-only", action="store_true",
                    help="c3: skip the eval-batch-256 and hipGraph legs (profiling runs: one call shape in the kernel trace)")
    ap.add_argument("--sync-batches", action="store_true",
                    help="c3: a host synchronisation after every eval batch (never for a timed number: the reference's loop has none). "
                         "Under `rocprofv3 --pmc` it bounds the queue depth the profiler's packet interception has to carry "
                         "(tools/pmc_scaling_probe.py)")
    ap.add_argument("--all-legs", action="store_true",
                    help="with more than one rank, also run the secondary legs (bf16_mode, fp32_split_mode); default: headline only")
    ap.add_argument("--precision", default="fp32", choices=["bf16", "fp32"],
                    help="headline precision; fp32 = the reference's (default).  bf16 here is for kernel work only")
    ap.add_argument("--no-bf16-mode", action="store_true", help="skip the secondary bf16-operand run")
    ap.add_argument("--no-split-mode", action="store_true", help="skip the secondary split-fp32 run (fp32_split_mode)")
    ap.add_argument("--split6", action="store_true", help="also run the six-product split mode of rounds 2-4 (fp32_split6_mode, precision fp32x6)")
    ap.add_argument("--clips", type=int, default=256, help="c2: clips (= captions) per GPU per step")
    ap.add_argument("--frames", type=int, default=None, help="frames per clip (default 8; c4: 16)")
    ap.add_argument("--chunk-frames", type=int, default=0)
    ap.add_argument("--split-chunk-frames", type=int, default=0, help="frames per pass of the fp32_split_mode leg (0 = the library's default, 768)")
    ap.add_argument("--gemm-tile", type=int, default=0)
    ap.add_argument("--cpu-sample-clips", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--truth-clips", type=int, default=4, help="clips of the CPU sample also evaluated in float64 (0 = skip)")
    ap.add_argument("--no-train-leg", action="store_true",
                    help="skip the (untimed, N = 1 only) KD training-step measurement reported under `kd_training_step`")
    ap.add_argument("--no-plant", action="store_true", help="keep the purely random towers (chance-level retrieval)")
    ap.add_argument("--prune-last-block", action="store_true",
                    help="opt-in: only the pooled rows go through the MLP of the last block (identical embeddings)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only to rehearse N > 1 on one GPU")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check only: no GPU work, rank 0 prints {world, sum of ranks}")
    args = ap.parse_args()
    conf = CONFIG_DEFAULTS[args.config]
    if args.frames is None:
        args.frames = conf["frames"]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:]))  # nothing has touched the GPU in this process

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run:
        # launcher, rendezvous and shard plumbing on the CPU: every rank derives its shard with the code the real run uses and the
        # plans meet on rank 0 through a collective (what an N-rank line would print about its sharding, without the GPU)
        total = torch.tensor([float(rank)])
        counts, eval_batch = plan_shards(args, world)
        mine = Shards(counts, rank, eval_batch)
        calls = 1 if mine.eval_batch is None else -(-mine.n_local // mine.eval_batch)
        plans = [[mine.rank, mine.offset, mine.n_local, calls]]
        if world > 1:
            dist.init_process_group("gloo")
            dist.all_reduce(total)
            gathered = [None] * world
            dist.all_gather_object(gathered, plans[0])
            plans = gathered
            dist.barrier()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "sum_of_ranks": float(total), "config": args.config,
                              "metric": metric_name(args.config, args.frames), "frames": args.frames, "scaling": conf["scaling"],
                              "n_total": mine.n_total, "shards [rank, first clip, clips, encoder calls]": plans}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    # one rank per GPU; modulo the visible devices, so that a launcher that shows each rank only its own GPU (index 0)
    # and the single-GPU gloo rehearsal (every rank on device 0) both work
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # FITCLIP_FORCE_COLLECTIVES=1 under a one-rank torch.distributed.run: the single-GPU rehearsal of the RCCL path
    # (process group, broadcast, all-gathers, barrier and all-reduce all go through the library)
    grouped = world > 1 or (os.environ.get("FITCLIP_FORCE_COLLECTIVES") == "1" and "WORLD_SIZE" in os.environ)
    if grouped:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from fitclip_amd import distributed as D
    from fitclip_amd import synth

    dims = synth.VIT_B_16
    sd = synth.make_state_dict(dims, seed=42)
    counts, eval_batch = plan_shards(args, world)
    shards = Shards(counts, rank, eval_batch)
    n_local, n_total = shards.n_local, shards.n_total
    sharding = (f"{n_total} clips in exact contiguous shards {counts if len(set(counts)) > 1 else f'of {counts[0]}'} over {world} "
                f"rank(s)")
    base = {"metric": metric_name(args.config, args.frames), "unit": "pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "higher_is_better": True, "scaling": conf["scaling"], "vs_baseline": None,
            "dtype": "fp32", "data": "synthetic"}
    collectives = f"{args.backend} process group, {world} rank(s)" if grouped else "none (single process)"

    if args.config == "c5":
        kd = run_kd_config(sd, dims, args, shards, device, args.backend)
        result = {**base, "value": kd["value"], "ms_per_step": kd["ms_per_step"],
                  "config": {"workload": f"KD training step, teacher + student CLIP ViT-B/16 dual forward, {n_total} clips x "
                                         f"{args.frames} frames x 224^2 + {n_total} x 77-token texts in total, half labeled on "
                                         f"every rank -> NCE + KD losses -> student backward -> AdamW (BASELINE configs[4])",
                             "total_clips": n_total, "frames": args.frames, "sharding": sharding,
                             "exchange": "one packed all-gather of the four [n_local, 512] embedding matrices + three "
                                         "all-reduces of the flat gradient buffer per step",
                             "weights": "random init (seed 42) teacher, student = teacher perturbed by 5 %; AdamW lr 3e-7",
                             "arithmetic": "fp32-input MFMA (v_mfma_f32_16x16x4_f32), fp32 accumulate",
                             "collectives": collectives},
                  "roofline": kd["roofline"], "split_step": kd["split_step"], "losses": kd["losses"], "kd_forward": kd["kd_forward"],
                  "peak_memory_gb": kd["peak_memory_gb"]}
        if rank == 0:
            print(json.dumps(result), flush=True)
        if grouped:
            dist.destroy_process_group()
        return

    if args.config == "c3":
        w3, enc3, video3, ids3 = run_wise_config(dims, args, shards, device, args.backend)
        result = {**base, "value": w3["value"], "ms_per_step": w3["ms_per_step"],
                  "config": {"workload": f"encoder=wise (0.5 CLIP + 0.5 student, ViT-B/16, blended on the device), WebVid-val shape: "
                                         f"{n_total} clips x {args.frames} frames x 224^2 + {n_total} x 77-token texts in total "
                                         f"through TextVideoRetrievalModule in eval batches of {shards.eval_batch or n_local} clips "
                                         f"({(shards.eval_batch or n_local) * args.frames} frames per encoder call) -> per-batch NCE "
                                         f"loss -> one all-gather -> T@V^T -> R@k / MedR (BASELINE configs[2])",
                             "total_clips": n_total, "frames": args.frames, "eval_batch": shards.eval_batch or n_local,
                             "sharding": sharding, "weights": "random init (seed 42) CLIP, student = CLIP perturbed by 5 %, "
                                                              "weight_for_2 = 0.5",
                             "arithmetic": "fp32-input MFMA (v_mfma_f32_16x16x4_f32), fp32 accumulate",
                             "collectives": collectives, "encoder_build_s (two models + WiSE blend)": w3["encoder_build_s"]},
                  **{k: w3[k] for k in ("roofline", "roofline_all_gemms", "roofline_whole_path", "time_split", "eval_batch_256",
                                        "hipgraph", "fp32_split_mode") if k in w3},
                  "retrieval": {**{k: v for k, v in w3["metrics"].items()}, "n": n_total, "path": "device, fp32",
                                "note": "purely random towers: chance-level recall; parity is in cpu_baseline"}}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from oracle import clip_oracle as O
            cores = min(host_cores(), 64)
            torch.set_num_threads(cores)
            k = min(args.cpu_sample_clips, n_local)
            sd_t = {kk: vv.detach().float().cpu() for kk, vv in enc3.model.state_dict().items()}
            with torch.inference_mode():
                ev_g, et_g = enc3(video=video3[:k], text={"input_ids": ids3[:k]})
                v_cpu, ids_cpu = video3[:k].cpu(), ids3[:k].cpu()
                O.encode_image(sd_t, v_cpu[0, :1])  # warm the thread pool
                t0 = time.perf_counter()
                ev_ref, et_ref = O.forward(sd_t, v_cpu, {"input_ids": ids_cpu})
                cpu_s = time.perf_counter() - t0
            result["cpu_baseline"] = {"value": round(k / cpu_s, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
                                      "sample": f"{k} clips x {args.frames} frames + {k} texts of the same data through the "
                                                f"blended weights, fp32 PyTorch oracle, {cores} threads, {cpu_s:.1f} s"}
            result["parity_vs_oracle_on_sample"] = {
                "video_max_abs": float((ev_g.cpu() - ev_ref).abs().max()), "text_max_abs": float((et_g.cpu() - et_ref).abs().max()),
                "tolerance": "fp32 path: embeddings <= 2e-5 (SURVEY 8(c))"}
        if rank == 0:
            print(json.dumps(result), flush=True)
        if grouped:
            dist.destroy_process_group()
        return

    video = fill_video(n_local, args.frames, dims.image_resolution, seed=1000 + rank, device=device)
    ids = torch.from_numpy(synth.make_text(n_local, dims, seed=42, first_text=shards.offset)).to(device)
    text = {"input_ids": ids}

    weights_note = "random init (seed 42)"
    unplanted = {k: sd[k].copy() for k in PLANTED}  # the training leg uses the plain random towers
    if not args.no_plant:
        # one model for all ranks: rank 0 plants on (the first 256 of) its own clips, everybody receives the three tensors
        k_plant = min(n_local, 256)
        planted = plant_retrieval_weights(sd, video[:k_plant], ids[:k_plant], dims, device,
                                          block=args.cpu_sample_clips) if rank == 0 else None
        if grouped:
            for k in PLANTED:
                t = torch.from_numpy(planted[k]).to(device) if rank == 0 else torch.empty(sd[k].shape, device=device)
                if args.backend == "gloo":
                    t = t.cpu()
                dist.broadcast(t, src=0)
                sd[k] = t.cpu().numpy()
        else:
            sd.update(planted)
        weights_note += " + planted visual.proj / ln_post.bias / text_projection (retrieval task, see `retrieval`)"
        torch.cuda.empty_cache()

    head, (ev, et, all_ranks) = run_mode(args.precision, sd, video, text, args, shards, device, args.backend, True)
    metrics = D.metrics_from_ranks(all_ranks.cpu().numpy())
    if args.config == "c2":
        workload = (f"CLIP ViT-B/16 dual encoder, {n_local} clips x {args.frames} frames x 224^2 + {n_local} x 77-token texts per "
                    f"GPU -> T@V^T -> ranks (BASELINE configs[1])")
    else:
        workload = (f"CLIP ViT-B/16 dual encoder, {n_total} clips x {args.frames} frames x 224^2 + {n_total} x 77-token texts in "
                    f"total, sharded by clip, eval batches of {shards.eval_batch or n_local} clips -> one all-gather -> row-block "
                    f"T@V^T -> ranks (BASELINE configs[3])")
    result = {
        **base, "value": head["value"], "ms_per_step": head["ms_per_step"],
        "instrumented_repeat_ms_per_step": head["instrumented_repeat_ms_per_step"], "dtype": args.precision,
        "config": {"workload": workload, "clips_per_gpu": n_local if len(set(counts)) == 1 else counts, "frames": args.frames,
                   "weights": weights_note,
                   "arithmetic": "fp32-input MFMA (v_mfma_f32_16x16x4_f32), fp32 accumulate" if args.precision == "fp32"
                                 else "bf16 MFMA operands, fp32 accumulate / residual / LayerNorm / softmax statistics",
                   "prune_last_block": bool(args.prune_last_block),
                   "sharding": f"{sharding}, one RCCL all-gather of embeddings",
                   "collectives": collectives},
        "roofline": head["roofline"], "roofline_all_gemms": head["roofline_all_gemms"],
        "roofline_whole_path": head["roofline_whole_path"], "roofline_vit_forward": head["roofline_vit_forward"],
        "time_split": head["time_split"],
        "retrieval": {**metrics, "n": n_total, "path": f"device, {args.precision}"},
    }
    legs = Legs(result, world)
    secondary = args.config == "c2" and args.precision == "fp32" and (world == 1 or args.all_legs)
    if not secondary and world > 1:
        result["secondary_legs"] = "skipped with more than one rank (headline leg only); --all-legs runs them"

    ev16 = et16 = None

    def bf16_leg():
        torch.cuda.empty_cache()
        b16, (v16, t16, ranks16) = run_mode("bf16", sd, video, text, args, shards, device, args.backend, False)
        m16 = D.metrics_from_ranks(ranks16.cpu().numpy())
        b16["retrieval"] = m16
        b16["recall_delta_vs_fp32_path"] = {k: round(m16[k] - metrics[k], 6) for k in ("r1", "r5", "r10", "mr")}
        b16["embedding_max_abs_vs_fp32_path"] = {"video": float((v16 - ev).abs().max()), "text": float((t16 - et).abs().max())}
        b16["note"] = ("secondary mode: bf16 MFMA operands (fp32 accumulate, fp32 residual stream / LayerNorm / softmax "
                       "statistics); narrower than the reference's fp32, so it is never `value`")
        result["bf16_mode"] = b16
        return v16, t16

    if secondary and not args.no_bf16_mode:
        got = legs.run("bf16_mode", bf16_leg)
        if got is not None:
            ev16, et16 = got

    ev6 = et6 = None

    def split_leg(precision="fp32x3", key="fp32_split_mode"):
        torch.cuda.empty_cache()
        s6, (v6, t6, ranks6) = run_split_mode(sd, video, text, args, shards, device, args.backend, precision)
        m6 = D.metrics_from_ranks(ranks6.cpu().numpy())
        s6["retrieval"] = m6
        s6["recall_delta_vs_fp32_path"] = {k: round(m6[k] - metrics[k], 6) for k in ("r1", "r5", "r10", "mr")}
        s6["ranks_identical_to_fp32_path"] = bool(torch.equal(ranks6, all_ranks))
        s6["embedding_max_abs_vs_fp32_path"] = {"video": float((v6 - ev).abs().max()), "text": float((t6 - et).abs().max())}
        s6["speedup_vs_headline"] = round(s6["value"] / result["value"], 3)
        s6["note"] = ("secondary mode, never `value`: products are formed on the %s pipe, from splits of the fp32 operands that are "
                      "exact to %s - see parity on the CPU sample (`on_sample`) at the fp32 tolerances"
                      % (("fp16", "2^-22 per product (two fp16 planes, three products)") if precision == "fp32x3"
                         else ("bf16", "2^-26 per product (three bf16 planes, six products)")))
        result[key] = s6
        return v6, t6

    if secondary and not args.no_split_mode:
        got = legs.run("fp32_split_mode", split_leg)
        if got is not None:
            ev6, et6 = got
        if args.split6:
            legs.run("fp32_split6_mode", lambda: split_leg("fp32x6", "fp32_split6_mode"))

    if rank == 0 and world == 1 and args.config == "c2" and not args.no_train_leg and n_local >= 4:
        got = legs.run("kd_training_step", lambda: training_leg({**sd, **unplanted}, video, ids, args, dims, device))
        if got is not None:
            result["kd_training_step"] = got

    def cpu_leg():
        from oracle import clip_oracle as O
        cores = min(host_cores(), 64)
        torch.set_num_threads(cores)
        sd_t = O.to_torch(sd)
        k = min(args.cpu_sample_clips, n_local)
        v_cpu, ids_cpu = video[:k].cpu(), ids[:k].cpu()
        with torch.inference_mode():
            O.encode_image(sd_t, v_cpu[0, :1])  # warm the thread pool
            t0 = time.perf_counter()
            ev_ref, et_ref = O.forward(sd_t, v_cpu, {"input_ids": ids_cpu})
            cpu_s = time.perf_counter() - t0
        cpu_model = "unknown"
        try:
            for line in open("/proc/cpuinfo"):
                if line.lower().startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        result["cpu_baseline"] = {"value": round(k / cpu_s, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
                                  "cpu_model": cpu_model,
                                  "sample": f"{k} clips x {args.frames} frames + {k} texts of the same batch, "
                                            f"fp32 PyTorch oracle, {cores} threads, {cpu_s:.1f} s"}
        dv, dt = (ev[:k].cpu() - ev_ref).abs().max().item(), (et[:k].cpu() - et_ref).abs().max().item()
        s_ref = O.retrieval_scores(et_ref, ev_ref)
        ref_ranks = O.ranks_of_target(s_ref, torch.arange(k))
        ref_m = D.metrics_from_ranks(ref_ranks.numpy())

        def sample_metrics(e_t, e_v):
            from fitclip_amd import ops
            s = ops.similarity(e_t[:k].contiguous(), e_v[:k].contiguous())
            r = ops.ranks(s).cpu()
            return D.metrics_from_ranks(r.numpy()), s.cpu(), r

        gpu_m, s_gpu, gpu_ranks = sample_metrics(et, ev)
        result["parity_vs_oracle_on_sample"] = {
            "video_max_abs": dv, "text_max_abs": dt, "score_max_abs": float((s_gpu - s_ref).abs().max()),
            "ranks_identical": bool(torch.equal(gpu_ranks.long(), ref_ranks.long())),
            "tolerance": "fp32 path: embeddings <= 2e-5, scores <= 5e-5, identical ranks (SURVEY 8(c))"}
        top2 = s_ref.topk(2, dim=1).values
        result["retrieval"].update({
            "sample_clips": k, "gpu": gpu_m, "ref": ref_m,
            "delta": {kk: round(gpu_m[kk] - ref_m[kk], 6) for kk in ("r1", "r5", "r10", "mr")},
            "ref_min_top2_margin": float((top2[:, 0] - top2[:, 1]).min()),
            "ref_offdiag_cosine_std": float(torch.nn.functional.normalize(ev_ref, dim=1).mm(
                torch.nn.functional.normalize(ev_ref, dim=1).T)[~torch.eye(k, dtype=torch.bool)].std())})
        if ev6 is not None:
            m6s, s6s, r6 = sample_metrics(et6, ev6)
            result["fp32_split_mode"]["on_sample"] = {
                "gpu": m6s, "ref": ref_m, "delta": {kk: round(m6s[kk] - ref_m[kk], 6) for kk in ("r1", "r5", "r10", "mr")},
                "video_max_abs": (ev6[:k].cpu() - ev_ref).abs().max().item(),
                "text_max_abs": (et6[:k].cpu() - et_ref).abs().max().item(),
                "score_max_abs": float((s6s - s_ref).abs().max()),
                "ranks_identical": bool(torch.equal(r6.long(), ref_ranks.long())),
                "tolerance": "embeddings <= 2e-5, scores <= 5e-5, identical ranks (SURVEY 8(c), the fp32 bar)"}
        if ev16 is not None:
            m16s, s16, r16 = sample_metrics(et16, ev16)
            result["bf16_mode"]["on_sample"] = {
                "gpu": m16s, "ref": ref_m, "delta": {kk: round(m16s[kk] - ref_m[kk], 6) for kk in ("r1", "r5", "r10", "mr")},
                "video_max_abs": (ev16[:k].cpu() - ev_ref).abs().max().item(),
                "text_max_abs": (et16[:k].cpu() - et_ref).abs().max().item(),
                "ranks_identical": bool(torch.equal(r16.long(), ref_ranks.long()))}
        # distance of every path to the TRUTH: the oracle evaluated in float64 on the first clips of the sample (the same
        # check as tests/test_gpu_split2.py::test_mode_is_as_close_to_float64_as_fp32_arithmetic_itself, at the bench's size)
        kt = min(args.truth_clips, k)
        if kt > 0:
            sd64 = {kk: (v.double() if v.is_floating_point() else v) for kk, v in sd_t.items()}
            with torch.inference_mode():
                t0 = time.perf_counter()
                tv64, tt64 = O.forward(sd64, v_cpu[:kt].double(), {"input_ids": ids_cpu[:kt]})
                truth_s = time.perf_counter() - t0

            def dist64(e_v, e_t):
                return {"video_max_abs": float((e_v[:kt].cpu().double() - tv64).abs().max()),
                        "text_max_abs": float((e_t[:kt].cpu().double() - tt64).abs().max())}

            truth = {"sample": f"first {kt} clips + texts of the sample, oracle in float64, {truth_s:.1f} s",
                     "oracle_fp32": dist64(ev_ref, et_ref), "fp32": dist64(ev, et)}
            if ev6 is not None:
                truth[result["fp32_split_mode"]["precision"]] = dist64(ev6, et6)
            if ev16 is not None:
                truth["bf16"] = dist64(ev16, et16)
            result["distance_to_float64"] = truth
        return True

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        legs.run("cpu_baseline", cpu_leg)  # needs the device only to compare the embeddings it already holds
    if legs.failed:
        result["failed_legs"] = legs.failed
    if rank == 0:
        print(json.dumps(result), flush=True)
    if grouped:
        dist.destroy_process_group()
    if legs.failed:
        raise SystemExit(1)  # the headline line above is complete; the exit code says a secondary leg is broken


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Kernel-level GEMM microbenchmark on the shapes of the ViT-B/16 forward (tuning aid, not the headline bench).

    python tools/bench_gemm.py [--frames 256] [--precision bf16] [--reps 20]
Prints one line per (shape, epilogue, tile): ms, TFLOP/s, fraction of the dense MFMA peak.  Operands are random
(uniform-ish normal), never zero-filled (guide rule 25).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitclip_amd import ops  # noqa: E402

PEAK = {"bf16": 2500.0, "fp32": 157.3}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--tiles", default="1,2")
    ap.add_argument("--only", default="", help="comma list of shape names (default: all four)")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.precision == "bf16" else torch.float32
    M = args.frames * 197
    shapes = [("qkv", 2304, 768, ops.EPI_BIAS_T), ("out_proj", 768, 768, ops.EPI_RESID_F32),
              ("c_fc", 3072, 768, ops.EPI_GELU_T), ("c_proj", 768, 3072, ops.EPI_RESID_F32),
              # (lab: the projections' shapes without the residual read - only with --only)
              ("out_proj_bias", 768, 768, ops.EPI_BIAS_T), ("c_proj_bias", 768, 3072, ops.EPI_BIAS_T)]
    g = torch.Generator(device="cuda").manual_seed(0)
    only = [x for x in args.only.split(",") if x]
    for name, N, K, epi in shapes:
        if (only and name not in only) or (not only and name.endswith("_bias")):
            continue
        a = torch.randn((M, K), generator=g, device="cuda").to(dt)
        w = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).to(dt)
        bias = torch.randn((N,), generator=g, device="cuda")
        out = torch.zeros((M, N), device="cuda", dtype=torch.float32 if epi == ops.EPI_RESID_F32 else dt)
        for tile in [int(t) for t in args.tiles.split(",")]:
            for _ in range(3):
                ops.gemm(a, w, bias, epi, out=out, tile=tile)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.reps):
                ops.gemm(a, w, bias, epi, out=out, tile=tile)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / args.reps
            tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
            print(f"{args.precision} {name:9s} M={M} N={N} K={K} tile={tile}: {ms:8.3f} ms  {tf:8.1f} TF/s  "
                  f"{tf / PEAK[args.precision]:.3f} of peak", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""The four block GEMMs of the TEXT tower at small batches (32 / 64 / 128 captions = 2464 / 4928 / 9856 rows), fp32: the
one-tile-per-workgroup kernels (tile 1 = 128 x 128, 8 = 64 x 64 on a four-stage ring) and what `tile = 0` resolves to."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitclip_amd import ops  # noqa: E402


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for texts in (32, 64, 128):
    M = texts * 77
    for name, N, K, epi in (("qkv", 1536, 512, ops.EPI_BIAS_T), ("out_proj", 512, 512, ops.EPI_RESID_F32),
                            ("c_fc", 2048, 512, ops.EPI_GELU_T), ("c_proj", 512, 2048, ops.EPI_RESID_F32)):
        a = torch.randn((M, K), generator=g, device="cuda")
        w = torch.randn((N, K), generator=g, device="cuda") * K ** -0.5
        bias = torch.randn((N,), generator=g, device="cuda")
        out = torch.zeros((M, N), device="cuda")
        line = f"texts={texts} {name:8s} M={M} N={N} K={K}:"
        for tile in (0, 1, 8):
            us = timed(lambda: ops.gemm(a, w, bias, epi, out=out, tile=tile))
            line += f"  tile {tile}: {us:6.1f} us ({2.0 * M * N * K / us / 1e6 / 157.3:.3f})"
        print(line, flush=True)

#!/usr/bin/env python
"""What a phase of the persistent fp32 GEMM costs when every CU owns exactly ONE tile of a given height: N = 768 / 2304,
K = 768 / 3072, M = 85 x (64 h) rows (255 / 765 tiles), forced cuts (fc_gemm tile 4 = full tiles, 5..7 = tails of 1..3 units with
an empty head).  Prints us per launch next to the MFMA-issue time of the tile (tools/README.md)."""
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
for N, K in ((768, 768), (768, 3072), (2304, 768)):
    w = torch.randn((N, K), generator=g, device="cuda") * K ** -0.5
    bias = torch.randn((N,), generator=g, device="cuda")
    for h, tile in ((4, 4), (1, 5), (2, 6), (3, 7)):
        panels = 255 // (N // 256)
        M = panels * 64 * h  # one tile per CU (tail cuts: M < 256 rows per panel count keeps the head empty when hp = panels - 1 = 0?)
        a = torch.randn((M, K), generator=g, device="cuda")
        out = torch.zeros((M, N), device="cuda")
        us = timed(lambda: ops.gemm(a, w, bias, ops.EPI_RESID_F32, out=out, tile=tile))
        ideal = 2.0 * 64 * h * 256 * K / (157.3e12 / 256) * 1e6
        print(f"N={N} K={K} h={h} M={M} tiles={panels * (N // 256)}: {us:7.1f} us per launch; one tile's MFMA issue {ideal:6.1f} us; "
              f"plan {ops.gemm_plan(M, N, K)}", flush=True)

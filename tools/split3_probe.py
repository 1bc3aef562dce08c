#!/usr/bin/env python
"""The four three-plane split-fp32 GEMMs of one 768-frame pass (M = 151 296) through the C ABI: us per launch and share of the
bf16 MFMA peak (six products per fp32 product).  With the lab library (FITCLIP_HIP_LIB=tools/bin/libfitclip_hip_lab.so) the
variant comes from FITCLIP_LAB_SPLIT3 (bit 0: nt activation loads, bit 1: round-robin tile deal, 4: no N split, 8: N split over
2 XCD groups).  Under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` the per-kernel counters give the traffic per launch
(tools/split3_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitclip_amd import ops  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
only = sys.argv[3] if len(sys.argv) > 3 else ""
M = frames * 197
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in (("qkv", 2304, 768, ops.EPI_BIAS_F32), ("out_proj", 768, 768, ops.EPI_RESID3_F32),
                        ("c_fc", 3072, 768, ops.EPI_GELU_X3), ("c_proj", 768, 3072, ops.EPI_RESID3_F32)):
    if only and name != only:
        continue
    a3 = ops.split3(torch.randn((M, K), generator=g, device="cuda"))
    w3 = ops.split3(torch.randn((N, K), generator=g, device="cuda") * K ** -0.5)
    bias = torch.randn((N,), generator=g, device="cuda")
    out = torch.zeros((M, N), device="cuda") if epi == ops.EPI_RESID3_F32 else None
    for _ in range(3):
        ops.gemm_split3(a3, w3, bias, epi, out=out)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        ops.gemm_split3(a3, w3, bias, epi, out=out)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"variant {os.environ.get('FITCLIP_LAB_SPLIT3', '-')} {name:8s} M={M} N={N} K={K}: {ms * 1e3:8.1f} us  "
          f"{12.0 * M * N * K / ms / 1e9 / 2500:.4f} of the bf16 peak", flush=True)
    del a3, w3, out

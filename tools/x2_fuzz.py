#!/usr/bin/env python
"""Random-shape screen of the two kernels of precision "fp32x3" (GPU): `fc_gemm_split2` over random (M, N, K, epilogue, tile height)
and the three-product attention over random (sequences, S in 193..208, heads), every element against FLOAT64 (rocBLAS dgemm / a float64 softmax of
the same operands) next to the fp32-MFMA kernels' own error, guard rows / columns around every output, and a second run for bit-equality.  The parity tests
(tests/test_gpu_split2.py) pin chosen shapes; this looks for the shape nobody chose.

    python tools/x2_fuzz.py [--cases 150] [--seed 0]      ->  one line per case, a summary line, exit code 1 on any failure"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from fitclip_amd import ops  # noqa: E402

DEV = "cuda"
GUARD = 1234.0


def value(x2):
    h1, h2 = ops.x2_planes(x2)
    return h1.float() + h2.float() / 2048.0


def gemm_case(rng, i):
    K = 64 * int(rng.integers(2, 49))                      # 128 .. 3072
    N = 32 * int(rng.integers(1, 97))                      # 32 .. 3072
    M = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 5000)), int(rng.integers(5000, 70000))]))
    epi = str(rng.choice(["bias", "resid", "gelu"]))
    cut = int(rng.integers(0, 4))
    g = torch.Generator(device=DEV).manual_seed(1000 + i)
    a = torch.randn(M, K, device=DEV, generator=g) * float(10.0 ** rng.uniform(-2, 1.5))
    w = torch.randn(N, K, device=DEV, generator=g) * float(10.0 ** rng.uniform(-3, 0.5)) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    w2, sc = ops.split2_weight(w)
    a2 = ops.split2(a)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    z32 = ops.gemm(a, w, bias, ops.EPI_BIAS_T)
    z64 = torch.addmm(bias.double(), a.double(), w.double().T)          # the truth (rocBLAS dgemm of the same fp32 operands)
    mag = float(z64.abs().max()) + 1e-30
    e32 = float((z32.double() - z64).abs().max()) / mag                 # what the fp32-MFMA kernel makes of them
    pad = 5                                                  # guard rows behind the output
    ok, detail = True, ""
    for run in range(2):
        if epi == "gelu":
            buf = torch.full((M + pad, 2 * N), GUARD, device=DEV, dtype=torch.float16)
            ops.gemm_split2(a2, w2, sc, bias, ops.EPI_GELU_X2, out=buf[:M], cut=cut, flag=flag)
            got, want = value(buf[:M]), z64 * torch.sigmoid(1.702 * z64)
        else:
            buf = torch.full((M + pad, N), GUARD, device=DEV)
            if epi == "resid":
                x = torch.randn(M, N, device=DEV, generator=torch.Generator(device=DEV).manual_seed(7 + i)) * 3
                buf[:M] = x
                ops.gemm_split2(a2, w2, sc, bias, ops.EPI_RESID3_F32, out=buf[:M], cut=cut)
                got, want = buf[:M], x.double() + z64
            else:
                ops.gemm_split2(a2, w2, sc, bias, ops.EPI_BIAS_F32, out=buf[:M], cut=cut)
                got, want = buf[:M], z64
        err = float((got.double() - want.double()).abs().max()) / mag
        guards = bool((buf[M:].float() == GUARD).all())
        if run == 0:
            first, first_err = got.clone(), err
        else:
            same = bool(torch.equal(got, first))
            # as accurate as the fp32-MFMA kernel on the same operands (the x2 rows of the QuickGELU output hold 22 bits: + 2.4e-7)
            slack = 4e-7 if epi == "gelu" else 1e-7 + (1.2e-7 * float(want.abs().max()) / mag if epi == "resid" else 0.0)  # (+ the rounding of x + z)
            ok = first_err < 1.5 * e32 + slack and guards and same
            detail = f"err {first_err:.1e} (fp32-MFMA kernel {e32:.1e}) guards {guards} rerun-equal {same} flag {int(flag)}"
    return ok, f"gemm  M={M:6d} N={N:5d} K={K:5d} {epi:5s} cut={cut}  {detail}"


def attention_case(rng, i):
    S = int(rng.integers(193, 209))
    heads = int(rng.choice([1, 2, 3, 8, 12, 16]))
    n_seq = int(rng.choice([1, 2, int(rng.integers(3, 40)), int(rng.integers(40, 400))]))
    g = torch.Generator(device=DEV).manual_seed(5000 + i)
    qkv = torch.randn(n_seq * S, 3 * heads * 64, device=DEV, generator=g) * float(10.0 ** rng.uniform(-1, 0.5))
    o32 = ops.attention(qkv, n_seq, S, heads)
    D = heads * 64
    q, k, v = (t.double().view(n_seq, S, heads, 64).transpose(1, 2) for t in qkv.split(D, dim=1))
    o64 = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v).transpose(1, 2).reshape(n_seq * S, D)
    mag = float(o64.abs().max()) + 1e-30
    e32 = float((o32.double() - o64).abs().max()) / mag
    a = ops.attention(qkv, n_seq, S, heads, split=True, two_plane=True, three_products=True)
    b = ops.attention(qkv, n_seq, S, heads, split=True, two_plane=True, three_products=True)
    err = float((value(a).double() - o64).abs().max()) / mag
    same = bool(torch.equal(a, b))
    finite = bool(torch.isfinite(a.float()).all())
    return (err < 1.5 * e32 + 4e-7 and same and finite,
            f"attn  n_seq={n_seq:4d} S={S} heads={heads:2d}  err {err:.1e} (fp32-MFMA kernel {e32:.1e}) rerun-equal {same} finite {finite}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import numpy as np
    rng = np.random.default_rng(args.seed)
    bad, t0 = 0, time.time()
    for i in range(args.cases):
        ok, line = (attention_case if i % 3 == 2 else gemm_case)(rng, i)
        bad += not ok
        print(("ok   " if ok else "FAIL ") + line, flush=True)
    torch.cuda.synchronize()
    print(f"{args.cases} cases, {bad} failed, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

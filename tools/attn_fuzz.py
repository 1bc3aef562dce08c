#!/usr/bin/env python
"""Random-shape screen of the fp32 attention kernels (GPU): `fc_attention` precision fp32 over random (sequences, S, heads, causal,
magnitude) - every kernel behind it: the streaming-block kernel of 113..224 tokens (log2-domain softmax), the whole-K/V MFMA
kernels, the thread-per-query kernel beyond 288 - every element against a float64 softmax of the same operands computed on the
GPU (next to the error of the same expression in torch float32), a second run for bit-equality, and the result of a sub-batch against the same rows of the full batch.

    python tools/attn_fuzz.py [--cases 400] [--seed 0]"""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from fitclip_amd import ops  # noqa: E402

DEV = "cuda"


def case(rng, i):
    S = int(rng.choice([int(rng.integers(1, 97)), int(rng.integers(97, 225)), int(rng.integers(97, 225)), int(rng.integers(225, 289)),
                        int(rng.integers(289, 600))]))
    heads = int(rng.choice([1, 2, 3, 8, 12, 16]))
    causal = bool(rng.integers(0, 2)) and S <= 224
    n_seq = int(rng.choice([1, 2, int(rng.integers(3, 40)), int(rng.integers(40, 300))]))
    if n_seq * S * heads > 3_000_000:
        n_seq = max(1, 3_000_000 // (S * heads))
    g = torch.Generator(device=DEV).manual_seed(9000 + i)
    qkv = torch.randn(n_seq * S, 3 * heads * 64, device=DEV, generator=g) * float(10.0 ** rng.uniform(-1, 0.6))
    D = heads * 64
    q, k, v = (t.double().view(n_seq, S, heads, 64).transpose(1, 2) for t in qkv.split(D, dim=1))
    s = q @ k.transpose(-1, -2) * 0.125
    if causal:
        s = s + torch.full((S, S), float("-inf"), device=DEV, dtype=torch.float64).triu_(1)
    o64 = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(n_seq * S, D)
    mag = float(o64.abs().max()) + 1e-30
    # the same expression in float32 (rocBLAS sgemm + torch.softmax): what plain fp32 arithmetic makes of these operands
    q32, k32, v32 = (t.view(n_seq, S, heads, 64).transpose(1, 2) for t in qkv.split(D, dim=1))
    s32 = (q32 * 0.125) @ k32.transpose(-1, -2)
    if causal:
        s32 = s32 + torch.full((S, S), float("-inf"), device=DEV).triu_(1)
    o32 = (torch.softmax(s32, dim=-1) @ v32).transpose(1, 2).reshape(n_seq * S, D)
    e32 = float((o32.double() - o64).abs().max()) / mag
    a = ops.attention(qkv, n_seq, S, heads, causal=causal)
    b = ops.attention(qkv, n_seq, S, heads, causal=causal)
    err = float((a.double() - o64).abs().max()) / mag
    same = bool(torch.equal(a, b))
    few = max(1, n_seq // 3)
    sub = bool(torch.equal(ops.attention(qkv[: few * S].contiguous(), few, S, heads, causal=causal), a[: few * S]))
    # (the MFMA kernels stay within 2.3x torch's float32 error over 480 cases, median 1.0x; the thread-per-query kernel behind
    # S > 288, sequential over the keys, within 3.5x, median 1.2x)
    ok = err < (4.0 if S > 288 else 2.5) * e32 + 5e-7 and same and sub and bool(torch.isfinite(a).all())
    return ok, (f"attn  n_seq={n_seq:4d} S={S:3d} heads={heads:2d} causal={int(causal)}  err {err:.1e} (torch float32 {e32:.1e}) rerun-equal {same} "
                f"sub-batch-equal {sub}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad, t0, worst = 0, time.time(), 0.0
    for i in range(args.cases):
        ok, line = case(rng, i)
        bad += not ok
        print(("ok   " if ok else "FAIL ") + line, flush=True)
    torch.cuda.synchronize()
    print(f"{args.cases} cases, {bad} failed, {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

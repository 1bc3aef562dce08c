"""Probe for DESIGN.md 9, "exact-fp32 results from the bf16 matrix pipe": an fp32 number is the sum of three bf16
numbers (a = a1 + a2 + a3), so a.b is recovered in fp32 accumulation from six bf16 products (a1b1 a1b2 a2b1 a2b2 a1b3
a3b1; the dropped terms are below 2^-26 per product).  Laid out along K - every 32 columns of A replaced by the six
planes [a1 a1 a2 a2 a1 a3], of W by [b1 b2 b1 b2 b3 b1] - this is an ordinary bf16 GEMM with K' = 6 K, so the EXISTING
bf16 kernel measures what the idea is worth before any kernel is written for it:
  accuracy  fp32-MFMA GEMM and the 6-product bf16 GEMM (fp32 output) against float64, on the c_fc / c_proj shapes
  speed     the persistent pipelined bf16 kernel at K' = 6 K (bf16 output) against the fp32-MFMA kernel
Nothing in the product uses this.    python tools/x3_probe.py [frames=512]
"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fitclip_amd import ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 512
M = frames * 197
g = torch.Generator(device='cuda').manual_seed(0)


def planes(x):
    p1 = x.bfloat16()
    r = x - p1.float()
    p2 = r.bfloat16()
    p3 = (r - p2.float()).bfloat16()
    return p1, p2, p3


def expand(ps, order):
    rows, K = ps[0].shape
    chunks = [ps[i].view(rows, K // 32, 1, 32) for i in order]
    return torch.cat(chunks, dim=2).reshape(rows, 6 * K).contiguous()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best


for name, N, K in (("c_fc", 3072, 768), ("c_proj", 768, 3072)):
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    a6 = expand(planes(a), (0, 0, 1, 1, 0, 2))
    w6 = expand(planes(w), (0, 1, 0, 1, 2, 0))
    rows = slice(0, 2048)
    ref = a[rows].double() @ w.double().T
    scale = float(ref.abs().max())
    f32 = ops.gemm(a[rows].contiguous(), w, None, ops.EPI_STORE_F32)
    x6 = ops.gemm(a6[rows].contiguous(), w6, None, ops.EPI_STORE_F32)
    b16 = ops.gemm(a[rows].bfloat16().contiguous(), w.bfloat16(), None, ops.EPI_STORE_F32)
    err = lambda y: float((y.double() - ref).abs().max()) / scale
    bias32, bias16 = torch.zeros(N, device='cuda'), torch.zeros(N, device='cuda')
    t_f32 = timed(lambda: ops.gemm(a, w, bias32, ops.EPI_BIAS_T))
    t_x6 = timed(lambda: ops.gemm(a6, w6, bias16, ops.EPI_BIAS_T))
    fl = 2.0 * M * N * K
    print("%-7s M=%d N=%d K=%d | max err / max|ref|: fp32 MFMA %.2e, six bf16 products %.2e, plain bf16 %.2e | "
          "fp32 MFMA %.3f ms = %.1f TF/s; bf16 kernel at K'=6K %.3f ms = %.1f TF/s fp32-equivalent (%.0f TF/s of bf16 MFMA), x%.2f"
          % (name, M, N, K, err(f32), err(x6), err(b16), t_f32 * 1e3, fl / t_f32 / 1e12, t_x6 * 1e3, fl / t_x6 / 1e12,
             6 * fl / t_x6 / 1e12, t_f32 / t_x6))

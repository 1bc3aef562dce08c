"""Op-level check and timing of the split-fp32 ("x6") GEMM path: fc_split6, fc_gemm epilogues EPI_BIAS_F32 / EPI_GELU_X6."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fitclip_amd import ops

g = torch.Generator(device='cuda').manual_seed(0)


def planes(x):
    p1 = x.bfloat16(); r = x - p1.float(); p2 = r.bfloat16(); p3 = (r - p2.float()).bfloat16()
    return p1, p2, p3


def expand(ps, order):
    rows, K = ps[0].shape
    return torch.cat([ps[i].view(rows, K // 32, 1, 32) for i in order], dim=2).reshape(rows, 6 * K).contiguous()


def unpack(o6):  # [M, 6 N] activation-side image -> the six planes [M, N]
    M = o6.shape[0]
    v = o6.view(M, -1, 6, 32)
    return [v[:, :, i].reshape(M, -1) for i in range(6)]


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
    return best


M = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 197
for name, N, K in (("qkv", 2304, 768), ("c_fc", 3072, 768), ("c_proj", 768, 3072), ("out_proj", 768, 768)):
    a = torch.randn(M, K, device='cuda', generator=g)
    w = torch.randn(N, K, device='cuda', generator=g) / K ** 0.5
    bias = torch.randn(N, device='cuda', generator=g)
    a6, w6 = ops.split6(a), ops.split6(w, weight=True)
    assert torch.equal(a6, expand(planes(a), (0, 0, 1, 1, 0, 2))) and torch.equal(w6, expand(planes(w), (0, 1, 0, 1, 2, 0)))
    rows = slice(M - 3000, M)  # includes the ragged last tile
    ref = a[rows].double() @ w.double().T + bias.double()
    y = ops.gemm(a6, w6, bias, ops.EPI_BIAS_F32)
    y32 = ops.gemm(a, w, bias, ops.EPI_BIAS_T)
    scale = float(ref.abs().max())
    e6, e32 = float((y[rows].double() - ref).abs().max()) / scale, float((y32[rows].double() - ref).abs().max()) / scale
    msg = "%-8s M=%d N=%d K=%d  err/max: x6 %.2e fp32-MFMA %.2e" % (name, M, N, K, e6, e32)
    t32 = timed(lambda: ops.gemm(a, w, bias, ops.EPI_BIAS_T))
    t6 = timed(lambda: ops.gemm(a6, w6, bias, ops.EPI_BIAS_F32))
    msg += " | bias->f32: fp32-MFMA %.3f ms, x6 %.3f ms (x%.2f, %.0f TF/s fp32-equivalent)" % (t32 * 1e3, t6 * 1e3, t32 / t6, 2.0 * M * N * K / t6 / 1e12)
    if name == "c_fc":
        h6 = ops.gemm(a6, w6, bias, ops.EPI_GELU_X6)
        p = unpack(h6)
        assert torch.equal(p[0], p[1]) and torch.equal(p[0], p[4]) and torch.equal(p[2], p[3])
        h = p[0].float() + p[2].float() + p[5].float()              # exact: the three planes add up to an fp32 number
        q1, q2, q3 = planes(h)
        assert torch.equal(q1, p[0]) and torch.equal(q2, p[2]) and torch.equal(q3, p[5])  # and ARE its canonical split
        x = y.double()
        want = x * torch.sigmoid(1.702 * x)
        eg = float((h.double() - want).abs().max() / want.abs().max())
        tg32 = timed(lambda: ops.gemm(a, w, bias, ops.EPI_GELU_T))
        tg6 = timed(lambda: ops.gemm(a6, w6, bias, ops.EPI_GELU_X6))
        msg += " | gelu: err %.2e, fp32-MFMA %.3f ms, x6 -> six planes %.3f ms (x%.2f)" % (eg, tg32 * 1e3, tg6 * 1e3, tg32 / tg6)
    print(msg)
ts = timed(lambda: ops.split6(a))
print("split6 of [%d, %d]: %.3f ms = %.2f TB/s" % (M, a.shape[1], ts * 1e3, M * a.shape[1] * 16 / ts / 1e12))

"""Would the TEXT tower's block GEMMs gain from the three-product kernel?  The four shapes of a CLIP text block (width 512) at the bench
batch (256 captions x 77 tokens = 19 712 rows) and at the reference's eval batch (32 captions = 2 464 rows): ms per launch of the fp32
path's kernel (what the tower runs today) against fc_gemm_split2 on the same operands.
    python tools/text_gemm_x2_probe.py"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import ops


def timed(fn, reps=20, rounds=5):
    fn()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[len(ts) // 2]


g = torch.Generator(device="cuda").manual_seed(0)
for captions in (256, 128, 64, 54, 32, 24, 16, 8):
    M = captions * 77
    tot32 = tot2 = 0.0
    for name, N, K, epi32, epi2 in (("qkv", 1536, 512, ops.EPI_BIAS_T, ops.EPI_BIAS_F32), ("out_proj", 512, 512, ops.EPI_BIAS_T, ops.EPI_BIAS_F32),
                                    ("c_fc", 2048, 512, ops.EPI_GELU_T, ops.EPI_GELU_X2), ("c_proj", 512, 2048, ops.EPI_BIAS_T, ops.EPI_BIAS_F32)):
        a = torch.randn(M, K, device="cuda", generator=g)
        w = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
        bias = torch.zeros(N, device="cuda")
        t32 = timed(lambda: ops.gemm(a, w, bias, epi32))
        a2 = ops.split2(a)
        w2, sc = ops.split2_weight(w)
        t2 = timed(lambda: ops.gemm_split2(a2, w2, sc, bias, epi2))
        tot32 += t32
        tot2 += t2
        print(f"{captions:3d} captions {name:8s} M={M} N={N} K={K}: fp32 kernel {t32 * 1e3:7.1f} us, three-product kernel {t2 * 1e3:7.1f} us ({t32 / t2:.2f}x)", flush=True)
    print(f"{captions:3d} captions: the four GEMMs of a block {tot32 * 1e3:.0f} -> {tot2 * 1e3:.0f} us; x 12 blocks {12 * tot32:.2f} -> {12 * tot2:.2f} ms", flush=True)

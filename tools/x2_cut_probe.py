"""The four block GEMMs of the split_gemm = 2 mode at the reference-shaped call's size (M rows, default 25 216 = 128 frames) under
the forced row cuts of fc_gemm_split2: 1 = 256-row tiles only, 2 = 128-row tiles only, 3 = head of whole tile rounds + 128-row
tail, 0 = what the launcher plans.  ms per launch (median of rounds).     python tools/x2_cut_probe.py [M]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25216
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in (("qkv", 2304, 768, ops.EPI_BIAS_F32), ("out_proj", 768, 768, ops.EPI_RESID3_F32),
                        ("c_fc", 3072, 768, ops.EPI_GELU_X2), ("c_proj", 768, 3072, ops.EPI_RESID3_F32)):
    a2 = ops.split2(torch.randn(M, K, device="cuda", generator=g))
    w2, sc = ops.split2_weight(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    x = torch.zeros(M, N, device="cuda")
    res = {}
    for rnd in range(5):
        for cut in (1, 2, 0):
            kw = {"out": x} if epi == ops.EPI_RESID3_F32 else {}
            ops.gemm_split2(a2, w2, sc, bias, epi, cut=cut, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_split2(a2, w2, sc, bias, epi, cut=cut, **kw)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(cut, []).append(e0.elapsed_time(e1) / 20)
    print(f"{name:9s} M={M} N={N} K={K}: " + "  ".join(f"cut {c}: {sorted(v)[len(v) // 2]:.4f} ms" for c, v in res.items()), flush=True)

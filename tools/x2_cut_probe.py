"""The block GEMMs of the split_gemm = 2 mode at M rows under the forced tile heights of fc_gemm_split2: cut 1 = 256-row tiles, 3 = 192-row
tiles, 2 = 128-row tiles, 0 = what the launcher picks.  ms per launch (median of interleaved rounds); the outputs of the forced heights
are compared bitwise (the result does not depend on the tile height).
    python tools/x2_cut_probe.py [M] [visual|text]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 25216
which = sys.argv[2] if len(sys.argv) > 2 else "visual"
w = 768 if which == "visual" else 512
g = torch.Generator(device="cuda").manual_seed(0)
for name, N, K, epi in (("qkv", 3 * w, w, ops.EPI_BIAS_F32), ("out_proj", w, w, ops.EPI_RESID3_F32),
                        ("c_fc", 4 * w, w, ops.EPI_GELU_X2), ("c_proj", w, 4 * w, ops.EPI_RESID3_F32)):
    a2 = ops.split2(torch.randn(M, K, device="cuda", generator=g))
    w2, sc = ops.split2_weight(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
    bias = torch.randn(N, device="cuda", generator=g)
    x = torch.zeros(M, N, device="cuda")
    outs = {}
    for cut in (1, 3, 2, 0):
        x.zero_()
        kw = {"out": x} if epi == ops.EPI_RESID3_F32 else {}
        outs[cut] = ops.gemm_split2(a2, w2, sc, bias, epi, cut=cut, **kw).clone()
    same = all(torch.equal(outs[1], outs[c]) for c in (3, 2, 0))
    res = {}
    for rnd in range(5):
        for cut in (1, 3, 2, 0):
            kw = {"out": x} if epi == ops.EPI_RESID3_F32 else {}
            ops.gemm_split2(a2, w2, sc, bias, epi, cut=cut, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.gemm_split2(a2, w2, sc, bias, epi, cut=cut, **kw)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(cut, []).append(e0.elapsed_time(e1) / 20)
    med = {c: sorted(v)[len(v) // 2] for c, v in res.items()}
    print(f"{name:9s} M={M} N={N} K={K}: 256 rows {med[1]:.4f}  192 rows {med[3]:.4f}  128 rows {med[2]:.4f}  launcher {med[0]:.4f} ms;  bitwise equal {same}", flush=True)

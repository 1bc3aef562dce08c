"""What the attention of precision "fp32x3" costs per frame on the reference's CLIP geometries: ViT-B/16
(197 tokens, 12 heads) has the fused three-product kernel (attention_split2.hip, x2 rows out); B/32 (50 tokens), L/14 (257 tokens,
16 heads) and L/14@336 (577) take the fp32 attention kernel + a split pass (fc_split2) over its output.  Prints ms per launch, us
per frame, and the share of a block's time the attention has next to the block's four three-product GEMMs at the same batch.
    python tools/attn_geometry_probe.py [frames per launch]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import ops

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 512


def timed(fn, reps=10, rounds=5):
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


for name, S, width in (("ViT-B/32", 50, 768), ("ViT-B/16", 197, 768), ("ViT-L/14", 257, 1024), ("ViT-L/14@336", 577, 1024)):
    heads = width // 64
    n = frames if S < 300 else max(64, frames // 2)
    g = torch.Generator(device="cuda").manual_seed(S)
    qkv = torch.randn(n * S, 3 * width, device="cuda", generator=g)
    t_f32 = timed(lambda: ops.attention(qkv, n, S, heads))
    o = ops.attention(qkv, n, S, heads)
    t_split = timed(lambda: ops.split2(o))
    line = f"{name:13s} S={S:3d} heads={heads:2d} frames={n}: fp32 attention {t_f32:7.3f} ms + split pass {t_split:6.3f} ms = {1e3 * (t_f32 + t_split) / n:6.2f} us/frame"
    if 193 <= S <= 208:
        t_fused = timed(lambda: ops.attention(qkv, n, S, heads, split=True, two_plane=True, three_products=True))
        line += f"; fused three-product kernel {t_fused:7.3f} ms = {1e3 * t_fused / n:6.2f} us/frame ({(t_f32 + t_split) / t_fused:.2f}x)"
    # the block's four GEMMs on three products at the same batch
    M = n * S
    a2 = ops.split2(torch.randn(M, width, device="cuda", generator=g))
    h2 = ops.split2(torch.randn(M, 4 * width, device="cuda", generator=g))
    t_gemm = 0.0
    for (N, K, epi, src) in ((3 * width, width, ops.EPI_BIAS_F32, a2), (width, width, ops.EPI_BIAS_F32, a2), (4 * width, width, ops.EPI_GELU_X2, a2),
                             (width, 4 * width, ops.EPI_BIAS_F32, h2)):
        w2, sc = ops.split2_weight(torch.randn(N, K, device="cuda", generator=g) / K ** 0.5)
        bias = torch.zeros(N, device="cuda")
        t_gemm += timed(lambda: ops.gemm_split2(src, w2, sc, bias, epi), reps=5, rounds=3)
    share = (t_f32 + t_split) / (t_f32 + t_split + t_gemm)
    line += f"; the block's four GEMMs {t_gemm:7.3f} ms -> attention = {100 * share:4.1f} % of (attention + GEMMs)"
    if 193 <= S <= 208:
        line += f" unfused, {100 * t_fused / (t_fused + t_gemm):4.1f} % fused"
    print(line, flush=True)

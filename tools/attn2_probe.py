"""The attention of the split-fp32 modes on one bench pass: the fp32-MFMA kernel, the six-product bf16 kernel (x3 and x2 rows out)
and the three-product fp16 kernel (attention_split2.hip) - error against float64 on sampled (sequence, head) pairs and ms per launch.
    python tools/attn2_probe.py [n_seq] [S] [scale of the random qkv]"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import ops
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 768
S = int(sys.argv[2]) if len(sys.argv) > 2 else 197
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
heads = 12
g = torch.Generator(device="cuda").manual_seed(S)
qkv = torch.randn(n_seq * S, 3 * heads * 64, device="cuda", generator=g) * scale
qkv[:S, : heads * 64] *= 3.0   # peaked rows too


def f64(q, n):
    a, b, c = (t.reshape(n, S, heads, 64).permute(0, 2, 1, 3).double() for t in q.chunk(3, dim=1))
    p = torch.softmax(a @ b.transpose(-1, -2) / 8.0, dim=-1)
    return (p @ c).permute(0, 2, 1, 3).reshape(n * S, heads * 64)


def x3val(o):
    p = ops.x3_planes(o)
    return p[0].double() + p[1].double() + p[2].double()


def x2val(o):
    h1, h2 = ops.x2_planes(o)
    return h1.double() + h2.double() / 2048.0


forms = {
    "fp32-MFMA kernel (fp32 rows)": (lambda: ops.attention(qkv, n_seq, S, heads), lambda o: o.double()),
    "six bf16 products, x3 rows": (lambda: ops.attention(qkv, n_seq, S, heads, split=True), x3val),
    "six bf16 products, x2 rows": (lambda: ops.attention(qkv, n_seq, S, heads, split=True, two_plane=True), x2val),
    "three fp16 products, x2 rows": (lambda: ops.attention(qkv, n_seq, S, heads, split=True, two_plane=True, three_products=True), x2val),
}
some = torch.cat([torch.arange(0, 3 * S), torch.arange((n_seq // 2) * S, (n_seq // 2 + 1) * S), torch.arange((n_seq - 2) * S, n_seq * S)]).cuda()
ref = f64(qkv[some], some.numel() // S)
big = float(ref.abs().max())
for name, (fn, val) in forms.items():
    out = fn()
    v = val(out)
    err = float((v[some] - ref).abs().max()) / big
    rms = float((v[some] - ref).pow(2).mean().sqrt()) / big
    again = fn()
    same = bool(torch.equal(out, again))
    times = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 10)
    print(f"{name:32s} max err {err:.2e} rms {rms:.2e} of the largest output; run-to-run equal {same}; {sorted(times)[2]:.3f} ms per {n_seq} x {S} x {heads}", flush=True)

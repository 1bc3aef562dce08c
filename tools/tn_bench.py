"""Weight-gradient GEMM (fc_gemm_tn) at the ViT-B/16 block shapes of one training micro-batch (512 frames x 197 tokens)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fitclip_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 512 * 197
# optional: `save=<file>` keeps the four results, `compare=<file>` checks them bitwise against a saved run (another library build)
opts = dict(a.split("=", 1) for a in sys.argv[2:] if "=" in a)
kept = {}
g = torch.Generator(device='cuda').manual_seed(0)
for name, n1, n2 in (("c_fc", 3072, 768), ("c_proj", 768, 3072), ("qkv", 2304, 768), ("out_proj", 768, 768)):
    dy = torch.randn(M, n1, device='cuda', generator=g)
    x = torch.randn(M, n2, device='cuda', generator=g)
    out = ops.gemm_tn(dy, x)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            out = ops.gemm_tn(dy, x)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    ref = (dy[:4096].double().T @ x[:4096].double())
    chk = float((ops.gemm_tn(dy[:4096].contiguous(), x[:4096].contiguous()).double() - ref).abs().max() / ref.abs().max())
    kept[name] = out.cpu()
    print("%-9s M=%d N1=%d N2=%d  %.3f ms  %.1f TF/s (%.3f of 157.3)  rel err on 4096 rows %.1e" % (
        name, M, n1, n2, best * 1e3, 2.0 * M * n1 * n2 / best / 1e12, 2.0 * M * n1 * n2 / best / 157.3e12, chk))
if "save" in opts:
    torch.save(kept, opts["save"])
if "compare" in opts:
    want = torch.load(opts["compare"])
    for name, t in kept.items():
        print("%-9s bitwise equal to %s: %s" % (name, opts["compare"], bool(torch.equal(t, want[name]))))

"""fp32 attention at 512 frames x 197 tokens x 12 heads: fp32 output vs x3 (three-plane) output (split-fp32 mode)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fitclip_amd import ops
n_seq, S, heads = 512, 197, 12
g = torch.Generator(device='cuda').manual_seed(0)
qkv = torch.randn(n_seq * S, 3 * heads * 64, device='cuda', generator=g)
for six in (False, True):
    fn = lambda: ops.attention(qkv, n_seq, S, heads, three_plane=six)
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
    print("attention three_plane=%s: %.3f ms" % (six, best * 1e3))

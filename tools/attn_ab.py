"""fp32 attention at one bench pass (1024 images x 197 tokens x 12 heads): time and checksum.  A/B the first-generation
kernel with FITCLIP_ATTN_F32_ONE_BLOCK=1."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fitclip_amd import ops
n_seq, S, heads = 1024, 197, 12
g = torch.Generator(device='cuda').manual_seed(0)
qkv = torch.randn(n_seq * S, 3 * heads * 64, device='cuda', generator=g)
out = ops.attention(qkv, n_seq, S, heads)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = ops.attention(qkv, n_seq, S, heads)
torch.cuda.synchronize()
print("attention fp32 1024 x 197 x 12 heads: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3), float(out.double().abs().sum()))

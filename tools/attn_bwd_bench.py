"""fc_attention_backward at the ViT-B/16 shape of one training micro-batch (512 frames x 197 tokens x 12 heads): ms per launch,
TF/s of the five algorithmic products, optional bitwise comparison with a saved run of another library build.

    python tools/attn_bwd_bench.py [n_seq=512] [save=<file> | compare=<file>]
"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fitclip_amd import _lib, ops

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 512
opts = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
S, heads = 197, 12
D = heads * 64
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(n_seq * S, 3 * D, device="cuda", generator=g)
d_o = torch.randn(n_seq * S, D, device="cuda", generator=g)
out = ops.attention(qkv, n_seq, S, heads)
dqkv = torch.empty_like(qkv)
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def run():
    _lib.check(lib.fc_attention_backward(_lib.PREC_F32, qkv.data_ptr(), out.data_ptr(), d_o.data_ptr(), dqkv.data_ptr(), n_seq, S,
                                         heads, 0, st))


for _ in range(3):
    run()
torch.cuda.synchronize()
best = 1e9
for _ in range(3):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        run()
    b.record()
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b) / 10)
flops = n_seq * heads * 5 * 2.0 * S * S * 64
print("attention backward %d x %d x %d heads: %.3f ms  %.1f TF/s of the five products (%.3f of 157.3)" % (
    n_seq, S, heads, best, flops / best / 1e9, flops / best / 1e9 / 157.3))
if "save" in opts:
    torch.save(dqkv.cpu(), opts["save"])
if "compare" in opts:
    print("bitwise equal to %s: %s" % (opts["compare"], bool(torch.equal(dqkv.cpu(), torch.load(opts["compare"])))))

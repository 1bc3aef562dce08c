import torch, time, sys
sys.path.insert(0, '.')
from fitclip_amd import ops
dev='cuda'
n_seq, S, heads, D = 512, 197, 12, 768
M = n_seq*S
qkv = (torch.randn(M, 3*D, device=dev)).to(torch.bfloat16)
x = torch.randn(M, D, device=dev); delta = torch.randn(M, D, device=dev).to(torch.bfloat16)
g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(mode, iters=20):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(iters):
        if mode=='seq':
            ops.attention(qkv, n_seq, S, heads, False); ops.add_layernorm(x, delta, g, b)
        elif mode=='par':
            s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s1): ops.attention(qkv, n_seq, S, heads, False)
            with torch.cuda.stream(s2): ops.add_layernorm(x, delta, g, b)
            torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        elif mode=='att': ops.attention(qkv, n_seq, S, heads, False)
        elif mode=='ln': ops.add_layernorm(x, delta, g, b)
        elif mode=='ln2':
            s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s1): ops.add_layernorm(x, delta, g, b)
            with torch.cuda.stream(s2): ops.add_layernorm(x2, delta, g, b)
            torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    torch.cuda.synchronize(); return (time.perf_counter()-t)/iters*1e6
x2 = x.clone()
for m in ('att','ln','seq','par','ln2','seq','par'):
    run(m, 3); print(m, round(run(m),1), 'us')

import torch, time, sys
sys.path.insert(0, '.')
from fitclip_amd import ops
dev='cuda'
def bench(n_seq,S,heads,dtype,iters=20):
    qkv=torch.randn(n_seq*S,3*heads*64,device=dev).to(dtype)
    for _ in range(3): ops.attention(qkv,n_seq,S,heads,False)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(iters): ops.attention(qkv,n_seq,S,heads,False)
    torch.cuda.synchronize(); us=(time.perf_counter()-t)/iters*1e6
    fl=4*S*S*64*heads*n_seq; by=qkv.numel()*qkv.element_size()*4/3
    print(f"S={S:4d} heads={heads:2d} n={n_seq:4d} {str(dtype)[6:]:9s} {us:8.1f} us  {fl/us/1e6:7.1f} TF/s  {by/us/1e6:5.2f} TB/s")
for dt in (torch.bfloat16, torch.float32):
    bench(512,197,12,dt); bench(384,257,16,dt); bench(128,577,16,dt); bench(2048,50,12,dt); bench(2048,77,8,dt)

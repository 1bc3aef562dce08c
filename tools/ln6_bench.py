"""add+LayerNorm at 100864 rows x 768: fp32 output (16 B per element) vs x3 (three-plane) output (18 B per element)."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
rows, D = 100864, 768
x = torch.randn(rows, D, device="cuda", generator=g); delta = torch.randn(rows, D, device="cuda", generator=g)
gamma, beta = torch.randn(D, device="cuda", generator=g), torch.randn(D, device="cuda", generator=g)
for six in (False, True):
    fn = lambda: ops.add_layernorm(x, delta, gamma, beta, write_x=True, three_plane=six)
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 10)
    nbytes = rows * D * (18 if six else 16)
    print("add_layernorm three_plane=%s: %.3f ms = %.2f TB/s" % (six, best * 1e3, nbytes / best / 1e12))

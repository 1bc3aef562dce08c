import sys, torch
sys.path.insert(0, sys.argv[1])
from fitclip_amd import synth
from fitclip_amd.clip_model import build_clip
d = synth.VIT_B_16
m = build_clip(synth.make_state_dict(d, seed=42), precision="fp32", device="cuda:0")
for n in (128, 2048):
    x = torch.randn(n, 3, 224, 224, device="cuda:0")
    m.encode_image(x); m.profile(4096); m.profile_select(kind_mask=1, epilogue_mask=1 << 3); m.profile_reset()
    for _ in range(5): m.encode_image(x)
    torch.cuda.synchronize()
    recs = m.profile_records()
    ms = [r["ms"] for r in recs if r["epilogue"] == 3]
    r = recs[0]
    print(f"{n} frames: patch GEMM tile {r['tile']} M={r['M']} avg {sum(ms)/len(ms)*1e3:.1f} us ({2.0*r['M']*r['N']*r['K']/(sum(ms)/len(ms))/1e9/157.3:.3f} of peak)")
    m.profile(0)

"""End-to-end comparison of the three precisions on 64 clips x 8 frames (ViT-B/16, random towers): time per call and the
embedding distance of the split-fp32 (fp32x6) and bf16 modes to the fp32-MFMA path."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import synth
from fitclip_amd.clip_model import build_clip
from fitclip_amd.encoder import ClipVideoTextEncoder
d = synth.VIT_B_16
sd = synth.make_state_dict(d, seed=42)
g = torch.Generator(device="cuda").manual_seed(0)
video = torch.randn((64, 8, 3, 224, 224), generator=g, device="cuda").clamp_(-2.5, 2.5)
ids = torch.from_numpy(synth.make_text(64, d, seed=1)).cuda()
out = {}
for prec in ("fp32", "fp32x6", "bf16"):
    enc = ClipVideoTextEncoder(build_clip(sd, precision=prec, device="cuda:0"), num_frames=8)
    with torch.no_grad():
        v, t = enc(video=video, text={"input_ids": ids})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            v, t = enc(video=video, text={"input_ids": ids})
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
    out[prec] = (v.clone(), t.clone())
    print(prec, "%.1f ms per 64 clips x 8 frames -> %.0f pairs/s" % (dt * 1e3, 64 / dt))
    del enc
for prec in ("fp32x6", "bf16"):
    print(prec, "vs fp32: video max abs", float((out[prec][0] - out["fp32"][0]).abs().max()), "text", float((out[prec][1] - out["fp32"][1]).abs().max()))

"""Embeddings do not depend on the batch they are computed in: a sub-batch against the same rows of the full batch (bit-equal)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import synth
from fitclip_amd.clip_model import build_clip
from fitclip_amd.encoder import ClipVideoTextEncoder
d = synth.VIT_B_16
sd = synth.make_state_dict(d, seed=42)
enc = ClipVideoTextEncoder(build_clip(sd, precision="fp32", device="cuda:0"), num_frames=8)
g = torch.Generator(device="cuda").manual_seed(0)
video = torch.randn((64, 8, 3, 224, 224), generator=g, device="cuda").clamp_(-2.5, 2.5)
ids = torch.from_numpy(synth.make_text(64, d, seed=1)).cuda()
with torch.no_grad():
    full_v, full_t = enc(video=video, text={"input_ids": ids})
    half_v, half_t = enc(video=video[32:], text={"input_ids": ids[32:]})
    print("video max abs diff", float((full_v[32:] - half_v).abs().max()), "text", float((full_t[32:] - half_t).abs().max()))
    for n in (1, 3, 16, 33):
        v = enc.encode_video(video[:n])
        print(n, "clips: diff vs full", float((full_v[:n] - v).abs().max()))

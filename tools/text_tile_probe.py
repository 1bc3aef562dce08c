"""Does the fp32 bench step gain when the TEXT tower's block GEMMs use small-LDS tile forms that can share a CU with the visual tower's
attention / LayerNorm workgroups?  Run with the lab library: FITCLIP_HIP_LIB=tools/bin/libfitclip_hip_lab.so FITCLIP_LAB_TEXT_TILE={0,1,8}.
    python tools/text_tile_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import synth
from fitclip_amd.clip_model import build_clip
from fitclip_amd.encoder import ClipVideoTextEncoder
d = synth.VIT_B_16
sd = synth.make_state_dict(d, seed=42)
g = torch.Generator(device="cuda").manual_seed(0)
N = 256
video = torch.randn((N, 8, 3, 224, 224), generator=g, device="cuda").clamp_(-2.5, 2.5)
ids = torch.from_numpy(synth.make_text(N, d, seed=1)).cuda()


def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    return sorted(ts)[1]


enc = ClipVideoTextEncoder(build_clip(sd, precision="fp32", device="cuda:0"), num_frames=8)
with torch.no_grad():
    both = t(lambda: enc(video=video, text={"input_ids": ids}))
    v = t(lambda: enc.encode_video(video))
    tx = t(lambda: enc.encode_text({"input_ids": ids}), reps=20)
    enc.overlap_text = False
    serial = t(lambda: enc(video=video, text={"input_ids": ids}))
print(f"FITCLIP_LAB_TEXT_TILE={os.environ.get('FITCLIP_LAB_TEXT_TILE', '0')}: forward (two streams) {both:.2f} ms, one stream {serial:.2f} ms; "
      f"encode_video alone {v:.2f} ms, encode_text alone {tx:.2f} ms -> the text tower costs {both - v:.2f} ms beside the visual tower", flush=True)

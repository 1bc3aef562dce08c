"""What the text tower costs the bench step (256 clips x 8 frames + 256 captions, ViT-B/16): the forward with the towers on two
streams and on one, and each tower alone - fp32x3 and fp32.  Also encode_text alone at 32 / 54 / 128 captions (fp32x3 switches the
text blocks to the three-product GEMMs from 2048 token rows per call on).
    python tools/text_exposure.py"""
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
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for prec in ("fp32x3", "fp32"):
    enc = ClipVideoTextEncoder(build_clip(sd, precision=prec, device="cuda:0"), num_frames=8)
    with torch.no_grad():
        both = t(lambda: enc(video=video, text={"input_ids": ids}))
        v = t(lambda: enc.encode_video(video))
        tx = t(lambda: enc.encode_text({"input_ids": ids}))
        enc.overlap_text = False
        serial = t(lambda: enc(video=video, text={"input_ids": ids}))
    print(f"{prec}: forward (two streams) {both:.1f} ms, one stream {serial:.1f} ms; encode_video alone {v:.1f} ms, encode_text alone {tx:.1f} ms -> the text tower costs {both - v:.1f} ms beside the visual tower ({serial - v:.1f} serial)", flush=True)
models = {prec: build_clip(sd, precision=prec, device="cuda:0") for prec in ("fp32x3", "fp32")}
for rnd in range(2):
    for n in (8, 16, 26, 27, 28, 30, 32, 33, 36, 40, 48, 64, 128, 256):
        with torch.no_grad():
            row = [t(lambda: models[prec].encode_text(ids[:n]), reps=10) for prec in ("fp32x3", "fp32")]
        print(f"round {rnd}: encode_text, {n:3d} captions: fp32x3 {row[0]:.2f} ms, fp32 {row[1]:.2f} ms", flush=True)

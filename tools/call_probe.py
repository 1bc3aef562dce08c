#!/usr/bin/env python
"""The reference-shaped call (32 clips x 4 frames + 32 captions) taken apart: both towers (two streams), both towers on one
stream, the visual tower alone, the text tower alone - ms per call (tools/README.md); and the whole call captured in a hipGraph and
replayed.     python tools/call_probe.py [clips per call] [precision]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitclip_amd import synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402

bs, frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4
dev = torch.device("cuda", 0)
d = synth.VIT_B_16
precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
enc = ClipVideoTextEncoder(build_clip(synth.make_state_dict(d, seed=42), precision=precision, device=dev), num_frames=frames)
g = torch.Generator(device=dev).manual_seed(0)
video = torch.randn((bs, frames, 3, 224, 224), generator=g, device=dev).clamp_(-2.5, 2.5)
text = {"input_ids": torch.from_numpy(synth.make_text(bs, d, seed=1)).to(dev)}


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


with torch.inference_mode():
    both = timed(lambda: enc(video=video, text=text))
    enc.overlap_text = False
    seq = timed(lambda: enc(video=video, text=text))
    vis = timed(lambda: enc.encode_video(video))
    txt = timed(lambda: enc.encode_text(text))
    # the whole call (both towers on two streams) as ONE graph launch
    enc.overlap_text = True
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        enc(video=video, text=text)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out = enc(video=video, text=text)
        replay = timed(graph.replay)
        eager_side = timed(lambda: enc(video=video, text=text))
    torch.cuda.current_stream().wait_stream(side)
flops = bs * (frames * 35.127e9 + 5.960e9)
print(f"{precision}: hipGraph replay {replay:.3f} ms against eager {eager_side:.3f} ms per call ({bs / replay * 1e3:.0f} vs {bs / eager_side * 1e3:.0f} pairs/s)")
print(f"batch {bs} x {frames} frames: two streams {both:.3f} ms ({flops / both / 1e9 / 157.3:.4f} of peak), one stream {seq:.3f} ms, "
      f"visual tower alone {vis:.3f} ms ({bs * frames * 35.127e9 / vis / 1e9 / 157.3:.4f}), text tower alone {txt:.3f} ms "
      f"({bs * 5.960e9 / txt / 1e9 / 157.3:.4f})")

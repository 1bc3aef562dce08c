#!/usr/bin/env python
"""Does HIP stream priority change what the text tower costs next to the visual tower?  The reference-shaped call (32 clips x 4
frames + 32 captions) with the two towers on (a) the encoder's default pair of streams, (b) visual tower on a HIGH-priority
stream, text on the caller's stream, (c) text tower on a LOW-priority stream (same as (a) if the range has no level below the
default).  ms per call."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fitclip_amd import ops, synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402

bs, frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 4
dev = torch.device("cuda", 0)
d = synth.VIT_B_16
enc = ClipVideoTextEncoder(build_clip(synth.make_state_dict(d, seed=42), precision="fp32", device=dev), num_frames=frames)
g = torch.Generator(device=dev).manual_seed(0)
video = torch.randn((bs, frames, 3, 224, 224), generator=g, device=dev).clamp_(-2.5, 2.5)
text = {"input_ids": torch.from_numpy(synth.make_text(bs, d, seed=1)).to(dev)}
print("priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def two_streams(vis_stream, txt_stream):
    main = torch.cuda.current_stream()

    def call():
        enc.model._ensure_ready()
        outs = {}
        for s, fn, key in ((txt_stream, lambda: enc.encode_text(text), "t"), (vis_stream, lambda: enc.encode_video(video), "v")):
            if s is None:
                outs[key] = fn()
            else:
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    outs[key] = fn()
        for s in (txt_stream, vis_stream):
            if s is not None:
                main.wait_stream(s)
        return outs["v"], outs["t"]
    return call


with torch.inference_mode():
    ref = enc(video=video, text=text)
    base = timed(lambda: enc(video=video, text=text))
    print(f"(a) encoder default (text on a side stream of default priority): {base:.3f} ms")
    for name, vp, tp in (("(b) visual on a priority -1 stream, text on the caller's", -1, None),
                         ("(c) visual on the caller's stream, text on a priority -1 stream", None, -1),
                         ("(d) visual on priority -1, text on priority 0 side stream", -1, 0)):
        vs = torch.cuda.Stream(device=dev, priority=vp) if vp is not None else None
        ts = torch.cuda.Stream(device=dev, priority=tp) if tp is not None else None
        call = two_streams(vs, ts)
        out = call()
        torch.cuda.synchronize()
        ok = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
        print(f"{name}: {timed(call):.3f} ms  (bitwise equal: {ok})")
    enc.overlap_text = False
    print(f"(e) one stream: {timed(lambda: enc(video=video, text=text)):.3f} ms; visual alone {timed(lambda: enc.encode_video(video)):.3f} ms")

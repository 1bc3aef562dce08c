"""The reference-shaped call (32 clips x 4 frames + 32 captions, ViT-B/16) eager against its hipGraph replay: ms per call, per precision.
    python tools/graph_probe.py [clips] [frames]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import synth
from fitclip_amd.clip_model import build_clip
from fitclip_amd.encoder import ClipVideoTextEncoder
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = synth.VIT_B_16
sd = synth.make_state_dict(d, seed=42)
video = torch.from_numpy(synth.make_video(clips, frames, d, seed=3)).cuda()
text = {"input_ids": torch.from_numpy(synth.make_text(clips, d, seed=3)).cuda()}


def timed(fn, reps=30):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps * 1e3)
    return sorted(ts)[2]


for prec in ("fp32x3", "fp32", "bf16"):
    enc = ClipVideoTextEncoder(build_clip(sd, precision=prec, device="cuda:0"), num_frames=frames)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.inference_mode():
        eager = timed(lambda: enc(video=video, text=text))
        ev, et = enc(video=video, text=text)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            gv, gt = enc(video=video, text=text)
        replay = timed(graph.replay)
        same = bool(torch.equal(ev, gv) and torch.equal(et, gt))
    torch.cuda.current_stream().wait_stream(side)
    print(f"{prec}: {clips} clips x {frames} frames + {clips} captions: eager {eager:.3f} ms, graph replay {replay:.3f} ms ({eager / replay:.3f}x), bitwise equal {same}", flush=True)

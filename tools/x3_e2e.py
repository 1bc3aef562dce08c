"""End-to-end timing of the split-fp32 modes on the bench batch (256 clips x 8 frames + 256 texts, ViT-B/16, random towers): ms per
step and pairs/s of fp32x6 (six bf16 products) and fp32x3 (three fp16 products) at several pass sizes, the per-kernel time split
of the fp32x3 step from the library's event records, and the embedding distance of either mode to the fp32-MFMA path.
    python tools/x3_e2e.py [chunk sizes for fp32x3, comma separated]"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fitclip_amd import synth, _lib
from fitclip_amd.clip_model import build_clip
from fitclip_amd.encoder import ClipVideoTextEncoder
import ctypes as C
d = synth.VIT_B_16
sd = synth.make_state_dict(d, seed=42)
g = torch.Generator(device="cuda").manual_seed(0)
N = 256
video = torch.randn((N, 8, 3, 224, 224), generator=g, device="cuda").clamp_(-2.5, 2.5)
ids = torch.from_numpy(synth.make_text(N, d, seed=1)).cuda()
chunks = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "768,1024,2048").split(",")]
out = {}
def run(prec, chunk, profile=False):
    enc = ClipVideoTextEncoder(build_clip(sd, precision=prec, device="cuda:0", chunk_frames=chunk), num_frames=8)
    with torch.no_grad():
        v, t = enc(video=video, text={"input_ids": ids})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            v, t = enc(video=video, text={"input_ids": ids})
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        enc.model.check_range()
        print(f"{prec:7s} chunk {chunk:5d}: {dt * 1e3:8.1f} ms per step -> {N / dt:7.1f} pairs/s", flush=True)
        if profile:
            lib = _lib.load()
            h = enc.model._rt.handle
            _lib.check(lib.fc_profile_enable(h, 4096))
            enc.model.encode_image(video.reshape(-1, 3, 224, 224))
            torch.cuda.synchronize()
            recs = (_lib.fc_prof_record * 4096)()
            n = lib.fc_profile_read(h, recs, 4096)
            agg = {}
            for r in recs[:n]:
                key = (r.kind, r.precision, r.epilogue, r.M, r.N, r.K)
                a = agg.setdefault(key, [0.0, 0])
                a[0] += r.ms; a[1] += 1
            tot = sum(a[0] for a in agg.values())
            for key, (ms, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                kind, p, epi, M, Nn, K = key
                extra = ""
                if kind == 0 and p == 2:
                    extra = f"  {2.0 * M * Nn * K / (ms / cnt * 1e-3) / 1e12:7.1f} TF/s on the fp16 pipe = {2.0 * M * Nn * K / (ms / cnt * 1e-3) / 2.5e15:.3f} of peak"
                print(f"   kind {kind} prec {p} epi {epi:2d} M={M} N={Nn} K={K}: {cnt:3d} x {ms / cnt:7.3f} ms = {ms:7.2f} ms ({ms / tot:.3f}){extra}")
            print(f"   visual tower, summed kernel time {tot:.1f} ms")
    res = (v.clone(), t.clone())
    del enc
    return res
out["fp32"] = run("fp32", 0)
out["fp32x6"] = run("fp32x6", 0)
for i, c in enumerate(chunks):
    out["fp32x3"] = run("fp32x3", c, profile=(i == len(chunks) - 1))
for prec in ("fp32x6", "fp32x3"):
    print(prec, "vs fp32: video max abs", float((out[prec][0] - out["fp32"][0]).abs().max()), "text", float((out[prec][1] - out["fp32"][1]).abs().max()))

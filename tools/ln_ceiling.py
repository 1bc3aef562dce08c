"""What fusing the block LayerNorms into the GEMMs could buy the fp32 headline AT MOST: the bench step (256 clips x 8 frames + 256
texts, fp32) with the lab library (`python -m fitclip_amd.build --lab`), as shipped, with FITCLIP_LAB_SKIP_LN=1 (the LayerNorm
launches of blocks 1.. skipped: the ceiling) and with FITCLIP_LAB_LN_FUSE=1 (a statistics-only pass in place of each of them and the
correction of the fusion in the QKV / c_fc epilogues, on stand-in vectors: what the simple form of the fusion would really give);
and FITCLIP_LAB_GELU=2 / 3
(the c_fc epilogue without its QuickGELU / with the plain form: what the function costs where it runs); results meaningless, timing
valid.  One child process per run (the switches are read once).
    python tools/ln_ceiling.py"""
import json, os, subprocess, sys
repo = os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
child = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from fitclip_amd import synth
from fitclip_amd.clip_model import build_clip
from fitclip_amd.encoder import ClipVideoTextEncoder
d = synth.VIT_B_16
enc = ClipVideoTextEncoder(build_clip(synth.make_state_dict(d, seed=42), precision="fp32", device="cuda:0"), num_frames=8)
g = torch.Generator(device="cuda").manual_seed(0)
video = torch.randn((256, 8, 3, 224, 224), generator=g, device="cuda").clamp_(-2.5, 2.5)
ids = torch.from_numpy(synth.make_text(256, d, seed=1)).cuda()
with torch.no_grad():
    for _ in range(2): enc(video=video, text={"input_ids": ids})
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): enc(video=video, text={"input_ids": ids})
    torch.cuda.synchronize(); print((time.perf_counter() - t0) / 5 * 1e3)
''' % repo
res = {}
for mode in ("shipped", "skip", "fuse", "nogelu", "plaingelu") * 2:
    env = {**os.environ, "FITCLIP_HIP_LIB": os.path.join(repo, "tools/bin/libfitclip_hip_lab.so"),
           "FITCLIP_LAB_SKIP_LN": "1" if mode == "skip" else "0", "FITCLIP_LAB_LN_FUSE": "1" if mode == "fuse" else "0",
           "FITCLIP_LAB_GELU": {"nogelu": "2", "plaingelu": "3"}.get(mode, "0")}
    out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res.setdefault(mode, []).append(float(out.stdout.strip().splitlines()[-1]))
a, b, c = min(res["shipped"]), min(res["skip"]), min(res["fuse"])
ng, pg = min(res["nogelu"]), min(res["plaingelu"])
print(json.dumps({"fp32_step_ms": res["shipped"], "fp32_step_ms_without_block_layernorms": res["skip"],
                  "fp32_step_ms_with_statistics_passes_and_corrected_epilogues": res["fuse"],
                  "fp32_step_ms_without_quickgelu_in_the_c_fc_epilogue": res["nogelu"], "fp32_step_ms_with_the_plain_quickgelu": res["plaingelu"],
                  "quickgelu_in_the_c_fc_epilogue": f"{(a - ng) / a * 100:.2f} % of the step ({a:.1f} -> {ng:.1f} ms without it); the plain form x / (1 + 2^(-1.702 log2 e x)), no "
                                                    f"compensated exponent: {(a - pg) / a * 100:.2f} % ({pg:.1f} ms)",
                  "ceiling_of_layernorm_fusion": f"{(a - b) / a * 100:.2f} % of the step ({a:.1f} -> {b:.1f} ms; 256 / {a / 1e3:.4f} = {256e3 / a:.1f} -> {256e3 / b:.1f} pairs/s)",
                  "statistics_pass_variant": f"{(a - c) / a * 100:.2f} % of the step ({a:.1f} -> {c:.1f} ms = {256e3 / c:.1f} pairs/s): a statistics-only pass (mean, 1 / std per row) in place of 22 of "
                                             "the 24 block LayerNorms + the correction rstd (acc - mean g) + c in the QKV / c_fc epilogues"}))

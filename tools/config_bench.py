#!/usr/bin/env python
"""Timed `command=evaluate` runs at the BASELINE.json configurations other than the bench's (configs[1]):

  wise      configs[2]: `encoder=wise` (0.5 CLIP + 0.5 student, ViT-B/16), WebVid-val shape = 4096 clips x 4 frames,
            one caption per clip, through the evaluate loop (TextVideoRetrievalModule) - includes the WiSE blend
  shard     configs[3] per GPU: 1024 clips x 16 frames + 1024 captions (one rank's share of 8192 x 16 over 8 GPUs)

Inputs are generated on the device before timing (the reference's data pipeline is out of scope); one JSON line each.

    python tools/config_bench.py [--precision fp32] [--eval-batch-size 32 256]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from fitclip_amd import synth  # noqa: E402
from fitclip_amd.__main__ import instantiate, load_encoder_config  # noqa: E402
from fitclip_amd.retrieval import TextVideoRetrievalModule  # noqa: E402

GF_PER_FRAME, GF_PER_TEXT = 35.127e9, 5.960e9


def run(name, encoder_cfg, n, f, bs, precision, dev):
    cfg = {"precision": precision, "num_frames": f, "weight_for_2": 0.5 if encoder_cfg == "wise" else None}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enc = instantiate(load_encoder_config(encoder_cfg, cfg, dev)).to(dev)
    enc.num_frames = f
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    d = enc.model.dims
    g = torch.Generator(device=dev).manual_seed(0)
    video = torch.randn((n, f, 3, d.image_resolution, d.image_resolution), generator=g, device=dev).clamp_(-2.5, 2.5)
    ids = torch.from_numpy(synth.make_text(n, d, seed=1)).to(dev)
    module = TextVideoRetrievalModule(enc, init_temperature=0.015, n_total=n)

    def epoch():
        for s in range(0, n, bs):
            module.validation_step_end(module.validation_step(
                {"video": video[s:s + bs], "text": {"input_ids": ids[s:s + bs]}, "video_id": list(range(s, min(n, s + bs)))}))
        return module.validation_epoch_end()

    with torch.inference_mode():
        epoch()  # warm-up (weight packing, workspaces)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        metrics = epoch()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    tf = n * (f * GF_PER_FRAME + GF_PER_TEXT) / el / 1e12
    peak = 157.3 if precision == "fp32" else 2500.0
    print(json.dumps({"config": name, "encoder": encoder_cfg, "precision": precision, "clips": n, "frames": f,
                      "eval_batch_size": bs, "epoch_s": round(el, 3), "pairs_per_s": round(n / el, 1),
                      "tflops": round(tf, 1), "frac_of_peak": round(tf / peak, 4),
                      "encoder_build_s (incl. WiSE blend)": round(build_s, 2), "metrics": metrics}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--eval-batch-size", type=int, nargs="+", default=[32, 256])
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    for bs in a.eval_batch_size:
        run("configs[2] wise, WebVid-val shape", "wise", 4096, 4, bs, a.precision, dev)
    run("configs[3] one rank's share (1024 x 16 of 8192 x 16)", "clip_vit_b_16", 1024, 16, 128, a.precision, dev)


if __name__ == "__main__":
    main()

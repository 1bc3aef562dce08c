#!/usr/bin/env python
"""Throughput of the visual tower for the reference's CLIP ViT geometries (random weights): frames/s and TFLOP/s.

    python tools/model_probe.py [--precision bf16|fp32|fp32x6|fp32x3] [B/16 B/32 L/14 L/14@336]

With a split precision the embeddings are also compared with the fp32-MFMA path on 8 frames (the geometries whose sequence length
has no fused split attention - 50, 257, 577 tokens - take the fp32 attention + a split pass) and the range flag is checked."""
import sys
import time

import torch

sys.path.insert(0, ".")
from fitclip_amd import synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402

GEOM = {
    "B/16": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=16,
                 transformer_width=512, transformer_heads=8),
    "B/32": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
                 transformer_width=512, transformer_heads=8),
    "L/14": dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
                 transformer_width=768, transformer_heads=12),
    "L/14@336": dict(embed_dim=768, image_resolution=336, vision_layers=24, vision_width=1024, vision_patch_size=14,
                     transformer_width=768, transformer_heads=12),
}


def flops_per_frame(d):
    g = d.image_resolution // d.vision_patch_size
    S, D = g * g + 1, d.vision_width
    per_layer = 2 * 12 * S * D * D + 4 * S * S * D
    return d.vision_layers * per_layer + 2 * g * g * 3 * d.vision_patch_size ** 2 * D + 2 * D * d.embed_dim


def main():
    args = sys.argv[1:]
    precision = "bf16"
    if args and args[0] == "--precision":
        precision, args = args[1], args[2:]
    names = args or list(GEOM)
    for name in names:
        d = synth.ClipDims(context_length=77, vocab_size=49408, transformer_layers=2, **GEOM[name])  # text tower unused
        sd = synth.make_state_dict(d, seed=1)
        model = build_clip(sd, precision=precision, device="cuda")
        frames = 1024 if "336" not in name else 256
        x = torch.randn(frames, 3, d.image_resolution, d.image_resolution, device="cuda")
        for _ in range(4):
            model.encode_image(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        reps = 6
        for _ in range(reps):
            model.encode_image(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        extra = ""
        if precision.startswith("fp32x"):
            model.check_range()
            plain = build_clip(sd, precision="fp32", device="cuda")
            a, b = model.encode_image(x[:8].contiguous()), plain.encode_image(x[:8].contiguous())
            extra = f"  max |emb - fp32 path| {float((a - b).abs().max()):.2e} of max |emb| {float(b.abs().max()):.2f}"
            del plain
        print(f"ViT-{name:9s} {precision:7s} {frames / dt:9.0f} frames/s  {flops_per_frame(d) * frames / dt / 1e12:7.1f} TFLOP/s "
              f"({flops_per_frame(d) / 1e9:.1f} GF/frame){extra}", flush=True)
        del model, x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

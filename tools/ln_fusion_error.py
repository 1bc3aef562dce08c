#!/usr/bin/env python
"""What the algebraic LayerNorm fusion (verdict round 4, item 3a) costs in accuracy: LN(x) W^T computed as
rstd (x (gamma . W)^T - mu g) + c with g = sum_j gamma_j W_nj, in float32 with float32 accumulation, against the two-pass form
(LayerNorm, then the product) in float32 and against float64, for rows of increasing |mean| / std.  CPU, numpy; no GPU.

    python tools/ln_fusion_error.py        ->  a table; also |mean| / std of the residual stream of the synthetic ViT-B/16 and of
                                                the heavy-tailed variant of tests/test_gpu_split2.py (first frame, every block)"""
import sys

import numpy as np

sys.path.insert(0, ".")


def study():
    rng = np.random.default_rng(0)
    D, N, R = 768, 256, 512
    w = (rng.standard_normal((N, D)) / np.sqrt(D)).astype(np.float32)
    gamma = (1.0 + 0.1 * rng.standard_normal(D)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(D)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    wg = (w * gamma).astype(np.float32)                      # gamma . W, packed once
    g = wg.astype(np.float64).sum(1).astype(np.float32)      # (packed in float64, stored in float32)
    c = (w.astype(np.float64) @ beta.astype(np.float64) + bias).astype(np.float32)
    print(f"{'|mean|/std':>10s} {'two-pass fp32':>14s} {'fused fp32':>12s} {'ratio':>7s}   (max |y - y64| / max |y64| over {R} rows x {N} columns)")
    for ratio in (0.0, 0.3, 1.0, 3.0, 10.0, 30.0, 100.0, 1000.0):
        x = rng.standard_normal((R, D)).astype(np.float32)
        x = (x + np.float32(ratio)).astype(np.float32)
        x64 = x.astype(np.float64)
        mu64 = x64.mean(1, keepdims=True)
        var64 = x64.var(1, keepdims=True)
        y64 = ((x64 - mu64) / np.sqrt(var64 + 1e-5) * gamma + beta) @ w.astype(np.float64).T + bias
        # two-pass float32 (what the LayerNorm kernel + GEMM do)
        mu = x.mean(1, keepdims=True, dtype=np.float32)
        var = ((x - mu) ** 2).mean(1, keepdims=True, dtype=np.float32)
        h = ((x - mu) / np.sqrt(var + np.float32(1e-5)) * gamma + beta).astype(np.float32)
        y2 = (h @ w.T + bias).astype(np.float32)
        # fused float32: statistics from sum x, sum x^2 (one pass, as an epilogue would produce them), product on raw x
        s1 = x.sum(1, keepdims=True, dtype=np.float32)
        s2 = (x * x).sum(1, keepdims=True, dtype=np.float32)
        muf = s1 / np.float32(D)
        varf = np.maximum(s2 / np.float32(D) - muf * muf, np.float32(0))
        rstd = (1.0 / np.sqrt(varf + np.float32(1e-5))).astype(np.float32)
        acc = (x @ wg.T).astype(np.float32)
        yf = (rstd * (acc - muf * g) + c).astype(np.float32)
        scale = np.abs(y64).max()
        e2 = np.abs(y2 - y64).max() / scale
        ef = np.abs(yf - y64).max() / scale
        print(f"{ratio:10.1f} {e2:14.2e} {ef:12.2e} {ef / e2:7.1f}")


def streams():
    import torch
    import torch.nn.functional as F
    from fitclip_amd import synth
    from oracle import clip_oracle as O
    d = synth.VIT_B_16
    base = synth.make_state_dict(d, seed=42)

    def heavy(base):  # the planted outliers of tests/test_gpu_split2.py::test_heavy_tailed_weights_keep_the_fp32_accuracy
        rng = np.random.default_rng(11)
        sd = {k: np.array(v, copy=True) for k, v in base.items()}
        hot = rng.choice(d.vision_width, 6, replace=False)
        sd["visual.ln_pre.weight"][hot] *= 80.0
        for layer in range(d.vision_layers):
            pre = f"visual.transformer.resblocks.{layer}."
            sd[pre + "attn.out_proj.weight"][hot[:3]] *= 25.0
            sd[pre + "mlp.c_proj.weight"][hot[3:]] *= 25.0
        return sd

    video = torch.from_numpy(synth.make_video(1, 1, d, seed=78))
    for name, sdn in (("synthetic init", base), ("heavy-tailed", heavy(base))):
        sd = O.to_torch(sdn)
        w = sd["visual.conv1.weight"]
        worst = 0.0
        with torch.inference_mode():
            x = F.conv2d(video[0], w, None, stride=16).reshape(1, 768, -1).permute(0, 2, 1)
            x = torch.cat([sd["visual.class_embedding"].expand(1, 1, 768), x], 1) + sd["visual.positional_embedding"]
            x = O.layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
            for i in range(12):
                r = (x.mean(-1).abs() / x.std(-1)).max().item()
                worst = max(worst, r)
                x = O.residual_block(x, sd, f"visual.transformer.resblocks.{i}", 12, None)
        print(f"residual stream, {name}: largest |mean| / std of a row over the 12 blocks = {worst:.3f}")


if __name__ == "__main__":
    study()
    streams()

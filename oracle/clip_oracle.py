"""CPU oracle for the FitCLIP encode-and-score path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this module; the product
package `fitclip_amd` never does (it fails loudly when the HIP library is missing instead of falling back here).

It is a plain PyTorch fp32 restatement, written from the architecture description, of what the reference computes on
this path.  Each function cites the reference lines it follows (paths relative to /root/reference):

  * LayerNorm-in-fp32 / QuickGELU / pre-LN residual block / Transformer ... aligner/encoder/slip.py:350-396
  * causal mask, text encode, EOT argmax ................................ aligner/encoder/slip.py:454-480
  * visual stem (conv1 no bias -> CLS|patches + pos -> ln_pre -> blocks -> ln_post(CLS) -> @proj): the third-party
    `clip.model.VisionTransformer` (openai/CLIP@b46f5ac, environment.yml:7; source NOT under /root/reference; its
    published architecture is restated here; hyper-parameters config/encoder/clip_from_scratch_vit_b_16.yaml:5-16)
  * frame flatten -> encode_image -> L2 normalise -> mean over frames; text L2 normalise:
    aligner/encoder/clip_video_text_encoder.py:80-94
  * WiSE ................................................................ aligner/wise.py:10-23
  * NCE / teacher-student KD losses ..................................... aligner/loss.py:13-39
  * score matrix, Recall@k, rank, median rank ........................... aligner/text_video_retrieval.py:67-83,
                                                                          aligner/metrics.py:16-36
  * all_gather flattening of [world, B, ...] ............................ util/tensor_utils.py:48-66
  * KD training loss (per-dataset NCE / KD * tau^2, loss shares) ........ aligner/teacher_student.py:142-176
    (its gradients are torch autograd through this restatement; pinned against autograd through the reference's
    own slip / loss classes by tests/golden/make_goldens.py::golden_training)

Pinning (see tests/golden/make_goldens.py and DESIGN.md "Oracle"): checked in the build container against the
reference's own `aligner.wise` (direct import), `aligner.loss` and `aligner.encoder.slip` transformer / text tower
(imported with inert decorator shims) and against HuggingFace `CLIPModel` built from a local config.  The reference has
no tests or golden vectors of its own for this path.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Sequence, Tuple

import torch
import torch.nn.functional as F

TensorDict = Mapping[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------- transformer blocks
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """slip.py:350-356 (LayerNorm computed in float32, eps = nn.LayerNorm default 1e-5)."""
    return F.layer_norm(x.to(w.dtype), (x.shape[-1],), w, b, 1e-5)  # float32 weights: the reference's fp32 LayerNorm; float64 weights: the truth run of the tests


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """slip.py:359-361."""
    return x * torch.sigmoid(1.702 * x)


def multi_head_attention(x: torch.Tensor, in_w, in_b, out_w, out_b, heads: int, mask) -> torch.Tensor:
    """What `nn.MultiheadAttention(x, x, x, need_weights=False, attn_mask=mask)` computes (slip.py:367,378-380):
    packed in_proj, q scaled by 1/sqrt(head_dim), additive mask, softmax, out_proj.  x: [N, L, D] (batch first)."""
    n, l, d = x.shape
    hd = d // heads
    qkv = F.linear(x, in_w, in_b)
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(n, l, heads, hd).transpose(1, 2)
    k = k.view(n, l, heads, hd).transpose(1, 2)
    v = v.view(n, l, heads, hd).transpose(1, 2)
    s = (q * (1.0 / math.sqrt(hd))) @ k.transpose(-1, -2)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(n, l, d)
    return F.linear(o, out_w, out_b)


def residual_block(x: torch.Tensor, sd: TensorDict, prefix: str, heads: int, mask) -> torch.Tensor:
    """slip.py:364-385: x + attn(ln_1(x)); x + mlp(ln_2(x))."""
    g = lambda n: sd[f"{prefix}.{n}"]  # noqa: E731
    h = layer_norm(x, g("ln_1.weight"), g("ln_1.bias"))
    x = x + multi_head_attention(h, g("attn.in_proj_weight"), g("attn.in_proj_bias"),
                                 g("attn.out_proj.weight"), g("attn.out_proj.bias"), heads, mask)
    h = layer_norm(x, g("ln_2.weight"), g("ln_2.bias"))
    h = quick_gelu(F.linear(h, g("mlp.c_fc.weight"), g("mlp.c_fc.bias")))
    return x + F.linear(h, g("mlp.c_proj.weight"), g("mlp.c_proj.bias"))


def transformer(x: torch.Tensor, sd: TensorDict, prefix: str, layers: int, heads: int, mask=None) -> torch.Tensor:
    """slip.py:388-396."""
    for i in range(layers):
        x = residual_block(x, sd, f"{prefix}.resblocks.{i}", heads, mask)
    return x


def causal_mask(n_ctx: int) -> torch.Tensor:
    """slip.py:454-460: -inf strictly above the diagonal."""
    return torch.full((n_ctx, n_ctx), float("-inf")).triu_(1)


def _count_layers(sd: TensorDict, prefix: str) -> int:
    return len({k.split(".resblocks.")[1].split(".")[0] for k in sd if k.startswith(prefix + ".resblocks.")})


# ----------------------------------------------------------------------------------------------------------- towers
def encode_image(sd: TensorDict, images: torch.Tensor) -> torch.Tensor:
    """clip.model.CLIP.encode_image (called at clip_video_text_encoder.py:84).  images f32 [N,3,H,W] -> [N,E]."""
    w = sd["visual.conv1.weight"]
    width, patch = w.shape[0], w.shape[-1]
    heads = width // 64
    x = F.conv2d(images.to(w.dtype), w, None, stride=patch)  # [N, width, g, g]
    x = x.reshape(x.shape[0], width, -1).permute(0, 2, 1)  # [N, g*g, width]
    cls = sd["visual.class_embedding"].expand(x.shape[0], 1, width)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
    x = transformer(x, sd, "visual.transformer", _count_layers(sd, "visual.transformer"), heads)
    x = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
    return x @ sd["visual.proj"]


def encode_text_tokens(sd: TensorDict, ids: torch.Tensor) -> torch.Tensor:
    """slip.py:468-480 (== clip.model.CLIP.encode_text).  ids int [N, n_ctx] -> [N,E]."""
    width = sd["token_embedding.weight"].shape[1]
    layers = _count_layers(sd, "transformer")
    heads = width // 64
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"]
    x = transformer(x, sd, "transformer", layers, heads, causal_mask(ids.shape[1]))
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    x = x[torch.arange(x.shape[0]), ids.argmax(dim=-1)]  # EOT has the highest id in each sequence
    return x @ sd["text_projection"]


def encode_video(sd: TensorDict, video: torch.Tensor, batch: int = 32) -> torch.Tensor:
    """clip_video_text_encoder.py:80-89.  video f32 [B,F,3,H,W] -> [B,E]: per-frame unit vectors, averaged over
    frames and NOT re-normalised."""
    b = video.shape[0]
    images = video.reshape(-1, *video.shape[2:])
    enc = torch.cat([encode_image(sd, images[i:i + batch]) for i in range(0, images.shape[0], batch)])
    enc = enc / enc.norm(dim=-1, keepdim=True)
    return enc.view(b, -1, enc.shape[-1]).mean(dim=1)


def encode_text(sd: TensorDict, text: Mapping[str, torch.Tensor], batch: int = 64) -> torch.Tensor:
    """clip_video_text_encoder.py:92-94."""
    ids = text["input_ids"]
    enc = torch.cat([encode_text_tokens(sd, ids[i:i + batch]) for i in range(0, ids.shape[0], batch)])
    return enc / enc.norm(dim=-1, keepdim=True)


def forward(sd: TensorDict, video: torch.Tensor, text: Mapping[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """video_text_encoder.py:21-22."""
    return encode_video(sd, video), encode_text(sd, text)


# ------------------------------------------------------------------------------------------------------------- WiSE
def wise_state_dict(sd1: TensorDict, sd2: TensorDict, weight_for_2: float = 0.5) -> Dict[str, torch.Tensor]:
    """wise.py:10-16."""
    assert set(sd1) == set(sd2)
    return {k: (1 - weight_for_2) * sd1[k] + weight_for_2 * sd2[k] for k in sd1}


# ----------------------------------------------------------------------------------------------------------- losses
def nce_loss(scores: torch.Tensor) -> torch.Tensor:
    """loss.py:13-26 with reduction="mean"."""
    return (-F.log_softmax(scores, dim=-1).diag()).mean() + (-F.log_softmax(scores.T, dim=-1).diag()).mean()


def teacher_student_nce_loss(scores: torch.Tensor, teacher_scores: torch.Tensor,
                             reduction: str = "batchmean") -> torch.Tensor:
    """loss.py:29-39; the reference instantiates it with reduction="batchmean" (teacher_student.py:72-73)."""
    rows = F.kl_div(F.log_softmax(scores, dim=-1), F.softmax(teacher_scores, dim=-1), reduction=reduction)
    cols = F.kl_div(F.log_softmax(scores.T, dim=-1), F.softmax(teacher_scores.T, dim=-1), reduction=reduction)
    return rows + cols


def step_scores(encoded_video: torch.Tensor, encoded_text: torch.Tensor, init_temperature: float) -> torch.Tensor:
    """video_text_module.py:32,62-63: logit_scale = -log(T); scores = exp(logit_scale) * V @ T^T."""
    logit_scale = torch.tensor([-math.log(init_temperature)]).exp()
    return logit_scale * encoded_video @ encoded_text.T


def teacher_student_training_loss(student_out: Mapping[str, Tuple[torch.Tensor, torch.Tensor]],
                                  teacher_out: Mapping[str, Tuple[torch.Tensor, torch.Tensor]],
                                  logit_scale: torch.Tensor, teacher_student_logit_scale: torch.Tensor,
                                  dataset_loss_share: Mapping[str, float], labeled_dataset_name: str = "labeled"
                                  ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """teacher_student.py:142-176 (`_dataset_step_end` + `training_step_end`) on already gathered embeddings:
    per dataset `scores = exp(logit_scale) * V @ T^T`; NCE on the labeled dataset, KD("batchmean") * exp(ts_scale)^2 on
    the other; total = sum_d share_d * loss_d.  Differentiable (plain torch): autograd through it and the towers above is
    the oracle of the training step."""
    losses = {}
    for name, (video, text) in student_out.items():
        scores = logit_scale.exp() * video @ text.T
        if name == labeled_dataset_name:
            losses[name] = nce_loss(scores)
        else:
            t_video, t_text = teacher_out[name]
            ts = teacher_student_logit_scale.exp()
            teacher_scores = ts * t_video @ t_text.T
            losses[name] = teacher_student_nce_loss(scores, teacher_scores) * ts ** 2
    total = sum(losses[name] * dataset_loss_share[name] for name in losses)
    return total, losses


# ---------------------------------------------------------------------------------------------------------- metrics
def retrieval_scores(encoded_texts: torch.Tensor, encoded_videos: torch.Tensor) -> torch.Tensor:
    """text_video_retrieval.py:74."""
    return encoded_texts @ encoded_videos.T


def ranks_of_target(scores: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """metrics.py:16-20: column at which `target` appears in the descending argsort of each row.  Ties are broken by
    index (stable sort), which is the order the HIP kernel reproduces."""
    order = scores.argsort(dim=1, descending=True, stable=True)
    return torch.where(order == target.unsqueeze(-1))[1]


def retrieval_metrics(scores: torch.Tensor) -> Dict[str, float]:
    """text_video_retrieval.py:76-83 + metrics.py:33-36: R@1/5/10 (torchmetrics Recall(top_k): target among the k
    highest scores of its row) and MedianRank (= torch `median`, lower middle, + 1)."""
    target = torch.arange(scores.shape[-1])
    ranks = ranks_of_target(scores, target)
    out = {f"r{k}": (ranks < k).float().mean().item() for k in (1, 5, 10)}
    out["mr"] = float(ranks.median().item() + 1)
    return out


def zero_shot_label_embeddings(encoded_prompts: torch.Tensor, template_count: int) -> torch.Tensor:
    """video_text_classification.py:88-90: mean of the prompt embeddings over the templates of each label."""
    return encoded_prompts.reshape(-1, template_count, encoded_prompts.shape[1]).mean(dim=1)


def zero_shot_metrics(scores: torch.Tensor, label_id: torch.Tensor) -> Dict[str, float]:
    """video_text_classification.py:62-63,110-118: Accuracy@1 / @5 (label among the k best columns) and the median
    rank (+1) of the true label."""
    ranks = ranks_of_target(scores, label_id)
    return {"a1": (ranks < 1).float().mean().item(), "a5": (ranks < 5).float().mean().item(),
            "mr": float(ranks.median().item() + 1)}


def flatten_gathered(t: torch.Tensor) -> torch.Tensor:
    """tensor_utils.py:58-60: [world, B, ...] -> [world * B, ...]."""
    return t.view(-1, *t.shape[2:])


def to_torch(sd_np: Mapping[str, "object"]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(v) if not isinstance(v, torch.Tensor) else v for k, v in sd_np.items()}

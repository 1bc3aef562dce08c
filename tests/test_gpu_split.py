"""The split-fp32 mode (`precision="fp32x6"`, fc_config.split_gemm): the block GEMMs of the visual tower on the bf16 matrix
cores over six-plane operands - every fp32 value as three bf16 numbers, every product as six bf16 products accumulated
in fp32.  It must meet the fp32 tolerances of SURVEY.md section 8(c): embeddings <= 2e-5, scores <= 5e-5, identical
ranks - against the same fixtures (pinned to the reference's slip / HF CLIP) the fp32-MFMA path is tested with."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fitclip_amd import ops, synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402
from fitclip_amd.retrieval import TextVideoRetrievalModule  # noqa: E402

DEV = "cuda"
F32_TOL = 2e-5


def _planes(x):
    p1 = x.bfloat16()
    r = x - p1.float()
    p2 = r.bfloat16()
    return p1, p2, (r - p2.float()).bfloat16()


def _expand(ps, order):
    rows, K = ps[0].shape
    return torch.cat([ps[i].view(rows, K // 32, 1, 32) for i in order], dim=2).reshape(rows, 6 * K).contiguous()


def _unpack(o6):
    v = o6.view(o6.shape[0], -1, 6, 32)
    return [v[:, :, i].reshape(o6.shape[0], -1) for i in range(6)]


def _check_image(o6, want):
    """o6 is the activation-side six-plane image of the fp32 tensor `want`: [p1 p1 p2 p2 p1 p3], the canonical split."""
    p = _unpack(o6)
    q1, q2, q3 = _planes(want)
    assert torch.equal(p[0], q1) and torch.equal(p[1], q1) and torch.equal(p[4], q1)
    assert torch.equal(p[2], q2) and torch.equal(p[3], q2) and torch.equal(p[5], q3)
    assert torch.equal(p[0].float() + p[2].float() + p[5].float(), want)  # the three planes add up to the value exactly


def test_split6_is_the_exact_three_term_split():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(300, 96, device=DEV, generator=g) * torch.logspace(-6, 6, 96, device=DEV)
    x[0, :4] = torch.tensor([0.0, -0.0, 1.0, 2.0 ** -120], device=DEV)
    a6, w6 = ops.split6(x), ops.split6(x, weight=True)
    assert torch.equal(a6, _expand(_planes(x), (0, 0, 1, 1, 0, 2)))
    assert torch.equal(w6, _expand(_planes(x), (0, 1, 0, 1, 2, 0)))
    _check_image(a6, x)
    with pytest.raises(ValueError):
        ops.split6(x[:, :40].contiguous())


@pytest.mark.parametrize("M,N,K", [(20000, 768, 768), (16389, 2304, 768), (12345, 768, 3072)])
def test_six_product_gemm_has_fp32_accuracy(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(M)
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    y6 = ops.gemm(ops.split6(a), ops.split6(w, weight=True), bias, ops.EPI_BIAS_F32)
    y32 = ops.gemm(a, w, bias, ops.EPI_BIAS_T)
    rows = torch.cat([torch.arange(0, 512), torch.arange(M - 600, M)]).to(DEV)  # first tiles and the ragged last one
    ref = a[rows].double() @ w.double().T + bias.double()
    scale = float(ref.abs().max())
    e6 = float((y6[rows].double() - ref).abs().max()) / scale
    e32 = float((y32[rows].double() - ref).abs().max()) / scale
    assert e6 < 3e-6 and e6 < 1.5 * e32 + 1e-7, (e6, e32)
    assert float((y6 - y32).abs().max()) / scale < 6e-6


def test_quickgelu_epilogue_writes_the_canonical_planes():
    M, N, K = 16500, 1024, 256
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    a6, w6 = ops.split6(a), ops.split6(w, weight=True)
    pre = ops.gemm(a6, w6, bias, ops.EPI_BIAS_F32)
    h6 = ops.gemm(a6, w6, bias, ops.EPI_GELU_X6)
    p = _unpack(h6)
    h = p[0].float() + p[2].float() + p[5].float()
    _check_image(h6, h)                                      # a valid six-plane image of an fp32 tensor h ...
    want = pre.double() * torch.sigmoid(1.702 * pre.double())
    assert float((h.double() - want).abs().max() / want.abs().max()) < 3e-7   # ... and h = QuickGELU(pre) to fp32 accuracy
    # feeding it to the next GEMM equals feeding split6(h)
    w2 = torch.randn(256, N, device=DEV, generator=g) / N ** 0.5
    b2 = torch.zeros(256, device=DEV)
    assert torch.equal(ops.gemm(h6, ops.split6(w2, weight=True), b2, ops.EPI_BIAS_F32),
                       ops.gemm(ops.split6(h), ops.split6(w2, weight=True), b2, ops.EPI_BIAS_F32))


def test_layernorm_and_attention_write_six_plane_rows():
    g = torch.Generator(device=DEV).manual_seed(2)
    rows, D = 777, 768
    x = torch.randn(rows, D, device=DEV, generator=g) * 3
    delta = torch.randn(rows, D, device=DEV, generator=g)
    gamma, beta = torch.randn(D, device=DEV, generator=g), torch.randn(D, device=DEV, generator=g)
    _check_image(ops.layernorm(x, gamma, beta, out_dtype="x6"), ops.layernorm(x, gamma, beta))
    x1, x2 = x.clone(), x.clone()
    y6, y = ops.add_layernorm(x1, delta, gamma, beta, six_plane=True), ops.add_layernorm(x2, delta, gamma, beta)
    _check_image(y6, y)
    assert torch.equal(x1, x2) and torch.equal(x1, x + delta)
    n_seq, S, heads = 5, 197, 12
    qkv = torch.randn(n_seq * S, 3 * heads * 64, device=DEV, generator=g)
    _check_image(ops.attention(qkv, n_seq, S, heads, six_plane=True), ops.attention(qkv, n_seq, S, heads))
    with pytest.raises(Exception):
        ops.attention(qkv[: 50 * n_seq].contiguous(), n_seq, 50, heads, six_plane=True)  # not a streaming-block length


@pytest.mark.parametrize("tag,dims", [("tiny", synth.TINY), ("vitb16", synth.VIT_B_16)])
def test_towers_match_reference_fixtures(golden_dir, tag, dims, request):
    """The raw visual tower in split mode vs the fixtures produced by the reference's slip classes and by HF CLIP
    (tiny: 17 tokens -> fp32 attention + split pass; ViT-B/16: the fused six-plane attention output)."""
    g = np.load(golden_dir / f"towers_{tag}.npz")
    sd = request.getfixturevalue(f"{tag}_state_dict")
    model = build_clip(sd, precision="fp32x6", device=DEV)
    plain = build_clip(sd, precision="fp32", device=DEV)
    video = torch.from_numpy(synth.make_video(int(g["n_clip"]), int(g["n_frames"]), dims, seed=int(g["seed"])))
    frames = video.reshape(-1, *video.shape[2:]).to(DEV)
    img = model.encode_image(frames)
    scale = max(1.0, float(np.abs(g["image_features_oracle"]).max()))
    for ref in ("oracle", "slip", "hf"):
        assert np.abs(img.cpu().numpy() - g[f"image_features_{ref}"]).max() < F32_TOL * scale, ref
    assert float((img - plain.encode_image(frames)).abs().max()) < 2e-6 * scale       # next to the fp32-MFMA path
    ids = torch.from_numpy(g["ids"]).to(DEV)
    assert torch.equal(model.encode_text(ids), plain.encode_text(ids))                 # the text tower IS the fp32 path


def test_evaluate_goldens_in_split_mode(golden_dir, vitb16_state_dict):
    """`command=evaluate` end to end at ViT-B/16: embeddings, scores, ranks and metrics at the fp32 tolerances."""
    g = np.load(golden_dir / "evaluate_config1.npz")
    n, f = int(g["n_clips"]), int(g["n_frames"])
    video = torch.from_numpy(synth.make_video(n, f, synth.VIT_B_16, seed=42)).to(DEV)
    ids = torch.from_numpy(synth.make_text(n, synth.VIT_B_16, seed=42)).to(DEV)
    module = TextVideoRetrievalModule(ClipVideoTextEncoder(build_clip(vitb16_state_dict, precision="fp32x6", device=DEV)),
                                      init_temperature=0.015)
    for s in range(0, n, 4):
        module.validation_step_end(module.validation_step({"video": video[s:s + 4], "text": {"input_ids": ids[s:s + 4]},
                                                           "video_id": list(range(s, s + 4))}))
    ev = torch.cat([o[0] for o in module._outputs])
    et = torch.cat([o[1] for o in module._outputs])
    metrics = module.validation_epoch_end()
    assert np.abs(ev.cpu().numpy() - g["encoded_videos"]).max() < F32_TOL
    assert np.abs(et.cpu().numpy() - g["encoded_texts"]).max() < F32_TOL
    scores = ops.similarity(et, ev)
    assert np.abs(scores.cpu().numpy() - g["scores"]).max() < 5e-5
    assert ops.ranks(scores).tolist() == g["ranks"].tolist()
    for k in ("r1", "r5", "r10", "mr"):
        assert metrics[k] == pytest.approx(float(g[k])), k


def test_big_pass_and_ragged_tail_agree_with_the_fp32_path(vitb16_state_dict):
    """600 frames: one 512-frame pass + an 88-frame pass, against the fp32-MFMA path on the same frames, and a batch
    computed in pieces equals the batch computed at once (rows are independent)."""
    split = build_clip(vitb16_state_dict, precision="fp32x6", device=DEV)
    plain = build_clip(vitb16_state_dict, precision="fp32", device=DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    base = torch.randn(24, 3, 224, 224, device=DEV, generator=g).clamp_(-2.5, 2.5)
    frames = base[torch.arange(600, device=DEV) % 24].contiguous()
    got = split.encode_image(frames)
    diff = float((got - plain.encode_image(frames)).abs().max()) / max(1.0, float(got.abs().max()))
    assert diff < 1e-5, diff   # two fp32-accurate evaluations of un-normalised features (each <= 2e-5 from the oracle)
    assert torch.equal(got[:24], got[24:48]) and torch.equal(got[:24], got[576:600])   # same frame, same bits, any pass
    assert torch.equal(split.encode_image(frames[:7].contiguous()), got[:7])


def test_split_mode_needs_fp32_and_cannot_train(tiny_state_dict):
    from fitclip_amd.training import StudentTrainer
    model = build_clip(tiny_state_dict, precision="fp32x6", device=DEV)
    with pytest.raises(ValueError):
        StudentTrainer(ClipVideoTextEncoder(model))


def test_split_mode_is_as_close_to_float64_as_fp32_arithmetic_itself(vitb16_state_dict):
    """Distance to the TRUTH (the oracle evaluated in float64) of three fp32-grade evaluations of ViT-B/16 embeddings:
    the oracle in float32 (the reference's arithmetic on a CPU), the fp32-MFMA path and the split-fp32 path.  The split
    path must not be further from the truth than ordinary fp32 arithmetic is."""
    from oracle import clip_oracle as O
    d = synth.VIT_B_16
    sd32 = O.to_torch(vitb16_state_dict)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
    video = torch.from_numpy(synth.make_video(6, 2, d, seed=77))
    with torch.inference_mode():
        truth = O.encode_video(sd64, video.double())
        cpu32 = O.encode_video(sd32, video)
    err = {"oracle fp32": float((cpu32.double() - truth).abs().max())}
    for precision in ("fp32", "fp32x6"):
        enc = ClipVideoTextEncoder(build_clip(vitb16_state_dict, precision=precision, device=DEV))
        err[precision] = float((enc.encode_video(video.to(DEV)).cpu().double() - truth).abs().max())
    print("max |embedding - float64 truth|:", err)
    assert err["fp32x6"] < 1e-6 and err["fp32"] < 1e-6
    assert err["fp32x6"] <= 2.0 * max(err["oracle fp32"], err["fp32"]) + 5e-8, err

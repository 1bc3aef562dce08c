"""The split-fp32 mode (`precision="fp32x6"`, fc_config.split_gemm): the block GEMMs of the visual tower on the bf16 matrix
cores over three-plane operands - every fp32 value as three bf16 numbers, every product as six bf16 products (hence the
mode's name) formed from registers and accumulated in fp32.  It must meet the fp32 tolerances of SURVEY.md section 8(c): embeddings <= 2e-5, scores <= 5e-5, identical
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


def _check_image(o3, want):
    """o3 holds the x3 rows of the fp32 tensor `want`: per 16 columns one line [p1 | p2 | p3 | unused], the canonical split."""
    p = ops.x3_planes(o3)
    q1, q2, q3 = _planes(want)
    assert torch.equal(p[0], q1) and torch.equal(p[1], q2) and torch.equal(p[2], q3)
    assert torch.equal(p[0].float() + p[1].float() + p[2].float(), want)  # the three planes add up to the value exactly


def test_split3_is_the_exact_three_term_split():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(300, 96, device=DEV, generator=g) * torch.logspace(-6, 6, 96, device=DEV)
    x[0, :4] = torch.tensor([0.0, -0.0, 1.0, 2.0 ** -120], device=DEV)
    a3 = ops.split3(x)
    assert a3.shape == (300, 4 * 96) and a3.dtype == torch.bfloat16
    _check_image(a3, x)
    lines = a3.view(300, 6, 4, 16)
    assert not lines[:, :, 3].any()                           # the fourth quarter of a line is never written
    with pytest.raises(ValueError):
        ops.split3(x[:, :40].contiguous())


@pytest.mark.parametrize("M,N,K", [(20000, 768, 768), (16389, 2304, 768), (12345, 768, 3072), (300, 512, 64), (70000, 3072, 256)])
def test_six_product_gemm_has_fp32_accuracy(M, N, K):
    """fc_gemm_split3: six bf16 products per fp32 product, formed from registers over three-plane operands.  Against a float64
    product of the same fp32 operands it must be as accurate as the fp32-input MFMA kernel; shapes: whole tiles and ragged
    last row panels, more tiles than CUs (several tiles per workgroup, the prefetch across the tile boundary), the shortest K."""
    g = torch.Generator(device=DEV).manual_seed(M)
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    y3 = ops.gemm_split3(ops.split3(a), ops.split3(w), bias, ops.EPI_BIAS_F32)
    y32 = ops.gemm(a, w, bias, ops.EPI_BIAS_T)
    rows = torch.cat([torch.arange(0, min(512, M)), torch.arange(max(0, M - 600), M)]).unique().to(DEV)  # first tiles, ragged last one
    ref = a[rows].double() @ w.double().T + bias.double()
    scale = float(ref.abs().max())
    e3 = float((y3[rows].double() - ref).abs().max()) / scale
    e32 = float((y32[rows].double() - ref).abs().max()) / scale
    assert e3 < 3e-6 and e3 < 1.5 * e32 + 1e-7, (e3, e32)
    assert float((y3 - y32).abs().max()) / scale < 6e-6      # EVERY element, against the fp32-MFMA kernel
    assert torch.equal(ops.gemm_split3(ops.split3(a), ops.split3(w), bias, ops.EPI_BIAS_F32), y3)  # run to run
    # a row's result does not depend on the rows around it (tile order, K rotation): a slice equals the slice of the whole
    lo = max(0, M - 300)
    assert torch.equal(ops.gemm_split3(ops.split3(a[lo:].contiguous()), ops.split3(w), bias, ops.EPI_BIAS_F32), y3[lo:])


@pytest.mark.parametrize("M,N,K", [(20000, 768, 768), (12345, 768, 3072), (300, 512, 64)])
def test_residual_epilogue_is_the_bias_epilogue_plus_the_stream(M, N, K):
    """EPI_RESID3_F32 (C += A . W^T + bias in place: what out_proj / c_proj run in the split mode) must give the bits of
    EPI_BIAS_F32 followed by the fp32 add the fused add+LayerNorm did, on whole tiles and ragged last panels, and leave the
    rows beyond M alone."""
    g = torch.Generator(device=DEV).manual_seed(M + 1)
    a3 = ops.split3(torch.randn(M, K, device=DEV, generator=g))
    w3 = ops.split3(torch.randn(N, K, device=DEV, generator=g) / K ** 0.5)
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M + 3, N, device=DEV, generator=g) * 5
    want = x[:M] + ops.gemm_split3(a3, w3, bias, ops.EPI_BIAS_F32)
    got = x.clone()
    ops.gemm_split3(a3, w3, bias, ops.EPI_RESID3_F32, out=got[:M])
    assert torch.equal(got[:M], want) and torch.equal(got[M:], x[M:])


def test_gemm_split3_rejects_bad_operands():
    a3 = torch.zeros(64, 256, dtype=torch.bfloat16, device=DEV)
    bias = torch.zeros(48, device=DEV)
    with pytest.raises(Exception, match="N=48"):
        ops.gemm_split3(a3, torch.zeros(48, 256, dtype=torch.bfloat16, device=DEV), bias)   # N % 32
    with pytest.raises(Exception, match="K=32"):
        ops.gemm_split3(a3[:, :128].contiguous(), torch.zeros(64, 128, dtype=torch.bfloat16, device=DEV), torch.zeros(64, device=DEV))
    with pytest.raises(ValueError):
        ops.gemm(a3, a3, torch.zeros(64, device=DEV), ops.EPI_BIAS_F32)                      # not an epilogue of the plain GEMM


def test_quickgelu_epilogue_writes_the_canonical_planes():
    M, N, K = 16500, 1024, 256
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    a3, w3 = ops.split3(a), ops.split3(w)
    pre = ops.gemm_split3(a3, w3, bias, ops.EPI_BIAS_F32)
    h3 = ops.gemm_split3(a3, w3, bias, ops.EPI_GELU_X3)
    p = ops.x3_planes(h3)
    h = p[0].float() + p[1].float() + p[2].float()
    _check_image(h3, h)                                      # valid x3 rows of an fp32 tensor h ...
    want = pre.double() * torch.sigmoid(1.702 * pre.double())
    assert float((h.double() - want).abs().max() / want.abs().max()) < 3e-7   # ... and h = QuickGELU(pre) to fp32 accuracy
    # feeding it to the next GEMM equals feeding split3(h)
    w2 = torch.randn(256, N, device=DEV, generator=g) / N ** 0.5
    b2 = torch.zeros(256, device=DEV)
    assert torch.equal(ops.gemm_split3(h3, ops.split3(w2), b2), ops.gemm_split3(ops.split3(h), ops.split3(w2), b2))


def test_layernorm_and_attention_write_three_plane_rows():
    g = torch.Generator(device=DEV).manual_seed(2)
    rows, D = 777, 768
    x = torch.randn(rows, D, device=DEV, generator=g) * 3
    delta = torch.randn(rows, D, device=DEV, generator=g)
    gamma, beta = torch.randn(D, device=DEV, generator=g), torch.randn(D, device=DEV, generator=g)
    _check_image(ops.layernorm(x, gamma, beta, out_dtype="x3"), ops.layernorm(x, gamma, beta))
    x1, x2 = x.clone(), x.clone()
    y3, y = ops.add_layernorm(x1, delta, gamma, beta, three_plane=True), ops.add_layernorm(x2, delta, gamma, beta)
    _check_image(y3, y)
    assert torch.equal(x1, x2) and torch.equal(x1, x + delta)
    n_seq, S, heads = 5, 197, 12
    qkv = torch.randn(n_seq * S, 3 * heads * 64, device=DEV, generator=g)
    _check_image(ops.attention(qkv, n_seq, S, heads, three_plane=True), ops.attention(qkv, n_seq, S, heads))
    with pytest.raises(Exception):
        ops.attention(qkv[: 50 * n_seq].contiguous(), n_seq, 50, heads, three_plane=True)  # not a streaming-block length


def _attention_f64(qkv, n_seq, S, heads):
    q, k, v = (t.reshape(n_seq, S, heads, 64).permute(0, 2, 1, 3).double() for t in qkv.chunk(3, dim=1))
    p = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(n_seq * S, heads * 64)


@pytest.mark.parametrize("S", [197, 193, 208])
def test_split_attention_has_fp32_accuracy(S):
    """fc_attention precision 4: softmax(q k^T / 8) v with both products as six bf16 products per fp32 product (the attention
    of the split-fp32 mode).  Against float64 it must be as accurate as the fp32-MFMA kernel; its output must be the
    canonical x3 image of an fp32 tensor; a (sequence, head) pair's result must not depend on how many pairs the persistent
    workgroups walk over (300 sequences: 3600 pairs on 256 workgroups, the prefetch across passes) nor on the run."""
    heads = 12
    g = torch.Generator(device=DEV).manual_seed(S)
    n_seq = 300
    qkv = torch.randn(n_seq * S, 3 * heads * 64, device=DEV, generator=g) * 2.0   # scores of +-30: peaked rows too
    qkv[: S, : heads * 64] *= 3.0
    o3 = ops.attention(qkv, n_seq, S, heads, split=True)
    p = ops.x3_planes(o3)
    o = p[0].float() + p[1].float() + p[2].float()
    _check_image(o3, o)
    lines = o3.view(n_seq * S, heads * 4, 4, 16)
    assert not lines[:, :, 3].any()                              # the fourth quarter of every line is zero
    some = torch.cat([torch.arange(0, 3 * S), torch.arange(150 * S, 151 * S), torch.arange((n_seq - 2) * S, n_seq * S)]).to(DEV)
    n_some = some.numel() // S
    ref = _attention_f64(qkv[some], n_some, S, heads)
    o32 = ops.attention(qkv, n_seq, S, heads)
    scale = float(ref.abs().max())
    e_split = float((o[some].double() - ref).abs().max()) / scale
    e_f32 = float((o32[some].double() - ref).abs().max()) / scale
    assert e_split < 5e-6 and e_split < 1.2 * e_f32 + 1e-7, (e_split, e_f32)
    assert float((o - o32).abs().max()) / scale < 1e-5          # EVERY element, against the fp32-MFMA kernel
    assert torch.equal(ops.attention(qkv, n_seq, S, heads, split=True), o3)  # run to run
    few = 7                                                      # fewer pairs than workgroups: one pass each
    assert torch.equal(ops.attention(qkv[: few * S].contiguous(), few, S, heads, split=True), o3[: few * S])
    tail = qkv[(n_seq - few) * S:].contiguous()
    assert torch.equal(ops.attention(tail, few, S, heads, split=True), o3[(n_seq - few) * S:])


def test_split_attention_rejects_other_lengths():
    qkv = torch.zeros(2 * 192, 3 * 768, device=DEV)
    with pytest.raises(Exception, match="S=192"):
        ops.attention(qkv, 2, 192, 12, split=True)
    with pytest.raises(Exception, match="S=209"):
        ops.attention(torch.zeros(2 * 209, 3 * 768, device=DEV), 2, 209, 12, split=True)
    with pytest.raises(Exception, match="causal"):
        ops.attention(torch.zeros(2 * 197, 3 * 768, device=DEV), 2, 197, 12, causal=True, split=True)


@pytest.mark.parametrize("tag,dims", [("tiny", synth.TINY), ("vitb16", synth.VIT_B_16)])
def test_towers_match_reference_fixtures(golden_dir, tag, dims, request):
    """The raw visual tower in split mode vs the fixtures produced by the reference's slip classes and by HF CLIP
    (tiny: 17 tokens -> fp32 attention + split pass; ViT-B/16: the fused three-plane attention output)."""
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


def test_gemm_split3_race_screen():
    """The three-plane GEMM orders its LDS-DMA against fragment reads with counted vmcnt waits and one barrier per K-step
    (three-stage ring, pieces spread over MFMA groups, the epilogue borrowing the released stage).  A misplaced wait would show
    as rare wrong tiles, not as a failure of a single run: 40 launches of a multi-round shape (5 tiles per workgroup, ragged
    last panel, both epilogues) must reproduce the first result bitwise, with another GEMM's traffic in between."""
    g = torch.Generator(device=DEV).manual_seed(7)
    M, N, K = 256 * 105 + 77, 3072, 768
    a3 = ops.split3(torch.randn(M, K, device=DEV, generator=g))
    w3 = ops.split3(torch.randn(N, K, device=DEV, generator=g) / K ** 0.5)
    bias = torch.randn(N, device=DEV, generator=g)
    other_a = ops.split3(torch.randn(30000, 3072, device=DEV, generator=g))
    other_w = ops.split3(torch.randn(768, 3072, device=DEV, generator=g) / 3072 ** 0.5)
    other_b = torch.zeros(768, device=DEV)
    first = {epi: ops.gemm_split3(a3, w3, bias, epi) for epi in (ops.EPI_BIAS_F32, ops.EPI_GELU_X3)}
    first_other = ops.gemm_split3(other_a, other_w, other_b)
    for i in range(20):
        for epi in (ops.EPI_BIAS_F32, ops.EPI_GELU_X3):
            assert torch.equal(ops.gemm_split3(a3, w3, bias, epi), first[epi]), (i, epi)
        assert torch.equal(ops.gemm_split3(other_a, other_w, other_b), first_other), i


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two visible devices")
def test_second_device_gets_its_own_launch_attributes():
    """Launch attributes are per (device, kernel): the dynamic-LDS limit of the large kernels and the CU count that sizes the
    persistent grids.  A process that first ran on device 0 must produce the same bits on device 1 (round 3 cached both in
    function-local statics of the split attention: the second device's launch failed or used the wrong grid)."""
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(3 * 197, 3 * 2 * 64, generator=g)
    a, w, bias = torch.randn(25216, 768, generator=g), torch.randn(768, 768, generator=g) * 768 ** -0.5, torch.randn(768, generator=g)
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        with torch.cuda.device(dev):
            o = ops.attention(qkv.to(dev), 3, 197, 2, split=True)
            c = ops.gemm(a.to(dev), w.to(dev), bias.to(dev), ops.EPI_BIAS_T, tile=3)
            c3 = ops.gemm_split3(ops.split3(a.to(dev)), ops.split3(w.to(dev)), bias.to(dev))
            outs.append((o.cpu(), c.cpu(), c3.cpu()))
    for x, y in zip(*outs):
        assert torch.equal(x, y)

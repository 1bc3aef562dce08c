"""The three-product split-fp32 mode (`precision="fp32x3"`, fc_config.split_gemm = 2): the block GEMMs of the visual tower on the
fp16 matrix cores over TWO-plane operands - x = h1 + 2^-11 h2 in fp16, a weight s w = g1 + g2 with a power-of-two scale per
tensor, every fp32 product as the three fp16 products h1 g1 + h1 g2 + h2 (2^-11 g1) accumulated in fp32.  It must meet the fp32
tolerances of SURVEY.md section 8(c) - embeddings <= 2e-5, scores <= 5e-5, identical ranks - against the same fixtures (pinned to
the reference's slip / HF CLIP) as the fp32-MFMA path, and it must never be silently wrong outside fp16's range."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fitclip_amd import _lib, ops, synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402
from fitclip_amd.retrieval import TextVideoRetrievalModule  # noqa: E402

DEV = "cuda"
F32_TOL = 2e-5


def _planes(x):
    """The canonical two-plane image of an fp32 activation tensor."""
    h1 = x.half()
    return h1, ((x - h1.float()) * 2048.0).half()


def _value(x2):
    h1, h2 = ops.x2_planes(x2)
    return h1.double() + h2.double() / 2048.0


def _check_image(o2, want):
    """o2 holds the x2 rows of the fp32 tensor `want`: per 32 columns one line [h1 | h2]."""
    p = ops.x2_planes(o2)
    q1, q2 = _planes(want)
    assert torch.equal(p[0], q1) and torch.equal(p[1], q2)


def test_split2_is_the_two_plane_image_and_flags_what_it_cannot_hold():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(300, 96, device=DEV, generator=g) * torch.logspace(-9, 4, 96, device=DEV)
    x[0, :6] = torch.tensor([0.0, -0.0, 1.0, 65504.0, -2.0 ** -14, 3.0e-8], device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    a2 = ops.split2(x, flag)
    assert a2.shape == (300, 2 * 96) and a2.dtype == torch.float16
    _check_image(a2, x)
    # 22 bits wherever h1 is a normal fp16 number, 2^-36 absolute below
    err = (_value(a2) - x.double()).abs()
    assert bool((err <= torch.maximum(x.double().abs() * 2.0 ** -22, torch.full_like(err, 2.0 ** -35))).all())
    assert int(flag) == 0
    x[7, 5] = 7.0e4                                              # beyond fp16: flagged, and loud (an infinity in the plane)
    a2 = ops.split2(x, flag)
    assert int(flag) == 1 and bool(torch.isinf(ops.x2_planes(a2)[0][7, 5]))
    with pytest.raises(ValueError):
        ops.split2(x[:, :40].contiguous())


def test_weight_scale_is_the_power_of_two_that_fills_the_range():
    g = torch.Generator(device=DEV).manual_seed(1)
    for mag in (1e-4, 0.03, 1.0, 700.0):
        w = torch.randn(512, 256, device=DEV, generator=g) * mag
        w2, scale = ops.split2_weight(w)
        s, inv = (float(v) for v in scale)
        assert s * inv == 1.0 and np.log2(s) == round(np.log2(s))
        assert 2.0 ** 14 <= s * float(w.abs().max()) < 2.0 ** 15
        g1, g2 = ops.x2_planes(w2)
        assert torch.equal(g1, (w * s).half()) and torch.equal(g2, (w * s - g1.float()).half())
        rel = ((g1.double() + g2.double()) * inv - w.double()).abs().max() / w.abs().max()
        assert float(rel) < 2.0 ** -22
    z2, zs = ops.split2_weight(torch.zeros(32, 64, device=DEV))
    assert zs.tolist() == [1.0, 1.0] and not z2.any()


@pytest.mark.parametrize("M,N,K", [(20000, 768, 768), (16389, 2304, 768), (12345, 768, 3072), (300, 512, 128), (70000, 3072, 256),
                                   # ViT-L/14's block shapes (width 1024, 257 tokens; config/encoder/clip_vit_l_14.yaml of the reference)
                                   (16448, 1024, 1024), (16448, 3072, 1024), (8224, 4096, 1024), (8224, 1024, 4096)])
def test_three_product_gemm_has_fp32_accuracy(M, N, K):
    """fc_gemm_split2 against a float64 product of the same fp32 operands: as accurate as the fp32-input MFMA kernel; whole
    tiles and ragged last row panels, more tiles than CUs (the prefetch across the tile boundary), the shortest K."""
    g = torch.Generator(device=DEV).manual_seed(M)
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    w2, sc = ops.split2_weight(w)
    y2 = ops.gemm_split2(ops.split2(a), w2, sc, bias, ops.EPI_BIAS_F32)
    y32 = ops.gemm(a, w, bias, ops.EPI_BIAS_T)
    rows = torch.cat([torch.arange(0, min(512, M)), torch.arange(max(0, M - 600), M)]).unique().to(DEV)
    ref = a[rows].double() @ w.double().T + bias.double()
    scale = float(ref.abs().max())
    e2 = float((y2[rows].double() - ref).abs().max()) / scale
    e32 = float((y32[rows].double() - ref).abs().max()) / scale
    assert e2 < 3e-6 and e2 < 1.5 * e32 + 1e-7, (e2, e32)
    assert float((y2 - y32).abs().max()) / scale < 6e-6      # EVERY element, against the fp32-MFMA kernel
    assert torch.equal(ops.gemm_split2(ops.split2(a), w2, sc, bias, ops.EPI_BIAS_F32), y2)  # run to run
    lo = max(0, M - 300)                                     # a row's result does not depend on the rows around it
    assert torch.equal(ops.gemm_split2(ops.split2(a[lo:].contiguous()), w2, sc, bias, ops.EPI_BIAS_F32), y2[lo:])


@pytest.mark.parametrize("a_mag,w_mag", [(1e-3, 1e-3), (300.0, 0.02), (1e-6, 5.0), (30.0, 30.0)])
def test_three_product_gemm_keeps_its_accuracy_across_magnitudes(a_mag, w_mag):
    """The planes are robust, not tuned to unit-scale data: small activations (subnormal h1), large ones, tiny and large
    weights (the per-tensor scale), and a wide spread INSIDE one tensor."""
    M, N, K = 4096, 512, 768
    g = torch.Generator(device=DEV).manual_seed(5)
    a = torch.randn(M, K, device=DEV, generator=g) * a_mag * torch.logspace(-3, 0, K, device=DEV)
    w = torch.randn(N, K, device=DEV, generator=g) * w_mag
    w[::7] *= 1e-4                                           # rows far below the tensor's maximum
    bias = torch.zeros(N, device=DEV)
    w2, sc = ops.split2_weight(w)
    y2 = ops.gemm_split2(ops.split2(a), w2, sc, bias)
    ref = a.double() @ w.double().T
    # per element against the float64 product: a few ulp of the terms' magnitude, as fp32 arithmetic gives - down to the floor of
    # the fp16 planes: an activation below 2^-14 is held to 2^-36 ABSOLUTE (common.h), i.e. 2^-36 sum_k |w| per output element
    mag = a.double().abs() @ w.double().abs().T
    floor = 2.0 ** -35 * w.double().abs().sum(dim=1)[None, :]
    err = (y2.double() - ref).abs()
    assert bool((err <= 4e-7 * mag + floor).all()), float((err / (4e-7 * mag + floor)).max())
    if a_mag >= 1e-3:   # (activations of the size the towers produce: indistinguishable from the fp32-MFMA kernel)
        assert float((err / mag).max()) < 4e-7
        y32 = ops.gemm(a, w, bias, ops.EPI_BIAS_T)
        assert float((err / mag).max()) < 1.5 * float(((y32.double() - ref).abs() / mag).max()) + 5e-8


@pytest.mark.parametrize("M,N,K", [(20000, 768, 768), (12345, 768, 3072), (300, 512, 128)])
def test_residual_epilogue_is_the_bias_epilogue_plus_the_stream(M, N, K):
    g = torch.Generator(device=DEV).manual_seed(M + 1)
    a2 = ops.split2(torch.randn(M, K, device=DEV, generator=g))
    w2, sc = ops.split2_weight(torch.randn(N, K, device=DEV, generator=g) / K ** 0.5)
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M + 3, N, device=DEV, generator=g) * 5
    want = x[:M] + ops.gemm_split2(a2, w2, sc, bias, ops.EPI_BIAS_F32)
    got = x.clone()
    ops.gemm_split2(a2, w2, sc, bias, ops.EPI_RESID3_F32, out=got[:M])
    assert torch.equal(got[:M], want) and torch.equal(got[M:], x[M:])


@pytest.mark.parametrize("epi", ["bias", "resid", "gelu"])
@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (25216, 2304, 768), (25216, 768, 3072), (7000, 3072, 768), (130, 768, 768),
                                   (6425, 1024, 1024), (6425, 4096, 1024), (6425, 1024, 4096)])   # 25 frames of ViT-L/14
def test_tile_height_is_bit_invisible(M, N, K, epi):
    """The launcher picks 256-, 192- or 128-row tiles by the rounds of the busiest XCD: an element sees the same K order and the same
    product chain whatever tile holds it, so the automatic choice and the three forced heights give the same bits."""
    g = torch.Generator(device=DEV).manual_seed(M + N)
    a2 = ops.split2(torch.randn(M, K, device=DEV, generator=g))
    w2, sc = ops.split2_weight(torch.randn(N, K, device=DEV, generator=g) / K ** 0.5)
    bias = torch.randn(N, device=DEV, generator=g)
    x = torch.randn(M, N, device=DEV, generator=g)
    outs = []
    for cut in (1, 0, 2, 3):
        if epi == "resid":
            o = x.clone()
            ops.gemm_split2(a2, w2, sc, bias, ops.EPI_RESID3_F32, out=o, cut=cut)
        else:
            o = ops.gemm_split2(a2, w2, sc, bias, ops.EPI_GELU_X2 if epi == "gelu" else ops.EPI_BIAS_F32, cut=cut)
        outs.append(o)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("M,N,K", [(1000, 288, 128), (70, 32, 192), (513, 800, 256)])
def test_ragged_column_tiles(M, N, K):
    """N a multiple of 32 but not of the 256-column tile (partial last column tile: clamped weight rows, guarded stores), K at its
    minimum and at an odd multiple of 64; all three epilogues against float64 / each other, rows and columns beyond left alone."""
    g = torch.Generator(device=DEV).manual_seed(N)
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    a2 = ops.split2(a)
    w2, sc = ops.split2_weight(w)
    ref = a.double() @ w.double().T + bias.double()
    y = ops.gemm_split2(a2, w2, sc, bias)
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 3e-6
    x = torch.randn(M + 2, N, device=DEV, generator=g)
    got = x.clone()
    ops.gemm_split2(a2, w2, sc, bias, ops.EPI_RESID3_F32, out=got[:M])
    assert torch.equal(got[:M], x[:M] + y) and torch.equal(got[M:], x[M:])
    h2 = ops.gemm_split2(a2, w2, sc, bias, ops.EPI_GELU_X2)
    want = ref * torch.sigmoid(1.702 * ref)
    assert h2.shape == (M, 2 * N) and float((_value(h2) - want).abs().max() / want.abs().max()) < 1.5e-6   # (the GEMM's own error, through the GELU)
    for cut in (1, 2, 3):
        assert torch.equal(ops.gemm_split2(a2, w2, sc, bias, cut=cut), y)


def test_gemm_split2_rejects_bad_operands():
    a2 = torch.zeros(64, 256, dtype=torch.float16, device=DEV)
    sc = torch.ones(2, device=DEV)
    with pytest.raises(Exception, match="N=48"):
        ops.gemm_split2(a2, torch.zeros(48, 256, dtype=torch.float16, device=DEV), sc, torch.zeros(48, device=DEV))
    with pytest.raises(Exception, match="K=64"):
        ops.gemm_split2(a2[:, :128].contiguous(), torch.zeros(64, 128, dtype=torch.float16, device=DEV), sc, torch.zeros(64, device=DEV))
    with pytest.raises(ValueError):
        ops.gemm_split2(a2, a2, sc, torch.zeros(64, device=DEV), ops.EPI_GELU_X3)


def test_quickgelu_epilogue_writes_the_canonical_planes_and_flags_overflow():
    M, N, K = 16500, 1024, 256
    g = torch.Generator(device=DEV).manual_seed(1)
    a = torch.randn(M, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
    bias = torch.randn(N, device=DEV, generator=g)
    a2 = ops.split2(a)
    w2, sc = ops.split2_weight(w)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    pre = ops.gemm_split2(a2, w2, sc, bias, ops.EPI_BIAS_F32)
    h2 = ops.gemm_split2(a2, w2, sc, bias, ops.EPI_GELU_X2, flag=flag)
    want = pre.double() * torch.sigmoid(1.702 * pre.double())
    assert float((_value(h2) - want).abs().max() / want.abs().max()) < 4e-7    # QuickGELU(pre) to fp32 accuracy
    assert int(flag) == 0
    # ... and the rows are a valid operand of the next GEMM: against the float64 product of the values they hold
    wn32 = torch.randn(256, N, device=DEV, generator=g) / N ** 0.5
    wn, sn = ops.split2_weight(wn32)
    nxt = ops.gemm_split2(h2, wn, sn, torch.zeros(256, device=DEV))
    ref = _value(h2) @ wn32.double().T
    assert float((nxt.double() - ref).abs().max() / ref.abs().max()) < 3e-6
    # pre-activations beyond fp16's range: the flag goes up
    big_bias = bias.clone()
    big_bias[3] = 1.0e5
    ops.gemm_split2(a2, w2, sc, big_bias, ops.EPI_GELU_X2, flag=flag)
    assert int(flag) == 1


def test_layernorm_and_attention_write_two_plane_rows():
    g = torch.Generator(device=DEV).manual_seed(2)
    rows, D = 777, 768
    x = torch.randn(rows, D, device=DEV, generator=g) * 3
    delta = torch.randn(rows, D, device=DEV, generator=g)
    gamma, beta = torch.randn(D, device=DEV, generator=g), torch.randn(D, device=DEV, generator=g)
    _check_image(ops.layernorm(x, gamma, beta, out_dtype="x2"), ops.layernorm(x, gamma, beta))
    x1, x2 = x.clone(), x.clone()
    y2, y = ops.add_layernorm(x1, delta, gamma, beta, two_plane=True), ops.add_layernorm(x2, delta, gamma, beta)
    _check_image(y2, y)
    assert torch.equal(x1, x2) and torch.equal(x1, x + delta)
    for D2 in (256, 512, 1024):
        xx = torch.randn(130, D2, device=DEV, generator=g)
        gg, bb = torch.randn(D2, device=DEV, generator=g), torch.randn(D2, device=DEV, generator=g)
        _check_image(ops.layernorm(xx, gg, bb, out_dtype="x2"), ops.layernorm(xx, gg, bb))


def _attention_f64(qkv, n_seq, S, heads):
    q, k, v = (t.reshape(n_seq, S, heads, 64).permute(0, 2, 1, 3).double() for t in qkv.chunk(3, dim=1))
    p = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
    return (p @ v).permute(0, 2, 1, 3).reshape(n_seq * S, heads * 64)


@pytest.mark.parametrize("S", [197, 193, 208])
def test_split_attention_with_two_plane_output(S):
    """fc_attention precision 5: the split attention writing x2 rows - the SAME values as its x3 output (the kernel's fp32
    result, re-split), canonical planes, independent of the number of (sequence, head) pairs a launch walks over."""
    heads, n_seq = 12, 300
    g = torch.Generator(device=DEV).manual_seed(S)
    qkv = torch.randn(n_seq * S, 3 * heads * 64, device=DEV, generator=g) * 2.0
    o2 = ops.attention(qkv, n_seq, S, heads, split=True, two_plane=True)
    p3 = ops.x3_planes(ops.attention(qkv, n_seq, S, heads, split=True))
    o = p3[0].float() + p3[1].float() + p3[2].float()           # the fp32 value the kernel computed
    _check_image(o2, o)
    some = torch.cat([torch.arange(0, 3 * S), torch.arange((n_seq - 2) * S, n_seq * S)]).to(DEV)
    ref = _attention_f64(qkv[some], some.numel() // S, S, heads)
    assert float((_value(o2)[some] - ref).abs().max()) / float(ref.abs().max()) < 5e-6
    few = 7
    assert torch.equal(ops.attention(qkv[: few * S].contiguous(), few, S, heads, split=True, two_plane=True), o2[: few * S])


@pytest.mark.parametrize("S", [197, 193, 208])
def test_three_product_attention_has_fp32_accuracy(S):
    """fc_attention precision 6 (attention_split2.hip): softmax(q k^T / 8) v with both products as THREE fp16 products per fp32
    product (two planes per operand, scaled residuals, the cross terms in a second accumulator) - the attention of precision
    "fp32x3".  Against float64 it must be as accurate as the fp32-MFMA kernel; a (sequence, head) pair's result must not depend on
    how many pairs the persistent workgroups walk over (300 sequences: 3600 pairs on 256 workgroups) nor on the run."""
    heads, n_seq = 12, 300
    g = torch.Generator(device=DEV).manual_seed(S)
    qkv = torch.randn(n_seq * S, 3 * heads * 64, device=DEV, generator=g) * 2.0   # scores of +-30: peaked rows too
    qkv[: S, : heads * 64] *= 3.0
    o2 = ops.attention(qkv, n_seq, S, heads, split=True, two_plane=True, three_products=True)
    assert o2.shape == (n_seq * S, 2 * heads * 64) and o2.dtype == torch.float16 and bool(torch.isfinite(o2.float()).all())
    some = torch.cat([torch.arange(0, 3 * S), torch.arange(150 * S, 151 * S), torch.arange((n_seq - 2) * S, n_seq * S)]).to(DEV)
    ref = _attention_f64(qkv[some], some.numel() // S, S, heads)
    o32 = ops.attention(qkv, n_seq, S, heads)
    scale = float(ref.abs().max())
    e3 = float((_value(o2)[some] - ref).abs().max()) / scale
    e32 = float((o32[some].double() - ref).abs().max()) / scale
    assert e3 < 5e-6 and e3 < 1.2 * e32 + 1e-7, (e3, e32)
    assert float((_value(o2) - o32.double()).abs().max()) / scale < 1e-5          # EVERY element, against the fp32-MFMA kernel
    assert torch.equal(ops.attention(qkv, n_seq, S, heads, split=True, two_plane=True, three_products=True), o2)  # run to run
    few = 7                                                      # fewer pairs than workgroups: one pass each
    assert torch.equal(ops.attention(qkv[: few * S].contiguous(), few, S, heads, split=True, two_plane=True, three_products=True), o2[: few * S])
    tail = qkv[(n_seq - few) * S:].contiguous()
    assert torch.equal(ops.attention(tail, few, S, heads, split=True, two_plane=True, three_products=True), o2[(n_seq - few) * S:])
    # small and large operands alike (the planes are scale-free down to 2^-14): 1e-3 and 100 times the values above
    for mag in (1e-3, 1e2):
        q2 = qkv[: 4 * S].clone()
        q2[:, 2 * heads * 64:] *= mag                              # v: the output scales with it
        o = _value(ops.attention(q2, 4, S, heads, split=True, two_plane=True, three_products=True))
        r = _attention_f64(q2, 4, S, heads)
        assert float((o - r).abs().max() / r.abs().max()) < 5e-6, mag
    with pytest.raises(Exception, match="S=192"):
        ops.attention(torch.zeros(2 * 192, 3 * 768, device=DEV), 2, 192, 12, split=True, two_plane=True, three_products=True)
    with pytest.raises(Exception, match="causal"):
        ops.attention(torch.zeros(2 * 197, 3 * 768, device=DEV), 2, 197, 12, causal=True, split=True, two_plane=True, three_products=True)


@pytest.mark.parametrize("tag,dims", [("tiny", synth.TINY), ("vitb16", synth.VIT_B_16)])
def test_towers_match_reference_fixtures(golden_dir, tag, dims, request):
    """The raw visual tower vs the fixtures produced by the reference's slip classes and by HF CLIP (tiny: 17 tokens -> fp32
    attention + split pass; ViT-B/16: the split attention's x2 output)."""
    g = np.load(golden_dir / f"towers_{tag}.npz")
    sd = request.getfixturevalue(f"{tag}_state_dict")
    model = build_clip(sd, precision="fp32x3", device=DEV)
    plain = build_clip(sd, precision="fp32", device=DEV)
    video = torch.from_numpy(synth.make_video(int(g["n_clip"]), int(g["n_frames"]), dims, seed=int(g["seed"])))
    frames = video.reshape(-1, *video.shape[2:]).to(DEV)
    img = model.encode_image(frames)
    model.check_range()
    scale = max(1.0, float(np.abs(g["image_features_oracle"]).max()))
    for ref in ("oracle", "slip", "hf"):
        assert np.abs(img.cpu().numpy() - g[f"image_features_{ref}"]).max() < F32_TOL * scale, ref
    assert float((img - plain.encode_image(frames)).abs().max()) < 2e-6 * scale       # next to the fp32-MFMA path
    ids = torch.from_numpy(g["ids"]).to(DEV)
    assert torch.equal(model.encode_text(ids), plain.encode_text(ids))                 # (a call below 2048 token rows: the fp32 kernels)


def test_evaluate_goldens_in_three_product_mode(golden_dir, vitb16_state_dict):
    """`command=evaluate` end to end at ViT-B/16: embeddings, scores, ranks and metrics at the fp32 tolerances."""
    g = np.load(golden_dir / "evaluate_config1.npz")
    n, f = int(g["n_clips"]), int(g["n_frames"])
    video = torch.from_numpy(synth.make_video(n, f, synth.VIT_B_16, seed=42)).to(DEV)
    ids = torch.from_numpy(synth.make_text(n, synth.VIT_B_16, seed=42)).to(DEV)
    module = TextVideoRetrievalModule(ClipVideoTextEncoder(build_clip(vitb16_state_dict, precision="fp32x3", device=DEV)),
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
    """1100 frames: one 1024-frame pass + a 76-frame pass, against the fp32-MFMA path on the same frames, and a batch computed
    in pieces equals the batch computed at once (rows are independent)."""
    split = build_clip(vitb16_state_dict, precision="fp32x3", device=DEV, chunk_frames=1024)   # (the default pass holds 2048 frames)
    plain = build_clip(vitb16_state_dict, precision="fp32", device=DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    base = torch.randn(25, 3, 224, 224, device=DEV, generator=g).clamp_(-2.5, 2.5)
    frames = base[torch.arange(1100, device=DEV) % 25].contiguous()
    got = split.encode_image(frames)
    split.check_range()
    diff = float((got - plain.encode_image(frames)).abs().max()) / max(1.0, float(got.abs().max()))
    assert diff < 1e-5, diff
    assert torch.equal(got[:25], got[25:50]) and torch.equal(got[:25], got[1075:1100])  # same frame, same bits, any pass
    assert torch.equal(split.encode_image(frames[:7].contiguous()), got[:7])
    one_pass = build_clip(vitb16_state_dict, precision="fp32x3", device=DEV)
    assert torch.equal(one_pass.encode_image(frames), got)                               # ... and any pass size


def test_mode_is_as_close_to_float64_as_fp32_arithmetic_itself(vitb16_state_dict):
    """Distance to the TRUTH (the oracle evaluated in float64) of ViT-B/16 embeddings: the oracle in float32 (the reference's
    arithmetic on a CPU), the fp32-MFMA path, the six-product and the three-product split paths."""
    from oracle import clip_oracle as O
    d = synth.VIT_B_16
    sd32 = O.to_torch(vitb16_state_dict)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
    video = torch.from_numpy(synth.make_video(6, 2, d, seed=77))
    with torch.inference_mode():
        truth = O.encode_video(sd64, video.double())
        cpu32 = O.encode_video(sd32, video)
    err = {"oracle fp32": float((cpu32.double() - truth).abs().max())}
    for precision in ("fp32", "fp32x6", "fp32x3"):
        enc = ClipVideoTextEncoder(build_clip(vitb16_state_dict, precision=precision, device=DEV))
        err[precision] = float((enc.encode_video(video.to(DEV)).cpu().double() - truth).abs().max())
    print("max |embedding - float64 truth|:", err)
    assert err["fp32x3"] < 1e-6
    assert err["fp32x3"] <= 2.0 * max(err["oracle fp32"], err["fp32"]) + 5e-8, err


def test_heavy_tailed_weights_keep_the_fp32_accuracy(vitb16_state_dict):
    """Trained CLIP towers are not the benign synthetic init: a few residual channels carry activations in the hundreds, the
    LayerNorm gains of those channels are large, and some MLP units saturate.  The same shape of trouble, planted: outlier
    channels out of ln_pre and in out_proj / c_proj (a residual stream of several hundred), LayerNorm gains up to 12 and
    offsets up to 3, a c_fc bias that drives units far into both GELU tails (|pre-activation| to 17), Q and K rows scaled so
    that the softmax is peaked (largest probability of a row up to 0.98).  The per-tensor weight scales, the two planes and the
    range flag must hold: as close to float64 as the fp32 paths."""
    from oracle import clip_oracle as O
    d = synth.VIT_B_16
    rng = np.random.default_rng(11)
    sd = {k: np.array(v, copy=True) for k, v in vitb16_state_dict.items()}
    hot = rng.choice(d.vision_width, 6, replace=False)
    sd["visual.ln_pre.weight"][hot] *= 80.0                      # a residual stream of 250..370 in six channels (measured on the oracle)
    for layer in range(d.vision_layers):
        pre = f"visual.transformer.resblocks.{layer}."
        sd[pre + "attn.out_proj.weight"][hot[:3]] *= 25.0
        sd[pre + "mlp.c_proj.weight"][hot[3:]] *= 25.0
        for ln in ("ln_1", "ln_2"):
            gain = np.ones(d.vision_width, np.float32)
            gain[rng.choice(d.vision_width, 8, replace=False)] = rng.uniform(4.0, 12.0, 8).astype(np.float32)
            gain[hot] = 0.05                                       # (what training does to the outlier channels)
            sd[pre + ln + ".weight"] = sd[pre + ln + ".weight"] * gain
            sd[pre + ln + ".bias"] = sd[pre + ln + ".bias"] + rng.uniform(-3.0, 3.0, d.vision_width).astype(np.float32) * (gain > 1)
        sd[pre + "mlp.c_fc.bias"] = sd[pre + "mlp.c_fc.bias"] + rng.normal(0.0, 4.0, 4 * d.vision_width).astype(np.float32)
        sd[pre + "attn.in_proj_weight"][: 2 * d.vision_width] *= 3.0
    sd32 = O.to_torch(sd)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
    video = torch.from_numpy(synth.make_video(3, 2, d, seed=78))
    with torch.inference_mode():
        truth = O.encode_video(sd64, video.double())
        cpu32 = O.encode_video(sd32, video)
    assert torch.isfinite(truth).all()
    err = {"oracle fp32": float((cpu32.double() - truth).abs().max())}
    for precision in ("fp32", "fp32x3"):
        model = build_clip(sd, precision=precision, device=DEV)
        err[precision] = float((ClipVideoTextEncoder(model).encode_video(video.to(DEV)).cpu().double() - truth).abs().max())
        model.check_range()                                          # nothing left fp16's range
    print("heavy-tailed weights, max |embedding - float64 truth|:", err)
    assert err["fp32x3"] <= 2.0 * max(err["oracle fp32"], err["fp32"]) + 5e-8, err


def test_values_beyond_fp16_raise_instead_of_passing_silently(tiny_state_dict):
    """Never silently wrong: (1) LayerNorm weights whose outputs could leave fp16's range are refused when the weights are
    packed; (2) an activation that overflows at run time raises FC_ERANGE from `check_range()` and from the NEXT encode call."""
    d = synth.TINY
    frames = torch.from_numpy(synth.make_video(2, 2, d, seed=9)).reshape(-1, 3, d.image_resolution, d.image_resolution).to(DEV)
    model = build_clip(tiny_state_dict, precision="fp32x3", device=DEV)
    model.encode_image(frames)
    model.check_range()                                          # a sane model: nothing to report
    sd = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v.clone()) for k, v in tiny_state_dict.items()}
    key = "visual.transformer.resblocks.0.ln_2.weight"
    sd[key] = sd[key] * 1.0e4                                    # sqrt(256) * 1e4 > 65504
    bad = build_clip(sd, precision="fp32x3", device=DEV)
    bad.encode_image(frames)                                     # (the flag is raised at pack time, on the device)
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        bad.check_range()
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        bad.encode_image(frames)
    sd = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v.clone()) for k, v in tiny_state_dict.items()}
    key = "visual.transformer.resblocks.1.mlp.c_fc.bias"
    sd[key] = sd[key] + 1.0e5                                    # QuickGELU(1e5) = 1e5: beyond fp16 at run time
    hot = build_clip(sd, precision="fp32x3", device=DEV)
    out = hot.encode_image(frames)
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        hot.check_range()
    assert not bool(torch.isfinite(out).all())                   # and loud in the values themselves


def test_attention_operands_beyond_fp16_raise_too(vitb16_state_dict):
    """q, k, v enter fp16 planes inside the attention kernel (ViT-B/16: the fused three-product attention): a QKV bias that pushes
    them beyond 65504 raises FC_ERANGE as well."""
    d = synth.VIT_B_16
    frames = torch.from_numpy(synth.make_video(1, 2, d, seed=9)).reshape(-1, 3, d.image_resolution, d.image_resolution).to(DEV)
    ok = build_clip(vitb16_state_dict, precision="fp32x3", device=DEV)
    ok.encode_image(frames)
    ok.check_range()
    sd = {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v.clone()) for k, v in vitb16_state_dict.items()}
    key = "visual.transformer.resblocks.2.attn.in_proj_bias"
    sd[key] = sd[key] + 1.0e5
    hot = build_clip(sd, precision="fp32x3", device=DEV)
    hot.encode_image(frames)
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        hot.check_range()


def _copy_sd(sd):
    return {k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v.clone()) for k, v in sd.items()}


def _hot_sd(sd):
    """QuickGELU(1e5) = 1e5 in block 1: an activation no fp16 plane can hold, at run time."""
    hot = _copy_sd(sd)
    hot["visual.transformer.resblocks.1.mlp.c_fc.bias"] = hot["visual.transformer.resblocks.1.mlp.c_fc.bias"] + 1.0e5
    return hot


def test_nan_and_infinity_raise_the_flag_like_an_overflow(tiny_state_dict):
    """A running maximum drops NaN operands (`fmaxf`, `v_max3_f32`): the range tests look at bit patterns / at the planes, and every
    visual-tower call scans its embeddings - a NaN or infinite activation or weight is FC_ERANGE, not rc 0."""
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    x = torch.randn(64, 256, device=DEV)
    ops.split2(x, flag=flag)
    assert int(flag) == 0
    for poison in (float("nan"), float("inf"), -float("inf"), 7.0e4):
        y = x.clone()
        y[17, 33] = poison
        flag.zero_()
        ops.split2(y, flag=flag)
        assert int(flag) == 1, poison
    w = torch.randn(96, 128, device=DEV)
    flag.zero_()
    _, sc = ops.split2_weight(w, flag=flag)
    assert int(flag) == 0 and sc[0] * sc[1] == 1
    for poison in (float("nan"), float("inf")):
        v = w.clone()
        v[5, 7] = poison
        flag.zero_()
        w2, sc = ops.split2_weight(v, flag=flag)
        assert int(flag) == 1 and sc.tolist() == [1.0, 1.0], poison        # no scale can be chosen: 1, and the flag
        assert not bool(torch.isfinite(ops.x2_planes(w2)[0].float()).all())  # the planes carry it: loud in every product
    # the QuickGELU epilogue tests the planes it writes: a NaN pre-activation (NaN bias) raises the flag as an overflow does
    a2 = ops.split2(torch.randn(300, 256, device=DEV))
    w2, sc = ops.split2_weight(torch.randn(128, 256, device=DEV) / 16)
    for poison, want in ((float("nan"), 1), (1.0e5, 1), (0.0, 0)):
        bias = torch.zeros(128, device=DEV)
        bias[3] = poison
        flag.zero_()
        ops.gemm_split2(a2, w2, sc, bias, ops.EPI_GELU_X2, flag=flag)
        assert int(flag) == want, poison
    d = synth.TINY
    frames = torch.from_numpy(synth.make_video(2, 2, d, seed=9)).reshape(-1, 3, d.image_resolution, d.image_resolution).to(DEV)
    # a NaN weight of a block GEMM (flag at pack time), of the fp32 part of the tower and a NaN pixel (flag from the output scan)
    for key in ("visual.transformer.resblocks.0.mlp.c_fc.weight", "visual.conv1.weight", "visual.ln_post.weight", None):
        sd = _copy_sd(tiny_state_dict)
        f = frames.clone()
        if key is None:
            f[1, 2, 5, 5] = float("nan")
        else:
            t = torch.as_tensor(sd[key]).clone()
            t.view(-1)[3] = float("nan")
            sd[key] = t
        model = build_clip(sd, precision="fp32x3", device=DEV)
        model.encode_image(f)
        with pytest.raises(_lib.FitclipHipError, match="fp16"):
            model.check_range()


def test_strict_range_makes_every_call_answer_for_itself(tiny_state_dict):
    """`strict_range=True` (fc_range_strict): the call whose activations left fp16's range raises FC_ERANGE ITSELF."""
    d = synth.TINY
    frames = torch.from_numpy(synth.make_video(2, 2, d, seed=9)).reshape(-1, 3, d.image_resolution, d.image_resolution).to(DEV)
    ok = build_clip(tiny_state_dict, precision="fp32x3", device=DEV, strict_range=True)
    lazy = build_clip(tiny_state_dict, precision="fp32x3", device=DEV)
    assert torch.equal(ok.encode_image(frames), lazy.encode_image(frames))
    hot = build_clip(_hot_sd(tiny_state_dict), precision="fp32x3", device=DEV, strict_range=True)
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        hot.encode_image(frames)
    deferred = build_clip(_hot_sd(tiny_state_dict), precision="fp32x3", device=DEV)
    deferred.encode_image(frames)                                  # default: rc 0 here ...
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        deferred.check_range()                                     # ... and the consumer's check raises


def test_text_tower_joins_the_mode_for_large_calls(vitb16_state_dict):
    """fc_encode_text in fp32x3: a call of >= 2048 token rows (27 captions of 77 tokens) runs the text blocks' four GEMMs on the
    three-product kernel, a smaller call the fp32 kernels (a launch of the plane kernel is one tile's latency there).  One arithmetic per CALL: a small call has
    the fp32 path's bits, a large one its values at fp32 accuracy, and rows depend neither on the batch nor on the pass size."""
    from oracle import clip_oracle as O
    d = synth.VIT_B_16
    split = build_clip(vitb16_state_dict, precision="fp32x3", device=DEV)
    plain = build_clip(vitb16_state_dict, precision="fp32", device=DEV)
    ids = torch.from_numpy(synth.make_text(300, d, seed=5)).to(DEV)
    assert torch.equal(split.encode_text(ids[:26]), plain.encode_text(ids[:26]))        # 2002 rows: the fp32 kernels
    big, ref = split.encode_text(ids), plain.encode_text(ids)
    split.check_range()
    scale = max(1.0, float(ref.abs().max()))
    assert float((big - ref).abs().max()) < 2e-6 * scale
    assert not torch.equal(big, ref)                                                     # (it IS the other arithmetic)
    assert torch.equal(split.encode_text(ids[:27]), big[:27])                            # 2079 rows: same bits in any large batch
    passes = build_clip(vitb16_state_dict, precision="fp32x3", device=DEV, chunk_texts=128)
    assert torch.equal(passes.encode_text(ids), big)                                     # passes of 128, 128, 44 captions
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in O.to_torch(vitb16_state_dict).items()}
    with torch.inference_mode():
        truth = O.encode_text_tokens(sd64, ids[:54].cpu())
    err3 = float((big[:54].cpu().double() - truth).abs().max())
    err1 = float((ref[:54].cpu().double() - truth).abs().max())
    print(f"text tower, max |features - float64 truth|: fp32x3 {err3:.3e}, fp32 {err1:.3e}")
    assert err3 < 2e-6 * scale and err3 <= 2.0 * err1 + 1e-7 * scale


def test_text_tower_answers_for_fp16s_range_when_it_uses_it(vitb16_state_dict):
    """A text block whose QuickGELU output no fp16 plane can hold: a small call (fp32 kernels) is unaffected, a large call raises the flag,
    with `strict_range` from the call itself."""
    d = synth.VIT_B_16
    hot_sd = _copy_sd(vitb16_state_dict)
    key = "transformer.resblocks.1.mlp.c_fc.bias"
    hot_sd[key] = hot_sd[key] + 1.0e5
    ids = torch.from_numpy(synth.make_text(64, d, seed=6)).to(DEV)
    hot = build_clip(hot_sd, precision="fp32x3", device=DEV)
    plain = build_clip(hot_sd, precision="fp32", device=DEV)
    assert torch.equal(hot.encode_text(ids[:8]), plain.encode_text(ids[:8]))
    hot.check_range()                                              # nothing entered an fp16 plane
    hot.encode_text(ids)                                           # default: rc 0 here ...
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        hot.check_range()                                          # ... and the consumer's check raises
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        hot.encode_text(ids)                                       # so does the next call that would use the planes
    strict = build_clip(hot_sd, precision="fp32x3", device=DEV, strict_range=True)
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        strict.encode_text(ids)


def test_predict_saves_nothing_when_the_last_batch_left_fp16s_range(tmp_path, tiny_state_dict, capsys):
    """`command=predict` (aligner/__main__.py:70-91) writes its file only after the range flag of the LAST batch has been seen."""
    import yaml
    from fitclip_amd.__main__ import main

    def config(name, sd):
        torch.save({k: torch.as_tensor(v) for k, v in sd.items()}, tmp_path / f"{name}.pt")
        path = tmp_path / f"{name}.yaml"
        path.write_text(yaml.safe_dump({"_target_": "fitclip_amd.encoder.ClipVideoTextEncoder", "num_frames": 2, "bpe_path": None,
                                        "model": {"_target_": "fitclip_amd.clip_model.load_clip_model", "name": str(tmp_path / f"{name}.pt"),
                                                  "precision": "fp32x3"}}))
        return path

    out = tmp_path / "ok.out.pt"
    main(["command=predict", f"encoder={config('ok', tiny_state_dict)}", "n_clips=3", "num_frames=2", "eval_batch_size=2", f"output_path={out}"])
    capsys.readouterr()
    assert torch.load(out)["encoded_videos"].shape == (3, synth.TINY.embed_dim)
    bad = tmp_path / "hot.out.pt"
    for extra in ([], ["strict_range=true"]):
        with pytest.raises(_lib.FitclipHipError, match="fp16"):
            main(["command=predict", f"encoder={config('hot', _hot_sd(tiny_state_dict))}", "n_clips=3", "num_frames=2", "eval_batch_size=2",
                  f"output_path={bad}", *extra])
        assert not bad.exists()


def test_classification_reports_no_accuracy_from_out_of_range_embeddings(tiny_state_dict):
    """`VideoTextClassificationModule.validation_epoch_end` / `predict_step` (aligner/video_text_classification.py:86-118)."""
    from fitclip_amd.classification import VideoTextClassificationModule
    d = synth.TINY
    labels, video = ["cat", "dog", "guitar"], torch.from_numpy(synth.make_video(4, 2, d, seed=4)).to(DEV)
    target = (None, torch.arange(4) % 3)
    ok = VideoTextClassificationModule(ClipVideoTextEncoder(build_clip(tiny_state_dict, precision="fp32x3", device=DEV)), labels)
    ok.validation_step({"video": video, "target": target})
    assert set(ok.validation_epoch_end()) == {"a1", "a5", "mr"}
    hot = VideoTextClassificationModule(ClipVideoTextEncoder(build_clip(_hot_sd(tiny_state_dict), precision="fp32x3", device=DEV)), labels)
    hot.validation_step({"video": video, "target": target})
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        hot.validation_epoch_end()
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        hot.predict_step({"video": video, "target": target})


def test_no_gradient_step_from_an_out_of_range_fp32x3_teacher(tiny_state_dict):
    """`TeacherStudentTrainer.fit_step` with a frozen teacher in precision fp32x3 (teacher_student.py:93-96): FC_ERANGE before the
    backward, the student's parameters untouched."""
    from fitclip_amd.training import TeacherStudentTrainer
    d = synth.TINY
    student_np = synth.perturbed_state_dict(tiny_state_dict, d, seed=5, rel=0.3)
    video, ids = torch.from_numpy(synth.make_video(8, 2, d, seed=30)).to(DEV), torch.from_numpy(synth.make_text(8, d, seed=30)).to(DEV)
    batch = {"video_student": video, "text_student": {"input_ids": ids}, "video_teacher": video, "text_teacher": {"input_ids": ids},
             "dataset": ["labeled"] * 4 + ["unlabeled"] * 4}

    def trainer(teacher_sd):
        student = ClipVideoTextEncoder(build_clip(student_np, precision="fp32", device=DEV))
        teacher = ClipVideoTextEncoder(build_clip(teacher_sd, precision="fp32x3", device=DEV))
        return TeacherStudentTrainer(student, teacher, init_temperature=0.05, lr=1e-4)

    ok = trainer(tiny_state_dict)
    assert np.isfinite(ok.fit_step(dict(batch)))
    hot = trainer(_hot_sd(tiny_state_dict))
    before = {k: v.clone() for k, v in hot.student.model.state_dict().items()}
    with pytest.raises(_lib.FitclipHipError, match="fp16"):
        hot.fit_step(dict(batch))
    assert all(torch.equal(v, before[k]) for k, v in hot.student.model.state_dict().items())


def test_gemm_split2_race_screen():
    """Counted vmcnt waits, one barrier per K-step, two LDS stages, pieces spread over MFMA groups, half-pass output patches: a
    misplaced wait shows as rare wrong tiles.  20 rounds of a multi-round shape (ragged last panel, all three epilogues) must
    reproduce the first result bitwise, with another GEMM's traffic in between."""
    g = torch.Generator(device=DEV).manual_seed(7)
    M, N, K = 256 * 105 + 77, 3072, 768
    a2 = ops.split2(torch.randn(M, K, device=DEV, generator=g))
    w2, sc = ops.split2_weight(torch.randn(N, K, device=DEV, generator=g) / K ** 0.5)
    bias = torch.randn(N, device=DEV, generator=g)
    other_a = ops.split2(torch.randn(30000, 3072, device=DEV, generator=g))
    other_w, other_s = ops.split2_weight(torch.randn(768, 3072, device=DEV, generator=g) / 3072 ** 0.5)
    other_b = torch.zeros(768, device=DEV)
    first = {epi: ops.gemm_split2(a2, w2, sc, bias, epi) for epi in (ops.EPI_BIAS_F32, ops.EPI_GELU_X2)}
    x0 = torch.randn(30000, 768, device=DEV, generator=g)
    first_other = x0.clone()
    ops.gemm_split2(other_a, other_w, other_s, other_b, ops.EPI_RESID3_F32, out=first_other)
    for i in range(20):
        for epi in (ops.EPI_BIAS_F32, ops.EPI_GELU_X2):
            assert torch.equal(ops.gemm_split2(a2, w2, sc, bias, epi), first[epi]), (i, epi)
        again = x0.clone()
        ops.gemm_split2(other_a, other_w, other_s, other_b, ops.EPI_RESID3_F32, out=again)
        assert torch.equal(again, first_other), i

"""GPU parity of the single HIP operators, called through the C ABI, against fp64 / oracle references on the same
seeded inputs.  Tolerances are stated per test: the fp32 path is an exact-fp32 fma chain (summation-order noise only),
the bf16 path rounds operands and stored activations to bf16 (relative 2^-9 per rounding)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fitclip_amd import ops  # noqa: E402
from oracle import clip_oracle as O  # noqa: E402

DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


GEMM_SHAPES = [(300, 256, 256, 1), (1000, 512, 768, 2), (37, 128, 3072, 1), (515, 768, 128, 2), (16, 132, 64, 1),
               (5000, 768, 256, 3), (700, 264, 768, 3), (70000, 512, 192, 3)]


@pytest.mark.parametrize("M,N,K,tile", GEMM_SHAPES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_all_epilogues(M, N, K, tile, dtype):
    a, w, bias = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3)
    ad, wd = a.to(DEV).to(dtype), w.to(DEV).to(dtype)
    a64, w64 = ad.double().cpu(), wd.double().cpu()  # the operands the kernel actually sees
    ref = a64 @ w64.T
    tol = 5e-6 if dtype == torch.float32 else 6e-3   # f32: summation-order noise over K <= 3072; bf16: one output rounding (2^-9)
    bd = bias.to(DEV)
    out = ops.gemm(ad, wd, bd, ops.EPI_BIAS_T, tile=tile)
    assert out.dtype == dtype and _rel(out, ref + bias.double()) < tol
    out = ops.gemm(ad, wd, bd, ops.EPI_GELU_T, tile=tile)
    z = ref + bias.double()
    assert _rel(out, z * torch.sigmoid(1.702 * z)) < tol
    resid = _rand(M, N, seed=4)
    if tile == 3:  # the persistent / pipelined kernel has the two store epilogues and the residual update
        # C(f32) += acc + bias in place: the bits of the plain kernels (a row does not depend on the kernel the batch
        # selects); in fp32 also the bits of the bias epilogue followed by the fp32 add (what the fused add+LayerNorm did)
        acc = resid.to(DEV).clone()
        ops.gemm(ad, wd, bd, ops.EPI_RESID_F32, out=acc, tile=3)
        assert _rel(acc, resid.double() + z) < 2e-6
        if dtype == torch.float32:
            assert torch.equal(acc, resid.to(DEV) + ops.gemm(ad, wd, bd, ops.EPI_BIAS_T, tile=3))
        for other in (1, 2) + ((8,) if dtype == torch.float32 else ()):  # (8: 64 x 64 tiles on a four-stage ring, fp32)
            acc2 = resid.to(DEV).clone()
            ops.gemm(ad, wd, bd, ops.EPI_RESID_F32, out=acc2, tile=other)
            assert torch.equal(acc2, acc)
        return
    acc = resid.to(DEV).clone()
    ops.gemm(ad, wd, bd, ops.EPI_RESID_F32, out=acc, tile=tile)
    assert _rel(acc, resid.double() + z) < (2e-6 if dtype == torch.float32 else 2e-6 + 0)
    out = ops.gemm(ad, wd, None, ops.EPI_STORE_F32, alpha=0.5, tile=tile)
    assert out.dtype == torch.float32 and _rel(out, 0.5 * ref) < 2e-6
    if M % 4 == 0:
        P = M // 4
        pos = _rand(P + 1, N, seed=5).to(DEV)
        out = ops.gemm(ad, wd, None, ops.EPI_PATCH_F32, aux=pos, patches=P, tile=tile,
                       out=torch.full((M + 4, N), 7.0, device=DEV))
        want = torch.full((4, P + 1, N), 7.0, dtype=torch.float64)
        want[:, 1:] = ref.view(4, P, N) + pos.double().cpu()[1:]
        assert _rel(out.view(4, P + 1, N), want) < 2e-6


# the reference's eval batch (32 clips x 4 frames = 128 frames = 25 216 token rows) at the three output widths of a block: on
# 256 compute units the planner picks a tail of 1, 2 and 3 units for them; 25 179 rows end inside a 64-row unit
@pytest.mark.parametrize("M,N,K", [(25216, 768, 768), (25216, 2304, 768), (25216, 3072, 768), (25179, 768, 3072),
                                   (900, 768, 256), (70000, 512, 192)])
def test_pipelined_gemm_row_cut_is_bit_invisible(M, N, K):
    """The persistent fp32 GEMM cuts its rows into a head of whole rounds of 256-row tiles and a tail of 64 / 128 / 192-row
    tiles (csrc/gemm_kernel.h, HT): every forced cut, the planned one and the one-tile-per-workgroup kernels must produce the
    SAME BITS for all three block epilogues - an output element sees its K-tiles and its MFMA chain in the same order whatever
    tile it falls into."""
    a, w, bias = _rand(M, K, seed=11).to(DEV), _rand(N, K, seed=12, scale=K ** -0.5).to(DEV), _rand(N, seed=13).to(DEV)
    resid = _rand(M, N, seed=14).to(DEV)
    hp, ht = ops.gemm_plan(M, N, K)
    assert 0 <= hp <= (M + 255) // 256 and 0 <= ht <= 3
    if torch.cuda.get_device_properties(0).multi_processor_count == 256 and M == 25216:
        assert (hp, ht) == {768: (85, 1), 2304: (85, 2), 3072: (85, 3)}[N]  # (K = 768)
    for epi in (ops.EPI_BIAS_T, ops.EPI_GELU_T, ops.EPI_RESID_F32):
        def run(tile):
            out = resid.clone() if epi == ops.EPI_RESID_F32 else None
            return ops.gemm(a, w, bias, epi, out=out, tile=tile)
        want = run(2)
        if M <= 1000:  # (small enough for float64 on the host: the plain kernel itself against the definition)
            z = a.double().cpu() @ w.double().cpu().T + bias.double().cpu()
            ref = {ops.EPI_BIAS_T: z, ops.EPI_GELU_T: z * torch.sigmoid(1.702 * z), ops.EPI_RESID_F32: resid.double().cpu() + z}[epi]
            assert _rel(want, ref) < 5e-6
        for tile in (3, 4, 5, 6, 7, 1, 8):
            assert torch.equal(run(tile), want), (epi, tile)


def test_gemm_rejects_bad_arguments():
    from fitclip_amd._lib import FitclipHipError
    a, w = torch.zeros(8, 48, device=DEV), torch.zeros(8, 48, device=DEV)
    with pytest.raises(FitclipHipError, match="multiple of 32"):
        ops.gemm(a, w, None, ops.EPI_STORE_F32)
    with pytest.raises(FitclipHipError, match="bias"):
        ops.gemm(torch.zeros(8, 64, device=DEV), torch.zeros(8, 64, device=DEV), None, ops.EPI_BIAS_T)


@pytest.mark.parametrize("D", [128, 256, 512, 768])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_layernorm(D, out_dtype):
    x = _rand(333, D, seed=D) * 3 + 0.7
    g, b = _rand(D, seed=1) * 0.1 + 1, _rand(D, seed=2) * 0.1
    ref = O.layer_norm(x, g, b).double()
    y = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), out_dtype)
    assert _rel(y, ref) < (3e-6 if out_dtype == torch.float32 else 5e-3)
    # strided rows (ln_post on the CLS rows) and gathered rows (ln_final on the EOT rows)
    xs = x[:330].reshape(110, 3 * D)
    y = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), torch.float32, row_stride=3 * D, rows=110)
    assert _rel(y, O.layer_norm(xs[:, :D], g, b)) < 3e-6
    idx = torch.tensor([5, 0, 332, 17, 17], dtype=torch.int32)
    y = ops.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), torch.float32, gather=idx.to(DEV))
    assert _rel(y, ref[idx.long()]) < 3e-6


@pytest.mark.parametrize("D", [128, 512, 768])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_add_layernorm(D, dtype):
    x, d = _rand(257, D, seed=D) * 2 + 0.3, _rand(257, D, seed=D + 1)
    g, b = _rand(D, seed=1) * 0.1 + 1, _rand(D, seed=2) * 0.1
    dd = d.to(DEV).to(dtype)
    want_x = x.double() + dd.double().cpu()
    ref = O.layer_norm(want_x.float(), g, b).double()
    xd = x.to(DEV).clone()
    y = ops.add_layernorm(xd, dd, g.to(DEV), b.to(DEV))
    assert y.dtype == dtype and _rel(xd, want_x) < 1e-6
    assert _rel(y, ref) < (3e-6 if dtype == torch.float32 else 5e-3)
    xd2 = x.to(DEV).clone()
    y2 = ops.add_layernorm(xd2, dd, g.to(DEV), b.to(DEV), write_x=False)
    assert torch.equal(xd2.cpu(), x) and torch.equal(y2, y)


def _attention_ref(qkv, n, S, heads, causal):
    D = heads * 64
    q, k, v = qkv.double().view(n, S, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = (q * 0.125) @ k.transpose(-1, -2)
    if causal:
        s = s + O.causal_mask(S).double()
    return (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(n * S, D)


@pytest.mark.parametrize("n,S,heads,causal", [(3, 197, 12, False), (5, 77, 8, True), (2, 17, 4, False),
                                              (4, 16, 2, True), (1, 224, 1, False), (2, 33, 2, True),
                                              # > 224 tokens: the streaming kernel (ViT-L/14: 257, @336px: 577) and
                                              # tile-boundary cases of its 64-key ring / 16-query tiles
                                              (2, 257, 16, False), (1, 577, 3, False), (2, 225, 1, False),
                                              (1, 256, 2, False), (1, 320, 1, False), (1, 321, 2, False),
                                              # the streaming-block kernel of 97..224 tokens (log2-domain softmax, round 5) at the ends of
                                              # its range, at whole and ragged key blocks of 64 and with the last block almost empty
                                              (2, 97, 2, False), (3, 113, 3, False), (2, 128, 12, False), (2, 129, 4, False),
                                              (2, 192, 2, False), (3, 193, 12, False), (2, 208, 1, False), (2, 209, 3, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention(n, S, heads, causal, dtype):
    qkv = _rand(n * S, 3 * heads * 64, seed=S)
    qkv[:, :heads * 64] *= 2.0  # sharper softmax
    qd = qkv.to(DEV).to(dtype)
    ref = _attention_ref(qd.cpu(), n, S, heads, causal)
    out = ops.attention(qd, n, S, heads, causal)
    assert out.dtype == dtype
    assert _rel(out, ref) < (3e-6 if dtype == torch.float32 else 1.5e-2)


def test_similarity_ranks_and_losses():
    t, v = _rand(50, 512, seed=1), _rand(70, 512, seed=2)
    s = ops.similarity(t.to(DEV), v.to(DEV), alpha=1.0)
    assert s.shape == (50, 70) and _rel(s, t.double() @ v.double().T) < 2e-6
    s2 = ops.similarity(t.to(DEV), v[:49].to(DEV), alpha=2.5)  # column count not a multiple of 4
    assert s2.shape == (50, 49) and _rel(s2, 2.5 * (t.double() @ v[:49].double().T)) < 2e-6
    # ranks with ties: same known answers as the oracle test
    m = torch.tensor([[9., 1., 2., 3., 4.], [5., 5., 5., 5., 5.], [1., 2., 0., 4., 3.], [7., 3., 7., 7., 1.],
                      [0., 0., 0., 0., 1.]])
    assert ops.ranks(m.to(DEV)).tolist() == [0, 1, 4, 2, 0]
    big = _rand(300, 1000, seed=3)
    big[:, ::7] = big[:, 3:4]  # plant ties
    want = O.ranks_of_target(big[:, 100:], torch.arange(300) + 0)  # offset handled below
    got = ops.ranks(big.to(DEV)[:, 100:], 0)
    assert got.tolist() == want.tolist()
    got = ops.ranks(big.to(DEV), 100)
    assert got.tolist() == O.ranks_of_target(big, torch.arange(300) + 100).tolist()
    for n, scale in ((8, 1.0), (64, 30.0), (301, 66.7)):
        a, b = _rand(n, n, seed=n) * scale, _rand(n, n, seed=n + 1) * scale
        assert abs(float(ops.nce_loss(a.to(DEV))) - float(O.nce_loss(a.double()))) < 1e-4 * max(1, scale)
        kd, ref = float(ops.teacher_student_nce_loss(a.to(DEV), b.to(DEV))), float(O.teacher_student_nce_loss(a.double(), b.double()))
        assert abs(kd - ref) < 2e-5 * max(1.0, abs(ref))


@pytest.mark.parametrize("nt,nv,dim,offset", [(5, 5, 32, 0), (300, 1000, 512, 100), (129, 257, 64, 7), (1000, 1000, 512, 0),
                                              (64, 4099, 512, 2000)])
def test_similarity_ranks_equal_the_materialised_path(nt, nv, dim, offset):
    """fc_similarity_ranks (ranks from the scoring GEMM's epilogue, no [nt, nv] matrix) against fc_similarity + fc_ranks /
    fc_ranks_of: IDENTICAL ranks, with exact ties - duplicated video rows, and embeddings of small integers whose scores are
    exact in fp32 - deciding many positions (metrics.py:16-20: stable order, a tie goes to the lower column)."""
    g = torch.Generator().manual_seed(nt * 7 + nv)
    t = torch.randn(nt, dim, generator=g)
    v = torch.randn(nv, dim, generator=g)
    v[::3] = v[1:2]                                  # every third video is the SAME row: exact ties in every text row
    ti, vi = torch.randint(-2, 3, (nt, dim), generator=g).float(), torch.randint(-2, 3, (nv, dim), generator=g).float()
    for tt, vv, alpha in ((t, v, 1.0), (ti, vi, 1.0), (t, v, 66.666)):
        td, vd = tt.to(DEV), vv.to(DEV)
        scores = ops.similarity(td, vd, alpha=alpha)
        assert ops.similarity_ranks(td, vd, offset, alpha=alpha).tolist() == ops.ranks(scores, offset).tolist()
        tgt = torch.randint(0, nv, (nt,), generator=g)
        assert ops.similarity_ranks(td, vd, targets=tgt, alpha=alpha).tolist() == ops.ranks_of(scores, tgt).tolist()
    # and against the oracle on the integer case (scores exact in fp32: no arithmetic to argue about)
    want = O.ranks_of_target(ti @ vi.T, torch.arange(nt) + offset)
    assert ops.similarity_ranks(ti.to(DEV), vi.to(DEV), offset).tolist() == want.tolist()


def test_row_cut_on_random_shapes():
    """Forty random problems (M up to 60 000 rows, N a multiple of 8 up to 3072, K a multiple of 32 from the pipelined kernel's
    minimum up to 1024; seeded): whatever head / tail the planner picks (`ops.gemm_plan`), and under every forced tail height,
    the persistent kernel's rows equal the plain 128 x 128 kernel's bit for bit - residual epilogue (read-modify-write: a row
    touched twice or not at all would show)."""
    rng = np.random.default_rng(7)
    seen = set()
    for _ in range(40):
        M = int(rng.integers(1, 60000))
        N = int(rng.integers(1, 385)) * 8
        K = int(rng.integers(3, 33)) * 32
        a, w, bias = _rand(M, K, seed=M).to(DEV), _rand(N, K, seed=N, scale=K ** -0.5).to(DEV), _rand(N, seed=K).to(DEV)
        resid = _rand(M, N, seed=1).to(DEV)
        want = ops.gemm(a, w, bias, ops.EPI_RESID_F32, out=resid.clone(), tile=1)
        hp, ht = ops.gemm_plan(M, N, K)
        seen.add(ht)
        for tile in (3, 5 + int(rng.integers(0, 3))):
            got = ops.gemm(a, w, bias, ops.EPI_RESID_F32, out=resid.clone(), tile=tile)
            assert torch.equal(got, want), (M, N, K, tile, hp, ht)
    assert len(seen) >= 3, seen   # the planner used several cuts over the sample


def test_new_gemm_paths_at_the_edges():
    """Ragged and tiny problems through every fp32 kernel of the block epilogues: fewer rows than one 64-row unit, rows that end
    inside a unit, one column tile, K at the pipelined kernel's minimum (3 K-tiles) - forced row cuts (tiles 4..7), the 64 x 64
    ring (8) and the planned cut (3) against the plain 128 x 128 kernel, bitwise; and the scoring epilogue on empty / one-row /
    one-column inputs."""
    for M, N, K in ((1, 256, 96), (63, 256, 96), (65, 512, 128), (257, 264, 96), (700, 768, 2048)):
        a, w, bias = _rand(M, K, seed=M).to(DEV), _rand(N, K, seed=N, scale=K ** -0.5).to(DEV), _rand(N, seed=3).to(DEV)
        resid = _rand(M, N, seed=4).to(DEV)
        for epi in (ops.EPI_BIAS_T, ops.EPI_GELU_T, ops.EPI_RESID_F32):
            def run(tile):
                return ops.gemm(a, w, bias, epi, out=resid.clone() if epi == ops.EPI_RESID_F32 else None, tile=tile)
            want = run(1)
            for tile in (3, 4, 5, 6, 7, 8, 2):
                assert torch.equal(run(tile), want), (M, N, K, epi, tile)
    t, v = _rand(5, 64, seed=1).to(DEV), _rand(9, 64, seed=2).to(DEV)
    assert ops.similarity_ranks(t[:0], v).shape == (0,)
    assert ops.similarity_ranks(t[:1], v[:1]).tolist() == [0]
    assert ops.similarity_ranks(t, v[:1], targets=torch.zeros(5, dtype=torch.int32)).tolist() == [0] * 5   # one column
    assert ops.similarity_ranks(t, v, 4).tolist() == ops.ranks(ops.similarity(t, v), 4).tolist()          # the last valid offset
    from fitclip_amd._lib import FitclipHipError
    with pytest.raises(FitclipHipError, match="offset"):   # targets past the last column: rejected, as fc_ranks does
        ops.similarity_ranks(t, v, 5)


def test_loss_matches_reference_fixture(golden_dir):
    g = np.load(golden_dir / "loss_ref.npz")
    for t in sorted({k.rsplit("_", 1)[0] for k in g.files}):
        s, te = torch.from_numpy(g[f"{t}_scores"]).to(DEV), torch.from_numpy(g[f"{t}_teacher"]).to(DEV)
        assert abs(float(ops.nce_loss(s)) - float(g[f"{t}_nce"])) < 1e-4 * max(1.0, abs(float(g[f"{t}_nce"])))
        assert abs(float(ops.teacher_student_nce_loss(s, te)) - float(g[f"{t}_kd"])) < 1e-4 * max(1.0, abs(float(g[f"{t}_kd"])))


def test_wise_is_bit_exact_against_reference_fixture(golden_dir):
    g = np.load(golden_dir / "wise_ref.npz")
    names = sorted(k[3:] for k in g.files if k.startswith("m1_"))
    for w in (0.0, 0.4, 0.5, 1.0):
        for n in names:
            a, b = torch.from_numpy(g[f"m1_{n}"]).to(DEV), torch.from_numpy(g[f"m2_{n}"]).to(DEV)
            assert np.array_equal(ops.wise_axpby(a, b, w).cpu().numpy(), g[f"w{w}_{n}"]), (w, n)
    a, b = _rand(1_000_003, seed=1).to(DEV), _rand(1_000_003, seed=2).to(DEV)  # ragged tail
    assert torch.equal(ops.wise_axpby(a, b, 0.4).cpu(), ((1 - 0.4) * a + 0.4 * b).cpu())


def test_pool_and_normalize():
    e = _rand(6 * 8, 512, seed=9) + 0.3
    ref = (e / e.norm(dim=-1, keepdim=True)).view(6, 8, 512).mean(1)
    assert _rel(ops.pool_normalize(e.to(DEV), 6, 8), ref) < 2e-6
    assert _rel(ops.l2_normalize(e.to(DEV)), e / e.norm(dim=-1, keepdim=True)) < 2e-6
    assert ops.pool_normalize(torch.zeros(0, 512, device=DEV), 0, 8).shape == (0, 512)


def test_ranks_of_and_group_mean():
    s = _rand(200, 37, seed=5)
    s[:, ::5] = s[:, 2:3]  # ties
    tgt = torch.randint(0, 37, (200,), generator=torch.Generator().manual_seed(0))
    assert ops.ranks_of(s.to(DEV), tgt).cpu().tolist() == O.ranks_of_target(s, tgt).tolist()
    x = _rand(7 * 5, 128, seed=6)
    assert torch.equal(ops.group_mean(x.to(DEV), 5).cpu(), O.zero_shot_label_embeddings(x, 5))
    assert torch.equal(ops.group_mean(x.to(DEV), 1).cpu(), x)


def test_quickgelu_is_accurate_per_element():
    """The fp32 QuickGELU of the GEMM epilogues / training kernels (csrc/common.h: exp2 with a compensated argument +
    v_rcp) against float64, ELEMENT-wise relative error over the whole useful range: better than 4e-7 (the plain
    float32 expression x / (1 + expf(-1.702f x)) is at 2.7e-6)."""
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.rand(100000, generator=g) * 120 - 60, torch.randn(100000, generator=g), torch.tensor([0.0, -0.0, 1e-30, -80.0, 80.0])])
    x = x[: x.numel() // 4 * 4]
    a = torch.zeros(x.numel(), 32)
    a[:, 0] = x
    w = torch.zeros(4, 32)
    w[:, 0] = 1.0
    out = ops.gemm(a.to(DEV), w.to(DEV), torch.zeros(4, device=DEV), ops.EPI_GELU_T)[:, 0].cpu().double()
    ref = x.double() * torch.sigmoid(1.702 * x.double())
    assert torch.isfinite(out).all()
    rel = (out - ref).abs() / ref.abs().clamp_min(1e-30)
    big = ref.abs() > 1e-30
    assert float(rel[big].max()) < 4e-7, float(rel[big].max())
    assert float((out - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("rows,D,vocab,pad_from", [(77, 64, 40, None), (1000, 512, 300, 5), (64 * 77, 512, 49408, 12),
                                                  (3 * 1024 + 5, 128, 7, None), (1, 64, 3, None)])
def test_token_embedding_backward_is_a_fixed_order_segmented_sum(rows, D, vocab, pad_from):
    """Embedding backward of `token_embedding(text)` (slip.py:469): BITWISE the sequential float32 sum of the gradient rows of
    each id in row order (numpy's unbuffered `np.add.at` is exactly that), for collision-heavy id patterns (SOT / EOT / pad
    columns shared by every caption, segments longer than one 1024-row chunk), ids outside the table clamped like the
    forward gather, `accumulate`, and two launches bit-identical."""
    rng = np.random.default_rng(rows + vocab)
    ids = rng.integers(0, vocab, size=rows, dtype=np.int64)
    if pad_from is not None:  # caption shape: [SOT, tokens..., EOT, pad...] per 77-row sequence
        pos = np.arange(rows) % 77
        ids[pos == 0] = vocab - 2
        ids[pos == pad_from] = vocab - 1
        ids[pos > pad_from] = 0
    if rows > 10:
        ids[3], ids[7] = -5, vocab + 11  # clamped to 0 and vocab - 1
    g = (rng.standard_normal((rows, D)) * np.exp(rng.uniform(-6, 6, size=(rows, 1)))).astype(np.float32)
    want = np.zeros((vocab, D), np.float32)
    np.add.at(want, np.clip(ids, 0, vocab - 1), g)
    ids_d, g_d = torch.from_numpy(ids).to(DEV), torch.from_numpy(g).to(DEV)
    got = ops.token_embedding_backward(ids_d, g_d, vocab)
    assert np.array_equal(got.cpu().numpy(), want)
    assert torch.equal(ops.token_embedding_backward(ids_d, g_d, vocab), got)
    base = torch.from_numpy(rng.standard_normal((vocab, D)).astype(np.float32)).to(DEV)
    acc = ops.token_embedding_backward(ids_d, g_d, vocab, out=base.clone(), accumulate=True)
    touched = torch.from_numpy(np.bincount(np.clip(ids, 0, vocab - 1), minlength=vocab) > 0).to(DEV)
    assert torch.equal(acc[~touched], base[~touched])
    assert torch.equal(acc[touched], base[touched] + got[touched])

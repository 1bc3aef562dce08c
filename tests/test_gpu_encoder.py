"""GPU parity of the whole encode-and-score path (plugin API -> C ABI -> HIP kernels) against the committed golden
fixtures (which the CPU oracle produced and which were pinned to the reference's own classes) and against the oracle
run on the same seeded inputs.

Tolerances:
  * fp32 path (exact-fp32 MFMA): embeddings max|d| <= 2e-5 (unit-norm vectors), scores <= 5e-5, identical ranks.
  * bf16 path: the embeddings of different synthetic clips differ by only ~0.1 in norm (random towers), so cosine
    similarity is uninformative; the error is measured RELATIVE TO THE SIGNAL: ||e_gpu - e_ref|| / ||e_ref - mean(e_ref)||
    per row, and must stay below 0.15; Recall@k must match the oracle to +-0.01 on the planted-projection task.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from fitclip_amd import ops, synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402
from fitclip_amd.retrieval import TeacherStudentModule, TextVideoRetrievalModule  # noqa: E402
from fitclip_amd.wise import wise  # noqa: E402
from oracle import clip_oracle as O  # noqa: E402

DEV = "cuda"
F32_TOL = 2e-5
BF16_TOL = 6e-3  # unit-norm embeddings, typical element 0.044; observed bf16 error 7e-4 .. 2.4e-3


def _encoder(sd, precision, **kw):
    return ClipVideoTextEncoder(build_clip(sd, precision=precision, device=DEV, **kw))


def _signal_rel_err(got: np.ndarray, ref: np.ndarray) -> float:
    signal = np.linalg.norm(ref - ref.mean(0, keepdims=True), axis=1)
    return float((np.linalg.norm(got - ref, axis=1) / signal).max())


@pytest.mark.parametrize("tag,dims", [("tiny", synth.TINY), ("vitb16", synth.VIT_B_16)])
def test_towers_fp32_match_reference_fixtures(golden_dir, tag, dims, request):
    """Raw tower outputs vs the fixtures produced by the reference's slip classes and by HF CLIP."""
    g = np.load(golden_dir / f"towers_{tag}.npz")
    sd = request.getfixturevalue(f"{tag}_state_dict")
    model = build_clip(sd, precision="fp32", device=DEV)
    video = torch.from_numpy(synth.make_video(int(g["n_clip"]), int(g["n_frames"]), dims, seed=int(g["seed"])))
    img = model.encode_image(video.reshape(-1, *video.shape[2:]).to(DEV)).cpu().numpy()
    txt = model.encode_text(torch.from_numpy(g["ids"]).to(DEV)).cpu().numpy()
    txt_rand = model.encode_text(torch.from_numpy(g["ids_rand"]).to(DEV)).cpu().numpy()
    scale = max(1.0, float(np.abs(g["image_features_oracle"]).max()))
    for ref in ("slip", "hf", "oracle"):
        assert np.abs(img - g[f"image_features_{ref}"]).max() < F32_TOL * scale, ref
        assert np.abs(txt - g[f"text_features_{ref}"]).max() < F32_TOL * scale, ref
    assert np.abs(txt_rand - g["text_features_rand_slip"]).max() < F32_TOL * scale  # argmax first-max semantics


@pytest.mark.parametrize("tag,dims", [("tiny", synth.TINY), ("config1", synth.VIT_B_16)])
def test_evaluate_goldens(golden_dir, tag, dims, request):
    """`command=evaluate` end to end (batches of 4, as the loop of the driver does) vs the oracle goldens:
    embeddings, T @ V^T, loss/val, R@1/5/10, median rank."""
    g = np.load(golden_dir / f"evaluate_{tag}.npz")
    sd = request.getfixturevalue("tiny_state_dict" if tag == "tiny" else "vitb16_state_dict")
    n, f = int(g["n_clips"]), int(g["n_frames"])
    video = torch.from_numpy(synth.make_video(n, f, dims, seed=42)).to(DEV)
    ids = torch.from_numpy(synth.make_text(n, dims, seed=42)).to(DEV)
    for precision in ("fp32", "bf16"):
        enc = _encoder(sd, precision)
        module = TextVideoRetrievalModule(enc, init_temperature=0.015)
        for s in range(0, n, 4):
            batch = {"video": video[s:s + 4], "text": {"input_ids": ids[s:s + 4]}, "video_id": list(range(s, s + 4))}
            module.validation_step_end(module.validation_step(batch))
        ev = torch.cat([o[0] for o in module._outputs]).cpu().numpy()
        et = torch.cat([o[1] for o in module._outputs]).cpu().numpy()
        metrics = module.validation_epoch_end()
        if precision == "fp32":
            assert np.abs(ev - g["encoded_videos"]).max() < F32_TOL
            assert np.abs(et - g["encoded_texts"]).max() < F32_TOL
            scores = ops.similarity(torch.from_numpy(et).to(DEV), torch.from_numpy(ev).to(DEV)).cpu().numpy()
            assert np.abs(scores - g["scores"]).max() < 5e-5
            assert ops.ranks(torch.from_numpy(g["scores"]).to(DEV)).tolist() == g["ranks"].tolist()
            for k in ("r1", "r5", "r10", "mr"):
                assert metrics[k] == pytest.approx(float(g[k])), k
            # loss/val is logged per batch of 4 here; recompute the golden's single-batch value on the full set
            full = float(ops.nce_loss(module.step_scores(torch.from_numpy(ev).to(DEV), torch.from_numpy(et).to(DEV))))
            assert abs(full - float(g["loss_val"])) < 2e-3
        else:
            assert _signal_rel_err(ev, g["encoded_videos"]) < 0.15
            assert _signal_rel_err(et, g["encoded_texts"]) < 0.15
            assert np.abs(ev - g["encoded_videos"]).max() < BF16_TOL and np.abs(et - g["encoded_texts"]).max() < BF16_TOL


def test_wise_encoder_matches_oracle(golden_dir, tiny_state_dict):
    """BASELINE config 3 in miniature: wise(teacher, student, 0.5) then encode."""
    g = np.load(golden_dir / "wise_encoder_tiny.npz")
    d = synth.TINY
    sd2 = synth.perturbed_state_dict(tiny_state_dict, d, seed=int(g["student_seed"]), rel=float(g["rel"]))
    enc = wise(_encoder(tiny_state_dict, "fp32"), _encoder(sd2, "fp32"), weight_for_2=float(g["weight_for_2"]))
    assert isinstance(enc, ClipVideoTextEncoder)
    ref_sd = O.wise_state_dict(O.to_torch(tiny_state_dict), O.to_torch(sd2), 0.5)
    for k, p in enc.model.named_parameters():
        assert torch.equal(p.cpu(), ref_sd[k]), k  # bit-exact blend
    video = torch.from_numpy(synth.make_video(6, 2, d, seed=42)).to(DEV)
    ids = torch.from_numpy(synth.make_text(6, d, seed=42)).to(DEV)
    ev, et = enc(video=video, text={"input_ids": ids})
    assert np.abs(ev.cpu().numpy() - g["encoded_videos"]).max() < F32_TOL
    assert np.abs(et.cpu().numpy() - g["encoded_texts"]).max() < F32_TOL
    with pytest.raises(AssertionError):
        wise(enc, enc.model)  # different classes, as aligner/wise.py:20


def test_recall_parity_on_an_ill_conditioned_task_fp32_exact_bf16_outside_north_star(tiny_state_dict):
    """A non-vacuous Recall@k check through both encoders.  Random towers give chance-level retrieval, so
    `text_projection` is FITTED (ridge regression on ORACLE features; 256 captions against a 128-wide tower, so the fit
    cannot interpolate) until captions retrieve their own clips ~89 % of the time; R@1/5/10/MedR of the fp32 and bf16
    HIP paths are then compared with the oracle's on the same weights.

    The synthetic clips are nearly collinear (cosine 0.96 between clips) and the winning margins are ~1e-2, so this
    task flips ranks under perturbations of 1e-3: fp32 may differ by one near-tie (1/256), bf16 by 0.03.  (The north
    star's +-0.01 is for real checkpoints, whose margins are an order of magnitude wider.)"""
    d = synth.TINY
    n, f = 256, 2
    sd = O.to_torch(dict(tiny_state_dict))
    video = torch.from_numpy(synth.make_video(n, f, d, seed=7))
    ids = torch.from_numpy(synth.make_text(n, d, seed=7))
    with torch.inference_mode():
        ev = O.encode_video(sd, video)
        eye = {**sd, "text_projection": torch.eye(d.transformer_width)}
        feats = O.encode_text_tokens(eye, ids)  # pre-projection text features
        target = ev + 4 * (ev - ev.mean(0, keepdim=True))
        proj = torch.linalg.solve(feats.T @ feats + torch.eye(feats.shape[1]), feats.T @ target)
        sd2 = {**sd, "text_projection": proj.float().contiguous()}
        ref = O.retrieval_metrics(O.retrieval_scores(O.encode_text(sd2, {"input_ids": ids}), ev))
    assert 0.5 < ref["r1"] < 0.99, ref  # neither trivial nor chance
    for precision, tol in (("fp32", 1 / 256 + 1e-9), ("bf16", 0.03)):
        enc = _encoder({k: v.numpy() for k, v in sd2.items()}, precision)
        module = TextVideoRetrievalModule(enc, init_temperature=0.015)
        for s in range(0, n, 32):
            module.validation_step_end(module.validation_step(
                {"video": video[s:s + 32].to(DEV), "text": {"input_ids": ids[s:s + 32].to(DEV)}}))
        got = module.validation_epoch_end()
        print(precision, got, ref)
        for k in ("r1", "r5", "r10"):
            assert abs(got[k] - ref[k]) <= tol, (precision, k, got, ref)
        assert abs(got["mr"] - ref["mr"]) <= 1, (precision, got, ref)


def test_recall_within_north_star_tolerance_with_wide_margins(tiny_state_dict):
    """The +-0.01 Recall bar of the north star, on a task whose winning margins are wide (as with trained checkpoints)
    but whose recall is not saturated.  The CLS features of the random tower are centred (their dataset mean is folded
    into `visual.ln_post.bias`) and `visual.proj` keeps their top principal directions, so clips spread over the sphere
    (max cosine 0.5 instead of 0.96); `text_projection` is fitted in the INTERPOLATING regime (96 captions, 128-wide
    tower) towards the embedding of the caption's own clip -- except for 30 % of the captions, which are planted on a
    WRONG clip.  The oracle then retrieves 71 % at rank 1 with score gaps >= 0.3, far above bf16 noise."""
    d = synth.TINY
    n, f = 96, 2
    sd = O.to_torch(dict(tiny_state_dict))
    video = torch.from_numpy(synth.make_video(n, f, d, seed=21))
    ids = torch.from_numpy(synth.make_text(n, d, seed=21))
    g = torch.Generator().manual_seed(5)
    with torch.inference_mode():
        cls = O.encode_image({**sd, "visual.proj": torch.eye(d.vision_width)}, video.flatten(0, 1)).double()
        mu = cls.mean(0)                                                  # ln_post(CLS) features, [n*f, width]
        _, _, vt = torch.linalg.svd(cls - mu, full_matrices=False)
        sd_v = {**sd, "visual.proj": vt[:d.embed_dim].T.float().contiguous(),
                "visual.ln_post.bias": (sd["visual.ln_post.bias"].double() - mu).float()}
        ev = O.encode_video(sd_v, video)
        wrong = torch.rand(n, generator=g) < 0.3
        perm = torch.where(wrong, torch.roll(torch.arange(n), 7), torch.arange(n))
        feats = O.encode_text_tokens({**sd_v, "text_projection": torch.eye(d.transformer_width)}, ids).double()
        proj = torch.linalg.lstsq(feats, ev[perm].double()).solution      # exact: n <= width
        sd2 = {**sd_v, "text_projection": proj.float().contiguous()}
        scores = O.retrieval_scores(O.encode_text(sd2, {"input_ids": ids}), ev)
        ref = O.retrieval_metrics(scores)
        top2 = scores.topk(2, dim=1).values
        assert float((top2[:, 0] - top2[:, 1]).min()) > 0.2, "margins are meant to be wide"
    assert 0.55 < ref["r1"] < 0.85, ref
    for precision in ("fp32", "bf16"):
        enc = _encoder({k: v.numpy() for k, v in sd2.items()}, precision)
        enc.num_frames = f
        module = TextVideoRetrievalModule(enc, init_temperature=0.015)
        for s0 in range(0, n, 32):
            module.validation_step_end(module.validation_step(
                {"video": video[s0:s0 + 32].to(DEV), "text": {"input_ids": ids[s0:s0 + 32].to(DEV)}}))
        got = module.validation_epoch_end()
        for kk in ("r1", "r5", "r10"):
            assert abs(got[kk] - ref[kk]) <= 0.01, (precision, kk, got, ref)
        assert got["mr"] == ref["mr"], (precision, got, ref)


def test_batch_and_chunk_invariance(vitb16_state_dict):
    """Size-independent property used at full size: a clip's embedding does not depend on what else is in the batch,
    on its position, or on how the batch is chunked / tiled (every output row only depends on its own input row and
    runs the same k-loop)."""
    d = synth.VIT_B_16
    base = torch.from_numpy(synth.make_video(3, 2, d, seed=11)).to(DEV)
    ids = torch.from_numpy(synth.make_text(5, d, seed=11)).to(DEV)
    for precision in ("fp32", "bf16"):
        ref_enc = _encoder(vitb16_state_dict, precision, gemm_tile=1)
        ref_v = ref_enc.encode_video(base)
        ref_t = ref_enc.encode_text({"input_ids": ids})
        big = base.repeat(11, 1, 1, 1, 1)[torch.randperm(33, generator=torch.Generator().manual_seed(0))]
        order = torch.randperm(33, generator=torch.Generator().manual_seed(0)) % 3
        for kw in (dict(gemm_tile=2), dict(gemm_tile=1, chunk_frames=7, chunk_texts=3), dict()):
            enc = _encoder(vitb16_state_dict, precision, **kw)
            out = enc.encode_video(big)
            assert torch.equal(out, ref_v[order.to(DEV)]), (precision, kw)
            assert torch.equal(enc.encode_text({"input_ids": ids.repeat(4, 1)}), ref_t.repeat(4, 1)), (precision, kw)


def test_teacher_student_forward_and_kd_loss(tiny_state_dict):
    """BASELINE config 5 in miniature: student + teacher dual forward, NCE on labeled, KD * tau^2 on unlabeled."""
    d = synth.TINY
    sd_student = synth.perturbed_state_dict(tiny_state_dict, d, seed=5, rel=0.3)
    video = torch.from_numpy(synth.make_video(8, 2, d, seed=3))
    ids = torch.from_numpy(synth.make_text(8, d, seed=3))
    module = TeacherStudentModule(_encoder(sd_student, "fp32"), _encoder(tiny_state_dict, "fp32"), init_temperature=0.05)
    batch = {"video_student": video.to(DEV), "video_teacher": video.to(DEV),
             "text_student": {"input_ids": ids.to(DEV)}, "text_teacher": {"input_ids": ids.to(DEV)}}
    out = module.step(batch)
    with torch.inference_mode():
        sv, st = O.forward(O.to_torch(sd_student), video, {"input_ids": ids})
        tv, tt = O.forward(O.to_torch(tiny_state_dict), video, {"input_ids": ids})
        s, t = O.step_scores(sv, st, 0.05), O.step_scores(tv, tt, 0.05)
        ref_labeled = float(O.nce_loss(s))
        ref_unlabeled = float(O.teacher_student_nce_loss(s, t) * (1 / 0.05) ** 2)
    assert abs(float(module.dataset_step_end(out, labeled=True)) - ref_labeled) < 1e-3 * max(1, abs(ref_labeled))
    got = float(module.dataset_step_end(out, labeled=False))
    assert abs(got - ref_unlabeled) < 2e-3 * max(1.0, abs(ref_unlabeled)), (got, ref_unlabeled)


def test_edge_cases(tiny_state_dict):
    enc = _encoder(tiny_state_dict, "bf16")
    d = synth.TINY
    assert enc.encode_video(torch.zeros(0, 2, 3, 64, 64, device=DEV)).shape == (0, d.embed_dim)
    assert enc.encode_text({"input_ids": torch.zeros(0, d.context_length, dtype=torch.long, device=DEV)}).shape == (0, d.embed_dim)
    one = enc.encode_video(torch.from_numpy(synth.make_video(1, 1, d, seed=1)).to(DEV))
    assert one.shape == (1, d.embed_dim) and abs(float(one.norm()) - 1.0) < 1e-5  # one frame: unit vector
    with pytest.raises(ValueError):
        enc.model.encode_image(torch.zeros(1, 3, 32, 32, device=DEV))
    tok = enc.get_tokenizer()(["a video of a cat", "x " * 100])
    assert tok["input_ids"].shape == (2, d.context_length) and int(tok["input_ids"][1].max()) == d.vocab_size - 1
    assert torch.isfinite(enc.encode_text({k: v.to(DEV) for k, v in tok.items()})).all()
    assert enc.should_pad_batch is True and enc.get_eval_frame_sampler()(0, 99, 30.0) == [12, 36, 62, 86]
    frames = torch.randint(0, 255, (2, 80, 120, 3), dtype=torch.uint8)
    assert enc.get_eval_transform(torch.float32)(frames).shape == (2, 3, 64, 64)


def test_evaluate_driver_reproduces_config1_golden(golden_dir, capsys):
    """`python -m fitclip_amd command=evaluate encoder=clip_vit_b_16 n_clips=16 num_frames=1` (BASELINE config 1) prints
    the metrics of the oracle golden: seed 42, one batch of 16, logit scale 1/0.015."""
    import json
    from fitclip_amd.__main__ import main
    g = np.load(golden_dir / "evaluate_config1.npz")
    main(["command=evaluate", "encoder=clip_vit_b_16", "n_clips=16", "num_frames=1", "precision=fp32"])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    for k in ("r1", "r5", "r10", "mr"):
        assert out[k] == pytest.approx(float(g[k])), (k, out)
    assert abs(out["loss/val"] - float(g["loss_val"])) < 2e-3


@pytest.mark.parametrize("precision", ["fp32", "fp32x6", "fp32x3", "bf16"])
@pytest.mark.parametrize("clips,frames,n_text", [(256, 8, 256), (1024, 16, 1024)])
def test_full_size_batches_by_invariance(vitb16_state_dict, clips, frames, n_text, precision):
    """BASELINE configs[1] (256 clips x 8 frames + 256 texts) and one rank's shard of configs[3] (1024 clips x 16
    frames + 1024 texts) at FULL size, checked through the size-independent property that a row's embedding does not
    depend on the rest of the batch: the big batch is built from 4 base clips / 8 base captions whose embeddings are
    computed alone (and those small cases are pinned to the oracle by the golden tests above)."""
    d = synth.VIT_B_16
    enc = _encoder(vitb16_state_dict, precision)
    base_v = torch.from_numpy(synth.make_video(4, frames, d, seed=21)).to(DEV)
    base_t = torch.from_numpy(synth.make_text(8, d, seed=21)).to(DEV)
    ref_v = enc.encode_video(base_v)
    ref_t = enc.encode_text({"input_ids": base_t})
    if precision == "fp32x3":
        # this mode picks the text tower's arithmetic per CALL (from 2048 token rows on: the three-product GEMMs; below, the fp32
        # kernels - tests/test_gpu_split2.py), so the captions "alone" are computed in a call of the big batch's kind: 8 x 8 of them
        ref_t = enc.encode_text({"input_ids": base_t.repeat(8, 1)})[:8]
    gen = torch.Generator().manual_seed(1)
    pick_v = torch.randint(0, 4, (clips,), generator=gen)
    pick_t = torch.randint(0, 8, (n_text,), generator=gen)
    out_v = enc.encode_video(base_v[pick_v.to(DEV)])
    out_t = enc.encode_text({"input_ids": base_t[pick_t.to(DEV)]})
    assert torch.equal(out_v, ref_v[pick_v.to(DEV)])
    assert torch.equal(out_t, ref_t[pick_t.to(DEV)])
    # scoring at full size: T @ V^T equals the 8 x 4 block matrix it must be, and every rank is consistent with it
    scores = ops.similarity(out_t, out_v)
    small = ops.similarity(ref_t, ref_v)
    assert torch.equal(scores, small[pick_t.to(DEV)][:, pick_v.to(DEV)])
    n = min(clips, n_text)
    ranks = ops.ranks(scores[:n, :n].contiguous())
    want = O.ranks_of_target(scores[:n, :n].cpu(), torch.arange(n))
    assert ranks.cpu().tolist() == want.tolist()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_teacher_student_full_size_by_invariance(vitb16_state_dict, precision):
    """BASELINE configs[4] at full size: teacher + student ViT-B/16 forward over 512 clips x 8 frames and the KD / NCE
    similarity losses on the 512 x 512 score matrices.  The oracle cannot run this size, so: (i) the batch is built
    from 4 base clips / 8 base captions whose embeddings are computed alone (batch invariance, bit-exact); (ii) the
    device losses equal the reference formulas (`aligner/loss.py:13-39`, `teacher_student.py:150-159`) evaluated in
    float64 on the CPU from the SAME device embeddings; (iii) distilling a model into itself costs nothing."""
    d = synth.VIT_B_16
    n, f = 512, 8
    student_sd = synth.perturbed_state_dict(vitb16_state_dict, d, seed=5, rel=0.05)
    student, teacher = _encoder(student_sd, precision), _encoder(vitb16_state_dict, precision)
    student.num_frames = teacher.num_frames = f
    base_v = torch.from_numpy(synth.make_video(4, f, d, seed=31)).to(DEV)
    base_t = torch.from_numpy(synth.make_text(8, d, seed=31)).to(DEV)
    gen = torch.Generator().manual_seed(2)
    pick_v, pick_t = torch.randint(0, 4, (n,), generator=gen).to(DEV), torch.randint(0, 8, (n,), generator=gen).to(DEV)
    batch = {"video_student": base_v[pick_v], "text_student": {"input_ids": base_t[pick_t]},
             "video_teacher": base_v[pick_v], "text_teacher": {"input_ids": base_t[pick_t]}}
    module = TeacherStudentModule(student, teacher, init_temperature=0.05)
    (sv, stx), (tv, ttx) = out = module.step(batch)
    for enc, v, t in ((student, sv, stx), (teacher, tv, ttx)):
        assert torch.equal(v, enc.encode_video(base_v)[pick_v]) and torch.equal(t, enc.encode_text({"input_ids": base_t})[pick_t])
    scale = 1 / 0.05
    kd = float(module.dataset_step_end(out, labeled=False))
    nce = float(module.dataset_step_end(out, labeled=True))
    s64 = scale * (sv.double().cpu() @ stx.double().cpu().T)
    t64 = scale * (tv.double().cpu() @ ttx.double().cpu().T)
    want_kd = float(O.teacher_student_nce_loss(s64, t64)) * scale ** 2
    want_nce = float(O.nce_loss(s64))
    assert abs(kd - want_kd) <= 2e-4 * max(1.0, abs(want_kd)), (kd, want_kd)
    assert abs(nce - want_nce) <= 2e-4 * max(1.0, abs(want_nce)), (nce, want_nce)
    self_distilled = TeacherStudentModule(teacher, teacher, init_temperature=0.05)
    assert abs(float(self_distilled.dataset_step_end(self_distilled.step(batch), labeled=False))) < 1e-6


def test_wise_full_size_is_bit_exact(vitb16_state_dict):
    """WiSE over all 149.6 M parameters of ViT-B/16 (BASELINE configs[2]): bit-identical to the reference expression
    `(1 - w) * p1 + w * p2` evaluated by torch on the CPU."""
    d = synth.VIT_B_16
    sd2 = synth.perturbed_state_dict(vitb16_state_dict, d, seed=9, rel=0.05)
    e1, e2 = _encoder(vitb16_state_dict, "bf16"), _encoder(sd2, "bf16")
    ens = wise(e1, e2, weight_for_2=0.5)
    for k in ("model.visual.conv1.weight", "model.token_embedding.weight", "model.transformer.resblocks.11.mlp.c_fc.bias",
              "model.visual.transformer.resblocks.0.attn.in_proj_weight", "model.ln_final.weight"):
        name = k[len("model."):]
        want = (1 - 0.5) * torch.from_numpy(vitb16_state_dict[name]) + 0.5 * torch.from_numpy(sd2[name])
        assert torch.equal(dict(ens.named_parameters())[k].cpu(), want), k
    video = torch.from_numpy(synth.make_video(2, 2, d, seed=3)).to(DEV)
    assert torch.isfinite(ens.encode_video(video)).all()


def test_zero_shot_classification_matches_oracle(tiny_state_dict):
    """labels x templates -> prompt embeddings -> template mean -> video @ labels^T -> Acc@1/5, MedR
    (aligner/video_text_classification.py:29-140), HIP path vs oracle on the same seeded model and clips."""
    from fitclip_amd.classification import VideoTextClassificationModule
    d = synth.TINY
    labels = ["cat", "dog", "guitar", "beach", "kitchen", "horse", "bicycle"]
    templates = ["a video of a {}", "a clip showing a {}", "{} in the wild"]
    enc = _encoder(tiny_state_dict, "fp32")
    module = VideoTextClassificationModule(enc, labels, templates)
    video = torch.from_numpy(synth.make_video(20, 2, d, seed=4))
    label_id = torch.arange(20) % len(labels)
    for s in range(0, 20, 8):
        module.validation_step({"video": video[s:s + 8].to(DEV), "target": (None, label_id[s:s + 8])})
    got = module.validation_epoch_end()
    sd = O.to_torch(tiny_state_dict)
    prompts = [t.format(l) for l in labels for t in templates]
    ids = enc.get_tokenizer()(prompts)["input_ids"]
    with torch.inference_mode():
        lab = O.zero_shot_label_embeddings(O.encode_text(sd, {"input_ids": ids}), len(templates))
        scores = O.encode_video(sd, video) @ lab.T
    ref = O.zero_shot_metrics(scores, label_id)
    assert np.abs(module.encoded_labels.cpu().numpy() - lab.numpy()).max() < F32_TOL
    assert np.abs(module(video[:4].to(DEV)).cpu().numpy() - scores[:4].numpy()).max() < 5e-5
    assert got == pytest.approx(ref)
    pred = module.predict_step({"video": video[:4].to(DEV), "target": (None, label_id[:4]), "video_id": list("abcd")})
    assert pred["predictions"].tolist() == scores[:4].argmax(-1).tolist()


@pytest.mark.parametrize("H,W", [(224, 224), (240, 320), (360, 202), (256, 256), (90, 130)])
def test_device_preprocessing_matches_the_eval_transform(H, W):
    """fc_preprocess_u8 vs the float64 restatement of the reference's eval transform under oracle/
    (BHWC->BCHW, /255, torchvision Resize(BICUBIC) of the shorter side, CenterCrop, CLIP mean/std -
    clip_video_text_encoder.py:125-133)."""
    from fitclip_amd.encoder import CLIP_MEAN, CLIP_STD
    from oracle.transform_oracle import eval_transform
    frames = torch.randint(0, 256, (3, H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(H * W))
    want = eval_transform(frames.numpy(), 224, CLIP_MEAN, CLIP_STD)
    got = ops.preprocess_u8(frames.to(DEV), 224, CLIP_MEAN, CLIP_STD).cpu().numpy()
    assert got.shape == want.shape == (3, 3, 224, 224)
    # the kernel computes source coordinates and tap weights in float32 (as torch's bicubic does): ~1e-4 on values in
    # [-1.8, 2.2] against the float64 restatement
    assert np.abs(got - want).max() < 2e-4


def test_encode_video_uint8_path(tiny_state_dict):
    enc = _encoder(tiny_state_dict, "fp32")
    frames = torch.randint(0, 256, (2, 3, 80, 96, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(3))
    tf = enc.get_eval_transform(torch.float32)
    video = torch.stack([tf(v) for v in frames])  # what the reference's dataset would hand over
    ref = enc.encode_video(video.to(DEV))
    got = enc.encode_video_uint8(frames.to(DEV))
    assert (got - ref).abs().max() < 2e-5


def test_small_workspace_means_more_passes_not_different_results(tiny_state_dict):
    """fc_encode_image / fc_encode_text accept any workspace that holds at least one item: fewer bytes -> more chunks,
    identical output; too few bytes -> FC_ENOMEM with a message, nothing launched."""
    from fitclip_amd import _lib
    d = synth.TINY
    model = build_clip(tiny_state_dict, precision="bf16", device=DEV)
    images = torch.from_numpy(synth.make_video(9, 1, d, seed=8)).reshape(9, 3, 64, 64).to(DEV)
    ref = model.encode_image(images)
    rt, lib = model._ensure_ready(), _lib.load()
    need = lib.fc_workspace_bytes(rt.handle, 0, 9)
    for frac in (0.5, 0.2):
        ws = torch.empty(int(need * frac), dtype=torch.uint8, device=DEV)
        out = torch.empty_like(ref)
        _lib.check(lib.fc_encode_image(rt.handle, images.data_ptr(), 9, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                       _lib.current_stream()))
        assert torch.equal(out, ref), frac
    tiny_ws = torch.empty(1024, dtype=torch.uint8, device=DEV)
    rc = lib.fc_encode_image(rt.handle, images.data_ptr(), 9, ref.data_ptr(), tiny_ws.data_ptr(), tiny_ws.numel(),
                             _lib.current_stream())
    assert rc == -3 and b"workspace too small" in lib.fc_last_error()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_pruned_last_block_is_bit_identical(vitb16_state_dict, precision):
    """`prune_last_block` only skips rows nothing reads after the last block: embeddings must not change by one bit."""
    d = synth.VIT_B_16
    video = torch.from_numpy(synth.make_video(5, 2, d, seed=13)).to(DEV)
    ids = torch.from_numpy(synth.make_text(7, d, seed=13)).to(DEV)
    full = _encoder(vitb16_state_dict, precision)
    pruned = _encoder(vitb16_state_dict, precision, prune_last_block=True)
    assert torch.equal(pruned.encode_video(video), full.encode_video(video))
    assert torch.equal(pruned.encode_text({"input_ids": ids}), full.encode_text({"input_ids": ids}))
    assert torch.equal(pruned.encode_text({"input_ids": torch.from_numpy(synth.make_text(7, d, 14, all_random=True)).to(DEV)}),
                       full.encode_text({"input_ids": torch.from_numpy(synth.make_text(7, d, 14, all_random=True)).to(DEV)}))


# The reference's other CLIP ViT configs (config/encoder/clip_vit_b_32.yaml, clip_vit_l_14.yaml, clip_vit_l_14_336px.yaml)
# go through the same ClipVideoTextEncoder.  Shrunk stand-ins with their awkward geometry: patch 14 (3*14*14 = 588
# columns: not a multiple of the GEMM's K tile -> zero-padded patch-embed), 257 and 577 tokens (> 224: the streaming
# attention kernel), width 1024 / 16 heads, a 768-wide text tower; patch 32 with 50 tokens.
_OTHER_VITS = {
    "L14-like": synth.ClipDims(embed_dim=768, image_resolution=224, vision_layers=2, vision_width=1024,
                               vision_patch_size=14, context_length=77, vocab_size=2048, transformer_width=768,
                               transformer_heads=12, transformer_layers=2),
    "L14@336-like": synth.ClipDims(embed_dim=256, image_resolution=336, vision_layers=1, vision_width=256,
                                   vision_patch_size=14, context_length=16, vocab_size=1024, transformer_width=128,
                                   transformer_heads=2, transformer_layers=1),
    "B32-like": synth.ClipDims(embed_dim=512, image_resolution=224, vision_layers=2, vision_width=768,
                               vision_patch_size=32, context_length=77, vocab_size=2048, transformer_width=512,
                               transformer_heads=8, transformer_layers=2),
}


@pytest.mark.parametrize("tag", sorted(_OTHER_VITS))
def test_other_clip_vit_geometries_match_oracle(tag):
    d = _OTHER_VITS[tag]
    sd = synth.make_state_dict(d, seed=7)
    video = synth.make_video(3, 2, d, seed=11)
    ids = synth.make_text(5, d, seed=12)
    ref_v = O.encode_video(O.to_torch(sd), torch.from_numpy(video)).numpy()
    ref_t = O.encode_text(O.to_torch(sd), {"input_ids": torch.from_numpy(ids)}).numpy()
    # the split-fp32 modes serve these geometries too, at the fp32 tolerances: their block GEMMs run on two / three planes whatever
    # the width (1024, 256) and the sequence length; sequence lengths without a fused split attention (50, 257, 577 tokens) take
    # the fp32 attention + a split pass (tools/attn_geometry_probe.py has what that costs)
    # 40 captions in one call: where the text tower has a width the plane kernels serve (768, 512) and the call has >= 2048 token rows,
    # fp32x3 runs its block GEMMs on three fp16 products too (the 5-caption call above keeps the fp32 kernels)
    many = torch.from_numpy(synth.make_text(40, d, seed=13)).to(DEV)
    many_f32 = None
    for precision in ("fp32", "fp32x6", "fp32x3", "bf16"):
        enc = _encoder(sd, precision)
        enc.num_frames = 2
        got_v = enc.encode_video(torch.from_numpy(video).to(DEV)).cpu().numpy()
        got_t = enc.encode_text({"input_ids": torch.from_numpy(ids).to(DEV)}).cpu().numpy()
        if precision == "fp32":
            many_f32 = enc.encode_text({"input_ids": many})
        if precision != "bf16":
            assert np.abs(got_v - ref_v).max() < F32_TOL and np.abs(got_t - ref_t).max() < F32_TOL, precision
            if precision == "fp32x3":
                got_many = enc.encode_text({"input_ids": many})
                assert float((got_many - many_f32).abs().max()) < 5e-6, tag
                joined = d.transformer_width % 256 == 0 and 40 * d.context_length >= 2048
                assert torch.equal(got_many, many_f32) != joined, tag      # (the other arithmetic exactly where it should be)
                enc.model.check_range()
        else:
            assert np.abs(got_v - ref_v).max() < BF16_TOL and np.abs(got_t - ref_t).max() < BF16_TOL
            assert _signal_rel_err(got_t, ref_t) < 0.15


def test_apply_wise_ft_files(tmp_path, tiny_state_dict):
    """`python -m fitclip_amd.checkpoint apply-wise-ft A B OUT --weight-for-2 w` (scripts/apply_wise_ft.py): two
    state-dict files in, the blended state dict out, bit-identical to the torch expression; a missing `logit_scale`
    comes out as NaN."""
    from fitclip_amd import checkpoint as C
    d = synth.TINY
    sd1 = {k: torch.from_numpy(v) for k, v in tiny_state_dict.items()}
    sd2 = {k: torch.from_numpy(v) for k, v in synth.perturbed_state_dict(tiny_state_dict, d, seed=4, rel=0.1).items()}
    p1, p2, out = tmp_path / "a.pt", tmp_path / "b.pt", tmp_path / "wise.pt"
    torch.save(sd1, p1)
    torch.save(sd2, p2)
    C.main(["apply-wise-ft", str(p1), str(p2), str(out), "--weight-for-2", "0.3"])
    got = torch.load(out, weights_only=False)
    assert torch.isnan(got.pop("logit_scale")).all()  # NaN blended with NaN
    assert set(got) == set(sd1)
    for k in sd1:
        assert torch.equal(got[k], (1 - 0.3) * sd1[k] + 0.3 * sd2[k]), k


def test_two_stream_forward_and_image_lanes_are_bit_identical(vitb16_state_dict, monkeypatch):
    """`forward` with the text tower on a side stream, and `encode_image` with slices of a big batch on several streams
    (`FITCLIP_IMAGE_STREAMS`), return exactly what the single-stream path returns (rows are independent; the caller's
    stream waits for the side streams)."""
    from fitclip_amd import clip_model
    d = synth.VIT_B_16
    enc = _encoder(vitb16_state_dict, "bf16", chunk_frames=32)
    enc.num_frames = 2
    video = torch.from_numpy(synth.make_video(70, 2, d, seed=41)).to(DEV)   # 140 frames: 4 lanes of 64 / 64 / 12 / -
    text = {"input_ids": torch.from_numpy(synth.make_text(33, d, seed=41)).to(DEV)}
    enc.overlap_text = False
    ref_v, ref_t = enc(video=video, text=text)
    enc.overlap_text = True
    for lanes in (1, 2, 4):
        monkeypatch.setattr(clip_model, "_IMAGE_STREAMS", lanes)
        got_v, got_t = enc(video=video, text=text)
        torch.cuda.synchronize()
        assert torch.equal(got_v, ref_v) and torch.equal(got_t, ref_t), lanes


def test_wise_evaluate_at_webvid_val_shape_by_invariance(vitb16_state_dict):
    """BASELINE configs[2]: `encoder=wise` (0.5 CLIP + 0.5 student, ViT-B/16) through the evaluate loop at the
    WebVid-val shape the survey fixes (N = 4096 clips x 4 frames, one caption per clip, eval batches of 32), in the
    reference's precision.  The oracle cannot run this size, so the 4096 clips / captions are drawn from 8 base clips /
    16 base captions: (i) every embedding must equal, bit for bit, the embedding of its base item computed alone by the
    same blended encoder (batch invariance); (ii) the blended parameters are bit-identical to `aligner.wise`'s
    expression; (iii) R@1/5/10 / MedR of the epoch end equal the oracle's metric code applied to the device scores
    (ties between duplicates are broken by index on both sides)."""
    d = synth.VIT_B_16
    n, f = 4096, 4
    sd2 = synth.perturbed_state_dict(vitb16_state_dict, d, seed=9, rel=0.05)
    enc = wise(_encoder(vitb16_state_dict, "fp32"), _encoder(sd2, "fp32"), weight_for_2=0.5)
    enc.num_frames = f
    name = "visual.transformer.resblocks.7.mlp.c_fc.weight"
    want = (1 - 0.5) * torch.from_numpy(vitb16_state_dict[name]) + 0.5 * torch.from_numpy(sd2[name])
    assert torch.equal(dict(enc.model.named_parameters())[name].cpu(), want)
    base_v = torch.from_numpy(synth.make_video(8, f, d, seed=51)).to(DEV)
    base_t = torch.from_numpy(synth.make_text(16, d, seed=51)).to(DEV)
    ref_v, ref_t = enc.encode_video(base_v), enc.encode_text({"input_ids": base_t})
    gen = torch.Generator().manual_seed(3)
    pick_v, pick_t = torch.randint(0, 8, (n,), generator=gen).to(DEV), torch.randint(0, 16, (n,), generator=gen).to(DEV)
    module = TextVideoRetrievalModule(enc, init_temperature=0.015)
    for s in range(0, n, 32):
        out = module.validation_step({"video": base_v[pick_v[s:s + 32]], "text": {"input_ids": base_t[pick_t[s:s + 32]]},
                                      "video_id": list(range(s, s + 32))})
        module.validation_step_end(out)
    ev = torch.cat([o[0] for o in module._outputs])
    et = torch.cat([o[1] for o in module._outputs])
    assert torch.equal(ev, ref_v[pick_v]) and torch.equal(et, ref_t[pick_t])
    got = module.validation_epoch_end()
    scores = ops.similarity(et, ev).cpu()
    ref = O.retrieval_metrics(scores)
    for k in ("r1", "r5", "r10", "mr"):
        assert got[k] == pytest.approx(ref[k]), (k, got, ref)
    assert np.isfinite(got["loss/val"])


def test_data_writes_are_picked_up_after_invalidate_weights(tiny_state_dict):
    """`param.data` writes do not bump the autograd version counter, so the kernel-layout weight copies (bf16 arena,
    transposed projections) would go stale: `CLIP.invalidate_weights()` (also called by load_state_dict / .to()) makes
    the next encode repack.  Both precisions: fp32 uses the caller's tensors directly except for the projections."""
    d = synth.TINY
    video = torch.from_numpy(synth.make_video(3, 2, d, seed=2)).to(DEV)
    ids = torch.from_numpy(synth.make_text(3, d, seed=2)).to(DEV)
    for precision in ("fp32", "bf16"):
        enc = _encoder(tiny_state_dict, precision)
        v0, t0 = enc.encode_video(video), enc.encode_text({"input_ids": ids})
        with torch.no_grad():
            enc.model.visual.proj.data.mul_(-1.0)
            enc.model.text_projection.data.mul_(-1.0)
            getattr(enc.model.visual.transformer.resblocks, "0").mlp.c_fc.weight.data.mul_(0.5)
        enc.model.invalidate_weights()
        v1, t1 = enc.encode_video(video), enc.encode_text({"input_ids": ids})
        assert torch.allclose(t1, -t0, atol=1e-6), precision          # the transposed projection copy was rebuilt
        assert (v1 + v0).abs().max() > 1e-3, precision                # and the block weight copy as well
        sd = {k: v.detach().cpu().numpy() for k, v in enc.model.state_dict().items()}
        fresh = _encoder(sd, precision)
        assert torch.equal(fresh.encode_video(video), v1) and torch.equal(fresh.encode_text({"input_ids": ids}), t1)
        enc.model.load_state_dict({k: torch.from_numpy(v) for k, v in tiny_state_dict.items()})  # invalidates by itself
        assert torch.equal(enc.encode_video(video), v0)


def test_similarity_rejects_mismatched_inner_dimensions():
    a, b = torch.zeros(4, 64, device=DEV), torch.zeros(4, 128, device=DEV)
    with pytest.raises(ValueError):
        ops.similarity(a, b)
    with pytest.raises(ValueError):
        ops.similarity(a, b[0])


def _run(cmd, timeout=900):
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, *cmd], capture_output=True, text=True, timeout=timeout, env=env,
                         cwd=str(__import__("pathlib").Path(__file__).resolve().parent.parent))
    assert res.returncode == 0, res.stderr[-3000:]
    import json
    return json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])


def test_two_rank_evaluate_equals_single_rank():
    """Rehearsal of the sharded path on ONE GPU: `python -m fitclip_amd command=evaluate gpus=2 backend=gloo` starts two
    fresh rank processes (both on device 0; collectives staged through the host), each encodes its own contiguous shard,
    embeddings are all-gathered once, ranks gathered - and the metrics equal the single-process run exactly."""
    # 24 clips in batches of 4: the per-batch `loss/val` terms are the same batches on 1 and on 2 ranks (ragged shards
    # are covered by tests/test_distributed_cpu.py)
    common = ["-m", "fitclip_amd", "command=evaluate", "encoder=clip_vit_b_16", "n_clips=24", "num_frames=1",
              "precision=fp32", "eval_batch_size=4"]
    one = _run(common)
    two = _run(common + ["gpus=2", "backend=gloo"])
    for k in ("r1", "r5", "r10", "mr"):
        assert one[k] == two[k], (k, one, two)
    assert abs(one["loss/val"] - two["loss/val"]) < 1e-5


def test_two_rank_evaluate_with_gathered_batches_logs_the_references_loss():
    """`gather_batches=true`: loss/val as the reference logs it on several ranks - the NCE of the per-step GATHERED batch
    (text_video_retrieval.py:44-58).  Two gloo ranks on one GPU, 21 clips = shards of 11 + 10 in batches of 4 (a ragged last
    step): step i scores rank 0's batch i together with rank 1's batch i.  Expected value: the same row sets scored in this
    process through the same operators; the retrieval metrics do not depend on the option."""
    import math
    common = ["-m", "fitclip_amd", "command=evaluate", "encoder=clip_vit_b_16", "n_clips=21", "num_frames=1",
              "precision=fp32", "eval_batch_size=4", "gpus=2", "backend=gloo"]
    plain = _run(common)
    gathered = _run(common + ["gather_batches=true"])
    for k in ("r1", "r5", "r10", "mr"):
        assert plain[k] == gathered[k], (k, plain, gathered)
    d = synth.VIT_B_16
    from fitclip_amd.__main__ import instantiate, load_encoder_config, parse_overrides
    cfg = parse_overrides(["encoder=clip_vit_b_16", "precision=fp32"])
    enc = instantiate(load_encoder_config("clip_vit_b_16", cfg, DEV)).to(DEV)
    video = torch.from_numpy(synth.make_video(21, 1, d, seed=42)).to(DEV)
    ids = torch.from_numpy(synth.make_text(21, d, seed=42)).to(DEV)
    ev, et = enc(video=video, text={"input_ids": ids})
    scale = 1.0 / 0.015

    def nce(rows):
        idx = torch.tensor(rows, device=DEV)
        return float(ops.nce_loss(ops.similarity(ev[idx].contiguous(), et[idx].contiguous(), alpha=scale))) * len(rows), len(rows)

    shard = [(0, 11), (11, 21)]
    num = den = 0.0
    for i in range(3):
        rows = [c for s, e in shard for c in range(min(e, s + 4 * i), min(e, s + 4 * i + 4))]
        a, b = nce(rows)
        num, den = num + a, den + b
    assert gathered["loss/val"] == pytest.approx(num / den, abs=2e-5)
    num = den = 0.0
    for s, e in shard:
        for lo in range(s, e, 4):
            a, b = nce(list(range(lo, min(e, lo + 4))))
            num, den = num + a, den + b
    assert plain["loss/val"] == pytest.approx(num / den, abs=2e-5)
    assert gathered["loss/val"] > plain["loss/val"] + 1e-3 and math.isfinite(gathered["loss/val"])


def test_bench_two_ranks_on_one_gpu_rehearsal():
    """`python bench.py --gpus 2 --backend gloo`: the bench launches its own ranks, shards the clips, gathers embeddings
    and ranks, and prints one line whose whole-job value counts both ranks' clips."""
    line = _run(["bench.py", "--gpus", "2", "--backend", "gloo", "--clips", "16", "--frames", "2", "--steps", "1",
                 "--warmup", "1", "--no-bf16-mode", "--no-split-mode", "--no-cpu-baseline", "--cpu-sample-clips", "8"])
    assert line["n_gpus"] == 2 and line["dtype"] == "fp32" and line["scaling"] == "weak"
    assert line["retrieval"]["n"] == 32 and line["value"] > 0
    assert line["retrieval"]["r1"] > 0.2  # rank 0's captions are planted on their clips; chance would be 1/32


def test_bench_config_c4_strong_scaling_rehearsal_with_a_ragged_shard():
    """`bench.py --config c4` (BASELINE configs[3], shrunk): a FIXED total of clips in exact contiguous shards - 21 clips over two
    ranks = 11 + 10 - encoded in eval batches of 4, one all-gather, row-block scoring; the retrieval metrics over all 21 clips
    equal the one-rank run of the same command (same inputs per global clip index is not required: metrics are per run)."""
    common = ["bench.py", "--config", "c4", "--total-clips", "21", "--frames", "2", "--eval-batch", "4", "--steps", "1",
              "--warmup", "1", "--no-cpu-baseline", "--no-plant"]
    two = _run(common + ["--gpus", "2", "--backend", "gloo"])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["retrieval"]["n"] == 21 and two["value"] > 0
    assert two["config"]["clips_per_gpu"] == [11, 10] and "configs[3]" in two["config"]["workload"]
    assert "bf16_mode" not in two and "fp32_split_mode" not in two and "skipped" in two["secondary_legs"]
    one = _run(common)
    assert one["n_gpus"] == 1 and one["retrieval"]["n"] == 21 and one["config"]["clips_per_gpu"] == 21
    assert one["roofline"]["frac"] > 0 and one["instrumented_repeat_ms_per_step"] > 0


def test_bench_config_c5_kd_training_step_rehearsal():
    """`bench.py --config c5` (BASELINE configs[4], shrunk): the teacher + student KD training step on 12 clips in total; on two
    gloo ranks (6 + 6, half labeled on each) the step exchanges the packed embeddings and the three gradient slices; the first
    loss equals the one-rank run's up to the summation order of the gathered batch."""
    common = ["bench.py", "--config", "c5", "--total-clips", "12", "--frames", "2", "--steps", "2", "--warmup", "1"]
    two = _run(common + ["--gpus", "2", "--backend", "gloo"])
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and two["value"] > 0 and "configs[4]" in two["config"]["workload"]
    assert len(two["losses"]) == 3 and all(np.isfinite(two["losses"]))
    assert two["roofline"]["frac"] > 0 and two["roofline"]["bound"] == "mfma"
    one = _run(common)
    assert one["n_gpus"] == 1 and len(one["losses"]) == 3 and all(np.isfinite(one["losses"]))


def test_bench_config_c3_wise_evaluate_rehearsal_with_a_ragged_shard():
    """`bench.py --config c3` (BASELINE configs[2], shrunk): `encoder=wise` through the evaluate loop in eval batches - 21 clips
    over two gloo ranks = 11 + 10, batches of 4 - one all-gather, ranks from the scoring epilogue; the retrieval metrics equal
    the one-rank run of the same command exactly (same seeded data per global clip is not required: the towers are random and
    both runs see their own shards; what must agree is that every one of the 21 clips is ranked once), and the one-rank line
    carries the roofline objects, the batch-256 key is absent at this size and the hipGraph leg replays bitwise."""
    common = ["bench.py", "--config", "c3", "--total-clips", "21", "--frames", "2", "--eval-batch", "4", "--steps", "1",
              "--warmup", "1", "--no-cpu-baseline"]
    two = _run(common + ["--gpus", "2", "--backend", "gloo"])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["retrieval"]["n"] == 21 and two["value"] > 0
    assert "configs[2]" in two["config"]["workload"] and two["config"]["eval_batch"] == 4
    one = _run(common)
    assert one["n_gpus"] == 1 and one["retrieval"]["n"] == 21
    assert one["roofline"]["frac"] > 0 and one["roofline_whole_path"]["frac"] > 0 and one["time_split"]["per_gemm"]
    assert one["hipgraph"]["bitwise_equal_to_eager"] is True and one["hipgraph"]["replay_ms"] > 0
    assert "eval_batch_256" not in one
    split = one["fp32_split_mode"]                     # the same epoch in fp32x3: a secondary leg of the one-rank line
    assert split["precision"] == "fp32x3" and split["value"] > 0 and split["metrics_identical_to_fp32_path"] is True
    assert max(split["embedding_max_abs_vs_fp32_path"].values()) < 2e-5 and "fp32_split_mode" not in two


def test_bench_config_c5_split_step_rehearsal():
    """`bench.py --config c5 --keep-clips 2 --micro-clips 3`: the KD step of a share that "does not fit" - 2 clips keep their
    activations, the other 8 are forwarded without and re-forwarded in micro-batches of 3; the losses equal the unsplit run's
    (the first one bitwise: same embeddings, same loss kernel; later ones to the AdamW noise of a split backward)."""
    common = ["bench.py", "--config", "c5", "--total-clips", "10", "--frames", "2", "--steps", "2", "--warmup", "1"]
    split = _run(common + ["--keep-clips", "2", "--micro-clips", "3"])
    plain = _run(common)
    assert split["split_step"] == {**split["split_step"], "kept_clips": 2, "micro_batch_clips": 3, "recomputed_clips": 8}
    assert plain["split_step"] is None
    assert split["losses"][0] == plain["losses"][0]
    assert all(abs(a - b) < 1e-4 * abs(b) for a, b in zip(split["losses"], plain["losses"]))
    assert split["roofline"]["executed_flops_per_step"] > split["roofline"]["flops_per_step"]
    # ... and on two ranks (5 + 5 clips): the gradient exchange happens once, in the last micro-batch's backward
    two_split = _run(common + ["--gpus", "2", "--backend", "gloo", "--keep-clips", "1", "--micro-clips", "2"])
    two_plain = _run(common + ["--gpus", "2", "--backend", "gloo"])
    assert two_split["split_step"]["recomputed_clips"] == 4 and two_plain["split_step"] is None
    assert two_split["losses"][0] == two_plain["losses"][0]
    assert all(abs(a - b) < 1e-4 * abs(b) for a, b in zip(two_split["losses"], two_plain["losses"]))


def test_bench_exits_nonzero_when_a_secondary_leg_fails():
    """A broken secondary leg must not hide behind rc 0: the headline line is still printed (with `failed_legs`), the exit
    code is 1.  The failure is injected from outside (an impossible tile for the bf16 leg only is not available, so the
    oracle import of the CPU leg is broken through PYTHONPATH shadowing)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = __import__("pathlib").Path(__file__).resolve().parent.parent
    with tempfile.TemporaryDirectory() as tmp:
        shadow = os.path.join(tmp, "sitecustomize.py")
        with open(shadow, "w") as f:  # makes `from oracle import clip_oracle` raise inside bench.py's cpu_baseline leg
            f.write("import sys, types\nm = types.ModuleType('oracle')\nm.__path__ = []\nsys.modules['oracle'] = m\n")
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        env["PYTHONPATH"] = tmp + os.pathsep + env.get("PYTHONPATH", "")
        res = subprocess.run([sys.executable, "bench.py", "--clips", "8", "--frames", "1", "--steps", "1", "--warmup", "1",
                              "--no-bf16-mode", "--no-split-mode", "--no-train-leg", "--cpu-sample-clips", "4"],
                             capture_output=True, text=True, timeout=900, env=env, cwd=str(root))
    assert res.returncode == 1, (res.returncode, res.stderr[-2000:])
    import json
    line = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert line["value"] > 0 and line["failed_legs"] == ["cpu_baseline"] and "error" in line["cpu_baseline"]


def test_debug_build_names_the_tower_that_produced_non_finite_values(tmp_path):
    """SURVEY.md section 5 (NaN / Inf self-check in debug builds): `python -m fitclip_amd.build --debug` builds
    tools/bin/libfitclip_hip_debug.so, whose tower calls scan their output on the device and fail with the call's name; the product
    library has no such scan (no allocation, no synchronisation on the hot path).  Run in child processes: the library is chosen
    once per process (FITCLIP_HIP_LIB)."""
    import os
    import subprocess
    import sys
    from fitclip_amd import build
    lib = build.DEBUG_LIB
    if not lib.exists():
        pytest.skip("tools/bin/libfitclip_hip_debug.so is not built (python -m fitclip_amd.build --debug)")
    child = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from fitclip_amd import synth, _lib
from fitclip_amd.clip_model import build_clip
d = synth.TINY
sd = synth.make_state_dict(d, seed=42)
frames = torch.from_numpy(synth.make_video(2, 2, d, seed=1)).reshape(-1, 3, d.image_resolution, d.image_resolution).cuda()
ids = torch.from_numpy(synth.make_text(3, d, seed=1)).cuda()
ok = build_clip(sd, precision="fp32", device="cuda")
assert torch.isfinite(ok.encode_image(frames)).all() and torch.isfinite(ok.encode_text(ids)).all()
bad = {k: np.array(v, copy=True) for k, v in sd.items()}
bad[sys.argv[1]][0] = np.nan
m = build_clip(bad, precision="fp32", device="cuda")
try:
    m.encode_image(frames) if sys.argv[1].startswith("visual") else m.encode_text(ids)
    print("NO ERROR")
except _lib.FitclipHipError as e:
    print("ERROR:", e)
''' % str(build.REPO)
    for key, call in (("visual.ln_post.bias", "fc_encode_image"), ("ln_final.bias", "fc_encode_text")):
        for path, expect_error in ((str(lib), True), (str(build.LIB), False)):
            res = subprocess.run([sys.executable, "-c", child, key], capture_output=True, text=True, timeout=600,
                                 env={**os.environ, "FITCLIP_HIP_LIB": path})
            assert res.returncode == 0, res.stderr[-2000:]
            last = res.stdout.strip().splitlines()[-1]
            if expect_error:
                assert last.startswith("ERROR:") and call in last and "NaN or Inf" in last, last
            else:
                assert last == "NO ERROR", last

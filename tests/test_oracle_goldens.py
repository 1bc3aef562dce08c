"""CPU: the oracle reproduces every golden fixture (which were themselves checked against the reference's own
`aligner.wise`, `aligner.loss`, `aligner.encoder.slip` classes and HF CLIP by tests/golden/make_goldens.py)."""
import json

import numpy as np
import pytest
import torch

from fitclip_amd import synth
from oracle import clip_oracle as O

TOL = 2e-5


def test_pinning_report_is_tight(golden_dir):
    rep = json.loads((golden_dir / "PINNING.json").read_text())
    assert rep["loss_vs_reference_aligner.loss_maxabs"] < 1e-5
    for tag in ("tiny", "vitb16"):
        d = rep[f"towers_{tag}_maxabs"]
        for k, v in d.items():
            if k != "output_abs_max":
                assert v < TOL * max(1.0, d["output_abs_max"]), (tag, k, v)


def test_wise_matches_reference_fixture(golden_dir):
    g = np.load(golden_dir / "wise_ref.npz")
    names = sorted(k[3:] for k in g.files if k.startswith("m1_"))
    sd1 = {n: torch.from_numpy(g[f"m1_{n}"]) for n in names}
    sd2 = {n: torch.from_numpy(g[f"m2_{n}"]) for n in names}
    for w in (0.0, 0.4, 0.5, 1.0):
        out = O.wise_state_dict(sd1, sd2, w)
        for n in names:
            assert np.array_equal(out[n].numpy(), g[f"w{w}_{n}"]), (w, n)


def test_wise_rejects_key_mismatch():
    with pytest.raises(AssertionError):
        O.wise_state_dict({"a": torch.zeros(1)}, {"b": torch.zeros(1)}, 0.5)


def test_losses_match_reference_fixture(golden_dir):
    g = np.load(golden_dir / "loss_ref.npz")
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files})
    assert tags
    for t in tags:
        s, te = torch.from_numpy(g[f"{t}_scores"]), torch.from_numpy(g[f"{t}_teacher"])
        assert abs(float(O.nce_loss(s)) - float(g[f"{t}_nce"])) < 1e-5
        assert abs(float(O.teacher_student_nce_loss(s, te)) - float(g[f"{t}_kd"])) < 1e-5


@pytest.mark.parametrize("tag,dims", [("tiny", synth.TINY), ("vitb16", synth.VIT_B_16)])
def test_towers_match_reference_slip_and_hf(golden_dir, tag, dims, request):
    g = np.load(golden_dir / f"towers_{tag}.npz")
    sd = O.to_torch(request.getfixturevalue(f"{tag}_state_dict"))
    video = torch.from_numpy(synth.make_video(int(g["n_clip"]), int(g["n_frames"]), dims, seed=int(g["seed"])))
    ids = torch.from_numpy(synth.make_text(int(g["n_text"]), dims, seed=int(g["seed"])))
    assert np.array_equal(ids.numpy(), g["ids"])
    with torch.inference_mode():
        img = O.encode_image(sd, video.reshape(-1, *video.shape[2:])).numpy()
        txt = O.encode_text_tokens(sd, ids).numpy()
        txt_rand = O.encode_text_tokens(sd, torch.from_numpy(g["ids_rand"])).numpy()
    for ref in ("slip", "hf"):
        assert np.abs(img - g[f"image_features_{ref}"]).max() < TOL * 4
        assert np.abs(txt - g[f"text_features_{ref}"]).max() < TOL * 4
    assert np.abs(txt_rand - g["text_features_rand_slip"]).max() < TOL * 4


def test_evaluate_tiny_golden(golden_dir, tiny_state_dict):
    g = np.load(golden_dir / "evaluate_tiny.npz")
    d = synth.TINY
    sd = O.to_torch(tiny_state_dict)
    n, f = int(g["n_clips"]), int(g["n_frames"])
    video = torch.from_numpy(synth.make_video(n, f, d, seed=42))
    ids = torch.from_numpy(synth.make_text(n, d, seed=42))
    with torch.inference_mode():
        ev, et = O.forward(sd, video, {"input_ids": ids})
    assert np.abs(ev.numpy() - g["encoded_videos"]).max() < TOL
    assert np.abs(et.numpy() - g["encoded_texts"]).max() < TOL
    scores = O.retrieval_scores(et, ev)
    m = O.retrieval_metrics(scores)
    for k in ("r1", "r5", "r10", "mr"):
        assert m[k] == pytest.approx(float(g[k]))
    assert abs(float(O.nce_loss(O.step_scores(ev, et, 0.015))) - float(g["loss_val"])) < 1e-3


def test_rank_known_answers_with_ties():
    # hand-built: row i's target is column i
    s = torch.tensor([[9., 1., 2., 3., 4.],     # target is the max           -> rank 0
                      [5., 5., 5., 5., 5.],     # all tied: one tie before it -> rank 1 (stable order)
                      [1., 2., 0., 4., 3.],     # 4 larger                    -> rank 4
                      [7., 3., 7., 7., 1.],     # two larger-or-tied-before   -> rank 2
                      [0., 0., 0., 0., 1.]])    # max                         -> rank 0
    r = O.ranks_of_target(s, torch.arange(5))
    assert r.tolist() == [0, 1, 4, 2, 0]
    m = O.retrieval_metrics(s)
    assert m["r1"] == pytest.approx(0.4) and m["r5"] == pytest.approx(1.0) and m["mr"] == 2.0


def test_median_rank_is_lower_middle_plus_one():
    s = torch.eye(4)
    s[2] = torch.tensor([0.9, 0.8, 0.1, 0.7])  # rank 3
    s[3] = torch.tensor([0.9, 0.8, 0.7, 0.75])  # rank 3
    m = O.retrieval_metrics(s)
    # ranks = [0, 0, 3, 3] -> torch.median lower middle = 0 -> +1
    assert m["mr"] == 1.0


def test_flatten_gathered_layout():
    t = torch.arange(2 * 3 * 4).view(2, 3, 4)
    assert torch.equal(O.flatten_gathered(t), t.view(6, 4))


def test_synth_is_deterministic_and_sliceable():
    a = synth.make_video(3, 2, synth.TINY, seed=5)
    b = synth.make_video(1, 2, synth.TINY, seed=5, first_clip=2)
    assert np.array_equal(a[2], b[0])
    t = synth.make_text(5, synth.TINY, seed=5)
    assert np.array_equal(t[3:], synth.make_text(2, synth.TINY, seed=5, first_text=3))
    eot = synth.TINY.vocab_size - 1
    assert (t.max(axis=1) == eot).all() and (t[:, 0] == eot - 1).all()
    assert len(synth.parameter_shapes(synth.VIT_B_16)) == 301
    assert sum(int(np.prod(s)) for s in synth.parameter_shapes(synth.VIT_B_16).values()) == 149_620_736


def test_bpe_tokenizer_matches_reference_fixture(golden_dir):
    """The C++ tokenizer (fc_bpe_* behind fitclip_amd.bpe) and the Python restatement under oracle/ vs ids produced by
    the reference's own SimpleTokenizer on the small synthetic merges file."""
    from fitclip_amd.bpe import ClipBpeTokenizer
    from oracle.bpe_oracle import ClipBpeTokenizer as OracleBpe
    g = json.loads((golden_dir / "bpe_toy.json").read_text())
    path = str(golden_dir / "bpe_toy_merges.txt.gz")
    for tok in (ClipBpeTokenizer(path, context_length=16), OracleBpe(path, context_length=16)):
        assert tok.vocab_size == g["vocab_size"]
        for text, ids, dec in zip(g["texts"], g["ids"], g["decoded"]):
            assert tok.encode(text) == ids, text
            assert tok.decode(ids) == dec
        out = tok(g["texts"])["input_ids"]
        sot, eot = g["vocab_size"] - 2, g["vocab_size"] - 1
        assert out.shape == (len(g["texts"]), 16) and (out[:, 0] == sot).all()
        long_row = out[4]  # 40 x's: truncated, EOT restored in the last slot (clip.tokenize(truncate=True))
        assert int(long_row[-1]) == eot and int(long_row.max()) == eot
        assert out[6][:3].tolist() == [sot, eot, 0]  # empty text


def test_decode_text_goes_through_the_bpe_decoder(golden_dir):
    """`ClipVideoTextEncoder.decode_text` (clip_video_text_encoder.py:100-103) = `clip._tokenizer.decode(ids)` per row: every id,
    SOT / EOT / padding included.  Rows tokenized by the plugin's own tokenizer decode to SOT + the reference's decoded
    string (fixture) + EOT + the pad token's text repeated; the batch mapping and per-instance mappings give the same."""
    from fitclip_amd import synth
    from fitclip_amd.clip_model import CLIP
    from fitclip_amd.encoder import ClipVideoTextEncoder
    from oracle.bpe_oracle import ClipBpeTokenizer as OracleBpe
    g = json.loads((golden_dir / "bpe_toy.json").read_text())
    path = str(golden_dir / "bpe_toy_merges.txt.gz")
    enc = ClipVideoTextEncoder(CLIP(synth.TINY), bpe_path=path)
    tokenizer = enc.get_tokenizer()
    assert enc.get_tokenizer() is tokenizer                     # one native handle per encoder
    L = enc.model.context_length
    keep = [i for i, ids in enumerate(g["ids"]) if len(ids) + 2 <= L]
    assert len(keep) >= 3
    batch = tokenizer([g["texts"][i] for i in keep])
    decoded = list(enc.decode_text(batch))
    oracle = OracleBpe(path, context_length=L)
    pad = oracle.decode([0])
    for row, i, text in zip(batch["input_ids"], keep, decoded):
        n_pad = L - 2 - len(g["ids"][i])
        assert text == "<|startoftext|>" + g["decoded"][i] + "<|endoftext|>" + pad * n_pad
        assert text == oracle.decode(row.tolist())
    assert list(enc.decode_text([{"input_ids": row} for row in batch["input_ids"]])) == decoded
    plain = ClipVideoTextEncoder(CLIP(synth.TINY))              # no vocabulary file: placeholders, padding dropped
    assert next(plain.decode_text({"input_ids": torch.tensor([[5, 9, 0, 0]])})) == "<5> <9>"


def test_bpe_tokenizer_at_full_vocabulary_size(golden_dir, tmp_path):
    """SURVEY 8(f) N2 at the real vocabulary size: a synthetic 48 894-merge file (regenerated from its seed; the
    published file is not available offline) -> SOT 49406 / EOT 49407, the 49152-256-2 cut, and ids identical to the
    reference's SimpleTokenizer (fixture) on 51 texts incl. contractions, digits, symbol runs, HTML entities, NBSP /
    control white space, long-s / Kelvin-sign case folding, the special tokens inside a text, and a 160-word text that
    must be cut to 77 with EOT in the last slot.  Random texts are cross-checked against the Python restatement."""
    import random
    import sys
    sys.path.insert(0, str(golden_dir))
    from synth_merges import write_synthetic_merges
    from fitclip_amd.bpe import ClipBpeTokenizer
    from oracle.bpe_oracle import ClipBpeTokenizer as OracleBpe
    g = json.loads((golden_dir / "bpe_full.json").read_text())
    path = str(tmp_path / "synthetic_full_merges.txt.gz")
    words = write_synthetic_merges(path, seed=g["merges_seed"])
    with __import__("gzip").open(path, "rt", encoding="utf-8") as f:
        lines = f.read().split("\n")
    assert len(lines) - 2 == g["n_merges"] == 48894
    tok = ClipBpeTokenizer(path, context_length=77)
    assert (tok.sot_token, tok.eot_token, tok.vocab_size) == (g["sot"], g["eot"], g["len_encoder"]) and g["sot"] == 49406
    for text, ids, dec in zip(g["texts"], g["ids"], g["decoded"]):
        assert tok.encode(text) == ids, text
        assert tok.decode(ids) == dec, text
    out = tok(g["texts"])["input_ids"]
    ref_call = torch.tensor(g["slip_call_context77"])          # slip's own __call__: cut at 77 WITHOUT restoring EOT
    assert out.shape == ref_call.shape == (len(g["texts"]), 77)
    for row, ref_row, ids in zip(out, ref_call, g["ids"]):
        if len(ids) + 2 <= 77:
            assert torch.equal(row, ref_row)
        else:                                                   # clip.tokenize(truncate=True): same, EOT in the last slot
            assert torch.equal(row[:76], ref_row[:76]) and int(row[76]) == 49407
    assert sum(len(ids) + 2 > 77 for ids in g["ids"]) >= 1
    # one more line in the file changes nothing (the cut at 49152 - 256 - 2 merges)
    longer = str(tmp_path / "longer.txt.gz")
    with __import__("gzip").open(longer, "wt", encoding="utf-8") as f:
        f.write("\n".join(lines[:-1] + ["q z", "z q"]) + "\n")
    tok2 = ClipBpeTokenizer(longer, context_length=77)
    assert (tok2.sot_token, tok2.eot_token) == (49406, 49407) and tok2.encode("qz zq") == tok.encode("qz zq")
    oracle = OracleBpe(path, context_length=77)
    rng = random.Random(11)
    pool = words[:3000] + ["it's", "they're", "12", "3.5", "!!", "...", "(", ")", "&amp;", "\u00e9t\u00e9", "\u65e5\u672c\u8a9e", "'ll", "I'M"]
    texts = [" ".join(rng.choice(pool) for _ in range(rng.randrange(0, 30))) for _ in range(300)]
    for t in texts:
        assert tok.encode(t) == oracle.encode(t), t
    assert torch.equal(tok(texts)["input_ids"], oracle(texts)["input_ids"])


def test_zero_shot_oracle_semantics():
    prompts = torch.tensor([[1., 0.], [0., 1.], [1., 1.], [3., 1.]])
    labels = O.zero_shot_label_embeddings(prompts, 2)
    assert torch.equal(labels, torch.tensor([[0.5, 0.5], [2., 1.]]))
    scores = torch.tensor([[0.9, 0.1, 0.3], [0.2, 0.2, 0.8]])
    m = O.zero_shot_metrics(scores, torch.tensor([0, 1]))
    assert m["a1"] == 0.5 and m["a5"] == 1.0 and m["mr"] == 1.0


@pytest.mark.parametrize("H,W", [(224, 224), (240, 320), (360, 202), (70, 64)])
def test_plugin_eval_transform_matches_the_transform_oracle(H, W):
    """The plugin's data-side `get_eval_transform` (torch on the CPU, what a dataset worker runs) vs the independent
    float64 restatement of the torchvision semantics in oracle/transform_oracle.py."""
    from fitclip_amd.encoder import CLIP_MEAN, CLIP_STD, ClipVideoTextEncoder
    from oracle.transform_oracle import eval_transform

    class _Visual:
        input_resolution = 64

    class _Model:
        visual = _Visual()

    enc = ClipVideoTextEncoder.__new__(ClipVideoTextEncoder)
    torch.nn.Module.__init__(enc)
    enc.__dict__["model"] = _Model()
    enc.mean, enc.std = CLIP_MEAN, CLIP_STD
    frames = torch.randint(0, 256, (2, H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(H + W))
    got = enc.get_eval_transform(torch.float32)(frames).numpy()
    want = eval_transform(frames.numpy(), 64, CLIP_MEAN, CLIP_STD)
    assert got.shape == want.shape and np.abs(got - want).max() < 2e-4  # float32 source coordinates on the torch side


def test_train_transform_is_a_random_resized_crop_with_flip():
    """`get_train_transform` (clip_video_text_encoder.py:113-122): output shape / dtype / normalisation, one crop box and one
    flip decision per CLIP (all frames move together), area in [0.5, 1] of the frame, reproducible under the global seed and
    different across seeds; a constant clip stays constant (crop + resize + flip of a constant image)."""
    from fitclip_amd import synth
    from fitclip_amd.clip_model import CLIP
    from fitclip_amd.encoder import CLIP_MEAN, CLIP_STD, ClipVideoTextEncoder
    enc = ClipVideoTextEncoder(CLIP(synth.TINY))
    size = enc.model.visual.input_resolution
    t = enc.get_train_transform(torch.float32)
    g = torch.Generator().manual_seed(0)
    clip = torch.randint(0, 256, (3, 90, 120, 3), generator=g, dtype=torch.uint8)
    clip[1:] = clip[:1]                                        # identical frames: identical outputs iff one box per clip
    torch.manual_seed(7)
    a = t(clip)
    torch.manual_seed(7)
    b = t(clip)
    torch.manual_seed(8)
    c = t(clip)
    assert a.shape == (3, 3, size, size) and a.dtype == torch.float32
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(a[0], a[1]) and torch.equal(a[0], a[2])
    flat = torch.full((2, 50, 70, 3), 128, dtype=torch.uint8)
    out = t(flat)
    want = (128 / 255 - torch.tensor(CLIP_MEAN)) / torch.tensor(CLIP_STD)
    assert torch.allclose(out, want.view(1, 3, 1, 1).expand_as(out), atol=1e-5)
    # a horizontal ramp: the output's left-to-right span tells the crop width (>= sqrt(0.5 * 3/4) of the frame) and the flip
    ramp = torch.arange(200, dtype=torch.float32).view(1, 1, 200, 1).expand(1, 150, 200, 3).contiguous() / 199
    spans, flips = [], 0
    for seed in range(40):
        torch.manual_seed(seed)
        o = t(ramp)[0, 0] * CLIP_STD[0] + CLIP_MEAN[0]
        span = float(o[0, -1] - o[0, 0])
        flips += span < 0
        spans.append(abs(span))
    assert 0.50 <= min(spans) and max(spans) <= 1.0 + 1e-5       # narrowest box: sqrt(0.5 * 3/4) = 0.61 of the height-limited width .. 0.53 here
    assert 8 <= flips <= 32                                     # p = 0.5 over 40 draws

"""Checkpoint formats either side of the path (CPU only): Lightning `.ckpt` -> prefix-stripped state dict, the
missing-`logit_scale` case, pipes, teacher-key filtering.  Mirrors `util/checkpoint_utils.py:9-12`,
`scripts/checkpoint_to_state_dict.py`, `clip_video_text_encoder.py:30-61`, `text_video_retrieval.py:101-131`."""
import io
import math
import os
import subprocess
import sys
import threading

import pytest
import torch

from fitclip_amd import checkpoint as C
from fitclip_amd import synth
from fitclip_amd.clip_model import build_clip, load_clip_model
from fitclip_amd.encoder import ClipVideoTextEncoder
from fitclip_amd.retrieval import TeacherStudentModule, TextVideoRetrievalModule

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _encoder(seed):
    return ClipVideoTextEncoder(build_clip(synth.make_state_dict(synth.TINY, seed=seed), precision="fp32"), num_frames=2)


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert torch.equal(torch.as_tensor(a[k]), torch.as_tensor(b[k])), k


def test_prefix_strip_adds_missing_dot(tmp_path):
    ckpt = {"state_dict": {"encoder.model.a": torch.ones(2), "encoder.model.b.c": torch.zeros(1),
                           "encoder.modelx": torch.ones(1), "logit_scale": torch.tensor(3.0)}}
    path = tmp_path / "m.ckpt"
    torch.save(ckpt, path)
    for prefix in ("encoder.model", "encoder.model."):
        assert sorted(C.state_dict_from_checkpoint_path(path, prefix)) == ["a", "b.c"]
    assert sorted(C.state_dict_from_checkpoint_path(path)) == sorted(ckpt["state_dict"])  # empty prefix: everything


def test_module_checkpoint_round_trip_and_cli(tmp_path):
    student, teacher = _encoder(1), _encoder(2)
    module = TeacherStudentModule(student, teacher, init_temperature=0.07)
    module.teacher_student_logit_scale = 1.25
    path = tmp_path / "ts.ckpt"
    C.save_checkpoint(module, path, epoch=3)
    sd = torch.load(path, weights_only=False)["state_dict"]
    assert "encoder.model.visual.conv1.weight" in sd and "teacher.model.ln_final.bias" in sd
    assert float(sd["teacher_student_logit_scale"]) == 1.25

    # the conversion script: stdout carries a bare OpenAI-named state dict of the STUDENT
    out = subprocess.run([sys.executable, "-m", "fitclip_amd.checkpoint", str(path)], cwd=REPO, check=True,
                         capture_output=True).stdout
    bare = torch.load(io.BytesIO(out), weights_only=False)
    _same(bare, student.model.state_dict())
    out = subprocess.run([sys.executable, "-m", "fitclip_amd.checkpoint", str(path), "--prefix", "teacher.model"],
                         cwd=REPO, check=True, capture_output=True).stdout
    _same(torch.load(io.BytesIO(out), weights_only=False), teacher.model.state_dict())

    # load_clip_model accepts both the bare file and the Lightning checkpoint
    bare_path = tmp_path / "student.pt"
    torch.save(bare, bare_path)
    # (the encoder drops CLIP's own `logit_scale`, so the files lack it and the loader re-creates it as NaN)
    for p in (bare_path, path):
        loaded = dict(load_clip_model(str(p), precision="fp32").state_dict())
        assert math.isnan(float(loaded.pop("logit_scale")))
        _same(loaded, student.model.state_dict())


def test_missing_logit_scale_becomes_nan(tmp_path):
    sd = {k: v for k, v in build_clip(synth.make_state_dict(synth.TINY, seed=3), precision="fp32").state_dict().items()
          if k != "logit_scale"}
    path = tmp_path / "no_scale.pt"
    torch.save(sd, path)
    model = load_clip_model(str(path), precision="fp32")
    assert math.isnan(float(model.logit_scale))  # clip_video_text_encoder.py:43-53


def test_pipe_is_accepted(tmp_path):
    sd = _encoder(4).model.state_dict()
    fifo = tmp_path / "fifo"
    os.mkfifo(fifo)
    buf = io.BytesIO()
    torch.save(sd, buf)

    def writer():
        with open(fifo, "wb") as f:
            f.write(buf.getvalue())

    t = threading.Thread(target=writer)
    t.start()
    model = load_clip_model(str(fifo), precision="fp32")
    t.join()
    loaded = dict(model.state_dict())
    assert math.isnan(float(loaded.pop("logit_scale")))
    _same(loaded, sd)


def test_remote_names_are_rejected():
    with pytest.raises(FileNotFoundError):
        load_clip_model("https://example.invalid/model.pt")
    with pytest.raises(FileNotFoundError):
        load_clip_model("ViT-B/16")


def test_retrieval_module_ignores_teacher_keys_only():
    student, teacher = _encoder(5), _encoder(6)
    ts = TeacherStudentModule(student, teacher)
    ts.logit_scale = 2.5
    sd = ts.state_dict()

    plain = TextVideoRetrievalModule(_encoder(7))
    res = plain.load_state_dict(sd)  # strict: teacher.* and teacher_student_logit_scale are dropped silently
    assert res.missing_keys == [] and res.unexpected_keys == []
    assert plain.logit_scale == 2.5
    _same(plain.encoder.state_dict(), student.state_dict())

    bad = dict(sd)
    bad["something_else"] = torch.zeros(1)
    del bad["encoder.model.ln_final.weight"]
    with pytest.raises(RuntimeError) as e:
        plain.load_state_dict(bad)
    msg = str(e.value)
    assert msg.startswith("Error(s) in loading state_dict for TextVideoRetrievalModule:")
    assert 'Unexpected key(s) in state_dict: "something_else". ' in msg
    assert 'Missing key(s) in state_dict: "encoder.model.ln_final.weight". ' in msg
    res = plain.load_state_dict(bad, strict=False)
    assert res.unexpected_keys == ["something_else"] and res.missing_keys == ["encoder.model.ln_final.weight"]

    # the distillation module itself does NOT ignore them: a plain checkpoint lacks its teacher
    with pytest.raises(RuntimeError, match="Missing key"):
        TeacherStudentModule(_encoder(8), _encoder(9)).load_state_dict(plain.state_dict())


def test_prepare_tools(tmp_path):
    """`prepare-clip` (bare state dict + NaN logit_scale) and `prepare` (checkpoint kept, state_dict stripped):
    scripts/prepare_trained_clip_checkpoint_for_evaluation.py, scripts/prepare_trained_checkpoint_for_evaluation.py."""
    student, teacher = _encoder(11), _encoder(12)
    path = tmp_path / "ts.ckpt"
    C.save_checkpoint(TeacherStudentModule(student, teacher), path, epoch=7, optimizer_states=["opaque"])
    out1, out2 = tmp_path / "clip.pt", tmp_path / "stripped.ckpt"
    C.main(["prepare-clip", str(path), str(out1)])
    sd = torch.load(out1, weights_only=False)
    assert math.isnan(float(sd.pop("logit_scale")))
    _same(sd, student.model.state_dict())
    _same({k: v for k, v in load_clip_model(str(out1), precision="fp32").state_dict().items() if k != "logit_scale"},
          student.model.state_dict())
    C.main(["prepare", str(path), str(out2), "--prefix", "teacher.model"])
    ck = torch.load(out2, weights_only=False)
    assert ck["epoch"] == 7 and ck["optimizer_states"] == ["opaque"]
    _same(ck["state_dict"], teacher.model.state_dict())


def test_open_clip_checkpoint_conversion(tmp_path):
    """scripts/open_clip_checkpoint_to_model.py: the prefix of the first key (`module.` under DDP, `model.` otherwise)
    is cut from every key."""
    for prefix in ("module", "model"):
        src, dst = tmp_path / f"{prefix}.pt", tmp_path / f"{prefix}_out.pt"
        torch.save({"epoch": 3, "state_dict": {f"{prefix}.visual.proj": torch.ones(2), f"{prefix}.logit_scale": torch.tensor(4.6)}}, src)
        C.main(["open-clip", str(src), str(dst)])
        out = torch.load(dst, weights_only=False)
        assert sorted(out) == ["logit_scale", "visual.proj"] and float(out["logit_scale"]) == pytest.approx(4.6)
    bad = tmp_path / "bad.pt"
    torch.save({"state_dict": {"encoder.x": torch.zeros(1)}}, bad)
    with pytest.raises(StopIteration):
        C.open_clip_checkpoint_to_model(bad, tmp_path / "never.pt")


def _reference_shaped_module(encoder, teacher, prompt_ids=None, fit_temperature=True):
    """A torch `nn.Module` whose attributes are registered in the order the reference's constructors register them
    (`VideoTextLightningModule.__init__`, video_text_module.py:28-35: encoder, logit_scale, max_logit_scale, loss;
    `TeacherStudentLightningModule.__init__`, teacher_student.py:52-91: teacher, teacher_student_logit_scale, frozen
    teacher, the two prompt `nn.ParameterDict`s) - so `state_dict()`, `parameters()` and `torch.optim.AdamW(parameters())`
    give the key order, shapes and parameter numbering of a reference checkpoint.  (The Lightning classes themselves are
    not importable here; nothing of them is copied - only the registration order is restated.)"""
    from torch import nn

    class ReferenceShaped(nn.Module):
        def __init__(self):
            super().__init__()
            self.encoder = encoder
            for p in self.encoder.parameters():   # a `clip.load`ed model is trainable (this repo's CLIP holds plain buffers-
                p.requires_grad = True            # like parameters: the HIP trainer owns the gradients)
            self.logit_scale = nn.Parameter(torch.tensor([-math.log(0.05)]), requires_grad=fit_temperature)
            self.max_logit_scale = nn.Parameter(torch.tensor([-math.log(0.001)]), requires_grad=False)
            self.loss = nn.Identity()
            self.metrics = nn.ModuleDict()
            self.teacher = teacher
            self.teacher_student_logit_scale = nn.Parameter(self.logit_scale.clone(), requires_grad=fit_temperature)
            self.teacher_student_loss = nn.Identity()
            for p in self.teacher.parameters():
                p.requires_grad = False
            if prompt_ids is not None:
                self.tokenized_prompts = nn.ParameterDict({"input_ids": nn.Parameter(prompt_ids, requires_grad=False)})
                self.teacher_tokenized_prompts = nn.ParameterDict({"input_ids": nn.Parameter(prompt_ids.clone(), requires_grad=False)})

    return ReferenceShaped()


@pytest.mark.parametrize("with_prompts,fit_temperature", [(False, True), (True, True), (False, False)])
def test_checkpoint_layout_is_the_reference_modules(with_prompts, fit_temperature):
    """`module_state_dict` / `reference_parameter_order` / `trainable_parameter_indices` against torch's own bookkeeping on a
    module with the reference's registration order: same state-dict keys in the same order with the same shapes, the same
    `parameters()` numbering, and optimiser `state` entries for exactly the parameters torch.optim.AdamW creates them for.
    And the other direction: a state dict of that module loads (strict) into this repo's module."""
    prompt_ids = torch.arange(3 * synth.TINY.context_length).view(3, -1) % 7 if with_prompts else None
    ref = _reference_shaped_module(_encoder(21), _encoder(22), prompt_ids, fit_temperature)
    ours = TeacherStudentModule(_encoder(23), _encoder(24), init_temperature=0.3)
    if with_prompts:
        ours.tokenized_prompts, ours.teacher_tokenized_prompts = prompt_ids + 1, prompt_ids + 1
    ref_sd = ref.state_dict()
    sd = C.module_state_dict(ours)
    assert list(sd) == list(ref_sd)
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: tuple(v.shape) for k, v in ref_sd.items()}
    assert "encoder.model.logit_scale" not in sd and tuple(sd["logit_scale"].shape) == (1,)
    assert C.reference_parameter_order(ours) == [n for n, _ in ref.named_parameters()]

    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3)
    sum((p.float() ** 2).sum() for p in ref.parameters() if p.requires_grad).backward()
    opt.step()
    opt_sd = opt.state_dict()
    indices = C.trainable_parameter_indices(ours, fit_temperature)
    assert sorted(opt_sd["state"]) == list(indices)
    assert opt_sd["param_groups"][0]["params"] == list(range(len(C.reference_parameter_order(ours))))
    names = [n for n, _ in ref.named_parameters()]
    assert all(names[i] == key for i, key in indices.items())

    res = C.load_module_state_dict(ours, ref_sd, strict=True)
    assert res.missing_keys == [] and res.unexpected_keys == []
    assert ours.logit_scale == pytest.approx(float(ref.logit_scale)) and ours.max_logit_scale == pytest.approx(-math.log(0.001))
    _same(ours.encoder.state_dict(), ref.encoder.state_dict())
    _same(ours.teacher.state_dict(), ref.teacher.state_dict())
    if with_prompts:
        assert torch.equal(ours.tokenized_prompts, prompt_ids) and torch.equal(ours.teacher_tokenized_prompts, prompt_ids)
    else:  # a module built without prompts rejects a checkpoint that has them, as nn.Module.load_state_dict does
        extra = dict(ref_sd)
        extra["tokenized_prompts.input_ids"] = torch.zeros(2, 4)
        with pytest.raises(RuntimeError, match='Unexpected key.*tokenized_prompts.input_ids'):
            C.load_module_state_dict(ours, extra, strict=True)
    # and the reference module accepts what this repo writes
    ref.load_state_dict(C.module_state_dict(ours), strict=True)

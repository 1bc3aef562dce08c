"""GPU parity of the KD training step (`aligner/teacher_student.py:99-183`, `aligner/loss.py:13-39`,
`aligner/video_text_module.py:94-97`, `aligner/cli.py:129`): every backward kernel against torch autograd on the CPU,
the gradients of every parameter of a whole student against autograd through the oracle (which is pinned to autograd
through the reference's own slip / loss classes: tests/golden/training_ref_tiny.npz), and optimiser steps against
`torch.optim.AdamW`.  fp32 mode; tolerance: max|g - g_ref| <= 1e-4 * max|g_ref| per tensor."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from fitclip_amd import _lib, ops, synth  # noqa: E402
from fitclip_amd.clip_model import build_clip  # noqa: E402
from fitclip_amd.encoder import ClipVideoTextEncoder  # noqa: E402
from fitclip_amd.training import StudentTrainer, TeacherStudentTrainer  # noqa: E402
from oracle import clip_oracle as O  # noqa: E402

DEV = "cuda"
GRAD_TOL = 1e-4


def _rel(got: torch.Tensor, ref: torch.Tensor) -> float:
    return float((got.double().cpu() - ref.double()).abs().max() / ref.double().abs().max().clamp_min(1e-30))


def _stream():
    return _lib.current_stream()


# ------------------------------------------------------------------------------------------------- single kernels
@pytest.mark.parametrize("M,N1,N2", [(5, 128, 128), (1000, 384, 128), (4097, 768, 3072), (3000, 3072, 768), (64, 512, 512)])
def test_gemm_tn_matches_float64(M, N1, N2):
    g = torch.Generator().manual_seed(M + N1)
    a, b = torch.randn(M, N1, generator=g), torch.randn(M, N2, generator=g)
    ref = a.double().T @ b.double()
    got = ops.gemm_tn(a.to(DEV), b.to(DEV))
    assert _rel(got, ref) < 2e-6 * math.sqrt(M)
    base = torch.randn(N1, N2, generator=g)
    acc = ops.gemm_tn(a.to(DEV), b.to(DEV), alpha=0.5, out=base.to(DEV).clone(), beta=1.0)
    assert _rel(acc, base.double() + 0.5 * ref) < 2e-6 * math.sqrt(M)
    assert torch.equal(ops.gemm_tn(a.to(DEV), b.to(DEV)), got)  # fixed reduction order: bit-reproducible


@pytest.mark.parametrize("M,N,K", [(300, 512, 128), (2000, 1024, 256), (50432, 3072, 768)])
def test_dgelu_epilogue(M, N, K):
    """dX = (dY . W) * quickgelu'(pre): the fused backward of c_proj's input (both GEMM kernels: small and pipelined)."""
    g = torch.Generator().manual_seed(N)
    dy, wt, pre = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * K ** -0.5, torch.randn(M, N, generator=g) * 2
    s = torch.sigmoid(1.702 * pre.double())
    ref = (dy.double() @ wt.double().T) * (s * (1 + 1.702 * pre.double() * (1 - s)))
    zeros = torch.zeros(N, device=DEV)
    got = ops.gemm(dy.to(DEV), wt.to(DEV), zeros, epilogue=_lib.EPI_DGELU_T, aux=pre.to(DEV))
    assert _rel(got, ref) < 1e-5


@pytest.mark.parametrize("n_seq,S,heads,causal", [(3, 17, 4, False), (2, 16, 2, True), (3, 77, 8, True), (2, 197, 12, False),
                                                 (1, 50, 3, False), (2, 224, 2, False)])
def test_attention_backward_matches_autograd(n_seq, S, heads, causal):
    D = heads * 64
    g = torch.Generator().manual_seed(S * heads)
    qkv = torch.randn(n_seq * S, 3 * D, generator=g)
    d_out = torch.randn(n_seq * S, D, generator=g)
    x = qkv.double().requires_grad_(True)
    q, k, v = (t.view(n_seq, S, heads, 64).transpose(1, 2) for t in x.split(D, dim=-1))
    s = (q / 8.0) @ k.transpose(-1, -2)
    if causal:
        s = s + O.causal_mask(S).double()
    o = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(n_seq * S, D)
    o.backward(d_out.double())
    qkv_d, dout_d = qkv.to(DEV), d_out.to(DEV)
    out = ops.attention(qkv_d, n_seq, S, heads, causal=causal)
    assert _rel(out, o.detach()) < 1e-5
    dqkv = torch.full_like(qkv_d, float("nan"))
    _lib.check(_lib.load().fc_attention_backward(_lib.PREC_F32, qkv_d.data_ptr(), out.data_ptr(), dout_d.data_ptr(),
                                                 dqkv.data_ptr(), n_seq, S, heads, int(causal), _stream()))
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _rel(dqkv[:, sl], x.grad[:, sl]) < 2e-5, name


@pytest.mark.parametrize("rows,D", [(7, 128), (1000, 256), (333, 512), (4100, 768), (64, 1024)])
def test_layernorm_backward_matches_autograd(rows, D):
    g = torch.Generator().manual_seed(D)
    x, dy = torch.randn(rows, D, generator=g) * 3 + 1, torch.randn(rows, D, generator=g)
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    xd, gd, bd = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    F.layer_norm(xd, (D,), gd, bd, 1e-5).backward(dy.double())
    lib = _lib.load()
    dx = torch.ones(rows, D, device=DEV)
    dgam, dbet = torch.empty(D, device=DEV), torch.empty(D, device=DEV)
    scratch = torch.empty(lib.fc_layernorm_backward_scratch_bytes(D), dtype=torch.uint8, device=DEV)
    x_d, dy_d, gamma_d = x.to(DEV), dy.to(DEV), gamma.to(DEV)  # named: a temporary's memory is reused by the next one
    _lib.check(lib.fc_layernorm_backward(x_d.data_ptr(), dy_d.data_ptr(), gamma_d.data_ptr(), dx.data_ptr(), 1,
                                         rows, D, dgam.data_ptr(), dbet.data_ptr(), scratch.data_ptr(), scratch.numel(), _stream()))
    assert _rel(dx - 1.0, xd.grad) < 2e-5     # accumulate = 1: added to what was there
    assert _rel(dgam, gd.grad) < 2e-5 and _rel(dbet, bd.grad) < 2e-5


@pytest.mark.parametrize("n,scale", [(8, 1.0), (37, 20.0), (512, 20.0)])
def test_loss_backward_matches_autograd(n, scale):
    g = torch.Generator().manual_seed(n)
    s, t = torch.randn(n, n, generator=g) * scale, torch.randn(n, n, generator=g) * scale
    lib = _lib.load()
    sd, td = s.to(DEV), t.to(DEV)
    ds, ws = torch.empty_like(sd), torch.empty(6 * n, device=DEV)
    x = s.double().requires_grad_(True)
    O.nce_loss(x).backward()
    _lib.check(lib.fc_nce_loss_backward(sd.data_ptr(), n, 0.25, ds.data_ptr(), ws.data_ptr(), _stream()))
    assert _rel(ds, 0.25 * x.grad) < 2e-5
    x, y = s.double().requires_grad_(True), t.double().requires_grad_(True)
    O.teacher_student_nce_loss(x, y).backward()
    _lib.check(lib.fc_kd_loss_backward(sd.data_ptr(), td.data_ptr(), n, n, 3.0, ds.data_ptr(), ws.data_ptr(), _stream()))
    assert _rel(ds, 3.0 * x.grad) < 2e-5
    out = torch.empty(1, device=DEV)
    _lib.check(lib.fc_kd_teacher_scale_grad(sd.data_ptr(), td.data_ptr(), n, n, out.data_ptr(), ws.data_ptr(), _stream()))
    assert abs(float(out) - float((y.grad * t.double()).sum())) < 1e-4 * max(1.0, float((y.grad * t.double()).abs().sum()))


@pytest.mark.parametrize("rows,cols", [(12, 5), (5, 12), (64, 300)])
def test_rectangular_kd_loss_and_backward(rows, cols):
    """The videos x prompts variant (teacher_student.py:111-138): KD on [rows, cols] score matrices, where each
    direction's "batchmean" divides by its own number of lines (F.kl_div(input) / input.size(0), loss.py:29-39)."""
    g = torch.Generator().manual_seed(rows * cols)
    s, t = torch.randn(rows, cols, generator=g) * 5, torch.randn(rows, cols, generator=g) * 5
    x, y = s.double().requires_grad_(True), t.double().requires_grad_(True)
    ref = O.teacher_student_nce_loss(x, y)
    ref.backward()
    sd, td = s.to(DEV), t.to(DEV)
    assert abs(float(ops.teacher_student_nce_loss(sd, td)) - float(ref.detach())) < 1e-5 * max(1.0, abs(float(ref.detach())))
    lib = _lib.load()
    ds, ws = torch.empty_like(sd), torch.empty(3 * (rows + cols), device=DEV)
    _lib.check(lib.fc_kd_loss_backward(sd.data_ptr(), td.data_ptr(), rows, cols, 2.0, ds.data_ptr(), ws.data_ptr(), _stream()))
    assert _rel(ds, 2.0 * x.grad) < 2e-5
    out = torch.empty(1, device=DEV)
    _lib.check(lib.fc_kd_teacher_scale_grad(sd.data_ptr(), td.data_ptr(), rows, cols, out.data_ptr(), ws.data_ptr(), _stream()))
    assert abs(float(out) - float((y.grad * t.double()).sum())) < 1e-4 * max(1.0, float((y.grad * t.double()).abs().sum()))


def test_pool_normalize_backward_matches_autograd():
    g = torch.Generator().manual_seed(3)
    z, dout = torch.randn(6 * 4, 128, generator=g), torch.randn(6, 128, generator=g)
    x = z.double().requires_grad_(True)
    ((x / x.norm(dim=-1, keepdim=True)).view(6, 4, 128).mean(1)).backward(dout.double())
    dz = torch.empty(24, 128, device=DEV)
    z_d, dout_d = z.to(DEV), dout.to(DEV)
    _lib.check(_lib.load().fc_pool_normalize_backward(z_d.data_ptr(), dout_d.data_ptr(), dz.data_ptr(), 6, 4, 128, _stream()))
    assert _rel(dz, x.grad) < 1e-5


def test_adamw_matches_torch_optimizer():
    g = torch.Generator().manual_seed(1)
    n = 10007
    p0 = torch.randn(n, generator=g)
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p = torch.zeros(10008, device=DEV)
    p[:n] = p0.to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (10.0 ** -step)
        ref.grad = grad.clone()
        opt.step()
        gd = torch.zeros_like(p)
        gd[:n] = grad.to(DEV)
        _lib.check(_lib.load().fc_adamw(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), n, 3e-3, 0.9, 0.999, 1e-8, 0.01,
                                        step, _stream()))
        assert (p[:n].cpu() - ref.detach()).abs().max() < 2e-7, step
    assert float(p[n]) == 0.0  # nothing past n is touched


# ----------------------------------------------------------------------------------------------- whole student
def _oracle_step(student_np, teacher_np, video, ids, n_lab, temp, share):
    """One KD training step on the CPU: autograd through the oracle (teacher_student.py:142-176)."""
    sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch(student_np).items()}
    ls = torch.tensor([-math.log(temp)], dtype=torch.float32, requires_grad=True)
    ts = torch.tensor([-math.log(temp)], dtype=torch.float32, requires_grad=True)
    with torch.no_grad():
        tv, tt = O.forward(O.to_torch(teacher_np), video, {"input_ids": ids})
    ev, et = O.forward(sd, video, {"input_ids": ids})
    loss, parts = O.teacher_student_training_loss(
        {"labeled": (ev[:n_lab], et[:n_lab]), "unlabeled": (ev[n_lab:], et[n_lab:])},
        {"labeled": (tv[:n_lab], tt[:n_lab]), "unlabeled": (tv[n_lab:], tt[n_lab:])}, ls, ts, share)
    loss.backward()
    return sd, ls, ts, loss, parts, (ev.detach(), et.detach())


def _batch(video, ids, n_lab):
    n = video.shape[0]
    return {"video_student": video.to(DEV), "text_student": {"input_ids": ids.to(DEV)}, "video_teacher": video.to(DEV),
            "text_teacher": {"input_ids": ids.to(DEV)}, "dataset": ["labeled"] * n_lab + ["unlabeled"] * (n - n_lab)}


def _trainer(student_np, teacher_np, temp, **kw):
    student = ClipVideoTextEncoder(build_clip(student_np, precision="fp32", device=DEV))
    teacher = ClipVideoTextEncoder(build_clip(teacher_np, precision="fp32", device=DEV))
    return TeacherStudentTrainer(student, teacher, init_temperature=temp, **kw)


def test_every_parameter_gradient_of_the_tiny_student(golden_dir, tiny_state_dict):
    """All 53 parameter tensors of the tiny student: HIP backward vs autograd through the oracle, vs the digest of
    autograd through the REFERENCE's slip / loss classes (fixture), plus both temperatures."""
    fx = np.load(golden_dir / "training_ref_tiny.npz")
    d = synth.TINY
    n, f, n_lab, temp = int(fx["n"]), int(fx["f"]), int(fx["n_labeled"]), float(fx["temperature"])
    student_np = synth.perturbed_state_dict(tiny_state_dict, d, seed=int(fx["student_seed"]), rel=float(fx["rel"]))
    video = torch.from_numpy(synth.make_video(n, f, d, seed=int(fx["video_seed"])))
    ids = torch.from_numpy(synth.make_text(n, d, seed=int(fx["video_seed"])))
    share = {"labeled": 0.5, "unlabeled": 0.5}
    sd, ls, ts, loss, parts, (ev_ref, et_ref) = _oracle_step(student_np, tiny_state_dict, video, ids, n_lab, temp, share)
    module = _trainer(student_np, tiny_state_dict, temp)
    out = module.training_step(_batch(video, ids, n_lab))
    sv = torch.cat([out["labeled"][0][0], out["unlabeled"][0][0]])
    assert (sv.cpu() - ev_ref).abs().max() < 2e-5   # the training forward is the parity-grade forward
    got_loss = module.training_step_end(out)
    assert abs(got_loss - float(loss)) < 1e-4 * abs(float(loss))
    assert abs(module.last_losses["labeled"] - float(fx["loss_labeled"])) < 1e-4 * float(fx["loss_labeled"])
    assert abs(module.last_losses["unlabeled"] - float(fx["loss_unlabeled"])) < 1e-4 * float(fx["loss_unlabeled"])
    module.backward()
    grads = {k: p.grad for k, p in module.encoder.model.named_parameters()}
    assert set(grads) == set(sd)
    worst = {}
    for k, p in sd.items():
        worst[k] = _rel(grads[k], p.grad)
        assert worst[k] < GRAD_TOL, (k, worst[k])
        assert abs(float(grads[k].double().norm()) - float(fx[f"norm/{k}"])) < 2e-4 * float(fx[f"norm/{k}"]), k
        if f"grad/{k}" in fx:
            assert _rel(grads[k], torch.from_numpy(fx[f"grad/{k}"])) < 2e-4, k
    print("worst relative gradient error:", max(worst.items(), key=lambda kv: kv[1]))
    assert abs(float(module.scale_grads[0]) - float(ls.grad)) < 1e-4 * abs(float(ls.grad))
    assert abs(float(module.scale_grads[1]) - float(ts.grad)) < 1e-4 * abs(float(ts.grad))


def test_one_vit_b16_block_gradients():
    """Full ViT-B/16 widths, token counts, vocabulary and head counts with ONE block per tower (what the CPU oracle can
    differentiate in seconds): the 197-token / 12-head and 77-token / 8-head causal attention backward, the 768- and
    512-wide LayerNorm / wgrad / dgrad shapes, patch-embed and embedding gradients at their real sizes."""
    d = synth.ClipDims(vision_layers=1, transformer_layers=1)
    teacher_np = synth.make_state_dict(d, seed=42)
    student_np = synth.perturbed_state_dict(teacher_np, d, seed=6, rel=0.2)
    n, f, n_lab, temp = 6, 2, 3, 0.05
    video = torch.from_numpy(synth.make_video(n, f, d, seed=4))
    ids = torch.from_numpy(synth.make_text(n, d, seed=4))
    share = {"labeled": 0.3, "unlabeled": 0.7}
    sd, ls, ts, loss, parts, _ = _oracle_step(student_np, teacher_np, video, ids, n_lab, temp, share)
    module = _trainer(student_np, teacher_np, temp, labeled_dataset_loss_share=0.3)
    got_loss = module.training_step_end(module.training_step(_batch(video, ids, n_lab)))
    assert abs(got_loss - float(loss)) < 1e-4 * abs(float(loss))
    module.backward()
    for k, p in module.encoder.model.named_parameters():
        assert _rel(p.grad, sd[k].grad) < GRAD_TOL, k


def test_optimizer_steps_match_torch_adamw(tiny_state_dict):
    """Two full steps (forward, loss, backward, AdamW, temperature clamp) vs autograd + torch.optim.AdamW on the CPU
    (cli.py:129 hands `self.parameters()` - encoder and temperatures - to one AdamW).  A large lr makes the update
    visible in fp32; the teacher stays frozen."""
    d = synth.TINY
    student_np = synth.perturbed_state_dict(tiny_state_dict, d, seed=5, rel=0.3)
    n, f, n_lab, temp, lr = 8, 2, 4, 0.05, 1e-3
    share = {"labeled": 0.5, "unlabeled": 0.5}
    module = _trainer(student_np, tiny_state_dict, temp, lr=lr)
    teacher_before = {k: p.detach().clone() for k, p in module.teacher.model.named_parameters()}
    ref_sd = {k: torch.nn.Parameter(v.clone()) for k, v in O.to_torch(student_np).items()}
    ref_ls = torch.nn.Parameter(torch.tensor([-math.log(temp)], dtype=torch.float32))
    ref_ts = torch.nn.Parameter(torch.tensor([-math.log(temp)], dtype=torch.float32))
    opt = torch.optim.AdamW([*ref_sd.values(), ref_ls, ref_ts], lr=lr)
    with torch.no_grad():
        t_sd = O.to_torch(tiny_state_dict)
    # AdamW's first steps move every weight by ~lr * sign(gradient) whatever the gradient's size, so where the true
    # gradient is ZERO (the key third of every in_proj_bias: softmax ignores a shift common to all keys) the update is
    # the sign of rounding noise.  Elements are compared only while their gradient has been well above noise level.
    solid = {k: torch.ones_like(v.data, dtype=torch.bool) for k, v in ref_sd.items()}
    for step in range(2):
        video = torch.from_numpy(synth.make_video(n, f, d, seed=20 + step))
        ids = torch.from_numpy(synth.make_text(n, d, seed=20 + step))
        with torch.no_grad():
            tv, tt = O.forward(t_sd, video, {"input_ids": ids})
        ev, et = O.forward(ref_sd, video, {"input_ids": ids})
        loss, _ = O.teacher_student_training_loss(
            {"labeled": (ev[:n_lab], et[:n_lab]), "unlabeled": (ev[n_lab:], et[n_lab:])},
            {"labeled": (tv[:n_lab], tt[:n_lab]), "unlabeled": (tv[n_lab:], tt[n_lab:])}, ref_ls, ref_ts, share)
        opt.zero_grad()
        loss.backward()
        for k, v in ref_sd.items():
            solid[k] &= v.grad.abs() > 2e-4 * v.grad.abs().max()
        opt.step()
        with torch.no_grad():  # video_text_module.py:94-97, teacher_student.py:179-183
            ref_ls.clamp_(max=-math.log(0.001))
            ref_ts.clamp_(max=-math.log(0.001))
        got = module.fit_step(_batch(video, ids, n_lab))
        assert abs(got - float(loss)) < 2e-4 * abs(float(loss)), step
        for k, p in module.encoder.model.named_parameters():
            # AdamW's first steps move every weight by ~lr regardless of the gradient's size: compare the UPDATE
            upd_ref = ref_sd[k].detach() - torch.from_numpy(student_np[k])
            upd = p.detach().cpu() - torch.from_numpy(student_np[k])
            assert solid[k].any(), k
            assert (upd - upd_ref)[solid[k]].abs().max() < 0.02 * lr * (step + 1) + 1e-7, (step, k)
            assert upd.abs().max() <= 1.05 * lr * (step + 1) + 1e-3 * lr, (step, k)
        assert abs(module.logit_scale - float(ref_ls)) < 0.02 * lr * (step + 1)
        assert abs(module.teacher_student_logit_scale - float(ref_ts)) < 0.02 * lr * (step + 1)
    for k, p in module.teacher.model.named_parameters():
        assert torch.equal(p.detach(), teacher_before[k]), k
    # the updated student is what the inference path now computes with
    video = torch.from_numpy(synth.make_video(3, f, d, seed=1))
    with torch.no_grad():
        want = O.encode_video({k: v.detach() for k, v in ref_sd.items()}, video)
    assert (module.encoder.encode_video(video.to(DEV)).cpu() - want).abs().max() < 2e-2  # weights differ by <= 2 lr where noise decided
    fresh = ClipVideoTextEncoder(build_clip({k: p.detach().cpu().numpy() for k, p in module.encoder.model.named_parameters()},
                                            precision="fp32", device=DEV))
    assert torch.equal(fresh.encode_video(video.to(DEV)), module.encoder.encode_video(video.to(DEV)))  # repacked after the step


def test_training_step_at_one_ranks_share_of_config5(vitb16_state_dict):
    """BASELINE configs[4] per GPU: 64 clips x 8 frames through teacher + student ViT-B/16 and a full backward (60 GB of
    kept activations).  The CPU oracle cannot run this size; checked by properties: (i) the training forward returns,
    bit for bit, the embeddings of the inference path; (ii) the loss equals the forward-only module's; (iii) the
    backward is deterministic (bit-identical gradients on a re-run); (iv) an AdamW
    step with lr = 0 and no weight decay leaves the model bit-identical, a real step changes every tensor."""
    from fitclip_amd.retrieval import TeacherStudentModule
    d = synth.VIT_B_16
    n, f, n_lab = 64, 8, 32
    student_np = synth.perturbed_state_dict(vitb16_state_dict, d, seed=5, rel=0.05)
    base_v = torch.from_numpy(synth.make_video(8, f, d, seed=31))
    base_t = torch.from_numpy(synth.make_text(n, d, seed=31))
    video = base_v[torch.arange(n) % 8] + 0.01 * torch.randn(n, f, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    module = _trainer(student_np, vitb16_state_dict, 0.05, lr=0.0, weight_decay=0.0)
    batch = _batch(video, base_t, n_lab)
    out = module.training_step(batch)
    sv = torch.cat([out["labeled"][0][0], out["unlabeled"][0][0]])
    st = torch.cat([out["labeled"][0][1], out["unlabeled"][0][1]])
    assert torch.equal(sv, module.encoder.encode_video(batch["video_student"]))
    assert torch.equal(st, module.encoder.encode_text(batch["text_student"]))
    loss = module.training_step_end(out)
    fwd = TeacherStudentModule(module.encoder, module.teacher, init_temperature=0.05)
    want = 0.5 * float(fwd.dataset_step_end(out["labeled"], labeled=True)) + \
        0.5 * float(fwd.dataset_step_end(out["unlabeled"], labeled=False))
    assert abs(loss - want) < 1e-5 * abs(want)
    module.backward()
    grads1 = module.student.grads.clone()
    assert torch.isfinite(grads1).all() and float(grads1.abs().max()) > 0
    module.training_step_end(module.training_step(batch))
    module.backward()
    assert torch.equal(module.student.grads, grads1)  # every sum of the step has a fixed order (no float atomics)
    before = module.student.params.clone()
    module.optimizer_step()                      # lr = 0, weight decay = 0: nothing may move
    assert torch.equal(module.student.params, before)
    module.student.lr = 1e-4
    module.training_step_end(module.training_step(batch))
    module.backward()
    module.optimizer_step()
    for k, p in module.encoder.model.named_parameters():
        o, cnt = module.student.offsets[k], p.numel()
        if k == "token_embedding.weight":
            continue                              # only the rows of tokens that occur move
        assert not torch.equal(module.student.params[o:o + cnt], before[o:o + cnt]), k


def test_two_rank_training_step_equals_single_rank(tmp_path):
    """The data-parallel training step (teacher_student.py:143: `all_gather(..., sync_grads=True)` + DDP): two fresh rank
    processes (gloo group on ONE GPU; the collectives are staged through the host, the rest is the production path) each
    hold half of the labeled and half of the unlabeled clips, gather the embeddings, compute the full-batch losses,
    back-propagate the LOCAL rows of the gathered-embedding gradient and sum the parameter gradients over the ranks.
    Loss, every gradient and the updated parameters must equal the single-process step over the whole batch."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    worker = str(Path(__file__).resolve().parent / "workers" / "train_two_ranks.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    r1 = subprocess.run([sys.executable, worker, "--out", one], capture_output=True, text=True, timeout=600, env=env)
    assert r1.returncode == 0, r1.stderr[-3000:]
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), worker, "--out", two], capture_output=True, text=True,
                        timeout=900, env=env)
    assert r2.returncode == 0, r2.stderr[-3000:]
    a, b = np.load(one), np.load(two)
    assert abs(float(a["loss"]) - float(b["loss"])) < 1e-5 * abs(float(a["loss"]))
    ga, gb = torch.from_numpy(a["grads"]), torch.from_numpy(b["grads"])
    assert float((ga - gb).abs().max()) < 1e-4 * float(ga.abs().max())
    # per tensor as well (small gradients must not hide behind the largest one)
    d = synth.TINY
    off = 0
    for name, shape in synth.parameter_shapes(d).items():
        cnt = int(np.prod(shape))
        sl = slice(off, off + cnt)
        assert _rel(gb[sl], ga[sl]) < 2e-4, name
        off += -(-cnt // 64) * 64
    assert np.allclose(a["scale_grads"], b["scale_grads"], rtol=1e-4, atol=1e-7)


def test_prompts_variant_of_the_distillation_step(tiny_state_dict):
    """`prompts` (teacher_student.py:79-91,111-138): the unlabeled videos are scored against a fixed prompt list instead
    of their own captions - the student encodes labeled captions + prompts, the unlabeled score matrix is
    [videos x prompts].  Loss and every parameter gradient vs autograd through the oracle."""
    d = synth.TINY
    student_np = synth.perturbed_state_dict(tiny_state_dict, d, seed=5, rel=0.3)
    n, f, n_lab, temp = 9, 2, 4, 0.05
    prompts = ["a video of a cat", "someone cooking", "people dancing outdoors"]
    video = torch.from_numpy(synth.make_video(n, f, d, seed=12))
    ids = torch.from_numpy(synth.make_text(n, d, seed=12))
    module = _trainer(student_np, tiny_state_dict, temp, prompts=prompts)
    prompt_ids = module.tokenized_prompts
    assert prompt_ids.shape == (3, d.context_length)
    share = {"labeled": 0.5, "unlabeled": 0.5}
    # oracle: labeled captions + prompts through the text tower, all videos through the visual tower
    sd = {k: v.clone().requires_grad_(True) for k, v in O.to_torch(student_np).items()}
    ls = torch.tensor([-math.log(temp)], dtype=torch.float32, requires_grad=True)
    ts = torch.tensor([-math.log(temp)], dtype=torch.float32, requires_grad=True)
    text_ids = torch.cat([ids[:n_lab], prompt_ids])
    with torch.no_grad():
        t_sd = O.to_torch(tiny_state_dict)
        tv, tt = O.encode_video(t_sd, video), O.encode_text(t_sd, {"input_ids": text_ids})
    ev, et = O.encode_video(sd, video), O.encode_text(sd, {"input_ids": text_ids})
    loss, parts = O.teacher_student_training_loss(
        {"labeled": (ev[:n_lab], et[:n_lab]), "unlabeled": (ev[n_lab:], et[n_lab:])},
        {"labeled": (tv[:n_lab], tt[:n_lab]), "unlabeled": (tv[n_lab:], tt[n_lab:])}, ls, ts, share)
    loss.backward()
    got = module.training_step_end(module.training_step(_batch(video, ids, n_lab)))
    assert abs(got - float(loss)) < 1e-4 * abs(float(loss))
    module.backward()
    for k, p in module.encoder.model.named_parameters():
        assert _rel(p.grad, sd[k].grad) < GRAD_TOL, k
    assert abs(float(module.scale_grads[0]) - float(ls.grad)) < 1e-4 * abs(float(ls.grad))
    assert abs(float(module.scale_grads[1]) - float(ts.grad)) < 1e-4 * abs(float(ts.grad))


def test_micro_batched_forward_backward_accumulates(tiny_state_dict):
    """A local batch larger than `max_frames_per_pass` / `max_texts_per_pass` goes through the towers in several passes
    (one activation arena each) and the backward ACCUMULATES the parameter gradients over the passes: same embeddings
    bit for bit, same gradients up to summation order."""
    d = synth.TINY
    student_np = synth.perturbed_state_dict(tiny_state_dict, d, seed=5, rel=0.3)
    n, f, n_lab = 8, 2, 4
    video = torch.from_numpy(synth.make_video(n, f, d, seed=9))
    ids = torch.from_numpy(synth.make_text(n, d, seed=9))
    whole = _trainer(student_np, tiny_state_dict, 0.05)
    parts = _trainer(student_np, tiny_state_dict, 0.05, max_frames_per_pass=6, max_texts_per_pass=3)
    outs = []
    for module in (whole, parts):
        out = module.training_step(_batch(video, ids, n_lab))
        outs.append(torch.cat([out["labeled"][0][0], out["unlabeled"][0][0], out["labeled"][0][1], out["unlabeled"][0][1]]))
        module.training_step_end(out)
        module.backward()
    assert torch.equal(outs[0], outs[1])
    for (k, a), (_, b) in zip(whole.encoder.model.named_parameters(), parts.encoder.model.named_parameters()):
        assert _rel(b.grad, a.grad.cpu()) < 1e-5, k


@pytest.mark.parametrize("keep,micro", [(4, 3), (0, 4), (6, 8), (10, 2)])
def test_split_step_for_batches_larger_than_memory(tiny_state_dict, keep, micro):
    """`TeacherStudentTrainer.split_step(keep, micro)` (BASELINE configs[4] on fewer than four GPUs: the activations of a rank's
    share do not fit): the first `keep` clips keep their activations, the others are forwarded without and re-forwarded micro-batch
    by micro-batch in the backward, gradients accumulated.  Embeddings are bit-invariant to their batch, so the LOSS and the
    temperature gradients are bitwise those of the unsplit step; the parameter gradients differ by summation order only; with
    everything kept (keep >= n) the step IS the unsplit one, bit for bit.  Two optimiser steps stay together."""
    d = synth.TINY
    student_np = synth.perturbed_state_dict(tiny_state_dict, d, seed=5, rel=0.3)
    n, f, n_lab = 10, 2, 4
    video = torch.from_numpy(synth.make_video(n, f, d, seed=11))
    ids = torch.from_numpy(synth.make_text(n, d, seed=11))
    whole, split = _trainer(student_np, tiny_state_dict, 0.05), _trainer(student_np, tiny_state_dict, 0.05)
    split.split_step(keep, micro)
    losses = []
    for module in (whole, split):
        losses.append(module.training_step_end(module.training_step(_batch(video, ids, n_lab))))
        module.backward()
    assert losses[0] == losses[1]
    assert torch.equal(whole.scale_grads, split.scale_grads)
    for (k, a), (_, b) in zip(whole.encoder.model.named_parameters(), split.encoder.model.named_parameters()):
        if keep >= n:
            assert torch.equal(a.grad, b.grad), k
        else:
            assert _rel(b.grad, a.grad.cpu()) < 1e-5, k
    for module in (whole, split):
        module.optimizer_step()
    l2 = [module.fit_step(_batch(video, ids, n_lab)) for module in (whole, split)]
    assert abs(l2[0] - l2[1]) < 1e-5 * abs(l2[0])
    # (AdamW divides by sqrt(v): on elements whose gradient is rounding noise the two runs may step in opposite directions,
    # one lr each and step)
    assert float((whole.student.params - split.student.params).abs().max()) <= 4.001 * whole.student.lr
    # the planner: everything fits on this device for a tiny model -> the plain step
    assert split.plan_split(n, f) == (n, n)


def test_train_command_reduces_the_distillation_loss(capsys):
    """`python -m fitclip_amd command=train encoder=teacher_student_tiny`: the distillation loop end to end (student
    forward / losses / backward / AdamW per step over synthetic mixed batches).  Repeating one batch (`repeat_batch`),
    the student must fit it: the training loss goes down step after step."""
    import json
    from fitclip_amd.__main__ import main
    main(["command=train", "encoder=teacher_student_tiny", "steps=25", "n_labeled=6", "n_unlabeled=6", "num_frames=2",
          "lr=1e-5", "init_temperature=0.05", "seed=3", "repeat_batch=true"])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    losses = out["loss/train"]
    assert len(losses) == 25 and all(np.isfinite(losses))
    assert np.mean(losses[-3:]) < 0.8 * np.mean(losses[:3]), losses
    assert sum(b < a for a, b in zip(losses, losses[1:])) >= 18, losses


def test_predict_command_writes_the_embeddings(tmp_path, capsys):
    import json
    from fitclip_amd.__main__ import main
    path = str(tmp_path / "predictions.pt")
    main(["command=predict", "encoder=clip_vit_b_16", "n_clips=5", "num_frames=1", "eval_batch_size=2", f"output_path={path}"])
    out = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    pred = torch.load(path)
    assert out["n"] == 5 and pred["encoded_videos"].shape == pred["encoded_texts"].shape == (5, 512)
    assert pred["video_ids"] == [f"clip{i}" for i in range(5)]


def test_checkpoint_and_resume_continue_the_same_run(tmp_path, tiny_state_dict):
    """Save after two steps, train two more; a FRESH trainer that loads the file and trains the same two steps ends with
    BITWISE the same weights, moments, temperatures and losses (every sum of a step has a fixed order).  The
    file has the reference's module keys and a torch.optim.AdamW-shaped optimiser state."""
    d = synth.TINY
    student_np = synth.perturbed_state_dict(tiny_state_dict, d, seed=5, rel=0.3)

    def batch(step):
        return _batch(torch.from_numpy(synth.make_video(8, 2, d, seed=30 + step)), torch.from_numpy(synth.make_text(8, d, seed=30 + step)), 4)

    a = _trainer(student_np, tiny_state_dict, 0.05, lr=1e-4)
    for step in range(2):
        a.fit_step(batch(step))
    path = str(tmp_path / "train.ckpt")
    torch.save(a.checkpoint(), path)
    losses_a = [a.fit_step(batch(step)) for step in range(2, 4)]
    ckpt = torch.load(path, weights_only=True)   # tensors and plain containers only
    assert {"state_dict", "optimizer_states", "global_step"} <= set(ckpt) and ckpt["global_step"] == 2
    assert "encoder.model.visual.proj" in ckpt["state_dict"] and "teacher.model.visual.proj" in ckpt["state_dict"]
    opt_state = ckpt["optimizer_states"][0]["state"]
    assert set(opt_state[0]) == {"step", "exp_avg", "exp_avg_sq"}
    # the reference's numbering (AdamW(self.parameters()), cli.py:129): 0 logit_scale, 1 max_logit_scale (frozen: no state),
    # 2 teacher_student_logit_scale, 3... the encoder's parameters, then the frozen teacher's (no state)
    from fitclip_amd.checkpoint import reference_parameter_order
    order = reference_parameter_order(a)
    assert order[:4] == ["logit_scale", "max_logit_scale", "teacher_student_logit_scale", "encoder.model.positional_embedding"]
    assert sorted(opt_state) == [i for i, k in enumerate(order) if k != "max_logit_scale" and not k.startswith("teacher.")]
    assert opt_state[3]["exp_avg"].shape == ckpt["state_dict"]["encoder.model.positional_embedding"].shape
    assert tuple(ckpt["state_dict"]["logit_scale"].shape) == (1,) and "max_logit_scale" in ckpt["state_dict"]
    assert ckpt["optimizer_states"][0]["param_groups"][0]["params"] == list(range(len(order)))
    b = _trainer(tiny_state_dict, tiny_state_dict, 0.3, lr=7.0)       # wrong weights, temperature and lr: all come from the file
    b.load_checkpoint(ckpt)
    assert b.student.step_count == 2 and b.student.lr == 1e-4 and abs(b.logit_scale - a.checkpoint()["state_dict"]["logit_scale"]) < 1
    losses_b = [b.fit_step(batch(step)) for step in range(2, 4)]
    assert losses_b == losses_a
    for buf in ("params", "exp_avg", "exp_avg_sq"):
        assert torch.equal(getattr(a.student, buf), getattr(b.student, buf)), buf
    assert a.logit_scale == b.logit_scale and a.teacher_student_logit_scale == b.teacher_student_logit_scale
    # evaluating the resumed student through the plain retrieval module ignores the teacher keys (text_video_retrieval.py:101-131)
    from fitclip_amd.retrieval import TextVideoRetrievalModule
    plain = TextVideoRetrievalModule(ClipVideoTextEncoder(build_clip(tiny_state_dict, precision="fp32", device=DEV)), init_temperature=0.05)
    plain.load_state_dict(ckpt["state_dict"])
    video = torch.from_numpy(synth.make_video(3, 2, d, seed=1)).to(DEV)
    want = ClipVideoTextEncoder(build_clip({k[len("encoder.model."):]: v.numpy() for k, v in ckpt["state_dict"].items()
                                            if k.startswith("encoder.model.")}, precision="fp32", device=DEV)).encode_video(video)
    assert torch.equal(plain.encoder.encode_video(video), want)


@pytest.mark.parametrize("labeled_first", [True, False])
def test_teacher_skips_the_labeled_rows_without_changing_the_step(tiny_state_dict, labeled_first):
    """The labeled part's loss never reads the teacher (teacher_student.py:150-160), so by default the teacher only
    encodes the unlabeled rows.  Loss, both per-dataset losses, every gradient and the temperature gradients must equal
    the reference's literal schedule (`teacher_on_labeled=True`), whichever dataset comes first in the batch; the
    unlabeled teacher embeddings are the same, the labeled ones are zeros."""
    d = synth.TINY
    student_np = synth.perturbed_state_dict(tiny_state_dict, d, seed=5, rel=0.3)
    n, n_lab = 10, 4
    video = torch.from_numpy(synth.make_video(n, 2, d, seed=21))
    ids = torch.from_numpy(synth.make_text(n, d, seed=21))
    batch = _batch(video, ids, n_lab)
    if not labeled_first:
        batch["dataset"] = ["unlabeled"] * (n - n_lab) + ["labeled"] * n_lab
    steps = {}
    for literal in (True, False):
        module = _trainer(student_np, tiny_state_dict, 0.05, teacher_on_labeled=literal)
        out = module.training_step(dict(batch))
        loss = module.training_step_end(out)
        module.backward()
        steps[literal] = (loss, dict(module.last_losses), module.student.grads.clone(), module.scale_grads.clone(), out)
    a, b = steps[True], steps[False]
    assert a[0] == b[0] and a[1] == b[1]
    assert torch.equal(a[2], b[2])
    assert torch.equal(a[3], b[3])
    assert torch.equal(a[4]["unlabeled"][1][0], b[4]["unlabeled"][1][0]) and torch.equal(a[4]["unlabeled"][1][1], b[4]["unlabeled"][1][1])
    assert not b[4]["labeled"][1][0].any() and not b[4]["labeled"][1][1].any() and a[4]["labeled"][1][0].any()

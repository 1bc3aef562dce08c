"""The KD fine-tuning step on the HIP path: what `TeacherStudentLightningModule.training_step / training_step_end /
optimizer_step` do through autograd + `torch.optim.AdamW` + DDP in the reference
(`aligner/teacher_student.py:93-183`, `aligner/video_text_module.py:55-97`, `aligner/cli.py:129`,
`config/trainer.yaml:21-23`), as explicit forward / backward / update calls into libfitclip_hip.so.

    student = ClipVideoTextEncoder(build_clip(sd_student, precision="fp32", device="cuda"))
    teacher = ClipVideoTextEncoder(build_clip(sd_teacher, precision="bf16", device="cuda"))   # frozen: any precision
    module = TeacherStudentTrainer(student, teacher, init_temperature=0.05, lr=3e-6)
    loss = module.fit_step({"video_student": ..., "text_student": {"input_ids": ...}, "video_teacher": ...,
                            "text_teacher": {"input_ids": ...}, "dataset": ["labeled"] * a + ["unlabeled"] * b})

`StudentTrainer` owns the native training state of one CLIP student: parameters, gradients and the two AdamW moments
live in four flat fp32 buffers (every `nn.Parameter` of the model is a view into the first, its `.grad` a view into the
second), so the optimiser is ONE kernel launch and the data-parallel gradient exchange is an all-reduce of slices of one
buffer.  Only fp32 models train (the reference trains in float32: Trainer precision 32).
"""
from __future__ import annotations

import itertools
import math
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import torch

from . import _lib
from . import distributed as D
from . import ops
from .encoder import ClipVideoTextEncoder
from .retrieval import TeacherStudentModule

_ALIGN = 64  # floats: every parameter starts on a 256-byte boundary of the flat buffers


class StudentTrainer:
    def __init__(self, encoder: ClipVideoTextEncoder, lr: float = 3e-6, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 1e-2, max_frames_per_pass: int = 1024,
                 max_texts_per_pass: int = 4096) -> None:
        model = encoder.model
        if model.precision not in ("fp32", "f32", "float32"):
            raise ValueError("only precision='fp32' models train (the reference trains in float32)")
        self.encoder, self.model = encoder, model
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.max_frames, self.max_texts = max_frames_per_pass, max_texts_per_pass
        self.step_count = 0
        dev = model._device()
        if dev.type != "cuda":
            raise _lib.FitclipHipError("move the student to the ROCm device before training (no CPU fallback)")
        named = model._named_weights()
        offsets, total = {}, 0
        for name, p in named:
            offsets[name] = total
            total += -(-p.numel() // _ALIGN) * _ALIGN
        self.params = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros_like(self.params)
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.offsets, self.total = offsets, total
        with torch.no_grad():
            for name, p in named:
                o, n = offsets[name], p.numel()
                view = self.params[o:o + n].view(p.shape)
                view.copy_(p.data)
                p.data = view                                   # the module's parameters ARE the flat buffer from now on
                p.requires_grad_(True)
                p.grad = self.grads[o:o + n].view(p.shape)
        # text-tower parameters sit at both ends of named_parameters(): three slices for the bucketed gradient exchange
        vis = [offsets[n] for n, _ in named if n.startswith("visual.")]
        after = [offsets[n] for n, _ in named if not n.startswith("visual.") and offsets[n] > max(vis)]
        self.visual_span = (min(vis), min(after) if after else total)
        model.invalidate_weights()
        self._wt_arena = None
        self._state: Optional[Dict[str, Any]] = None
        self._scratch: Dict[int, torch.Tensor] = {}
        self._bind()

    # ------------------------------------------------------------------------------------------------- native state
    def _bind(self) -> None:
        """(Re)packs the weights, hands the gradient views to the handle and refreshes the transposed weight copies
        the dgrad GEMMs read.  Needed once, and again after every parameter update."""
        lib = _lib.load()
        rt = self.model._ensure_ready()
        with torch.cuda.device(self.params.device):
            for name, p in self.model._named_weights():
                _lib.check(lib.fc_set_grad(rt.handle, name.encode(), p.grad.data_ptr()), f"fc_set_grad({name})")
            need = lib.fc_train_weights_bytes(rt.handle)
            if self._wt_arena is None or self._wt_arena.numel() < need:
                self._wt_arena = torch.empty(need, dtype=torch.uint8, device=self.params.device)
            _lib.check(lib.fc_train_prepare(rt.handle, self._wt_arena.data_ptr(), self._wt_arena.numel(),
                                            _lib.current_stream()), "fc_train_prepare")
        self._rt = rt

    def _scratch_for(self, tower: int, n: int) -> torch.Tensor:
        need = _lib.load().fc_train_scratch_bytes(self._rt.handle, tower, n)
        buf = self._scratch.get(tower)
        if buf is None or buf.numel() < need:
            self._scratch[tower] = buf = torch.empty(need, dtype=torch.uint8, device=self.params.device)
        return buf

    # ------------------------------------------------------------------------------------------------------ forward
    def forward(self, video: torch.Tensor, text: Mapping[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        """`encoder(video=..., text=...)` with every activation kept for `backward`: f32 [B, F, 3, R, R] and
        {"input_ids": int [B, L]} -> (video embeddings [B, E], text embeddings [B, E]), same values as the inference
        path of the fp32 mode."""
        lib, rt, dev = _lib.load(), self._rt, self.params.device
        if rt.fingerprint is None:
            raise _lib.FitclipHipError("weights changed since the last bind: call StudentTrainer.rebind()")
        d = self.model.dims
        b, f = video.shape[:2]
        frames = video.reshape(b * f, *video.shape[2:]).to(device=dev, dtype=torch.float32).contiguous()
        ids = text["input_ids"].to(device=dev, dtype=torch.int64).contiguous()
        zv = torch.empty((b * f, d.embed_dim), dtype=torch.float32, device=dev)
        zt = torch.empty((ids.shape[0], d.embed_dim), dtype=torch.float32, device=dev)
        v_chunks, t_chunks = [], []
        with torch.cuda.device(dev):
            for s in range(0, b * f, self.max_frames):
                n = min(self.max_frames, b * f - s)
                arena = torch.empty(lib.fc_train_arena_bytes(rt.handle, 0, n), dtype=torch.uint8, device=dev)
                _lib.check(lib.fc_encode_image_train(rt.handle, frames[s:].data_ptr(), n, zv[s:].data_ptr(),
                                                     arena.data_ptr(), arena.numel(), _lib.current_stream()),
                           "fc_encode_image_train")
                v_chunks.append((s, n, arena))
            for s in range(0, ids.shape[0], self.max_texts):
                n = min(self.max_texts, ids.shape[0] - s)
                arena = torch.empty(lib.fc_train_arena_bytes(rt.handle, 1, n), dtype=torch.uint8, device=dev)
                _lib.check(lib.fc_encode_text_train(rt.handle, ids[s:].data_ptr(), n, zt[s:].data_ptr(),
                                                    arena.data_ptr(), arena.numel(), _lib.current_stream()),
                           "fc_encode_text_train")
                t_chunks.append((s, n, arena))
        self._state = {"zv": zv, "zt": zt, "ids": ids, "frames": f, "clips": b, "v": v_chunks, "t": t_chunks}
        return ops.pool_normalize(zv, b, f), ops.l2_normalize(zt)

    # ----------------------------------------------------------------------------------------------------- backward
    def backward(self, d_video: torch.Tensor, d_text: torch.Tensor, accumulate: bool = False,
                 reduce_across_ranks: bool = True) -> None:
        """d(loss)/d(video embeddings) [B, E] and d(loss)/d(text embeddings) [B, E] of the last `forward` -> parameter
        gradients (`p.grad` of every parameter, i.e. `self.grads`).  With several ranks the gradients are summed over
        the ranks (text-tower slices first, overlapping the visual backward), which together with the unscaled local
        slice of the gathered-embedding gradient is what DDP's average of `all_gather(sync_grads=True)` gradients
        amounts to (`util/tensor_utils.py:48-66`)."""
        st = self._state
        if st is None:
            raise _lib.FitclipHipError("backward() needs a forward() first")
        lib, rt, dev = _lib.load(), self._rt, self.params.device
        e = self.model.dims.embed_dim
        d_video = ops._dev(d_video.contiguous(), "d_video", torch.float32)
        d_text = ops._dev(d_text.contiguous(), "d_text", torch.float32)
        dzv, dzt = torch.empty_like(st["zv"]), torch.empty_like(st["zt"])
        acc = int(accumulate)
        handles = []
        with torch.cuda.device(dev):
            stream = _lib.current_stream()
            _lib.check(lib.fc_pool_normalize_backward(st["zt"].data_ptr(), d_text.data_ptr(), dzt.data_ptr(),
                                                      dzt.shape[0], 1, e, stream), "fc_pool_normalize_backward")
            _lib.check(lib.fc_pool_normalize_backward(st["zv"].data_ptr(), d_video.data_ptr(), dzv.data_ptr(),
                                                      st["clips"], st["frames"], e, stream),
                       "fc_pool_normalize_backward")
            for i, (s, n, arena) in enumerate(st["t"]):
                scratch = self._scratch_for(1, n)
                _lib.check(lib.fc_encode_text_backward(rt.handle, st["ids"][s:].data_ptr(), dzt[s:].data_ptr(), n,
                                                       arena.data_ptr(), arena.numel(), scratch.data_ptr(),
                                                       scratch.numel(), int(acc or i > 0), stream),
                           "fc_encode_text_backward")
            lo, hi = self.visual_span
            if reduce_across_ranks and D.collectives_active():
                handles += [D.all_reduce_sum_(self.grads[:lo], async_op=True),
                            D.all_reduce_sum_(self.grads[hi:], async_op=True)]
            for i, (s, n, arena) in enumerate(st["v"]):
                scratch = self._scratch_for(0, n)
                _lib.check(lib.fc_encode_image_backward(rt.handle, dzv[s:].data_ptr(), n, arena.data_ptr(),
                                                        arena.numel(), scratch.data_ptr(), scratch.numel(),
                                                        int(acc or i > 0), stream), "fc_encode_image_backward")
            if reduce_across_ranks and D.collectives_active():
                handles.append(D.all_reduce_sum_(self.grads[lo:hi], async_op=True))
                for h in handles:
                    if h is not None:
                        h.wait()
        self._state = None  # the activation arenas are released

    # ------------------------------------------------------------------------------------------------------- update
    def step(self) -> None:
        """One `torch.optim.AdamW` step over all parameters (aligner/cli.py:129; lr 3e-6, config/trainer.yaml:21-23)."""
        self.step_count += 1
        with torch.cuda.device(self.params.device):
            _lib.check(_lib.load().fc_adamw(self.params.data_ptr(), self.grads.data_ptr(), self.exp_avg.data_ptr(),
                                            self.exp_avg_sq.data_ptr(), self.total, self.lr, self.betas[0],
                                            self.betas[1], self.eps, self.weight_decay, self.step_count,
                                            _lib.current_stream()), "fc_adamw")
        self.rebind()

    def rebind(self) -> None:
        self.model.invalidate_weights()
        self._bind()

    def zero_grad(self) -> None:
        self.grads.zero_()


def _lengths(datasets: Sequence[str]) -> Tuple[Tuple[str, ...], Tuple[int, ...]]:
    keys, lengths = zip(*((k, sum(1 for _ in grp)) for k, grp in itertools.groupby(datasets)))
    return keys, lengths


class TeacherStudentTrainer(TeacherStudentModule):
    """`TeacherStudentLightningModule` as a training loop body (teacher_student.py:44-183): student forward (kept),
    frozen teacher forward, per-dataset gather + scores + loss (NCE on the labeled part, KD * tau_ts^2 on the unlabeled
    part, weighted by `dataset_loss_share`), backward into the student, AdamW, temperature clamps."""

    def __init__(self, encoder: ClipVideoTextEncoder, teacher: ClipVideoTextEncoder, labeled_dataset_name: str = "labeled",
                 labeled_dataset_loss_share: Optional[float] = None,
                 dataset_names: Iterable[str] = ("labeled", "unlabeled"), init_temperature: float = 0.05,
                 min_temperature: float = 0.001, fit_temperature: bool = True, lr: float = 3e-6,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 prompts: Optional[Iterable[str]] = None, teacher_on_labeled: bool = False,
                 **trainer_kwargs: Any) -> None:
        super().__init__(encoder, teacher, init_temperature, min_temperature)
        # The reference runs the teacher over the whole batch (teacher_student.py:94-96) but only ever reads its
        # embeddings of the UNLABELED part (`_dataset_step_end`, :150-160: the labeled part's loss is `self.loss(scores)`
        # of the student alone).  By default the teacher therefore skips the labeled rows (same loss, same gradients,
        # half the teacher forward at the usual 1:1 composition); their teacher entries in `training_step`'s output
        # are zeros.  `teacher_on_labeled=True` restores the reference's literal schedule.
        self.teacher_on_labeled = bool(teacher_on_labeled)
        self.dataset_names = list(dataset_names)
        assert len(self.dataset_names) == 2, "The current implementation needs exactly 2 datasets."  # :57
        if labeled_dataset_loss_share is None:                                                       # :61-67
            self.dataset_loss_share = {name: 1 / len(self.dataset_names) for name in self.dataset_names}
        else:
            self.dataset_loss_share = {labeled_dataset_name: labeled_dataset_loss_share}
            self.dataset_loss_share.update((name, (1 - labeled_dataset_loss_share) / (len(self.dataset_names) - 1))
                                           for name in self.dataset_names if name != labeled_dataset_name)
        self.labeled_dataset_name = labeled_dataset_name
        self.unlabeled_dataset_name = next(k for k in self.dataset_names if k != labeled_dataset_name)
        self.fit_temperature = fit_temperature
        # `prompts` (teacher_student.py:79-91): the unlabeled videos are scored against this fixed list instead of
        # their own captions; each tower's tokenizer tokenizes it once
        self.tokenized_prompts = self.teacher_tokenized_prompts = None
        if prompts is not None:
            prompts = list(prompts)
            self.tokenized_prompts = encoder.get_tokenizer()(prompts)["input_ids"]
            self.teacher_tokenized_prompts = teacher.get_tokenizer()(prompts)["input_ids"]
        self.student = StudentTrainer(encoder, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, **trainer_kwargs)
        dev = self.student.params.device
        # [logit_scale, teacher_student_logit_scale]: value, grad, AdamW moments (the reference hands them to the same
        # optimiser as the encoder: `self.parameters()`, cli.py:129)
        self.scales = torch.tensor([self.logit_scale, self.teacher_student_logit_scale] + [0.0] * 2, dtype=torch.float32,
                                   device=dev)
        self.scale_grads = torch.zeros(4, dtype=torch.float32, device=dev)
        self.scale_m = torch.zeros(4, dtype=torch.float32, device=dev)
        self.scale_v = torch.zeros(4, dtype=torch.float32, device=dev)
        self._pending: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
        self.last_losses: Dict[str, float] = {}
        # Local batches whose activations do not fit the device (BASELINE configs[4] on one or two GPUs: 512 clips x 8 frames keep
        # 116 MB per frame = 476 GB): `split_step(keep_clips, micro_clips)` - see `_student_forward`.
        self.keep_clips: Optional[int] = None
        self.micro_clips: Optional[int] = None
        self._split: Optional[Dict[str, Any]] = None

    # ------------------------------------------------------------------------------------------------ training_step
    def training_step(self, batch: Mapping[str, Any]) -> Dict[str, Tuple[Tuple[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]]:
        """Student (activations kept) and teacher forward over the whole local batch, split per dataset
        (teacher_student.py:99-140).  With `prompts` the captions of the unlabeled part are replaced by the prompt list
        before the forward, so that part has `len(prompts)` text rows for its videos."""
        keys, lengths = _lengths(batch["dataset"])
        assert len(keys) == len(self.dataset_names), "All datasets should be present in each batch."
        text_student, text_teacher = batch["text_student"]["input_ids"], batch["text_teacher"]["input_ids"]
        text_lengths = list(lengths)
        if self.tokenized_prompts is not None:
            idx = keys.index(self.unlabeled_dataset_name)
            start, end = sum(lengths[:idx]), sum(lengths[:idx + 1])
            dev = text_student.device

            def replace(ids: torch.Tensor, new: torch.Tensor) -> torch.Tensor:  # _replace_in_tokenized_text, equal widths
                assert ids.shape[1] == new.shape[1], "prompts and captions must share the context length"
                return torch.cat((ids[:start], new.to(device=dev, dtype=ids.dtype), ids[end:]))

            text_student = replace(text_student, self.tokenized_prompts)
            text_teacher = replace(text_teacher, self.teacher_tokenized_prompts)
            text_lengths[idx] = self.tokenized_prompts.shape[0]
        sv, st = self._student_forward(batch["video_student"], text_student)
        with torch.no_grad():
            if self.teacher_on_labeled:
                tv, tt = self.teacher(video=batch["video_teacher"], text={"input_ids": text_teacher})
            else:
                # rows of the datasets whose loss reads the teacher; each dataset is one contiguous run of the batch
                vr, tr, v0, t0 = [], [], 0, 0
                for key, nv, nt in zip(keys, lengths, text_lengths):
                    if key != self.labeled_dataset_name:
                        vr.append((v0, v0 + nv))
                        tr.append((t0, t0 + nt))
                    v0, t0 = v0 + nv, t0 + nt
                pick = lambda x, runs: x[runs[0][0]:runs[0][1]] if len(runs) == 1 else torch.cat([x[a:b] for a, b in runs])  # noqa: E731
                pv, pt = self.teacher(video=pick(batch["video_teacher"], vr), text={"input_ids": pick(text_teacher, tr)})
                tv = pv.new_zeros((sv.shape[0], pv.shape[1]))
                tt = pt.new_zeros((st.shape[0], pt.shape[1]))
                if vr:
                    o = 0
                    for a, b in vr:
                        tv[a:b] = pv[o:o + b - a]
                        o += b - a
                    o = 0
                    for a, b in tr:
                        tt[a:b] = pt[o:o + b - a]
                        o += b - a
        out, v0, t0 = {}, 0, 0
        self._layout = []
        for key, nv, nt in zip(keys, lengths, text_lengths):
            out[key] = ((sv[v0:v0 + nv], st[t0:t0 + nt]), (tv[v0:v0 + nv], tt[t0:t0 + nt]))
            self._layout.append((key, nv, nt))
            v0, t0 = v0 + nv, t0 + nt
        self._rows = (v0, t0)
        return out

    # ------------------------------------------------------------------------------------- a batch larger than memory
    def split_step(self, keep_clips: Optional[int], micro_clips: Optional[int]) -> None:
        """The contrastive losses couple every row of the batch, so a local batch cannot simply be cut into independent
        training steps.  With `micro_clips` set, a step runs as (the gradient-cache schedule):
          1. student forward of the first `keep_clips` clips WITH kept activations, of the other clips WITHOUT (the inference
             forward, in micro-batches) - embeddings are bit-invariant to the batch they are computed in, so the embedding
             matrix, the loss and its gradient w.r.t. the embeddings are exactly those of the unsplit step;
          2. backward of the kept part; then per micro-batch of `micro_clips` clips: forward again with kept activations,
             backward with its rows of the embedding gradient, parameter gradients ACCUMULATED; the gradient exchange with
             the other ranks happens once, in the last backward.
        Cost: one more student forward over the clips that were not kept.  `micro_clips=None` restores the plain step."""
        if micro_clips is not None and micro_clips <= 0:
            raise ValueError("micro_clips must be positive")
        self.keep_clips, self.micro_clips = keep_clips, micro_clips

    def plan_split(self, n_clips: int, frames: int, micro_clips: int = 32, margin_bytes: int = 24 << 30) -> Tuple[int, int]:
        """(keep_clips, micro_clips) for a local batch of `n_clips` x `frames` from the device memory that is free NOW: as many
        clips kept as fit next to one micro-batch's activations and `margin_bytes` (backward scratch, the teacher's
        workspace, losses); (n_clips, n_clips) - the plain step - when everything fits."""
        lib, rt = _lib.load(), self.student._rt
        per_clip = lib.fc_train_arena_bytes(rt.handle, 0, frames) + lib.fc_train_arena_bytes(rt.handle, 1, 1)
        free, _ = torch.cuda.mem_get_info(self.student.params.device)
        free += torch.cuda.memory_reserved(self.student.params.device) - torch.cuda.memory_allocated(self.student.params.device)
        if (n_clips * per_clip + margin_bytes) <= free:
            return n_clips, n_clips
        micro = max(1, min(micro_clips, n_clips))
        keep = int(max(0, free - margin_bytes - micro * per_clip) // per_clip)
        keep = min(n_clips, keep // micro * micro)
        return keep, micro

    def _student_forward(self, video: torch.Tensor, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        n = video.shape[0]
        keep = n if self.micro_clips is None else min(n, max(0, self.keep_clips or 0))
        self._split = None
        if keep >= n:
            return self.student.forward(video, {"input_ids": ids})
        if ids.shape[0] != n:
            raise _lib.FitclipHipError("the split step pairs every clip with one caption (no `prompts` variant)")
        sv_parts, st_parts, kept_state = [], [], None
        if keep > 0:
            v, t = self.student.forward(video[:keep], {"input_ids": ids[:keep]})
            sv_parts.append(v), st_parts.append(t)
            kept_state, self.student._state = self.student._state, None
        with torch.no_grad():
            for s0 in range(keep, n, self.micro_clips):
                v, t = self.encoder(video=video[s0:s0 + self.micro_clips], text={"input_ids": ids[s0:s0 + self.micro_clips]})
                sv_parts.append(v), st_parts.append(t)
        self._split = {"keep": keep, "micro": self.micro_clips, "video": video, "ids": ids, "state": kept_state}
        return torch.cat(sv_parts), torch.cat(st_parts)

    # -------------------------------------------------------------------------------------------- training_step_end
    def training_step_end(self, output: Mapping[str, Any]) -> float:
        """sum_d share_d * loss_d (teacher_student.py:142-176), and the gradient of that sum w.r.t. the LOCAL student
        embeddings (kept for `backward`) and the two temperatures."""
        lib = _lib.load()
        rank, world = D.world()
        dev = self.student.params.device
        e = self.encoder.model.dims.embed_dim
        d_video = torch.zeros((self._rows[0], e), dtype=torch.float32, device=dev)
        d_text = torch.zeros((self._rows[1], e), dtype=torch.float32, device=dev)
        self.scale_grads.zero_()
        ls, ts_ls = float(self.scales[0]), float(self.scales[1])
        scale, ts_scale = math.exp(ls), math.exp(ts_ls)
        total, v0, t0 = 0.0, 0, 0
        self.last_losses = {}
        with torch.cuda.device(dev):
            stream = _lib.current_stream()
            for name, nv, nt in self._layout:
                (v, t), (tv, tt) = output[name]
                # one collective per distinct row count (one in all when videos and texts pair up); the labeled part's
                # teacher embeddings are never read, so they do not travel
                if name == self.labeled_dataset_name and not self.teacher_on_labeled:
                    if nv == nt:
                        v_all, t_all = D.all_gather_many((v.contiguous(), t.contiguous()), [nv] * world)
                    else:
                        v_all, t_all = D.all_gather_rows(v.contiguous(), [nv] * world), D.all_gather_rows(t.contiguous(), [nt] * world)
                    tv_all = tt_all = None
                elif nv == nt:
                    v_all, t_all, tv_all, tt_all = D.all_gather_many((v.contiguous(), t.contiguous(), tv.contiguous(),
                                                                      tt.contiguous()), [nv] * world)
                else:
                    v_all, tv_all = D.all_gather_many((v.contiguous(), tv.contiguous()), [nv] * world)
                    t_all, tt_all = D.all_gather_many((t.contiguous(), tt.contiguous()), [nt] * world)
                rows, cols = v_all.shape[0], t_all.shape[0]
                share = self.dataset_loss_share[name]
                scores = ops.similarity(v_all, t_all, alpha=scale).contiguous()
                dscores = torch.empty_like(scores)
                ws = torch.empty(3 * (rows + cols), dtype=torch.float32, device=dev)
                if name == self.labeled_dataset_name:
                    loss = ops.nce_loss(scores)
                    _lib.check(lib.fc_nce_loss_backward(scores.data_ptr(), rows, share, dscores.data_ptr(),
                                                        ws.data_ptr(), stream), "fc_nce_loss_backward")
                else:
                    teacher_scores = ops.similarity(tv_all, tt_all, alpha=ts_scale).contiguous()
                    kd = ops.teacher_student_nce_loss(scores, teacher_scores)
                    loss = kd * ts_scale ** 2
                    _lib.check(lib.fc_kd_loss_backward(scores.data_ptr(), teacher_scores.data_ptr(), rows, cols,
                                                       share * ts_scale ** 2, dscores.data_ptr(), ws.data_ptr(), stream),
                               "fc_kd_loss_backward")
                    if self.fit_temperature:
                        # d/d(ts_ls) [kd(S, e^ts_ls X) e^(2 ts_ls)] = e^(2 ts_ls) (sum dkd/dT * T + 2 kd)
                        tmp = torch.empty(1, dtype=torch.float32, device=dev)
                        _lib.check(lib.fc_kd_teacher_scale_grad(scores.data_ptr(), teacher_scores.data_ptr(), rows, cols,
                                                                tmp.data_ptr(), ws.data_ptr(), stream),
                                   "fc_kd_teacher_scale_grad")
                        self.scale_grads[1] += share * ts_scale ** 2 * (tmp[0] + 2.0 * kd)
                if self.fit_temperature:  # scores = e^ls X  =>  d loss / d ls = sum(dscores * scores)
                    _lib.check(lib.fc_dot(dscores.data_ptr(), scores.data_ptr(), rows * cols, 1.0, 1.0,
                                          self.scale_grads.data_ptr(), stream), "fc_dot")
                # scores = scale V T^T  =>  dV = scale dS T,  dT = scale dS^T V   (full gathered matrices; local rows kept)
                dscores_t = torch.empty((cols, rows), dtype=torch.float32, device=dev)
                _lib.check(lib.fc_transpose(dscores.data_ptr(), dscores_t.data_ptr(), rows, cols, stream), "fc_transpose")
                dv_all = ops.gemm_tn(dscores_t, t_all, alpha=scale)
                dt_all = ops.gemm_tn(dscores, v_all, alpha=scale)
                d_video[v0:v0 + nv] = dv_all[rank * nv:(rank + 1) * nv]
                d_text[t0:t0 + nt] = dt_all[rank * nt:(rank + 1) * nt]
                self.last_losses[name] = float(loss)
                total += share * self.last_losses[name]
                v0, t0 = v0 + nv, t0 + nt
        self._pending = (d_video, d_text)
        return total

    def backward(self) -> None:
        if self._pending is None:
            raise _lib.FitclipHipError("backward() needs training_step_end() first")
        d_video, d_text = self._pending
        self._pending = None
        sp, self._split = self._split, None
        if sp is None:
            self.student.backward(d_video, d_text)
            return
        keep, micro, video, ids = sp["keep"], sp["micro"], sp["video"], sp["ids"]
        n = video.shape[0]
        if keep > 0:
            self.student._state = sp["state"]
            self.student.backward(d_video[:keep], d_text[:keep], accumulate=False, reduce_across_ranks=False)
        for s0 in range(keep, n, micro):
            e0 = min(n, s0 + micro)
            self.student.forward(video[s0:e0], {"input_ids": ids[s0:e0]})
            self.student.backward(d_video[s0:e0], d_text[s0:e0], accumulate=s0 > 0, reduce_across_ranks=e0 == n)

    # ----------------------------------------------------------------------------------------------- optimizer_step
    def optimizer_step(self) -> None:
        """AdamW over the encoder and (with `fit_temperature`) the two temperatures, then the clamps of
        video_text_module.py:94-97 and teacher_student.py:179-183."""
        self.student.step()
        if self.fit_temperature:
            s = self.student
            with torch.cuda.device(self.scales.device):
                _lib.check(_lib.load().fc_adamw(self.scales.data_ptr(), self.scale_grads.data_ptr(),
                                                self.scale_m.data_ptr(), self.scale_v.data_ptr(), 2, s.lr, s.betas[0],
                                                s.betas[1], s.eps, s.weight_decay, s.step_count, _lib.current_stream()),
                           "fc_adamw")
            self.scales[:2].clamp_(max=self.max_logit_scale)
        self.logit_scale, self.teacher_student_logit_scale = (float(x) for x in self.scales[:2].tolist())

    # ------------------------------------------------------------------------------------------ checkpoint / resume
    def _optimizer_slots(self) -> List[Tuple[int, str, Any]]:
        """(index, key, where) for every parameter that HAS optimiser state, numbered as the reference's
        `AdamW(self.parameters())` numbers them (`checkpoint.trainable_parameter_indices`): `where` is a temperature slot
        (0 / 1) or the encoder parameter's name in the flat buffers."""
        from .checkpoint import trainable_parameter_indices
        temps = {"logit_scale": 0, "teacher_student_logit_scale": 1}
        prefix = "encoder.model."
        return [(i, key, temps[key] if key in temps else key[len(prefix):])
                for i, key in trainable_parameter_indices(self, self.fit_temperature).items()]

    def checkpoint(self) -> Dict[str, Any]:
        """A training checkpoint a reference run could resume from and vice versa: `state_dict` with the keys, shapes and
        order of the reference module's Lightning checkpoint (`checkpoint.module_state_dict`: `logit_scale`,
        `max_logit_scale`, `teacher_student_logit_scale` of shape [1], `encoder.*`, `teacher.*`, the prompt ids),
        `optimizer_states[0]` in `torch.optim.AdamW.state_dict()` layout with the reference's parameter numbering
        (`self.parameters()` order, cli.py:129-132; `state` entries only for the parameters that receive gradients) and
        `global_step`."""
        from .checkpoint import module_state_dict, reference_parameter_order
        s = self.student
        state = {}
        step = torch.tensor(float(s.step_count))
        shapes = dict(s.model._named_weights())
        for i, _, where in self._optimizer_slots():
            if isinstance(where, int):
                state[i] = {"step": step.clone(), "exp_avg": self.scale_m[where:where + 1].cpu(),
                            "exp_avg_sq": self.scale_v[where:where + 1].cpu()}
            else:
                o, shape = s.offsets[where], shapes[where].shape
                n = int(torch.Size(shape).numel())
                state[i] = {"step": step.clone(), "exp_avg": s.exp_avg[o:o + n].view(shape).cpu(),
                            "exp_avg_sq": s.exp_avg_sq[o:o + n].view(shape).cpu()}
        group = {"lr": s.lr, "betas": s.betas, "eps": s.eps, "weight_decay": s.weight_decay, "amsgrad": False,
                 "maximize": False, "params": list(range(len(reference_parameter_order(self))))}
        return {"state_dict": {k: v.detach().cpu() for k, v in module_state_dict(self).items()},  # host tensors
                "optimizer_states": [{"state": state, "param_groups": [group]}], "global_step": s.step_count}

    def load_checkpoint(self, ckpt: Mapping[str, Any]) -> None:
        """Resumes from `checkpoint()` (or a reference Lightning checkpoint of the same module graph): weights, AdamW
        moments, step count and temperatures; every sum of a step has a fixed order, so the next `fit_step` continues
        bit-for-bit where the saved run would have."""
        from .checkpoint import load_module_state_dict, reference_parameter_order
        s = self.student
        load_module_state_dict(self, ckpt["state_dict"], strict=True)     # copies into the flat parameter buffer (views)
        self.scales[0], self.scales[1] = self.logit_scale, self.teacher_student_logit_scale
        opt = ckpt["optimizer_states"][0]
        g = opt["param_groups"][0]
        n_params = len(reference_parameter_order(self))
        if len(g["params"]) != n_params:
            raise ValueError(f"the checkpoint's optimiser numbers {len(g['params'])} parameters, this module has {n_params} "
                             "(logit scales, encoder, teacher, prompts): not a checkpoint of the same module graph")
        shapes = dict(s.model._named_weights())
        with torch.no_grad():
            for i, key, where in self._optimizer_slots():
                entry = opt["state"].get(i)
                if entry is None:                                   # saved before this parameter's first gradient
                    m = v = None
                else:
                    m, v = entry["exp_avg"], entry["exp_avg_sq"]
                if isinstance(where, int):
                    self.scale_m[where] = 0.0 if m is None else float(m)
                    self.scale_v[where] = 0.0 if v is None else float(v)
                    continue
                o, shape = s.offsets[where], shapes[where].shape
                n = int(torch.Size(shape).numel())
                for buf, src in ((s.exp_avg, m), (s.exp_avg_sq, v)):
                    if src is None:
                        buf[o:o + n].zero_()
                    else:
                        if tuple(src.shape) != tuple(shape):
                            raise ValueError(f"optimiser state {i} ({key}) has shape {tuple(src.shape)}, expected {tuple(shape)}")
                        buf[o:o + n].view(shape).copy_(src)
        s.step_count = int(ckpt["global_step"])
        s.lr, s.betas, s.eps, s.weight_decay = float(g["lr"]), tuple(map(float, g["betas"])), float(g["eps"]), float(g["weight_decay"])
        s.rebind()

    def fit_step(self, batch: Mapping[str, Any]) -> float:
        loss = self.training_step_end(self.training_step(batch))
        # a frozen teacher in precision fp32x3: its embeddings fed this loss - no gradient is applied if its fp16 planes
        # overflowed (FC_ERANGE raises here; `training_step_end` has synchronised with the stream already)
        teacher_model = getattr(self.teacher, "model", None)
        if getattr(teacher_model, "precision", None) == "fp32x3":
            teacher_model.check_range()
        self.backward()
        self.optimizer_step()
        return loss

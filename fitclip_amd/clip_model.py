"""`CLIP`: the dual-encoder model object the reference obtains from `clip.load(...)`, backed by the HIP library.

It is an `nn.Module` that holds the 301 parameters under the OpenAI-CLIP names (so `named_parameters()`,
`state_dict()`, `load_state_dict()`, `copy.deepcopy` and therefore `aligner.wise.wise`, the checkpoint scripts and
parameter-freezing regexes of the reference keep working) and exposes `encode_image`, `encode_text` and
`visual.input_resolution` - the only members `ClipVideoTextEncoder` touches
(/root/reference/aligner/encoder/clip_video_text_encoder.py:84,93,103,115,126).

All arithmetic happens in libfitclip_hip.so (`fc_encode_image` / `fc_encode_text`); there is no PyTorch fallback.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Mapping, Optional, Union

import numpy as np
import torch
from torch import nn

from . import _lib
from .synth import ClipDims, VIT_B_16, parameter_shapes

_PRECISIONS = {"fp32": _lib.PREC_F32, "f32": _lib.PREC_F32, "float32": _lib.PREC_F32,
               "bf16": _lib.PREC_BF16, "bfloat16": _lib.PREC_BF16,
               # fp32 results from the bf16 matrix cores: the block GEMMs of the visual tower over split-fp32 operands
               # (three bf16 numbers per value, six bf16 products per fp32 product: fc_config.split_gemm), all else fp32
               "fp32x6": _lib.PREC_F32,
               # ... from the fp16 matrix cores: two fp16 planes per value, THREE fp16 products per fp32 product (split_gemm = 2)
               "fp32x3": _lib.PREC_F32}
_SPLIT_GEMM = {"fp32x6": 1, "fp32x3": 2}   # precision -> fc_config.split_gemm
# Opt-in: slices of a big image batch go to this many HIP streams (`CLIP._encode_image_lanes`).  4 gives ~1 % more
# throughput at 2048 frames, but kernels of different slices then overlap in time, so per-kernel durations (the roofline
# evidence of bench.py and rocprofv3) stop describing a kernel that owns the chip: off (1) by default.
_IMAGE_STREAMS = max(1, int(os.environ.get("FITCLIP_IMAGE_STREAMS", "1")))
_IGNORED_KEYS = ("logit_scale", "input_resolution", "context_length", "vocab_size")  # JIT-archive leftovers


class _Bag(nn.Module):
    """Name-space node: only there so that parameters get their dotted OpenAI names."""


class _Runtime:
    """Per-model native state (handle, packed-weight arena, workspaces).  Never copied or pickled."""

    def __init__(self) -> None:
        self.handle = None
        self.key = None
        self.arena = None
        self.fingerprint = None
        self.workspace = {}

    def __deepcopy__(self, memo):
        return _Runtime()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self.__init__()

    def close(self) -> None:
        if self.handle is not None:
            try:
                _lib.load().fc_destroy(self.handle)
            finally:
                self.handle = None
        self.arena = None
        self.workspace = {}
        self.fingerprint = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass


class CLIP(nn.Module):
    def __init__(self, dims: ClipDims = VIT_B_16, precision: str = "bf16", chunk_frames: int = 0,
                 chunk_texts: int = 0, gemm_tile: int = 0, prune_last_block: bool = False, strict_range: bool = False) -> None:
        """`strict_range` (precision "fp32x3" only): every `encode_image` waits for its own range flag and raises FC_ERANGE
        itself (one host synchronisation per call, `fc_range_strict`); by default the flag is reported by the next call and
        by `check_range()`, which every consumer of the embeddings in this package calls before it uses or saves them."""
        super().__init__()
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
        self.dims = dims
        self.precision = precision
        self.chunk_frames, self.chunk_texts, self.gemm_tile = chunk_frames, chunk_texts, gemm_tile
        self.prune_last_block = bool(prune_last_block)
        self.strict_range = bool(strict_range)
        for name, shape in parameter_shapes(dims).items():
            *path, leaf = name.split(".")
            node: nn.Module = self
            for part in path:
                if part not in node._modules:
                    node.add_module(part, _Bag())
                node = node._modules[part]
            node.register_parameter(leaf, nn.Parameter(torch.zeros(shape, dtype=torch.float32), requires_grad=False))
        # The reference's wrapper deletes this attribute when present (clip_video_text_encoder.py:75-77).
        self.logit_scale = nn.Parameter(torch.tensor(float(np.log(1 / 0.07))), requires_grad=False)
        self.visual.input_resolution = dims.image_resolution
        self.context_length = dims.context_length
        self.vocab_size = dims.vocab_size
        self._rt = _Runtime()

    # ------------------------------------------------------------------------------------------------ state dicts
    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor], strict: bool = True):
        sd = OrderedDict((k, v) for k, v in state_dict.items() if k not in _IGNORED_KEYS or k in self.state_dict())
        if "logit_scale" not in sd and hasattr(self, "logit_scale"):
            # the reference re-creates a NaN logit_scale for checkpoints that lack it (clip_video_text_encoder.py:43-53)
            sd["logit_scale"] = torch.tensor(float("nan"))
        self.invalidate_weights()
        return super().load_state_dict(sd, strict=strict)

    def invalidate_weights(self) -> None:
        """Tells the native side that parameter VALUES changed, so the kernel-layout copies (bf16 arena, transposed
        projections) are rebuilt before the next encode.  Called by `load_state_dict`, `_apply` (`.to()`, `.float()`,
        ...) and the trainer's optimiser step.  Writers that go through `param.data` (EMA updates, in-place init: such
        writes do not bump `param._version`) must call it themselves."""
        rt = self.__dict__.get("_rt")
        if rt is not None:
            rt.fingerprint = None

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_weights()
        return super()._apply(fn, *args, **kwargs)

    # ----------------------------------------------------------------------------------------------------- native
    def _device(self) -> torch.device:
        return self.positional_embedding.device

    def _named_weights(self):
        return [(n, p) for n, p in self.named_parameters() if n not in _IGNORED_KEYS]

    def _ensure_ready(self) -> "_Runtime":
        dev = self._device()
        if dev.type != "cuda":
            raise _lib.FitclipHipError(
                f"CLIP parameters are on {dev}; move the model to the ROCm device first (no CPU fallback).")
        lib, rt = _lib.load(), self._rt
        key = (self.precision, self.chunk_frames, self.chunk_texts, self.gemm_tile, self.prune_last_block, dev.index, self.strict_range)
        if rt.handle is None or rt.key != key:
            rt.close()
            d = self.dims
            cfg = _lib.fc_config(d.embed_dim, d.image_resolution, d.vision_layers, d.vision_width, d.vision_patch_size,
                                 d.context_length, d.vocab_size, d.transformer_width, d.transformer_heads,
                                 d.transformer_layers, _PRECISIONS[self.precision], self.chunk_frames,
                                 self.chunk_texts, self.gemm_tile, int(self.prune_last_block),
                                 _SPLIT_GEMM.get(self.precision, 0))
            h = _lib._vp()
            _lib.check(lib.fc_create(cfg, h), "fc_create")
            rt.handle, rt.key = h, key
            if self.strict_range:
                _lib.check(lib.fc_range_strict(h, 1), "fc_range_strict")
        weights = self._named_weights()
        # (pointer, version counter) per parameter: catches re-allocation and ordinary in-place autograd-visible writes;
        # `.data` writes are announced with `invalidate_weights()`.  Inference tensors have no version counter: a model
        # built under torch.inference_mode() is simply repacked on every call.
        fp = None
        if not any(p.is_inference() for _, p in weights):
            fp = tuple((p.data_ptr(), p._version) for _, p in weights)
        if fp is None or fp != rt.fingerprint:
            with torch.cuda.device(dev):
                for name, p in weights:
                    if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                        raise _lib.FitclipHipError(f"parameter {name} must be contiguous float32 on {dev}")
                    shape = (_lib._i64 * p.dim())(*p.shape)
                    _lib.check(lib.fc_set_weight(rt.handle, name.encode(), p.data_ptr(), shape, p.dim()),
                               f"fc_set_weight({name})")
                need = lib.fc_packed_bytes(rt.handle)
                if rt.arena is None or rt.arena.numel() < need or rt.arena.device != dev:
                    rt.arena = torch.empty(need, dtype=torch.uint8, device=dev)
                _lib.check(lib.fc_pack_weights(rt.handle, rt.arena.data_ptr(), rt.arena.numel(),
                                               _lib.current_stream()), "fc_pack_weights")
            rt.fingerprint = fp
        return rt

    def _workspace(self, rt: "_Runtime", tower: int, n: int, lane: int = 0) -> torch.Tensor:
        need = _lib.load().fc_workspace_bytes(rt.handle, tower, n)
        ws = rt.workspace.get((tower, lane))
        if ws is None or ws.numel() < need:
            rt.workspace[(tower, lane)] = ws = torch.empty(need, dtype=torch.uint8, device=self._device())
        return ws

    # --------------------------------------------------------------------------------------------------- encoders
    @torch.no_grad()
    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        """`clip.model.CLIP.encode_image`: f32 [N, 3, R, R] -> f32 [N, embed_dim] (not normalised)."""
        rt = self._ensure_ready()
        d = self.dims
        if image.dim() != 4 or tuple(image.shape[1:]) != (3, d.image_resolution, d.image_resolution):
            raise ValueError(f"expected [N, 3, {d.image_resolution}, {d.image_resolution}], got {tuple(image.shape)}")
        image = image.to(device=self._device(), dtype=torch.float32).contiguous()
        n = image.shape[0]
        out = torch.empty((n, d.embed_dim), dtype=torch.float32, device=image.device)
        if n:
            with torch.cuda.device(image.device):
                lanes = _IMAGE_STREAMS if n >= _IMAGE_STREAMS * self._chunk_frames() else 1
                if lanes == 1:
                    ws = self._workspace(rt, 0, n)
                    _lib.check(_lib.load().fc_encode_image(rt.handle, image.data_ptr(), n, out.data_ptr(),
                                                           ws.data_ptr(), ws.numel(), _lib.current_stream()),
                               "fc_encode_image")
                else:
                    self._encode_image_lanes(rt, image, out, lanes)
        return out

    def check_range(self, wait: bool = True) -> None:
        """precision "fp32x3" keeps the block activations as fp16 planes (|x| <= 65504): raises `FitclipHipError` (FC_ERANGE) if a
        value beyond that - or an infinite / NaN activation or weight - was met since the weights were packed.  `wait`: synchronise with the current stream first (call it
        after the last batch of an evaluation); without it, what the calls completed so far have shown (`encode_image`
        itself checks that on entry).  A no-op in the other precisions."""
        rt = self._rt
        if rt.handle is not None:
            with torch.cuda.device(self._device()):
                _lib.check(_lib.load().fc_range_status(rt.handle, _lib.current_stream(), int(wait)), "fc_range_status")

    def _chunk_frames(self) -> int:
        if self.chunk_frames > 0:
            return self.chunk_frames
        if self.precision in _SPLIT_GEMM:
            return 768 if _SPLIT_GEMM[self.precision] == 1 else 2048
        return 2048  # order of the library's pass size (ViT-B/16; csrc/api.hip planned_chunk)

    def _encode_image_lanes(self, rt: "_Runtime", image: torch.Tensor, out: torch.Tensor, lanes: int) -> None:
        """Large batches (>= `lanes` chunks): contiguous slices of the frames go to `lanes` HIP streams (the caller's +
        side streams), each with its own workspace, so that kernels of different slices interleave: the tail of one
        slice's persistent GEMM overlaps the head of another slice's next kernel.  Measured with 2048 frames: 4 lanes
        -1 % step time, 2 lanes +0.4 % (two persistent GEMMs then mostly fight for the same CUs).  Rows are independent,
        so the result is bit-identical; the caller's stream waits for the side streams before returning."""
        n, res = image.shape[0], 3 * image.shape[2] * image.shape[3]
        chunk = self._chunk_frames()
        per = -(-(-(-n // lanes)) // chunk) * chunk  # ceil(n / lanes) rounded up to whole chunks
        main = torch.cuda.current_stream()
        if not hasattr(rt, "side_streams"):
            rt.side_streams = []
        while len(rt.side_streams) < lanes - 1:
            rt.side_streams.append(torch.cuda.Stream(device=image.device))
        lib = _lib.load()
        start = 0
        used = []
        for lane in range(lanes):
            cnt = min(per, n - start)
            if cnt <= 0:
                break
            stream = main if lane == 0 else rt.side_streams[lane - 1]
            if lane:
                stream.wait_stream(main)
                used.append(stream)
            ws = self._workspace(rt, 0, cnt, lane)
            _lib.check(lib.fc_encode_image(rt.handle, image.data_ptr() + start * res * 4, cnt,
                                           out.data_ptr() + start * out.shape[1] * 4, ws.data_ptr(), ws.numel(),
                                           stream.cuda_stream), "fc_encode_image")
            start += cnt
        for stream in used:
            main.wait_stream(stream)

    @torch.no_grad()
    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """`clip.model.CLIP.encode_text`: int [N, context_length] token ids -> f32 [N, embed_dim] (not normalised)."""
        rt = self._ensure_ready()
        d = self.dims
        if text.dim() != 2 or text.shape[1] != d.context_length:
            raise ValueError(f"expected [N, {d.context_length}] token ids, got {tuple(text.shape)}")
        text = text.to(device=self._device(), dtype=torch.int64).contiguous()
        n = text.shape[0]
        out = torch.empty((n, d.embed_dim), dtype=torch.float32, device=text.device)
        if n:
            with torch.cuda.device(text.device):
                ws = self._workspace(rt, 1, n)
                _lib.check(_lib.load().fc_encode_text(rt.handle, text.data_ptr(), n, out.data_ptr(), ws.data_ptr(),
                                                      ws.numel(), _lib.current_stream()), "fc_encode_text")
        return out

    def forward(self, image: torch.Tensor, text: torch.Tensor):
        return self.encode_image(image), self.encode_text(text)

    # ---------------------------------------------------------------------------------------------- kernel timing
    def profile(self, max_records: int) -> None:
        rt = self._ensure_ready()
        _lib.check(_lib.load().fc_profile_enable(rt.handle, max_records), "fc_profile_enable")

    def profile_select(self, kind_mask: int = 0xFFFFFFFF, epilogue_mask: int = 0xFFFFFFFF) -> None:
        """Restricts the recorded launches (kind 0 = GEMM, 1 = attention, 2 = add+LayerNorm; GEMM epilogue ids)."""
        _lib.check(_lib.load().fc_profile_select(self._ensure_ready().handle, kind_mask, epilogue_mask),
                   "fc_profile_select")

    def profile_reset(self) -> None:
        _lib.check(_lib.load().fc_profile_reset(self._ensure_ready().handle), "fc_profile_reset")

    def profile_records(self, max_records: int = 65536):
        rt = self._ensure_ready()
        buf = (_lib.fc_prof_record * max_records)()
        n = _lib.load().fc_profile_read(rt.handle, buf, max_records)
        if n < 0:
            _lib.check(n, "fc_profile_read")
        return [dict(kind=r.kind, precision=r.precision, epilogue=r.epilogue, tile=r.tile, M=r.M, N=r.N, K=r.K,
                     ms=r.ms) for r in buf[:n]]


def dims_from_state_dict(sd: Mapping[str, torch.Tensor]) -> ClipDims:
    """Infers the architecture from tensor shapes, like `clip.model.build_model` does for a bare state dict."""
    vw = sd["visual.conv1.weight"].shape[0]
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    vlayers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
    tw = sd["ln_final.weight"].shape[0]
    tlayers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    return ClipDims(embed_dim=sd["text_projection"].shape[1], image_resolution=grid * patch, vision_layers=vlayers,
                    vision_width=vw, vision_patch_size=patch, context_length=sd["positional_embedding"].shape[0],
                    vocab_size=sd["token_embedding.weight"].shape[0], transformer_width=tw,
                    transformer_heads=tw // 64, transformer_layers=tlayers)


def build_clip(state_dict: Mapping[str, Union[torch.Tensor, np.ndarray]], precision: str = "bf16",
               device: Optional[Union[str, torch.device]] = None, **kwargs) -> CLIP:
    sd = OrderedDict((k, torch.as_tensor(v)) for k, v in state_dict.items())
    model = CLIP(dims_from_state_dict(sd), precision=precision, **kwargs)
    model.load_state_dict(sd, strict=True)
    return model.to(device) if device is not None else model


def load_clip_model(name: str, precision: str = "bf16", device: Optional[Union[str, torch.device]] = None,
                    **kwargs) -> CLIP:
    """Counterpart of `load_clip_model` (clip_video_text_encoder.py:30-61) for LOCAL checkpoints.

    `name` is a path to a bare OpenAI-CLIP-named state dict (what `scripts/checkpoint_to_state_dict.py` of the
    reference writes), with or without `logit_scale`; or `synthetic:<seed>` / `synthetic-student:<seed>` for seeded ViT-B/16 weights (`synthetic-tiny[-student]:<seed>`: the miniature of the tests).  Model names
    and URLs ("ViT-B/16", "https://...") need a network fetch and are rejected: there is no egress here.
    """
    if name.startswith("synthetic"):
        from . import synth
        seed = int(name.split(":", 1)[1]) if ":" in name else 42
        kind = name.split(":", 1)[0]
        dims = synth.TINY if "-tiny" in kind else VIT_B_16  # `synthetic-tiny[-student]`: the miniature of the tests
        sd = synth.make_state_dict(dims, seed=seed)
        if kind.endswith("-student"):  # a perturbed copy: the "fine-tuned" model2 of a WiSE ensemble / the KD student
            sd = synth.perturbed_state_dict(sd, dims, seed=seed + 1, rel=0.3 if dims is synth.TINY else 0.05)
        return build_clip(sd, precision=precision, device=device, **kwargs)
    if "://" in name or not os.path.exists(name):
        raise FileNotFoundError(f"{name!r}: only local state-dict files or 'synthetic[:seed]' can be loaded offline")
    from .checkpoint import load_state_dict_file  # bare state dict, Lightning .ckpt (prefix stripped) or a pipe
    return build_clip(load_state_dict_file(name), precision=precision, device=device, **kwargs)

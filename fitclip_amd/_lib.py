"""ctypes binding of libfitclip_hip.so (C ABI: include/fitclip_hip.h).

There is NO fallback: if the shared library is missing or a call fails, this module raises.  (`cffi` is not installed
in the target image; the declarations below mirror the header one to one, and tests/test_abi.py checks that every
function the header declares is exported and bound.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

_CSRC = Path(__file__).resolve().parent / "csrc"
LIB_PATH = Path(os.environ.get("FITCLIP_HIP_LIB", _CSRC / "libfitclip_hip.so"))

PREC_F32, PREC_BF16 = 0, 1
EPI_BIAS_T, EPI_GELU_T, EPI_RESID_F32, EPI_PATCH_F32, EPI_STORE_F32, EPI_DGELU_T, EPI_BIAS_F32, EPI_GELU_X3, EPI_RESID3_F32 = range(9)
EPI_GELU_X2 = 10  # fc_gemm_split2: QuickGELU + x2 rows out
FC_ERANGE = -5   # split_gemm = 2: a value beyond fp16's range was met


class FitclipHipError(RuntimeError):
    pass


ABI_VERSION = 5  # FC_ABI_VERSION of include/fitclip_hip.h


class fc_config(C.Structure):
    """`fc_config`; positional arguments start at `embed_dim` - `struct_size` (the first field, which fc_create checks
    before it reads anything else) is filled in here."""
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "embed_dim", "image_resolution", "vision_layers", "vision_width", "vision_patch_size", "context_length",
        "vocab_size", "transformer_width", "transformer_heads", "transformer_layers", "precision", "chunk_frames",
        "chunk_texts", "gemm_tile", "prune_last_block", "split_gemm")]

    def __init__(self, *args, **kw):
        super().__init__(C.sizeof(type(self)), *args, **kw)


class fc_prof_record(C.Structure):
    _fields_ = [("kind", C.c_int32), ("precision", C.c_int32), ("epilogue", C.c_int32), ("tile", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ms", C.c_float)]


_vp, _i32, _i64, _sz, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_size_t, C.c_float, C.c_double

# name -> (restype, argtypes); mirrors include/fitclip_hip.h
SIGNATURES = {
    "fc_create": (_i32, [C.POINTER(fc_config), C.POINTER(_vp)]),
    "fc_destroy": (None, [_vp]),
    "fc_last_error": (C.c_char_p, []),
    "fc_version": (C.c_char_p, []),
    "fc_set_weight": (_i32, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i32]),
    "fc_num_weights": (_i32, [_vp]),
    "fc_weight_name": (C.c_char_p, [_vp, _i32]),
    "fc_packed_bytes": (_sz, [_vp]),
    "fc_pack_weights": (_i32, [_vp, _vp, _sz, _vp]),
    "fc_workspace_bytes": (_sz, [_vp, _i32, _i32]),
    "fc_encode_image": (_i32, [_vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "fc_encode_text": (_i32, [_vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "fc_preprocess_u8": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, C.POINTER(_f32), C.POINTER(_f32), _vp]),
    "fc_pool_normalize": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "fc_l2_normalize": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "fc_similarity": (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _vp, _i32, _vp]),
    "fc_similarity_ranks": (_i32, [_vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _vp]),
    "fc_ranks": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "fc_ranks_of": (_i32, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "fc_group_mean": (_i32, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "fc_nce_loss": (_i32, [_vp, _i32, _vp, _vp, _vp]),
    "fc_kd_loss": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "fc_kd_loss_rect": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "fc_wise": (_i32, [_vp, _vp, _f64, _vp, _sz, _vp]),
    "fc_gemm": (_i32, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "fc_gemm_plan": (_i32, [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "fc_gemm_split2_plan": (_i32, [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]),
    "fc_layernorm": (_i32, [_vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp]),
    "fc_add_layernorm": (_i32, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp]),
    "fc_attention": (_i32, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "fc_convert": (_i32, [_vp, _vp, _i32, _sz, _vp]),
    "fc_split3": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _vp]),
    "fc_gemm_split3": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    "fc_split2": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp]),
    "fc_split2_weight": (_i32, [_vp, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _vp]),
    "fc_gemm_split2": (_i32, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp]),
    "fc_range_status": (_i32, [_vp, _vp, _i32]),
    "fc_range_strict": (_i32, [_vp, _i32]),
    "fc_set_grad": (_i32, [_vp, C.c_char_p, _vp]),
    "fc_train_weights_bytes": (_sz, [_vp]),
    "fc_train_prepare": (_i32, [_vp, _vp, _sz, _vp]),
    "fc_train_arena_bytes": (_sz, [_vp, _i32, _i32]),
    "fc_train_scratch_bytes": (_sz, [_vp, _i32, _i32]),
    "fc_encode_image_train": (_i32, [_vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "fc_encode_text_train": (_i32, [_vp, _vp, _i32, _vp, _vp, _sz, _vp]),
    "fc_encode_image_backward": (_i32, [_vp, _vp, _i32, _vp, _sz, _vp, _sz, _i32, _vp]),
    "fc_encode_text_backward": (_i32, [_vp, _vp, _vp, _i32, _vp, _sz, _vp, _sz, _i32, _vp]),
    "fc_pool_normalize_backward": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "fc_nce_loss_backward": (_i32, [_vp, _i32, _f32, _vp, _vp, _vp]),
    "fc_kd_loss_backward": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp]),
    "fc_kd_teacher_scale_grad": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "fc_gemm_tn_scratch_bytes": (_sz, [_i32, _i32, _i32]),
    "fc_gemm_tn": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp, _i32, _vp, _sz, _vp]),
    "fc_attention_backward": (_i32, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "fc_layernorm_backward_scratch_bytes": (_sz, [_i32]),
    "fc_layernorm_backward": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "fc_token_embedding_backward_scratch_bytes": (_sz, [_i32, _i32]),
    "fc_token_embedding_backward": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _sz, _vp]),
    "fc_dot": (_i32, [_vp, _vp, _sz, _f32, _f32, _vp, _vp]),
    "fc_transpose": (_i32, [_vp, _vp, _i32, _i32, _vp]),
    "fc_adamw": (_i32, [_vp, _vp, _vp, _vp, _sz, _f64, _f64, _f64, _f64, _f64, _i32, _vp]),
    "fc_bpe_create": (_i32, [C.c_char_p, _i32, C.POINTER(_vp)]),
    "fc_bpe_destroy": (None, [_vp]),
    "fc_bpe_vocab_size": (_i32, [_vp]),
    "fc_bpe_sot": (_i32, [_vp]),
    "fc_bpe_eot": (_i32, [_vp]),
    "fc_bpe_encode": (_i32, [_vp, C.c_char_p, C.POINTER(_i64), _i32]),
    "fc_bpe_tokenize": (_i32, [_vp, C.POINTER(C.c_char_p), _i32, _i32, _vp]),
    "fc_bpe_decode": (_i32, [_vp, C.POINTER(_i64), _i32, C.c_char_p, _i32]),
    "fc_profile_enable": (_i32, [_vp, _i32]),
    "fc_profile_select": (_i32, [_vp, C.c_uint32, C.c_uint32]),
    "fc_profile_reset": (_i32, [_vp]),
    "fc_profile_read": (_i32, [_vp, C.POINTER(fc_prof_record), _i32]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Loads the library (once).  `import torch` first so that the HIP runtime torch ships is the one in the process."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise FitclipHipError(
                f"{LIB_PATH} not found: build it with `python -m fitclip_amd.build` (hipcc --offload-arch=gfx950). "
                "fitclip_amd has no CPU fallback.")
        import torch  # noqa: F401  (loads libamdhip64 with torch's RPATH before our DT_NEEDED is resolved)
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().fc_last_error().decode(errors="replace")
        raise FitclipHipError(f"{what or 'libfitclip_hip'} failed ({rc}): {msg}")


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu() -> None:
    import torch
    if not torch.cuda.is_available():
        raise FitclipHipError("fitclip_amd needs a ROCm GPU (gfx950); there is no CPU fallback.")

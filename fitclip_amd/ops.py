"""Python entry points of the single HIP operators (PyTorch-ROCm tensors in, tensors out).  Thin: every function
checks shapes / dtypes / device, allocates the output with torch, and makes ONE C-ABI call on the current stream."""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from ._lib import (EPI_BIAS_F32, EPI_BIAS_T, EPI_GELU_T, EPI_GELU_X2, EPI_GELU_X3, EPI_PATCH_F32, EPI_RESID3_F32, EPI_RESID_F32, EPI_STORE_F32,  # noqa
                   PREC_BF16, PREC_F32)

_KIND = {torch.float32: PREC_F32, torch.bfloat16: PREC_BF16}
KIND_X3 = 3
ATTN_SPLIT = 4  # fc_attention precision: split-fp32 attention (x3 rows out)
KIND_X2 = 4     # row kernels: x2 rows out (two fp16 planes, csrc/common.h)
ATTN_SPLIT_X2 = 5  # fc_attention precision: the split attention with x2 rows out
ATTN_SPLIT2 = 6    # fc_attention precision: three fp16 products per fp32 product (attention_split2.hip), x2 rows out
# KIND_X3: element-kind argument of the row kernels / attention: x3 rows out (three bf16 planes, csrc/common.h)


def _x3_empty(rows: int, cols: int, device) -> torch.Tensor:
    """x3 rows for `cols` fp32 columns: [rows, 4 cols] bf16 positions (every 16 columns one 128-byte line [p1 | p2 | p3 |
    32 bytes no kernel reads or writes]).  Zero-filled so that two images of the same values compare equal."""
    return torch.zeros((rows, 4 * cols), dtype=torch.bfloat16, device=device)


def _x2_empty(rows: int, cols: int, device) -> torch.Tensor:
    """x2 rows for `cols` fp32 columns: [rows, 2 cols] fp16 positions (every 32 columns one 128-byte line [h1 x32 | h2 x32])."""
    return torch.empty((rows, 2 * cols), dtype=torch.float16, device=device)


def _dev(t: torch.Tensor, name: str, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    if t.device.type != "cuda":
        raise _lib.FitclipHipError(f"{name} must live on the ROCm device (got {t.device}); there is no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, epilogue: int = EPI_BIAS_T,
         out: Optional[torch.Tensor] = None, aux: Optional[torch.Tensor] = None, alpha: float = 1.0, patches: int = 0,
         tile: int = 0) -> torch.Tensor:
    """out = epilogue(a[M,K] @ w[N,K]^T).  a / w float32 (exact fp32 MFMA) or bfloat16."""
    _dev(a, "a"), _dev(w, "w", a.dtype)
    M, K = a.shape
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError("inner dimensions differ")
    f32_out = epilogue in (EPI_RESID_F32, EPI_PATCH_F32, EPI_STORE_F32, EPI_BIAS_F32)
    if out is None:
        if epilogue == EPI_RESID_F32:
            raise ValueError("the residual epilogue accumulates into `out`")
        rows = M + M // patches if epilogue == EPI_PATCH_F32 else M
        if epilogue in (EPI_BIAS_F32, EPI_GELU_X3):
            raise ValueError("the split-fp32 epilogues belong to gemm_split3")
        cols = N
        out = torch.empty((rows, cols), dtype=torch.float32 if f32_out else a.dtype, device=a.device)
    _dev(out, "out", torch.float32 if f32_out else a.dtype)
    with torch.cuda.device(a.device):
        _lib.check(_lib.load().fc_gemm(_KIND[a.dtype], epilogue, a.data_ptr(), w.data_ptr(), _ptr(bias), out.data_ptr(),
                                       _ptr(aux), alpha, M, N, K, a.stride(0), w.stride(0), out.stride(0), patches,
                                       tile, _lib.current_stream()), "fc_gemm")
    return out


def gemm_plan(M: int, N: int, K: int) -> "tuple[int, int]":
    """(head_panels, tail_units): how the fp32 persistent GEMM cuts the rows of an [M, N] output over K columns on the current device -
    256-row panels in whole tile rounds, then tiles of `tail_units` x 64 rows (0: none).  fc_gemm_plan."""
    import ctypes as C
    hp, ht = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.load().fc_gemm_plan(M, N, K, C.byref(hp), C.byref(ht)), "fc_gemm_plan")
    return hp.value, ht.value


def gemm_split2_plan(M: int, N: int, compute_units: int = 0) -> "tuple[int, int]":
    """(tile_rows, workgroups): the tile height (256 / 192 / 128) `gemm_split2` picks for an [M, N] problem and the workgroups it launches
    on `compute_units` CUs (0: the current device).  Host arithmetic: needs no GPU when `compute_units` is given.  fc_gemm_split2_plan."""
    import ctypes as C
    rows, wgs = C.c_int32(0), C.c_int32(0)
    _lib.check(_lib.load().fc_gemm_split2_plan(M, N, compute_units, C.byref(rows), C.byref(wgs)), "fc_gemm_split2_plan")
    return rows.value, wgs.value


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_dtype: torch.dtype = torch.float32,
              gather: Optional[torch.Tensor] = None, row_stride: Optional[int] = None, rows: Optional[int] = None
              ) -> torch.Tensor:
    _dev(x, "x", torch.float32), _dev(gamma, "gamma", torch.float32), _dev(beta, "beta", torch.float32)
    D = gamma.numel()
    row_stride = D if row_stride is None else row_stride
    if gather is not None:
        _dev(gather, "gather", torch.int32)
        rows = gather.numel()
    elif rows is None:
        rows = x.numel() // row_stride
    x3 = out_dtype == "x3"  # three-plane rows (split3 layout)
    x2 = out_dtype == "x2"  # two fp16 planes (split2 layout)
    y = (_x3_empty(rows, D, x.device) if x3 else _x2_empty(rows, D, x.device) if x2
         else torch.empty((rows, D), dtype=out_dtype, device=x.device))
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().fc_layernorm(x.data_ptr(), row_stride, _ptr(gather), gamma.data_ptr(), beta.data_ptr(),
                                            y.data_ptr(), y.shape[1], KIND_X3 if x3 else KIND_X2 if x2 else _KIND[out_dtype], rows, D,
                                            _lib.current_stream()), "fc_layernorm")
    return y


def add_layernorm(x: torch.Tensor, delta: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                  write_x: bool = True, three_plane: bool = False, two_plane: bool = False) -> torch.Tensor:
    """x += delta (in place, if write_x); returns LayerNorm(x + delta) in delta's dtype, or (`three_plane`, fp32 delta) as
    x3 rows [rows, 4 D] (split3 layout), or (`two_plane`) as x2 rows [rows, 2 D] (split2 layout)."""
    planes = three_plane or two_plane
    _dev(x, "x", torch.float32), _dev(delta, "delta", torch.float32 if planes else None), _dev(gamma, "gamma", torch.float32)
    rows, D = x.shape
    y = (_x3_empty(rows, D, x.device) if three_plane else _x2_empty(rows, D, x.device) if two_plane
         else torch.empty((rows, D), dtype=delta.dtype, device=x.device))
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().fc_add_layernorm(x.data_ptr(), D, delta.data_ptr(), D, None, gamma.data_ptr(),
                                                beta.data_ptr(), y.data_ptr(), y.shape[1],
                                                KIND_X3 if three_plane else KIND_X2 if two_plane else _KIND[delta.dtype], rows, D,
                                                int(write_x), _lib.current_stream()), "fc_add_layernorm")
    return y


def attention(qkv: torch.Tensor, n_seq: int, seq_len: int, heads: int, causal: bool = False,
              three_plane: bool = False, split: bool = False, two_plane: bool = False, three_products: bool = False) -> torch.Tensor:
    """qkv [n_seq * seq_len, 3 * heads * 64] (float32 or bfloat16) -> [n_seq * seq_len, heads * 64]; `three_plane`
    (float32 qkv, non-causal, 113..224 tokens): x3 rows [.., 4 * heads * 64] of the fp32 result; `split` (193..208 tokens):
    x3 rows too, both products as six bf16 products per fp32 product on the bf16 matrix cores (fp32 accuracy); `split` with
    `two_plane`: the same kernel, x2 rows [.., 2 * heads * 64] out (split2 layout); `three_products` (with `split` and `two_plane`):
    both products as THREE fp16 products per fp32 product (attention_split2.hip: the attention of precision "fp32x3")."""
    if three_products and not (split and two_plane):
        raise ValueError("three_products is a form of the split attention with x2 rows out")
    if two_plane and not split:
        raise ValueError("x2 rows come from the split attention only (split=True)")
    three_plane = (three_plane or split) and not two_plane
    _dev(qkv, "qkv", torch.float32 if (three_plane or two_plane) else None)
    D = heads * 64
    if qkv.shape != (n_seq * seq_len, 3 * D):
        raise ValueError(f"qkv shape {tuple(qkv.shape)} != {(n_seq * seq_len, 3 * D)}")
    out = (_x3_empty(n_seq * seq_len, D, qkv.device) if three_plane else _x2_empty(n_seq * seq_len, D, qkv.device) if two_plane
           else torch.empty((n_seq * seq_len, D), dtype=qkv.dtype, device=qkv.device))
    with torch.cuda.device(qkv.device):
        _lib.check(_lib.load().fc_attention(ATTN_SPLIT2 if three_products else ATTN_SPLIT_X2 if two_plane else ATTN_SPLIT if split else KIND_X3 if three_plane else _KIND[qkv.dtype],
                                            qkv.data_ptr(), out.data_ptr(),
                                            n_seq, seq_len, heads, int(causal), _lib.current_stream()), "fc_attention")
    return out


def split3(x: torch.Tensor) -> torch.Tensor:
    """x3 rows [rows, 4 K bf16 positions] of fp32 rows [rows, K] (fc_split3): x = p1 + p2 + p3 exactly in three bf16 numbers;
    every 16 columns become one 128-byte line [p1 x16 | p2 x16 | p3 x16 | 32 unused bytes].  Operand format of `gemm_split3`
    (activations and weights alike)."""
    _dev(x, "x", torch.float32)
    if x.dim() != 2 or x.shape[1] % 16:
        raise ValueError("split3 needs [rows, K] with K a multiple of 16")
    out = _x3_empty(x.shape[0], x.shape[1], x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().fc_split3(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1],
                                         _lib.current_stream()), "fc_split3")
    return out


def x3_planes(x3: torch.Tensor):
    """The three planes [rows, K] (bfloat16) of x3 rows [rows, 4 K] (a view per plane; test / lab helper)."""
    v = x3.view(x3.shape[0], -1, 4, 16)
    return tuple(v[:, :, i].reshape(x3.shape[0], -1) for i in range(3))


def gemm_split3(a3: torch.Tensor, w3: torch.Tensor, bias: torch.Tensor, epilogue: int = EPI_BIAS_F32,
                out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """epilogue(A @ W^T) for x3 operands a3 [M, 4 K], w3 [N, 4 K] (`split3` images or the x3 outputs of LayerNorm / attention
    / a previous EPI_GELU_X3 GEMM): six bf16 MFMA products per fp32 product, fp32 accumulate (fc_gemm_split3).
    EPI_BIAS_F32 -> float32 [M, N]; EPI_GELU_X3 -> x3 rows [M, 4 N] of QuickGELU(A @ W^T + bias); EPI_RESID3_F32: `out`
    (float32 [M, N]) += A @ W^T + bias, in place."""
    _dev(a3, "a3", torch.bfloat16), _dev(w3, "w3", torch.bfloat16), _dev(bias, "bias", torch.float32)
    if a3.dim() != 2 or w3.dim() != 2 or a3.shape[1] != w3.shape[1] or a3.shape[1] % 64:
        raise ValueError(f"x3 operands need matching [rows, 4 K] shapes, got {tuple(a3.shape)} and {tuple(w3.shape)}")
    M, N, K = a3.shape[0], w3.shape[0], a3.shape[1] // 4
    if epilogue == EPI_GELU_X3:
        out = _x3_empty(M, N, a3.device)
    elif epilogue == EPI_BIAS_F32:
        out = torch.empty((M, N), dtype=torch.float32, device=a3.device)
    elif epilogue == EPI_RESID3_F32:
        if out is None or out.shape != (M, N):
            raise ValueError("the residual epilogue accumulates into `out` [M, N]")
        _dev(out, "out", torch.float32)
    else:
        raise ValueError("gemm_split3 has the epilogues EPI_BIAS_F32, EPI_GELU_X3 and EPI_RESID3_F32")
    with torch.cuda.device(a3.device):
        _lib.check(_lib.load().fc_gemm_split3(epilogue, a3.data_ptr(), w3.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K,
                                              a3.stride(0), w3.stride(0), out.stride(0), _lib.current_stream()),
                   "fc_gemm_split3")
    return out


def split2(x: torch.Tensor, flag: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x2 rows [rows, 2 K fp16 positions] of fp32 ACTIVATION rows [rows, K] (fc_split2): h1 = fp16(x), h2 = fp16((x - h1) 2^11);
    every 32 columns one 128-byte line [h1 x32 | h2 x32].  `flag` (int32 [1], optional): 1 is ORed in when |x| > 65504."""
    _dev(x, "x", torch.float32)
    if x.dim() != 2 or x.shape[1] % 32:
        raise ValueError("split2 needs [rows, K] with K a multiple of 32")
    if flag is not None:
        _dev(flag, "flag", torch.int32)
    out = _x2_empty(x.shape[0], x.shape[1], x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().fc_split2(x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), x.shape[0], x.shape[1],
                                         _ptr(flag), _lib.current_stream()), "fc_split2")
    return out


def split2_weight(w: torch.Tensor, flag: Optional[torch.Tensor] = None):
    """(x2 rows [N, 2 K], scale pair float32 [2] = {s, 1 / s}) of a WEIGHT [N, K] (fc_split2_weight): s the power of two that
    puts max |s w| into [2^14, 2^15); g1 = fp16(s w), g2 = fp16(s w - g1).  Operand pair of `gemm_split2`.  `flag` (int32 [1],
    optional): 1 is ORed in when the tensor holds an infinite or NaN weight."""
    _dev(w, "w", torch.float32)
    if flag is not None:
        _dev(flag, "flag", torch.int32)
    if w.dim() != 2 or w.shape[1] % 32:
        raise ValueError("split2_weight needs [N, K] with K a multiple of 32")
    out = _x2_empty(w.shape[0], w.shape[1], w.device)
    scale = torch.empty(2, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _lib.check(_lib.load().fc_split2_weight(w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), w.shape[0], w.shape[1],
                                                scale.data_ptr(), _ptr(flag), _lib.current_stream()), "fc_split2_weight")
    return out, scale


def x2_planes(x2: torch.Tensor):
    """The two planes [rows, K] (float16) of x2 rows [rows, 2 K] (a view per plane; test / lab helper).  Activations:
    x = h1 + h2 / 2048; weights: s w = g1 + g2."""
    v = x2.view(x2.shape[0], -1, 2, 32)
    return tuple(v[:, :, i].reshape(x2.shape[0], -1) for i in range(2))


def gemm_split2(a2: torch.Tensor, w2: torch.Tensor, scale: torch.Tensor, bias: torch.Tensor, epilogue: int = EPI_BIAS_F32,
                out: Optional[torch.Tensor] = None, flag: Optional[torch.Tensor] = None, cut: int = 0) -> torch.Tensor:
    """epilogue(A @ W^T) for x2 operands a2 [M, 2 K] (`split2`, LayerNorm / attention x2 outputs, a previous EPI_GELU_X2 GEMM) and
    (w2 [N, 2 K], scale) from `split2_weight`: three fp16 MFMA products per fp32 product, fp32 accumulate (fc_gemm_split2).
    EPI_BIAS_F32 -> float32 [M, N]; EPI_GELU_X2 -> x2 rows [M, 2 N] of QuickGELU(A @ W^T + bias); EPI_RESID3_F32: `out`
    (float32 [M, N]) += A @ W^T + bias, in place (the other two write into `out` when one is given).  `cut` (tests): tile height - 0 = the launcher's choice, 1 = 256 rows, 2 = 128 rows, 3 = 192 rows; the
    result does not depend on it."""
    _dev(a2, "a2", torch.float16), _dev(w2, "w2", torch.float16), _dev(bias, "bias", torch.float32), _dev(scale, "scale", torch.float32)
    if a2.dim() != 2 or w2.dim() != 2 or a2.shape[1] != w2.shape[1] or a2.shape[1] % 64:
        raise ValueError(f"x2 operands need matching [rows, 2 K] shapes, got {tuple(a2.shape)} and {tuple(w2.shape)}")
    M, N, K = a2.shape[0], w2.shape[0], a2.shape[1] // 2
    if epilogue == EPI_GELU_X2:
        if out is None:
            out = _x2_empty(M, N, a2.device)
        elif out.shape != (M, 2 * N) or out.dtype != torch.float16 or out.stride(1) != 1:
            raise ValueError("EPI_GELU_X2 writes x2 rows [M, 2 N] (float16)")
    elif epilogue == EPI_BIAS_F32:
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a2.device)
        elif out.shape != (M, N) or out.dtype != torch.float32 or out.stride(1) != 1:
            raise ValueError("EPI_BIAS_F32 writes float32 [M, N]")
    elif epilogue == EPI_RESID3_F32:
        if out is None or out.shape != (M, N):
            raise ValueError("the residual epilogue accumulates into `out` [M, N]")
        _dev(out, "out", torch.float32)
    else:
        raise ValueError("gemm_split2 has the epilogues EPI_BIAS_F32, EPI_GELU_X2 and EPI_RESID3_F32")
    if flag is not None:
        _dev(flag, "flag", torch.int32)
    with torch.cuda.device(a2.device):
        _lib.check(_lib.load().fc_gemm_split2(epilogue, a2.data_ptr(), w2.data_ptr(), scale.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                              M, N, K, a2.stride(0), w2.stride(0), out.stride(0), _ptr(flag), cut, _lib.current_stream()),
                   "fc_gemm_split2")
    return out


def to_bf16(x: torch.Tensor) -> torch.Tensor:
    _dev(x, "x", torch.float32)
    out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.load().fc_convert(x.data_ptr(), out.data_ptr(), PREC_BF16, x.numel(), _lib.current_stream()),
                   "fc_convert")
    return out


def preprocess_u8(frames: torch.Tensor, size: int, mean, std) -> torch.Tensor:
    """uint8 [n, H, W, 3] on the device -> float32 [n, 3, size, size]: the CLIP eval transform (/255, bicubic resize of
    the shorter side, centre crop, mean/std) in one kernel."""
    _dev(frames, "frames", torch.uint8)
    if frames.dim() != 4 or frames.shape[-1] != 3:
        raise ValueError("expected uint8 frames [n, H, W, 3]")
    n, H, W, _ = frames.shape
    out = torch.empty((n, 3, size, size), dtype=torch.float32, device=frames.device)
    m3, s3 = (_lib._f32 * 3)(*mean), (_lib._f32 * 3)(*std)
    if n:
        with torch.cuda.device(frames.device):
            _lib.check(_lib.load().fc_preprocess_u8(frames.data_ptr(), out.data_ptr(), n, H, W, size, m3, s3,
                                                    _lib.current_stream()), "fc_preprocess_u8")
    return out


def pool_normalize(frame_emb: torch.Tensor, n_clips: int, frames: int) -> torch.Tensor:
    """mean over frames of the L2-normalised frame embeddings (clip_video_text_encoder.py:85-89)."""
    _dev(frame_emb, "frame_emb", torch.float32)
    dim = frame_emb.shape[-1]
    out = torch.empty((n_clips, dim), dtype=torch.float32, device=frame_emb.device)
    if n_clips:
        with torch.cuda.device(frame_emb.device):
            _lib.check(_lib.load().fc_pool_normalize(frame_emb.data_ptr(), out.data_ptr(), n_clips, frames, dim,
                                                     _lib.current_stream()), "fc_pool_normalize")
    return out


def l2_normalize(x: torch.Tensor) -> torch.Tensor:
    _dev(x, "x", torch.float32)
    out = torch.empty_like(x)
    if x.shape[0]:
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fc_l2_normalize(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                                   _lib.current_stream()), "fc_l2_normalize")
    return out


def similarity(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """alpha * a @ b^T in exact fp32 (text_video_retrieval.py:50,74)."""
    _dev(a, "a", torch.float32), _dev(b, "b", torch.float32)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise ValueError(f"similarity needs [na, d] and [nb, d], got {tuple(a.shape)} and {tuple(b.shape)}")
    na, nb = a.shape[0], b.shape[0]
    ld = (nb + 3) // 4 * 4
    buf = torch.empty((na, ld), dtype=torch.float32, device=a.device)
    if na and nb:
        bb = b
        if nb != ld:  # the kernel writes 4 columns at a time: pad B's rows, drop the extra columns afterwards
            bb = torch.zeros((ld, b.shape[1]), dtype=torch.float32, device=b.device)
            bb[:nb] = b
        with torch.cuda.device(a.device):
            _lib.check(_lib.load().fc_similarity(a.data_ptr(), bb.data_ptr(), na, ld, a.shape[1], alpha, buf.data_ptr(),
                                                 ld, _lib.current_stream()), "fc_similarity")
    return buf[:, :nb]


def gemm_tn(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0, out: Optional[torch.Tensor] = None,
            beta: float = 0.0) -> torch.Tensor:
    """out[N1, N2] = beta * out + alpha * a[M, N1]^T @ b[M, N2] in exact fp32 (weight gradients dW = dY^T X, and the
    gradients of the similarity matrix w.r.t. the embeddings)."""
    _dev(a, "a", torch.float32), _dev(b, "b", torch.float32)
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise ValueError(f"gemm_tn needs [M, N1] and [M, N2], got {tuple(a.shape)} and {tuple(b.shape)}")
    M, N1 = a.shape
    N2 = b.shape[1]
    if N1 % 4 or N2 % 4:  # the kernel moves 16-byte chunks: zero-pad the column counts, drop the padding afterwards
        if out is not None:
            raise ValueError("column counts that are not multiples of 4 need out=None")
        pad = lambda t, n: torch.nn.functional.pad(t, (0, -n % 4)) if n % 4 else t  # noqa: E731
        return gemm_tn(pad(a, N1).contiguous(), pad(b, N2).contiguous(), alpha)[:N1, :N2].contiguous()
    if out is None:
        if beta != 0.0:
            raise ValueError("beta needs an `out` to accumulate into")
        out = torch.empty((N1, N2), dtype=torch.float32, device=a.device)
    _dev(out, "out", torch.float32)
    lib = _lib.load()
    scratch = torch.empty(lib.fc_gemm_tn_scratch_bytes(M, N1, N2), dtype=torch.uint8, device=a.device)
    with torch.cuda.device(a.device):
        _lib.check(lib.fc_gemm_tn(a.data_ptr(), b.data_ptr(), M, N1, N2, N1, N2, alpha, beta, out.data_ptr(), N2,
                                  scratch.data_ptr(), scratch.numel(), _lib.current_stream()), "fc_gemm_tn")
    return out


def token_embedding_backward(ids: torch.Tensor, d_rows: torch.Tensor, vocab: int, out: Optional[torch.Tensor] = None,
                             accumulate: bool = False) -> torch.Tensor:
    """d_table[id] (+)= sum of d_rows[r] over the rows with ids[r] == id, added in row order (bit-reproducible: no float
    atomics).  ids int64 [rows] (any shape, flattened), d_rows f32 [rows, D] -> f32 [vocab, D]."""
    _dev(ids, "ids", torch.int64), _dev(d_rows, "d_rows", torch.float32)
    rows, D = d_rows.shape
    if ids.numel() != rows:
        raise ValueError(f"{ids.numel()} ids for {rows} gradient rows")
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs an `out` to add to")
        out = torch.zeros((vocab, D), dtype=torch.float32, device=d_rows.device)
    _dev(out, "out", torch.float32)
    if tuple(out.shape) != (vocab, D):
        raise ValueError(f"out must be [{vocab}, {D}], got {tuple(out.shape)}")
    lib = _lib.load()
    scratch = torch.empty(lib.fc_token_embedding_backward_scratch_bytes(rows, vocab), dtype=torch.uint8, device=d_rows.device)
    with torch.cuda.device(d_rows.device):
        _lib.check(lib.fc_token_embedding_backward(ids.data_ptr(), d_rows.data_ptr(), out.data_ptr(), rows, D, vocab,
                                                   int(accumulate), scratch.data_ptr(), scratch.numel(),
                                                   _lib.current_stream()), "fc_token_embedding_backward")
    return out


def similarity_ranks(texts: torch.Tensor, videos: torch.Tensor, target_offset: int = 0,
                     targets: Optional[torch.Tensor] = None, alpha: float = 1.0) -> torch.Tensor:
    """ranks(similarity(texts, videos, alpha), target_offset) - or ranks_of(..., targets) - WITHOUT the [nt, nv] score matrix:
    the comparison against the target column's score runs in the epilogue of the scoring GEMM (fc_similarity_ranks;
    text_video_retrieval.py:70-80, metrics.py:16-20).  Identical ranks, ties included."""
    _dev(texts, "texts", torch.float32), _dev(videos, "videos", torch.float32)
    if texts.dim() != 2 or videos.dim() != 2 or texts.shape[1] != videos.shape[1]:
        raise ValueError(f"similarity_ranks needs [nt, d] and [nv, d], got {tuple(texts.shape)} and {tuple(videos.shape)}")
    nt, nv = texts.shape[0], videos.shape[0]
    if targets is not None:
        targets = _dev(targets.to(device=texts.device, dtype=torch.int32).contiguous(), "targets", torch.int32)
        if targets.numel() != nt:
            raise ValueError("one target per row")
    out = torch.empty((nt,), dtype=torch.int32, device=texts.device)
    if nt:
        with torch.cuda.device(texts.device):
            _lib.check(_lib.load().fc_similarity_ranks(texts.data_ptr(), videos.data_ptr(), nt, nv, texts.shape[1], alpha,
                                                       target_offset, _ptr(targets), out.data_ptr(), _lib.current_stream()),
                       "fc_similarity_ranks")
    return out


def ranks(scores: torch.Tensor, target_offset: int = 0) -> torch.Tensor:
    """Position of column (i + target_offset) in the stable descending order of row i (aligner/metrics.py:16-20)."""
    if scores.device.type != "cuda" or scores.dtype != torch.float32 or scores.stride(1) != 1:
        raise _lib.FitclipHipError("scores must be a float32 ROCm tensor with unit column stride")
    n_rows, n_cols = scores.shape
    out = torch.empty((n_rows,), dtype=torch.int32, device=scores.device)
    if n_rows:
        with torch.cuda.device(scores.device):
            _lib.check(_lib.load().fc_ranks(scores.data_ptr(), scores.stride(0), n_rows, n_cols, target_offset,
                                            out.data_ptr(), _lib.current_stream()), "fc_ranks")
    return out


def ranks_of(scores: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
    """Position of column targets[i] in the stable descending order of row i (zero-shot classification metrics)."""
    if scores.device.type != "cuda" or scores.dtype != torch.float32 or scores.stride(1) != 1:
        raise _lib.FitclipHipError("scores must be a float32 ROCm tensor with unit column stride")
    targets = _dev(targets.to(device=scores.device, dtype=torch.int32).contiguous(), "targets", torch.int32)
    n_rows, n_cols = scores.shape
    if targets.numel() != n_rows:
        raise ValueError("one target per row")
    out = torch.empty((n_rows,), dtype=torch.int32, device=scores.device)
    if n_rows:
        with torch.cuda.device(scores.device):
            _lib.check(_lib.load().fc_ranks_of(scores.data_ptr(), scores.stride(0), n_rows, n_cols, targets.data_ptr(),
                                               out.data_ptr(), _lib.current_stream()), "fc_ranks_of")
    return out


def group_mean(x: torch.Tensor, group: int) -> torch.Tensor:
    """[n * group, dim] -> [n, dim]: mean over each run of `group` consecutive rows."""
    _dev(x, "x", torch.float32)
    if group <= 0 or x.shape[0] % group:
        raise ValueError("row count must be a multiple of the group size")
    out = torch.empty((x.shape[0] // group, x.shape[1]), dtype=torch.float32, device=x.device)
    if out.shape[0]:
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().fc_group_mean(x.data_ptr(), out.data_ptr(), out.shape[0], group, x.shape[1],
                                                 _lib.current_stream()), "fc_group_mean")
    return out


def _square(scores: torch.Tensor, name: str) -> torch.Tensor:
    scores = _dev(scores.contiguous(), name, torch.float32)
    if scores.dim() != 2 or scores.shape[0] != scores.shape[1]:
        raise ValueError(f"{name} must be square")
    return scores


def nce_loss(scores: torch.Tensor) -> torch.Tensor:
    """aligner/loss.py:13-26, reduction "mean"."""
    scores = _square(scores, "scores")
    n = scores.shape[0]
    out = torch.empty((1,), dtype=torch.float32, device=scores.device)
    ws = torch.empty((2 * n,), dtype=torch.float32, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.check(_lib.load().fc_nce_loss(scores.data_ptr(), n, out.data_ptr(), ws.data_ptr(), _lib.current_stream()),
                   "fc_nce_loss")
    return out[0]


def teacher_student_nce_loss(scores: torch.Tensor, teacher_scores: torch.Tensor) -> torch.Tensor:
    """aligner/loss.py:29-39 with reduction "batchmean" (teacher_student.py:72-73); [rows, cols] matrices (square for
    video-caption batches, rectangular for the videos x prompts variant)."""
    scores = _dev(scores.contiguous(), "scores", torch.float32)
    teacher_scores = _dev(teacher_scores.contiguous(), "teacher_scores", torch.float32)
    if scores.dim() != 2 or scores.shape != teacher_scores.shape:
        raise ValueError("scores and teacher_scores must be 2-D and of the same shape")
    rows, cols = scores.shape
    out = torch.empty((1,), dtype=torch.float32, device=scores.device)
    ws = torch.empty((rows + cols,), dtype=torch.float32, device=scores.device)
    with torch.cuda.device(scores.device):
        _lib.check(_lib.load().fc_kd_loss_rect(scores.data_ptr(), teacher_scores.data_ptr(), rows, cols, out.data_ptr(),
                                               ws.data_ptr(), _lib.current_stream()), "fc_kd_loss_rect")
    return out[0]


def wise_axpby(a: torch.Tensor, b: torch.Tensor, weight_for_2: float, out: Optional[torch.Tensor] = None
               ) -> torch.Tensor:
    """(1 - w) * a + w * b (aligner/wise.py:16)."""
    _dev(a, "a", torch.float32), _dev(b, "b", torch.float32)
    if a.shape != b.shape:
        raise ValueError("shape mismatch")
    out = torch.empty_like(a) if out is None else out
    if a.numel():
        with torch.cuda.device(a.device):
            _lib.check(_lib.load().fc_wise(a.data_ptr(), b.data_ptr(), float(weight_for_2), out.data_ptr(), a.numel(),
                                           _lib.current_stream()), "fc_wise")
    return out

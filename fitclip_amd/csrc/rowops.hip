// HBM-bound row kernels around the GEMMs: LayerNorm, patch extraction, embeddings, pooling, conversions, WiSE.
// All are one-wave-per-row or flat grid-stride kernels with 16-byte accesses (guide: Appendix B, Guideline 13).
#include "common.h"

#include <type_traits>
#include <climits>
#include <cmath>

namespace fc {

namespace {

constexpr int kMaxBlocks = 2048;  // grid-stride cap (Guideline 11)

template <typename OutT> __device__ __forceinline__ void put(OutT* p, float v);
template <> __device__ __forceinline__ void put<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void put<bf16>(bf16* p, float v) { *p = static_cast<bf16>(v); }

template <typename OutT> __device__ __forceinline__ void put4(OutT* p, const f32x4& v);
template <> __device__ __forceinline__ void put4<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void put4<bf16>(bf16* p, const f32x4& v) {
  bf16x4 o;
  o[0] = static_cast<bf16>(v[0]); o[1] = static_cast<bf16>(v[1]);
  o[2] = static_cast<bf16>(v[2]); o[3] = static_cast<bf16>(v[3]);
  *reinterpret_cast<bf16x4*>(p) = o;
}

// Output kind "x3": a row of 4 D bf16 positions, the three-plane image of D fp32 values (common.h: every 16 columns one
// 128-byte line [p1 x16 | p2 x16 | p3 x16 | 32 unused bytes]).  x3_t is a 2-byte stand-in element so that row pointers and
// strides count bf16 positions.
struct x3_t { bf16 v; };
// Written by a FULL wave whose lanes 8k .. 8k + 7 hold the eight adjacent 4-column blocks of TWO 16-column lines (the
// LayerNorm kernels: lane l has columns 4 l .. 4 l + 3 of every 256-column stripe).  The octet exchanges its planes with DPP
// (quad_perm inside a quad, row_shr / row_shl by 4 between the two quads; no LDS) and writes each line WHOLE with one
// store instruction - lane j of the octet stores chunk j of [p1 cols 0-7 | p1 8-15 | p2 0-7 | p2 8-15 | p3 0-7 | p3 8-15 | 0 | 0],
// first of the even line, then of the odd one.  The last 32 bytes of a line carry no data, but a line written only in
// part costs a read-modify-write at the memory side (the 96-byte form of this store ran the add+LayerNorm pass at 2.5 TB/s),
// and half lines per instruction (a quad writing 64 bytes twice) reached 3.5 TB/s.
template <int CTRL> __device__ __forceinline__ unsigned dpp_mov(unsigned v) {
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ void store_x3_octet(bf16* row_out, int col, const f32x4& x, int lane) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  bf16x4 p1, p2, p3;
  split3(x, p1, p2, p3);
  const u32x2 a1 = __builtin_bit_cast(u32x2, p1), a2 = __builtin_bit_cast(u32x2, p2), a3 = __builtin_bit_cast(u32x2, p3);
  // inside a quad: lane jq takes its first 8 bytes from quad lane {0, 2, 0, 2}[jq] and its second 8 bytes from {1, 3, 1, 3}[jq],
  // i.e. it holds columns 8 (jq & 1) .. + 7 of its quad's line, of every plane
  auto gather = [&](const u32x2& v) {
    return u32x4{dpp_mov<0x88>(v[0]), dpp_mov<0x88>(v[1]), dpp_mov<0xDD>(v[0]), dpp_mov<0xDD>(v[1])};
  };
  auto shift = [&](const u32x4& v, auto CTRL) {
    constexpr int ctrl = decltype(CTRL)::value;
    return u32x4{dpp_mov<ctrl>(v[0]), dpp_mov<ctrl>(v[1]), dpp_mov<ctrl>(v[2]), dpp_mov<ctrl>(v[3])};
  };
  const u32x4 g1 = gather(a1), g2 = gather(a2), g3 = gather(a3);
  const u32x4 lo3 = shift(g3, std::integral_constant<int, 0x114>{});  // row_shr:4 - lanes 4, 5: the even line's p3 chunks
  const u32x4 hi1 = shift(g1, std::integral_constant<int, 0x104>{});  // row_shl:4 - lanes 0, 1: the odd line's p1 chunks
  const u32x4 hi2 = shift(g2, std::integral_constant<int, 0x104>{});  //             lanes 2, 3: the odd line's p2 chunks
  const int j = lane & 7;
  const u32x4 zero = {0u, 0u, 0u, 0u};
  const u32x4 even = j < 2 ? g1 : (j < 4 ? g2 : (j < 6 ? lo3 : zero));
  const u32x4 odd = j < 2 ? hi1 : (j < 4 ? hi2 : (j < 6 ? g3 : zero));
  char* line = reinterpret_cast<char*>(row_out) + ((col & ~31) / X3_GROUP) * X3_GROUP_BYTES + j * 16;
  *reinterpret_cast<u32x4*>(line) = even;
  *reinterpret_cast<u32x4*>(line + X3_GROUP_BYTES) = odd;
}
// Output kind "x2": a row of 2 D fp16 positions, the two-plane image of D fp32 values (common.h: every 32 columns one 128-byte
// line [h1 x32 | h2 x32], h2 the residual scaled by 2^11).  Written like the x3 rows by a FULL wave: lanes 8k .. 8k + 7 hold the
// eight adjacent 4-column blocks of ONE line; inside a quad the lanes gather 8 columns of either plane (quad_perm), the two
// quads exchange what the other one stores (row_shl / row_shr by 4), and lane j of the octet stores chunk j of
// [h1 cols 0-7 | 8-15 | 16-23 | 24-31 | h2 cols 0-7 | 8-15 | 16-23 | 24-31]: one whole line per store instruction and octet.
struct x2_t { _Float16 v; };
__device__ __forceinline__ void store_x2_octet(_Float16* row_out, int col, const f32x4& x, int lane) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  f16x4 h1, h2;
  split2(x, h1, h2);
  const u32x2 a1 = __builtin_bit_cast(u32x2, h1), a2 = __builtin_bit_cast(u32x2, h2);
  auto gather = [&](const u32x2& v) {  // lane jq of a quad: columns 8 (jq & 1) .. + 7 of the quad's 16
    return u32x4{dpp_mov<0x88>(v[0]), dpp_mov<0x88>(v[1]), dpp_mov<0xDD>(v[0]), dpp_mov<0xDD>(v[1])};
  };
  auto shift = [&](const u32x4& v, auto CTRL) {
    constexpr int ctrl = decltype(CTRL)::value;
    return u32x4{dpp_mov<ctrl>(v[0]), dpp_mov<ctrl>(v[1]), dpp_mov<ctrl>(v[2]), dpp_mov<ctrl>(v[3])};
  };
  const u32x4 g1 = gather(a1), g2 = gather(a2);
  const u32x4 up1 = shift(g1, std::integral_constant<int, 0x104>{});  // row_shl:4 - lanes 2, 3: h1 of the second quad (cols 16-31)
  const u32x4 dn2 = shift(g2, std::integral_constant<int, 0x114>{});  // row_shr:4 - lanes 4, 5: h2 of the first quad (cols 0-15)
  const int j = lane & 7;
  const u32x4 v = j < 2 ? g1 : (j < 4 ? up1 : (j < 6 ? dn2 : g2));
  char* line = reinterpret_cast<char*>(row_out) + (col / X2_GROUP) * X2_GROUP_BYTES + j * 16;
  *reinterpret_cast<u32x4*>(line) = v;
}
// 4 columns starting at column c of an output row (called by every lane of the wave, lane l on columns .. + 4 l)
template <typename OutT> __device__ __forceinline__ void put4_at(OutT* row, int c, const f32x4& v) { put4<OutT>(row + c, v); }
template <> __device__ __forceinline__ void put4_at<x3_t>(x3_t* row, int c, const f32x4& v) {
  store_x3_octet(reinterpret_cast<bf16*>(row), c, v, (int)(threadIdx.x & 63));
}
template <> __device__ __forceinline__ void put4_at<x2_t>(x2_t* row, int c, const f32x4& v) {
  store_x2_octet(reinterpret_cast<_Float16*>(row), c, v, (int)(threadIdx.x & 63));
}
template <typename OutT> constexpr int kOutCols = 1;       // output elements per input column
template <> constexpr int kOutCols<x3_t> = X3_GROUP_BYTES / 2 / X3_GROUP;  // 4 bf16 positions per column
template <> constexpr int kOutCols<x2_t> = X2_GROUP_BYTES / 2 / X2_GROUP;  // 2 fp16 positions per column
template <typename OutT> constexpr bool kPlanes = std::is_same_v<OutT, x3_t> || std::is_same_v<OutT, x2_t>;  // plane rows: widths of 256 k only

// ---------------------------------------------------------------------------------------------- LayerNorm
// One wave per row, the row lives in registers (D / 64 floats per lane), two-pass mean / centred variance in fp32
// (reference: slip.py:350-356 computes LayerNorm in float32, eps 1e-5).
template <int D, typename OutT>
__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x, long x_stride,
                                                        const int* __restrict__ gather,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, OutT* __restrict__ y,
                                                        long y_stride, int rows) {
  constexpr int V4 = D / 256;          // float4 per lane (D multiple of 256) ...
  constexpr int REM = (D % 256) / 64;  // ... plus scalars for D = 128 style widths
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
    const long src = gather ? gather[row] : row;
    const float* xr = x + src * x_stride;
    f32x4 v[V4 > 0 ? V4 : 1];
    float s[REM > 0 ? REM : 1];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      s[i] = xr[V4 * 256 + i * 64 + lane];
      sum += s[i];
    }
    const float mean = wave_sum(sum) * (1.f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] -= mean;
        sq += v[i][e] * v[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      s[i] -= mean;
      sq += s[i] * s[i];
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) * (1.f / D) + 1e-5f);
#ifdef FITCLIP_LAB
    // (tools/ only, FITCLIP_LAB_LN_FUSE=1: what a statistics-only pass would cost - mean and 1 / std of the row, no normalised row)
    if (y_stride < 0) {
      if constexpr (std::is_same<OutT, float>::value) {
        if (lane == 0) *reinterpret_cast<f32x2*>(y + (long)row * -y_stride) = f32x2{mean, rstd};
      }
      continue;
    }
#endif
    OutT* yr = y + (long)row * y_stride;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      const int c = i * 256 + lane * 4;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[i][e] * rstd * g4[e] + b4[e];
      put4_at<OutT>(yr, c, o);
    }
    if constexpr (REM > 0) {
#pragma unroll
      for (int i = 0; i < REM; ++i) {
        const int c = V4 * 256 + i * 64 + lane;
        if constexpr (!kPlanes<OutT>) put<OutT>(yr + c, s[i] * rstd * gamma[c] + beta[c]);
      }
    }
  }
}

// Entry of the visual tower in ONE pass over the fp32 stream:  x[r] = LN_pre(row r)  (written back: it is the residual
// stream) and  y[r] = LN_1(x[r])  of the first block, where row r is the patch-embed output, or class_embedding +
// positional_embedding[0] for the first token of every image (the CLS row is never materialised before).  Same
// arithmetic, in the same order, as cls_pos + layernorm(ln_pre) + layernorm(ln_1); one read of x instead of two and two
// launches fewer (clip.model.VisionTransformer.forward: cat CLS, + pos, ln_pre, then the first block's ln_1).
template <int D, typename OutT>
__global__ void __launch_bounds__(256) layernorm_pair_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                             const float* __restrict__ pos0, int tokens,
                                                             const float* __restrict__ g0, const float* __restrict__ b0,
                                                             const float* __restrict__ g1, const float* __restrict__ b1,
                                                             OutT* __restrict__ y, int rows) {
  constexpr int V4 = D / 256, REM = (D % 256) / 64;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
    float* xr = x + (long)row * D;
    const bool is_cls = row % tokens == 0;  // wave-uniform
    f32x4 v[V4 > 0 ? V4 : 1];
    float s[REM > 0 ? REM : 1];
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      const int c = i * 256 + lane * 4;
      v[i] = is_cls ? *reinterpret_cast<const f32x4*>(cls + c) + *reinterpret_cast<const f32x4*>(pos0 + c)
                    : *reinterpret_cast<const f32x4*>(xr + c);
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      const int c = V4 * 256 + i * 64 + lane;
      s[i] = is_cls ? cls[c] + pos0[c] : xr[c];
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const float* gamma = pass ? g1 : g0;
      const float* beta = pass ? b1 : b0;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < V4; ++i) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
      for (int i = 0; i < REM; ++i) sum += s[i];
      const float mean = wave_sum(sum) * (1.f / D);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < V4; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[i][e] -= mean;
          sq += v[i][e] * v[i][e];
        }
      }
#pragma unroll
      for (int i = 0; i < REM; ++i) {
        s[i] -= mean;
        sq += s[i] * s[i];
      }
      const float rstd = 1.f / sqrtf(wave_sum(sq) * (1.f / D) + 1e-5f);
#pragma unroll
      for (int i = 0; i < V4; ++i) {
        const int c = i * 256 + lane * 4;
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = v[i][e] * rstd * g4[e] + b4[e];
        if (pass == 0) *reinterpret_cast<f32x4*>(xr + c) = v[i];
        else put4_at<OutT>(y + (long)row * (D * kOutCols<OutT>), c, v[i]);
      }
#pragma unroll
      for (int i = 0; i < REM; ++i) {
        const int c = V4 * 256 + i * 64 + lane;
        s[i] = s[i] * rstd * gamma[c] + beta[c];
        if (pass == 0) xr[c] = s[i];
        else if constexpr (!kPlanes<OutT>) put<OutT>(y + (long)row * D + c, s[i]);
      }
    }
  }
}

// Fused residual add + LayerNorm:  v = x[r] + delta[r];  (x[r] = v);  y[i] = LN(v).  `delta` is the projection output
// a GEMM just wrote (element type T, as y), so the residual update costs no extra pass over the fp32 stream and the
// GEMM epilogue stays a pure store (reference: x = x + attn(ln_1(x)); x = x + mlp(ln_2(x)), slip.py:382-385).
template <int D, typename T, typename YT = T>
__global__ void __launch_bounds__(256) add_layernorm_kernel(float* __restrict__ x, long x_stride,
                                                            const T* __restrict__ delta, long d_stride,
                                                            const int* __restrict__ gather,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, YT* __restrict__ y,
                                                            long y_stride, int rows, int write_x,
                                                            int delta_compact, float* __restrict__ x_out) {
  static_assert(D % 256 == 0 || D == 128, "width");
  constexpr int V4 = D / 256, REM = (D % 256) / 64;
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (int row = blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += gridDim.x * wpb) {
    const long src = gather ? gather[row] : row;
    const float* xr = x + src * x_stride;
    float* xw = (x_out ? x_out : x) + src * x_stride;  // training keeps every block's input: the sum goes elsewhere
    const T* dr = delta + (delta_compact ? (long)row : src) * d_stride;
    f32x4 v[V4 > 0 ? V4 : 1];
    float s[REM > 0 ? REM : 1];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      const int c = i * 256 + lane * 4;
      v[i] = *reinterpret_cast<const f32x4*>(xr + c);
      if constexpr (sizeof(T) == 2) {
        const bf16x4 d4 = *reinterpret_cast<const bf16x4*>(dr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] += static_cast<float>(d4[e]);
      } else {
        v[i] += *reinterpret_cast<const f32x4*>(dr + c);
      }
      if (write_x) *reinterpret_cast<f32x4*>(xw + c) = v[i];
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      const int c = V4 * 256 + i * 64 + lane;
      s[i] = xr[c] + static_cast<float>(dr[c]);
      if (write_x) xw[c] = s[i];
      sum += s[i];
    }
    const float mean = wave_sum(sum) * (1.f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] -= mean;
        sq += v[i][e] * v[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      s[i] -= mean;
      sq += s[i] * s[i];
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) * (1.f / D) + 1e-5f);
    YT* yr = y + (long)row * y_stride;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      const int c = i * 256 + lane * 4;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + c);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(beta + c);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = v[i][e] * rstd * g4[e] + b4[e];
      put4_at<YT>(yr, c, o);
    }
    if constexpr (REM > 0) {
#pragma unroll
      for (int i = 0; i < REM; ++i) {
        const int c = V4 * 256 + i * 64 + lane;
        if constexpr (!kPlanes<YT>) put<YT>(yr + c, s[i] * rstd * gamma[c] + beta[c]);
      }
    }
  }
}

template <typename T, typename YT = T>
int add_layernorm_dispatch(float* x, long xs, const void* delta, long ds, const int* gather, const float* g,
                           const float* b, void* y, long ys, int rows, int D, int write_x, int delta_compact,
                           float* x_out, hipStream_t st) {
  const int blocks = min((rows + 3) / 4, kMaxBlocks);
  const T* dl = reinterpret_cast<const T*>(delta);
  YT* yo = reinterpret_cast<YT*>(y);
  constexpr bool kX3 = kPlanes<YT>;
#define FC_ADDLN(W) hipLaunchKernelGGL((add_layernorm_kernel<W, T, YT>), dim3(blocks), dim3(256), 0, st, x, xs, dl, ds, gather, g, b, yo, ys, rows, write_x, delta_compact, x_out)
  switch (D) {
    case 128: if constexpr (kX3) return fail(FC_EINVAL, "add_layernorm: three-plane output needs a width that is a multiple of 256"); else FC_ADDLN(128); break;
    case 256: FC_ADDLN(256); break;
    case 512: FC_ADDLN(512); break;
    case 768: FC_ADDLN(768); break;
    case 1024: FC_ADDLN(1024); break;
    default: return fail(FC_EINVAL, "add_layernorm: unsupported width %d", D);
  }
#undef FC_ADDLN
  FC_CHECK_LAUNCH("add_layernorm");
  return FC_OK;
}

template <typename OutT>
int layernorm_dispatch(const float* x, long xs, const int* gather, const float* g, const float* b, void* y, long ys,
                       int rows, int D, hipStream_t st) {
  const int blocks = min((rows + 3) / 4, kMaxBlocks);
  OutT* yo = reinterpret_cast<OutT*>(y);
  switch (D) {
    case 128:
      if constexpr (kPlanes<OutT>) return fail(FC_EINVAL, "layernorm: plane-row output needs a width that is a multiple of 256");
      else hipLaunchKernelGGL((layernorm_kernel<128, OutT>), dim3(blocks), dim3(256), 0, st, x, xs, gather, g, b, yo, ys, rows);
      break;
    case 256: hipLaunchKernelGGL((layernorm_kernel<256, OutT>), dim3(blocks), dim3(256), 0, st, x, xs, gather, g, b, yo, ys, rows); break;
    case 512: hipLaunchKernelGGL((layernorm_kernel<512, OutT>), dim3(blocks), dim3(256), 0, st, x, xs, gather, g, b, yo, ys, rows); break;
    case 768: hipLaunchKernelGGL((layernorm_kernel<768, OutT>), dim3(blocks), dim3(256), 0, st, x, xs, gather, g, b, yo, ys, rows); break;
    case 1024: hipLaunchKernelGGL((layernorm_kernel<1024, OutT>), dim3(blocks), dim3(256), 0, st, x, xs, gather, g, b, yo, ys, rows); break;
    default: return fail(FC_EINVAL, "layernorm: unsupported width %d", D);
  }
  FC_CHECK_LAUNCH("layernorm");
  return FC_OK;
}

// ------------------------------------------------------------------------------------------- patch extraction
// frames f32 [n, 3, R, R] -> patches T [n * g * g, 3 * p * p] with k = c * p * p + py * p + px (the flattening of
// conv1.weight [width, 3, p, p]).  Each thread moves 4 consecutive pixels of one patch row (16-byte read).
template <typename OutT>
__global__ void __launch_bounds__(256) im2col_kernel(const float* __restrict__ frames, OutT* __restrict__ out, int n,
                                                     int R, int p) {
  const int g = R / p;
  const long total4 = (long)n * 3 * R * R / 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const long e = i * 4;
    const int x = (int)(e % R);
    const int y = (int)((e / R) % R);
    const int c = (int)((e / ((long)R * R)) % 3);
    const long img = e / ((long)3 * R * R);
    const f32x4 v = *reinterpret_cast<const f32x4*>(frames + e);
    const int gy = y / p, py = y - gy * p, gx = x / p, px = x - gx * p;
    const long row = img * g * g + gy * g + gx;
    put4<OutT>(out + row * (3L * p * p) + c * p * p + py * p + px, v);
  }
}

// Same mapping for ANY patch size (14: 3 * 196 = 588 columns) and a row stride Kp >= 3 p^2 padded to the GEMM's K
// granule: one thread per output element, columns k >= 3 p^2 are written as zeros (the packed conv weight is padded
// with zeros as well, so the extra K steps add exact zeros).
template <typename OutT>
__global__ void __launch_bounds__(256) im2col_padded_kernel(const float* __restrict__ frames, OutT* __restrict__ out,
                                                            int n, int R, int p, int Kp) {
  const int g = R / p, K = 3 * p * p;
  const long total = (long)n * g * g * Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    const long row = i / Kp;
    float v = 0.f;
    if (k < K) {
      const int c = k / (p * p), rem = k - c * p * p, py = rem / p, px = rem - py * p;
      const int gx = (int)(row % g), gy = (int)((row / g) % g);
      const long img = row / ((long)g * g);
      v = frames[((img * 3 + c) * R + gy * p + py) * (long)R + gx * p + px];
    }
    put<OutT>(out + i, v);
  }
}

// out[r, 0..Kp) = in[r, 0..K) followed by zeros (packing of conv1.weight when 3 p^2 is not a multiple of the K granule)
template <typename OutT>
__global__ void __launch_bounds__(256) convert_rows_kernel(const float* __restrict__ in, OutT* __restrict__ out,
                                                           long rows, int K, int Kp) {
  const long total = rows * Kp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kp);
    put<OutT>(out + i, k < K ? in[(i / Kp) * K + k] : 0.f);
  }
}

// ---------------------------------------------------------------------------------- eval preprocessing (N1)
// uint8 frames [n, H, W, 3] -> fp32 NCHW [n, 3, R, R]: /255, bicubic resize of the shorter side to R (PyTorch
// semantics: A = -0.75, half-pixel centres, border taps clamped), centre crop, (x - mean) / std.  One thread per output
// pixel, all three channels (the reference does this on the CPU per sample: clip_video_text_encoder.py:125-133).
__device__ __forceinline__ void cubic_weights(float t, float w[4]) {
  const float A = -0.75f;
  const float t1 = t + 1.f, t2 = 1.f - t, t3 = 2.f - t;
  w[0] = ((A * t1 - 5.f * A) * t1 + 8.f * A) * t1 - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * t2 - (A + 3.f)) * t2 * t2 + 1.f;
  w[3] = ((A * t3 - 5.f * A) * t3 + 8.f * A) * t3 - 4.f * A;
}

__global__ void __launch_bounds__(256) preprocess_u8_kernel(const unsigned char* __restrict__ in,
                                                            float* __restrict__ out, int n, int H, int W, int nh,
                                                            int nw, int R, int top, int left, float3 mean,
                                                            float3 inv_std) {
  const float sy = (float)H / (float)nh, sx = (float)W / (float)nw;
  const long total = (long)n * R * R;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % R), y = (int)((i / R) % R);
    const long img = i / ((long)R * R);
    const float ry = sy * ((float)(y + top) + 0.5f) - 0.5f, rx = sx * ((float)(x + left) + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry), ix = (int)floorf(rx);
    float wy[4], wx[4];
    cubic_weights(ry - (float)iy, wy);
    cubic_weights(rx - (float)ix, wx);
    float acc[3] = {0.f, 0.f, 0.f};
    const unsigned char* base = in + img * (long)H * W * 3;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int yy = min(max(iy - 1 + a, 0), H - 1);
      float row[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int xx = min(max(ix - 1 + b, 0), W - 1);
        const unsigned char* p = base + ((long)yy * W + xx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) row[c] += wx[b] * ((float)p[c] / 255.f);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[c] += wy[a] * row[c];
    }
    const long plane = (long)R * R;
    float* o = out + img * 3 * plane + (long)y * R + x;
    o[0] = (acc[0] - mean.x) * inv_std.x;
    o[plane] = (acc[1] - mean.y) * inv_std.y;
    o[2 * plane] = (acc[2] - mean.z) * inv_std.z;
  }
}

// ------------------------------------------------------------------------------------------- text embedding
// x[n, l, :] = token_embedding[ids[n, l]] + positional_embedding[l]; eot[n] = n * L + argmax(ids[n, :]) (first
// maximum, as torch.argmax; reference slip.py:469-470,478).  One block per text.
__global__ void __launch_bounds__(256) text_embed_kernel(const int64_t* __restrict__ ids,
                                                         const float* __restrict__ tok,
                                                         const float* __restrict__ pos, float* __restrict__ x,
                                                         int* __restrict__ eot, int L, int D, int vocab) {
  const int n = blockIdx.x;
  const int64_t* row = ids + (long)n * L;
  if (threadIdx.x < 64) {
    long long best = LLONG_MIN;
    int besti = 0x7fffffff;
    for (int l = threadIdx.x; l < L; l += 64) {
      const long long v = row[l];
      if (v > best) { best = v; besti = l; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const long long ob = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(besti, o, 64);
      if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if (threadIdx.x == 0) eot[n] = n * L + besti;
  }
  const int d4 = D / 4;
  for (int i = threadIdx.x; i < L * d4; i += blockDim.x) {
    const int l = i / d4, c = (i - l * d4) * 4;
    long id = row[l];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);  // never read outside the table
    const f32x4 t = *reinterpret_cast<const f32x4*>(tok + id * D + c);
    const f32x4 p = *reinterpret_cast<const f32x4*>(pos + (long)l * D + c);
    *reinterpret_cast<f32x4*>(x + ((long)n * L + l) * D + c) = t + p;
  }
}

// ------------------------------------------------------------------------------------- pooling / normalisation
// out[b] = mean_f( e[b, f] / ||e[b, f]|| )   (clip_video_text_encoder.py:85,89: NOT re-normalised).  Block per clip.
__global__ void __launch_bounds__(256) pool_normalize_kernel(const float* __restrict__ e, float* __restrict__ out,
                                                             int frames, int dim) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int per = (dim + 255) / 256;  // <= 4 for dim <= 1024
  float accv[4] = {0.f, 0.f, 0.f, 0.f};
  for (int f = 0; f < frames; ++f) {
    const float* r = e + ((long)b * frames + f) * dim;
    float v[4], sq = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256;
      v[i] = (i < per && c < dim) ? r[c] : 0.f;
      sq += v[i] * v[i];
    }
    sq = wave_sum(sq);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    const float nrm = sqrtf(red[0] + red[1] + red[2] + red[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) accv[i] += v[i] / nrm;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256;
    if (i < per && c < dim) out[(long)b * dim + c] = frames == 1 ? accv[i] : accv[i] / (float)frames;
  }
}

// dst[i] = src[row(i)] for rows of `bytes` bytes (multiple of 16), row(i) = idx ? idx[i] : i * step.  Compacts the pooled
// token rows (CLS / EOT) in front of the row-pruned last block.
__global__ void __launch_bounds__(256) gather_rows_kernel(const char* __restrict__ src, const int* __restrict__ idx,
                                                          long step, char* __restrict__ dst, int n, int bytes) {
  const int per_row = bytes >> 4;
  const long total = (long)n * per_row;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / per_row;
    const int c = (int)(i - r * per_row);
    const long sr = idx ? idx[r] : r * step;
    reinterpret_cast<f32x4*>(dst + r * bytes)[c] = reinterpret_cast<const f32x4*>(src + sr * bytes)[c];
  }
}

// out[g] = mean over the `group` consecutive rows of group g (template averaging of the zero-shot label prompts,
// video_text_classification.py:88-90).  Sequential fp32 sum then one division, as torch.mean over a short axis.
__global__ void __launch_bounds__(256) group_mean_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                         int group, int dim) {
  const int g = blockIdx.x;
  for (int c = threadIdx.x; c < dim; c += blockDim.x) {
    float acc = 0.f;
    for (int k = 0; k < group; ++k) acc += in[((long)g * group + k) * dim + c];
    out[(long)g * dim + c] = acc / (float)group;
  }
}

template <typename OutT>
__global__ void __launch_bounds__(256) convert_kernel(const float* __restrict__ in, OutT* __restrict__ out, size_t n4,
                                                      size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    put4<OutT>(out + i * 4, *reinterpret_cast<const f32x4*>(in + i * 4));
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) put<OutT>(out + n4 * 4 + threadIdx.x, in[n4 * 4 + threadIdx.x]);
}

// out[c, r] = in[r, c]  (weight packing of visual.proj / text_projection: [K, N] -> [N, K]); tiny, run once.
template <typename OutT>
__global__ void __launch_bounds__(256) transpose_convert_kernel(const float* __restrict__ in, OutT* __restrict__ out,
                                                                int rows, int cols) {
  __shared__ float tile[32][33];
  const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (by + j < rows && bx + tx < cols) tile[j][tx] = in[(long)(by + j) * cols + bx + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (bx + j < cols && by + tx < rows) put<OutT>(out + (long)(bx + j) * rows + by + tx, tile[tx][j]);
}

// out = (1 - w) * a + w * b   (aligner/wise.py:16), evaluated exactly as torch does: two roundings of the products
// then one add (no fma contraction), so the result is bit-identical to the reference expression.
__global__ void __launch_bounds__(256) wise_kernel(const float* __restrict__ a, const float* __restrict__ b, float w1,
                                                   float w2, float* __restrict__ out, size_t n4, size_t n) {
#pragma clang fp contract(off)  // hipcc contracts a*b+c into fma by default; torch evaluates mul, mul, add
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(a + i * 4);
    const f32x4 y = *reinterpret_cast<const f32x4*>(b + i * 4);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p1 = w1 * x[e], p2 = w2 * y[e];
      o[e] = p1 + p2;
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    const float p1 = w1 * a[i], p2 = w2 * b[i];
    out[i] = p1 + p2;
  }
}

inline int flat_blocks(size_t work_items) {
  size_t b = (work_items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > kMaxBlocks ? kMaxBlocks : b));
}

}  // namespace

int launch_layernorm(const float* x, long x_stride, const int* gather, const float* gamma, const float* beta, void* y,
                     long y_stride, int out_kind, int rows, int D, hipStream_t stream) {
  if (rows <= 0) return FC_OK;
  if ((x_stride % 4) || (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y) & 15))
    return fail(FC_EINVAL, "layernorm: operands must be 16-byte aligned");
  if (out_kind == KIND_X3) return layernorm_dispatch<x3_t>(x, x_stride, gather, gamma, beta, y, y_stride, rows, D, stream);
  if (out_kind == KIND_X2) {
    if ((y_stride % 64) || y_stride < x2_row_elems(D) || ((uintptr_t)y & 127))
      return fail(FC_EINVAL, "layernorm: x2 rows need 128-byte aligned rows of >= 2 D fp16 positions");
    return layernorm_dispatch<x2_t>(x, x_stride, gather, gamma, beta, y, y_stride, rows, D, stream);
  }
  return out_kind == 1 ? layernorm_dispatch<bf16>(x, x_stride, gather, gamma, beta, y, y_stride, rows, D, stream)
                       : layernorm_dispatch<float>(x, x_stride, gather, gamma, beta, y, y_stride, rows, D, stream);
}

template <typename OutT>
int layernorm_pair_dispatch(float* x, const float* cls, const float* pos0, int tokens, const float* g0, const float* b0,
                            const float* g1, const float* b1, void* y, int rows, int D, hipStream_t st) {
  const int blocks = min((rows + 3) / 4, kMaxBlocks);
  OutT* yo = reinterpret_cast<OutT*>(y);
  switch (D) {
    case 128:
      if constexpr (kPlanes<OutT>) return fail(FC_EINVAL, "layernorm_pair: plane-row output needs a width that is a multiple of 256");
      else hipLaunchKernelGGL((layernorm_pair_kernel<128, OutT>), dim3(blocks), dim3(256), 0, st, x, cls, pos0, tokens, g0, b0, g1, b1, yo, rows);
      break;
    case 256: hipLaunchKernelGGL((layernorm_pair_kernel<256, OutT>), dim3(blocks), dim3(256), 0, st, x, cls, pos0, tokens, g0, b0, g1, b1, yo, rows); break;
    case 512: hipLaunchKernelGGL((layernorm_pair_kernel<512, OutT>), dim3(blocks), dim3(256), 0, st, x, cls, pos0, tokens, g0, b0, g1, b1, yo, rows); break;
    case 768: hipLaunchKernelGGL((layernorm_pair_kernel<768, OutT>), dim3(blocks), dim3(256), 0, st, x, cls, pos0, tokens, g0, b0, g1, b1, yo, rows); break;
    case 1024: hipLaunchKernelGGL((layernorm_pair_kernel<1024, OutT>), dim3(blocks), dim3(256), 0, st, x, cls, pos0, tokens, g0, b0, g1, b1, yo, rows); break;
    default: return fail(FC_EINVAL, "layernorm_pair: unsupported width %d", D);
  }
  FC_CHECK_LAUNCH("layernorm_pair");
  return FC_OK;
}

int launch_layernorm_pair(float* x, const float* cls, const float* pos0, int tokens, const float* g0, const float* b0,
                          const float* g1, const float* b1, void* y, int out_kind, int rows, int D,
                          hipStream_t stream) {
  if (rows <= 0) return FC_OK;
  if (tokens <= 0 ||
      (((uintptr_t)x | (uintptr_t)cls | (uintptr_t)pos0 | (uintptr_t)g0 | (uintptr_t)b0 | (uintptr_t)g1 | (uintptr_t)b1 |
        (uintptr_t)y) & 15))
    return fail(FC_EINVAL, "layernorm_pair: operands must be 16-byte aligned");
  if (out_kind == KIND_X3) return layernorm_pair_dispatch<x3_t>(x, cls, pos0, tokens, g0, b0, g1, b1, y, rows, D, stream);
  if (out_kind == KIND_X2) {
    if ((uintptr_t)y & 127) return fail(FC_EINVAL, "layernorm_pair: x2 rows must be 128-byte aligned");
    return layernorm_pair_dispatch<x2_t>(x, cls, pos0, tokens, g0, b0, g1, b1, y, rows, D, stream);
  }
  return out_kind == 1 ? layernorm_pair_dispatch<bf16>(x, cls, pos0, tokens, g0, b0, g1, b1, y, rows, D, stream)
                       : layernorm_pair_dispatch<float>(x, cls, pos0, tokens, g0, b0, g1, b1, y, rows, D, stream);
}

int launch_add_layernorm(float* x, long x_stride, const void* delta, long d_stride, const int* gather,
                         const float* gamma, const float* beta, void* y, long y_stride, int kind, int rows, int D,
                         int write_x, int delta_compact, hipStream_t stream, float* x_out) {
  if (rows <= 0) return FC_OK;
  if (kind == KIND_X3) {  // delta fp32 (a split-fp32 GEMM's output), y the three-plane image of the LayerNorm output
    if ((x_stride % 4) || (d_stride % 4) || (y_stride % 64) || y_stride < x3_row_elems(D) ||
        (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)delta | (uintptr_t)x_out) & 15))
      return fail(FC_EINVAL, "add_layernorm: operands must be 16-byte aligned");
    return add_layernorm_dispatch<float, x3_t>(x, x_stride, delta, d_stride, gather, gamma, beta, y, y_stride, rows, D,
                                               write_x, delta_compact, x_out, stream);
  }
  if (kind == KIND_X2) {  // delta fp32, y the two-plane fp16 image of the LayerNorm output
    if ((x_stride % 4) || (d_stride % 4) || (y_stride % 64) || y_stride < x2_row_elems(D) || ((uintptr_t)y & 127) ||
        (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)delta | (uintptr_t)x_out) & 15))
      return fail(FC_EINVAL, "add_layernorm: operands must be 16-byte aligned (x2 rows: 128)");
    return add_layernorm_dispatch<float, x2_t>(x, x_stride, delta, d_stride, gather, gamma, beta, y, y_stride, rows, D,
                                               write_x, delta_compact, x_out, stream);
  }
  const int esz = kind == 1 ? 2 : 4;
  if ((x_stride % 4) || (d_stride * esz) % 8 || (y_stride * esz) % 8 ||
      (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)y | (uintptr_t)delta) & 15))
    return fail(FC_EINVAL, "add_layernorm: operands must be 16-byte aligned");
  if (x_out && ((uintptr_t)x_out & 15)) return fail(FC_EINVAL, "add_layernorm: x_out must be 16-byte aligned");
  return kind == 1 ? add_layernorm_dispatch<bf16>(x, x_stride, delta, d_stride, gather, gamma, beta, y, y_stride, rows,
                                                  D, write_x, delta_compact, x_out, stream)
                   : add_layernorm_dispatch<float>(x, x_stride, delta, d_stride, gather, gamma, beta, y, y_stride, rows,
                                                   D, write_x, delta_compact, x_out, stream);
}

int launch_im2col(const float* frames, void* patches, int out_kind, int n, int res, int patch, int Kp,
                  hipStream_t stream) {
  if (n <= 0) return FC_OK;
  const int K = 3 * patch * patch;
  if (patch <= 0 || res % patch || Kp < K) return fail(FC_EINVAL, "im2col: resolution %d / patch %d / Kp %d", res, patch, Kp);
  if (patch % 4 == 0 && Kp == K) {  // 16-byte moves of 4 consecutive pixels of a patch row
    const int blocks = flat_blocks((size_t)n * 3 * res * res / 4);
    if (out_kind == 1)
      hipLaunchKernelGGL(im2col_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, frames, (bf16*)patches, n, res, patch);
    else
      hipLaunchKernelGGL(im2col_kernel<float>, dim3(blocks), dim3(256), 0, stream, frames, (float*)patches, n, res, patch);
  } else {
    const int g = res / patch;
    const int blocks = flat_blocks((size_t)n * g * g * Kp);
    if (out_kind == 1)
      hipLaunchKernelGGL(im2col_padded_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, frames, (bf16*)patches, n, res,
                         patch, Kp);
    else
      hipLaunchKernelGGL(im2col_padded_kernel<float>, dim3(blocks), dim3(256), 0, stream, frames, (float*)patches, n,
                         res, patch, Kp);
  }
  FC_CHECK_LAUNCH("im2col");
  return FC_OK;
}

int launch_convert_rows(const float* in, void* out, int out_kind, long rows, int K, int Kp, hipStream_t stream) {
  if (rows <= 0) return FC_OK;
  if (Kp < K) return fail(FC_EINVAL, "convert_rows: Kp %d < K %d", Kp, K);
  const int blocks = flat_blocks((size_t)rows * Kp);
  if (out_kind == 1)
    hipLaunchKernelGGL(convert_rows_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, in, (bf16*)out, rows, K, Kp);
  else
    hipLaunchKernelGGL(convert_rows_kernel<float>, dim3(blocks), dim3(256), 0, stream, in, (float*)out, rows, K, Kp);
  FC_CHECK_LAUNCH("convert_rows");
  return FC_OK;
}

int launch_preprocess_u8(const unsigned char* frames, float* out, int n, int H, int W, int R, const float* mean3,
                         const float* std3, hipStream_t stream) {
  if (n <= 0) return FC_OK;
  if (H <= 0 || W <= 0 || R <= 0) return fail(FC_EINVAL, "preprocess: %dx%d -> %d", H, W, R);
  // torchvision Resize(int): shorter side = R, longer side = int(R * long / short); then CenterCrop(R)
  const int nh = H <= W ? R : (int)((long)R * H / W), nw = H <= W ? (int)((long)R * W / H) : R;
  const int top = (int)lroundf((nh - R) / 2.f), left = (int)lroundf((nw - R) / 2.f);
  const float3 mean = make_float3(mean3[0], mean3[1], mean3[2]);
  const float3 inv = make_float3(1.f / std3[0], 1.f / std3[1], 1.f / std3[2]);
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(flat_blocks((size_t)n * R * R)), dim3(256), 0, stream, frames, out, n,
                     H, W, nh, nw, R, top, left, mean, inv);
  FC_CHECK_LAUNCH("preprocess_u8");
  return FC_OK;
}

int launch_text_embed(const int64_t* ids, const float* tok, const float* pos, float* x, int* eot, int n, int L, int D,
                      int vocab, hipStream_t stream) {
  if (n <= 0) return FC_OK;
  if (D % 4) return fail(FC_EINVAL, "text_embed: width %d", D);
  hipLaunchKernelGGL(text_embed_kernel, dim3(n), dim3(256), 0, stream, ids, tok, pos, x, eot, L, D, vocab);
  FC_CHECK_LAUNCH("text_embed");
  return FC_OK;
}

int launch_pool_normalize(const float* frame_emb, float* out, int n_clips, int frames, int dim, hipStream_t stream) {
  if (n_clips <= 0) return FC_OK;
  if (frames <= 0 || dim <= 0 || dim > 1024) return fail(FC_EINVAL, "pool_normalize: frames=%d dim=%d", frames, dim);
  hipLaunchKernelGGL(pool_normalize_kernel, dim3(n_clips), dim3(256), 0, stream, frame_emb, out, frames, dim);
  FC_CHECK_LAUNCH("pool_normalize");
  return FC_OK;
}

int launch_gather_rows(const void* src, const int* idx, long step, void* dst, int n, int row_bytes, hipStream_t stream) {
  if (n <= 0) return FC_OK;
  if (row_bytes % 16 || (((uintptr_t)src | (uintptr_t)dst) & 15)) return fail(FC_EINVAL, "gather_rows: alignment");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(flat_blocks((size_t)n * (row_bytes >> 4))), dim3(256), 0, stream,
                     (const char*)src, idx, step, (char*)dst, n, row_bytes);
  FC_CHECK_LAUNCH("gather_rows");
  return FC_OK;
}

int launch_group_mean(const float* in, float* out, int n_groups, int group, int dim, hipStream_t stream) {
  if (n_groups <= 0) return FC_OK;
  if (group <= 0 || dim <= 0) return fail(FC_EINVAL, "group_mean: group=%d dim=%d", group, dim);
  hipLaunchKernelGGL(group_mean_kernel, dim3(n_groups), dim3(256), 0, stream, in, out, group, dim);
  FC_CHECK_LAUNCH("group_mean");
  return FC_OK;
}

int launch_l2_normalize(const float* in, float* out, int n, int dim, hipStream_t stream) {
  return launch_pool_normalize(in, out, n, 1, dim, stream);
}

int launch_convert(const float* in, void* out, int out_kind, size_t n, hipStream_t stream) {
  if (n == 0) return FC_OK;
  const size_t n4 = n / 4;
  const int blocks = flat_blocks(n4 ? n4 : 1);
  if (out_kind == 1)
    hipLaunchKernelGGL(convert_kernel<bf16>, dim3(blocks), dim3(256), 0, stream, in, (bf16*)out, n4, n);
  else
    hipLaunchKernelGGL(convert_kernel<float>, dim3(blocks), dim3(256), 0, stream, in, (float*)out, n4, n);
  FC_CHECK_LAUNCH("convert");
  return FC_OK;
}

int launch_transpose_convert(const float* in, void* out, int out_kind, int rows, int cols, hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return FC_OK;
  dim3 grid((cols + 31) / 32, (rows + 31) / 32);
  if (out_kind == 1)
    hipLaunchKernelGGL(transpose_convert_kernel<bf16>, grid, dim3(256), 0, stream, in, (bf16*)out, rows, cols);
  else
    hipLaunchKernelGGL(transpose_convert_kernel<float>, grid, dim3(256), 0, stream, in, (float*)out, rows, cols);
  FC_CHECK_LAUNCH("transpose_convert");
  return FC_OK;
}

int launch_wise(const float* a, const float* b, double w, float* out, size_t n, hipStream_t stream) {
  if (n == 0) return FC_OK;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return fail(FC_EINVAL, "wise: unaligned operand");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(wise_kernel, dim3(flat_blocks(n4 ? n4 : 1)), dim3(256), 0, stream, a, b, (float)(1.0 - w),
                     (float)w, out, n4, n);
  FC_CHECK_LAUNCH("wise");
  return FC_OK;
}

}  // namespace fc

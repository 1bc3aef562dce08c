// Host-side dispatch of the two-plane fp16 split-fp32 GEMM (kernel: gemm_split2.h), the fp32 -> x2 row converter and the weight
// packer with its per-tensor power-of-two scale (computed on the device: no host synchronisation at pack time).
#include "gemm_split2.h"

#include <algorithm>

namespace fc {

namespace {

constexpr int split2_lds(int bm) { return 2 * (bm + 256) * 128 + 8 * 2048 + 2048; }  // two stages, eight patches, two bias slices
[[maybe_unused]] constexpr int kSplit2Lds = split2_lds(256);   // (tools/split2_lab)

// tiles of the busiest XCD under the kernel's schedule (gemm_split2.h: an XCD owns ceil(panels / (8 / groups)) M-panels of 1 / groups of
// the column tiles)
int x2_xcd_tiles(int M, int N, int bmt, int nsplit) {
  const int tilesM = (M + bmt - 1) / bmt, tilesN = (N + 255) / 256;
  const int ngrp = (nsplit > 1 && 8 % nsplit == 0 && tilesN % nsplit == 0) ? nsplit : 1;
  const int nx = 8 / ngrp;
  return ((tilesM + nx - 1) / nx) * (tilesN / ngrp);
}

template <int EPI, int RW, int BMT>
int launch_x2_tiles(const GemmArgs& a, hipStream_t stream) {
  // SPREAD = 5: the LDS-DMA pieces of a K-step are issued 3, 3, 2 behind the first three MFMA groups of the step before it is
  // needed, none in the hand-over (tools/x2k_lab: up to 1.7 % over two behind each of the first four groups, which round 5 had
  // measured 2 - 4 % over a burst at the hand-over)
  auto kern = gemm_split2_kernel<EPI, 0, 5, RW, 0, BMT>;
  constexpr int lds = split2_lds(BMT);
  if (raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "gemm_split2: cannot raise dynamic LDS to %d bytes", lds);
  // workgroups: the kernel's schedule gives every XCD (blockIdx & 7) a contiguous range of M-panels, so a launch with fewer tiles
  // than CUs needs as many workgroups per XCD as its BUSIEST XCD has tiles - min(tiles, CUs) left that XCD with a second round (20
  // panels x 2 column tiles on 40 workgroups: 6 tiles for 5 workgroups).  Workgroups without a tile return at once.
  hipLaunchKernelGGL(kern, dim3(std::min(8 * x2_xcd_tiles(a.M, a.N, BMT, a.nsplit), device_cus())), dim3(512), lds, stream, a);
  FC_CHECK_LAUNCH("gemm_split2");
  return FC_OK;
}

// Tile height: 256, 192 or 128 rows (wave tiles of 128 / 96 / 64 x 64).  The kernel is persistent; what a launch costs is the number of
// ROUNDS of its busiest XCD (cus / 8 workgroups each, x2_xcd_tiles) times the height of a tile, over the efficiency of that height (fewer
// MFMAs per staged weight tile: 0.93 for 192 rows, 0.83 for 128 - fitted below).  The big tile stays unless another height is at least 7 %
// cheaper by that count.  tools/x2_cut_probe.py (profiles/r06_x2_cut_probe.log; ms per launch at 256 / 192 / 128 rows; the model picks the
// measured best in every row, its ties included):
//   25 216 rows (the reference-shaped call of 128 frames): out_proj 0.119 / 0.097 / 0.108 (297 big tiles = two rounds for 1.16 rounds of
//     work), c_proj 0.373 / 0.320 / 0.398, QKV 0.215 / 0.211 / 0.249, c_fc 0.293 / 0.306 / 0.322;
//   12 608 rows: out_proj 0.056 / 0.049 / 0.071, c_proj 0.166 / 0.145 / 0.222, QKV 0.116 / 0.122 / 0.127;
//   6304 rows (a 32-frame call): QKV 0.098 / 0.079 / 0.067, out_proj 0.050 / 0.041 / 0.035, c_fc 0.104 / 0.084 / 0.096, c_proj 0.152 / 0.125 / 0.105;
//   the text tower at 256 captions (19 712 rows, width 512): out_proj 0.045 / 0.040 / 0.052, c_proj 0.116 / 0.103 / 0.154, QKV 0.083 / 0.089 / 0.092;
//   50 432 rows and more: the big tile (403 456 rows: c_fc 4.57 / 4.71 / 5.33).
// (Round 5 had measured the fp32 kernel's remedy for partial rounds - a head launch of whole rounds plus a tail launch of lower tiles - as
// slower than either: at the power limit a second launch only adds its fill and drain.)
// Rows are independent and an element's K order only depends on its column tile: the result does not depend on the tile height.
// `force`: 0 = by this rule, 1 = 256-row tiles, 2 = 128-row tiles, 3 = 192-row tiles (tests, tools/x2_cut_probe.py).
int x2_tile_height(int M, int N, int cus, int force, int nsplit) {
  if (force == 1) return 256;
  if (force == 2) return 128;
  if (force == 3) return 192;
  const int per_xcd = std::max(1, cus / 8);
  auto cost = [&](int bmt, float eff) {
    const int rounds = (x2_xcd_tiles(M, N, bmt, nsplit) + per_xcd - 1) / per_xcd;
    return (float)rounds * ((float)bmt / 256.f) / eff;
  };
  const float c192 = cost(192, 0.93f), c128 = cost(128, 0.83f);
  int best = 256;
  float cb = 0.93f * cost(256, 1.f);
  if (c192 < cb) best = 192, cb = c192;
  if (c128 < cb) best = 128;
  return best;
}

template <int EPI, int RW = 4>
int launch_x2_variant(const GemmArgs& a, hipStream_t stream, int force) {
  switch (x2_tile_height(a.M, a.N, device_cus(), force, a.nsplit)) {
    case 128: return launch_x2_tiles<EPI, RW, 128>(a, stream);
    case 192: return launch_x2_tiles<EPI, RW, 192>(a, stream);
    default: return launch_x2_tiles<EPI, RW, 256>(a, stream);
  }
}

// fp32 rows -> x2 rows: thread per (row, line of 32 columns, quarter): 8 values -> 16 bytes of either plane; the four threads of a
// line write its 128 bytes
__global__ void __launch_bounds__(256) split2_rows_kernel(const float* __restrict__ in, long ld_in, char* __restrict__ out,
                                                          long ld_out_bytes, long rows, int K, const float* __restrict__ scale2,
                                                          int* sat_flag) {
  const long per_row = K / 8;  // 8-column pieces
  const long total = rows * per_row;
  const bool weight = scale2 != nullptr;
  const float s = weight ? scale2[0] : 1.f;
  // largest |x| as a BIT PATTERN (integer order = float order for non-negative floats, and every NaN sorts above infinity: fmaxf
  // would drop a NaN operand and the flag would stay down)
  unsigned amax = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / per_row;
    const int piece = (int)(i - row * per_row), group = piece >> 2, quarter = piece & 3;
    const float* src = in + row * ld_in + piece * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(src) * s, hi = *reinterpret_cast<const f32x4*>(src + 4) * s;
    f16x4 a1, a2, b1, b2;
    if (weight) {
      split2w(lo, a1, a2);
      split2w(hi, b1, b2);
    } else {
      split2(lo, a1, a2);
      split2(hi, b1, b2);
#pragma unroll
      for (int e = 0; e < 4; ++e) amax = max(amax, max(__float_as_uint(lo[e]) & 0x7fffffffu, __float_as_uint(hi[e]) & 0x7fffffffu));
    }
    char* dst = out + row * ld_out_bytes + (long)group * X2_GROUP_BYTES + quarter * 16;
    auto put = [&](char* p, const f16x4& x, const f16x4& y) {
      f16x8 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = x[e]; v[4 + e] = y[e]; }
      *reinterpret_cast<f16x8*>(p) = v;
    };
    put(dst, a1, b1);
    put(dst + 64, a2, b2);
  }
  if (sat_flag && amax > 0x477fe000u) atomicOr(sat_flag, 1);   // 0x477fe000 = 65504.f; infinities and NaNs lie above
}

// The patch-embedding operand of the three-product mode: frames f32 NCHW [n, 3, R, R] -> x2 rows of the im2col matrix [n g^2, 2 * 3 p^2
// fp16 positions] (row = (image, gy, gx), column k = (channel, py, px), as `im2col` lays it out), thread per (row, 8 columns): two
// 16-byte loads of a patch row's pixels, 16 bytes to either plane.  p % 8 == 0 (a piece never crosses a patch row), R % 4 == 0.
__global__ void __launch_bounds__(256) im2col_x2_kernel(const float* __restrict__ frames, char* __restrict__ out, long ld_out_bytes, long rows,
                                                        int R, int p, int* sat_flag) {
  const int g = R / p, pp = p * p, per_row = 3 * pp / 8;
  const long total = rows * per_row;
  unsigned amax = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / per_row;
    const int piece = (int)(i - row * per_row), group = piece >> 2, quarter = piece & 3;
    const int k = piece * 8, c = k / pp, rem = k - c * pp, py = rem / p, px = rem - py * p;
    const int gx = (int)(row % g), gy = (int)((row / g) % g);
    const long img = row / ((long)g * g);
    const float* src = frames + ((img * 3 + c) * R + gy * p + py) * (long)R + gx * p + px;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
    f16x4 a1, a2, b1, b2;
    split2(lo, a1, a2);
    split2(hi, b1, b2);
#pragma unroll
    for (int e = 0; e < 4; ++e) amax = max(amax, max(__float_as_uint(lo[e]) & 0x7fffffffu, __float_as_uint(hi[e]) & 0x7fffffffu));
    char* dst = out + row * ld_out_bytes + (long)group * X2_GROUP_BYTES + quarter * 16;
    f16x8 v1, v2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v1[e] = a1[e]; v1[4 + e] = b1[e]; v2[e] = a2[e]; v2[4 + e] = b2[e]; }
    *reinterpret_cast<f16x8*>(dst) = v1;
    *reinterpret_cast<f16x8*>(dst + 64) = v2;
  }
  if (sat_flag && amax > 0x477fe000u) atomicOr(sat_flag, 1);
}

// max |w| over the tensor as the bits of a non-negative float (integer order = float order; a NaN anywhere in the tensor sorts
// above infinity and survives the reduction - fmaxf would drop it)
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ w, long ld_in, long rows, int K, unsigned* out) {
  const long per_row = K / 4, total = rows * per_row;
  unsigned m = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / per_row;
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + row * ld_in + (i - row * per_row) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) m = max(m, __float_as_uint(v[e]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0u) atomicMax(out, m);
}
// scale2 <- {s, 1 / s}: s = 2^(14 - floor(log2 max)), i.e. max |s w| in [2^14, 2^15); an all-zero tensor gets 1.  A tensor with
// an infinite or NaN weight cannot be split: it gets 1 too, its planes hold the infinities / NaNs (loud in every product), and the
// range flag - when the caller has one - is raised
__global__ void scale_from_absmax_kernel(float* scale2, int* sat_flag) {
  const unsigned bits = reinterpret_cast<const unsigned*>(scale2)[0];
  const float m = __uint_as_float(bits);
  float s = 1.f;
  if (bits >= 0x7f800000u) {
    if (sat_flag) atomicOr(sat_flag, 1);
  } else if (m > 0.f) {
    int e;
    frexpf(m, &e);                            // m = f 2^e, f in [0.5, 1)  =>  floor(log2 m) = e - 1
    const int k = max(-100, min(100, 15 - e));
    s = ldexpf(1.f, k);
  }
  scale2[0] = s;
  scale2[1] = 1.f / s;
}

// a LayerNorm output is bounded by sqrt(D) max|gamma| + max|beta| (|x - mean| / std <= sqrt(D - 1)): one wave decides
__global__ void __launch_bounds__(64) ln_bound_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int D, int* flag) {
  unsigned ug = 0u, ub = 0u;   // (bit patterns of |.|: a NaN sorts above infinity instead of being dropped by fmaxf)
  for (int i = threadIdx.x; i < D; i += 64) {
    ug = max(ug, __float_as_uint(gamma[i]) & 0x7fffffffu);
    ub = max(ub, __float_as_uint(beta[i]) & 0x7fffffffu);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    ug = max(ug, (unsigned)__shfl_xor((int)ug, o, 64));
    ub = max(ub, (unsigned)__shfl_xor((int)ub, o, 64));
  }
  const float mg = __uint_as_float(ug), mb = __uint_as_float(ub);
  if (threadIdx.x == 0 && !(sqrtf((float)D) * mg + mb <= 65504.f)) atomicOr(flag, 1);   // (false for a NaN or an infinity too)
}

}  // namespace

int launch_x2_ln_bound(const float* gamma, const float* beta, int D, int* sat_flag, hipStream_t stream) {
  if (!gamma || !beta || !sat_flag || D <= 0) return fail(FC_EINVAL, "x2_ln_bound: bad argument");
  hipLaunchKernelGGL(ln_bound_kernel, dim3(1), dim3(64), 0, stream, gamma, beta, D, sat_flag);
  FC_CHECK_LAUNCH("x2_ln_bound");
  return FC_OK;
}

bool gemm_split2_ok(const GemmArgs& a) {
  return a.M > 0 && a.N > 0 && a.K >= 128 && a.K % 64 == 0 && a.N % 32 == 0 && a.bias != nullptr && a.wscale != nullptr &&
         ((uintptr_t)a.bias & 15) == 0 && (size_t)256 * a.lda * 2 < (1ull << 32) && (size_t)a.N * a.ldw * 2 < (1ull << 32);
}

// the N range of a wide problem is split over XCD groups (gemm_split2.h: nsplit): the weights of a group's column tiles stay in its L2
static int x2_auto_nsplit(int N) { return (N / 256) % 4 == 0 && N / 256 >= 8 && N % 256 == 0 ? 4 : 0; }

// What launch_gemm_split2 will do with an [M, N] problem on `cus` compute units (<= 0: the current device): the tile height it picks and
// the number of workgroups it launches.  Pure host arithmetic (tests/test_launch_plan.py pins it without a GPU).
void gemm_split2_plan(int M, int N, int cus, int* tile_rows, int* workgroups) {
  if (cus <= 0) cus = device_cus();
  const int nsplit = x2_auto_nsplit(N);
  const int rows = x2_tile_height(M, N, cus, 0, nsplit);
  *tile_rows = rows;
  *workgroups = std::min(8 * x2_xcd_tiles(M, N, rows, nsplit), cus);
}

int launch_gemm_split2(int epilogue, const GemmArgs& a, hipStream_t stream, int force_cut) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fail(FC_EINVAL, "gemm_split2: empty problem %dx%dx%d", a.M, a.N, a.K);
  if (a.K % 64 || a.K < 128) return fail(FC_EINVAL, "gemm_split2: K=%d must be a multiple of 64, at least 128", a.K);
  if (a.N % 32) return fail(FC_EINVAL, "gemm_split2: N=%d must be a multiple of 32", a.N);
  const long need = x2_row_elems(a.K);
  if (a.lda < need || a.ldw < need || a.lda % 64 || a.ldw % 64)
    return fail(FC_EINVAL, "gemm_split2: lda=%d / ldw=%d must cover the %ld fp16 positions of an x2 row in whole 128-byte lines",
                a.lda, a.ldw, need);
  if (((uintptr_t)a.A | (uintptr_t)a.W) & 127 || ((uintptr_t)a.C & 15))
    return fail(FC_EINVAL, "gemm_split2: operands must be 128-byte aligned (x2 rows are made of whole lines)");
  if (!a.bias || ((uintptr_t)a.bias & 15)) return fail(FC_EINVAL, "gemm_split2: bias missing or unaligned");
  if (!a.wscale) return fail(FC_EINVAL, "gemm_split2: the weight's scale pair is missing");
  if ((size_t)256 * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldw * 2 >= (1ull << 32))
    return fail(FC_EINVAL, "gemm_split2: the weight (or a 256-row tile of the activations) exceeds the 4 GiB of the kernel's 32-bit row offsets");
  GemmArgs b = a;
  if (b.nsplit == 0) b.nsplit = x2_auto_nsplit(b.N);
  switch (epilogue) {
    case EPI_BIAS_F32:
      if (a.ldc % 4 || a.ldc < a.N) return fail(FC_EINVAL, "gemm_split2: ldc=%d", a.ldc);
      return launch_x2_variant<EPI_BIAS_F32>(b, stream, force_cut);
    case EPI_RESID3_F32:
      if (a.ldc % 4 || a.ldc < a.N) return fail(FC_EINVAL, "gemm_split2: ldc=%d", a.ldc);
      return launch_x2_variant<EPI_RESID3_F32>(b, stream, force_cut);
    case EPI_PATCH_F32:   // the patch embedding: C[m + m / P + 1] = acc + bias + aux[(m % P) + 1]  (aux = positional embedding [P + 1, N])
      if (a.ldc % 4 || a.ldc < a.N || !a.aux || a.P <= 0 || ((uintptr_t)a.aux & 15) || a.N % 4)
        return fail(FC_EINVAL, "gemm_split2: the patch epilogue needs the positional embedding, P > 0 and ldc=%d >= N", a.ldc);
      if (a.M >= (1 << 23)) return fail(FC_EINVAL, "gemm_split2: the patch epilogue indexes at most 2^23 - 1 patch rows per launch (M=%d)", a.M);
      return launch_x2_variant<EPI_PATCH_F32>(b, stream, force_cut);
    case EPI_GELU_X2:
      if (a.ldc % 64 || a.ldc < x2_row_elems(a.N) || ((uintptr_t)a.C & 127))
        return fail(FC_EINVAL, "gemm_split2: the x2 output needs 128-byte aligned rows of >= 2 N fp16 (ldc=%d)", a.ldc);
      return launch_x2_variant<EPI_GELU_X2>(b, stream, force_cut);
  }
  return fail(FC_EINVAL, "gemm_split2: epilogue %d", epilogue);
}

static int check_rows(const char* what, const float* in, long ld_in, void* out, long ld_out, int K) {
  if (K % X2_GROUP || ld_in % 4 || ld_out % 64 || ld_out < x2_row_elems(K) || ((uintptr_t)in & 15) || ((uintptr_t)out & 127))
    return fail(FC_EINVAL, "%s: K %% 32, alignment or row stride", what);
  return FC_OK;
}

int launch_split2_rows(const float* in, long ld_in, void* out, long ld_out, long rows, int K, int* sat_flag, hipStream_t stream) {
  if (rows <= 0) return FC_OK;
  if (int rc = check_rows("split2_rows", in, ld_in, out, ld_out, K)) return rc;
  const long total = rows * (K / 8);
  const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(split2_rows_kernel, dim3(blocks), dim3(256), 0, stream, in, ld_in, static_cast<char*>(out), ld_out * 2, rows, K,
                     (const float*)nullptr, sat_flag);
  FC_CHECK_LAUNCH("split2_rows");
  return FC_OK;
}

int launch_im2col_x2(const float* frames, void* out, long ld_out, int n, int res, int patch, int* sat_flag, hipStream_t stream) {
  if (n <= 0) return FC_OK;
  const int K = 3 * patch * patch;
  if (patch <= 0 || patch % 8 || res % patch || res % 4 || K % X2_GROUP || ld_out % 64 || ld_out < x2_row_elems(K) ||
      ((uintptr_t)frames & 15) || ((uintptr_t)out & 127))
    return fail(FC_EINVAL, "im2col_x2: resolution %d / patch %d (a multiple of 8), alignment or row stride", res, patch);
  const long rows = (long)n * (res / patch) * (res / patch), total = rows * (K / 8);
  hipLaunchKernelGGL(im2col_x2_kernel, dim3((int)std::min<long>((total + 255) / 256, 16384)), dim3(256), 0, stream, frames, static_cast<char*>(out),
                     ld_out * 2, rows, res, patch, sat_flag);
  FC_CHECK_LAUNCH("im2col_x2");
  return FC_OK;
}

int launch_split2_weight(const float* w, long ld_in, void* out, long ld_out, long rows, int K, float* scale2, int* sat_flag,
                         hipStream_t stream) {
  if (rows <= 0) return FC_OK;
  if (int rc = check_rows("split2_weight", w, ld_in, out, ld_out, K)) return rc;
  if (!scale2 || ((uintptr_t)scale2 & 7)) return fail(FC_EINVAL, "split2_weight: scale pair missing or unaligned");
  if (hipMemsetAsync(scale2, 0, 8, stream) != hipSuccess) return fail(FC_ELAUNCH, "split2_weight: memset");
  const long total4 = rows * (K / 4);
  hipLaunchKernelGGL(absmax_kernel, dim3((int)std::min<long>((total4 + 255) / 256, 2048)), dim3(256), 0, stream, w, ld_in, rows, K,
                     reinterpret_cast<unsigned*>(scale2));
  hipLaunchKernelGGL(scale_from_absmax_kernel, dim3(1), dim3(1), 0, stream, scale2, sat_flag);
  const long total = rows * (K / 8);
  hipLaunchKernelGGL(split2_rows_kernel, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, stream, w, ld_in,
                     static_cast<char*>(out), ld_out * 2, rows, K, (const float*)scale2, (int*)nullptr);
  FC_CHECK_LAUNCH("split2_weight");
  return FC_OK;
}

}  // namespace fc

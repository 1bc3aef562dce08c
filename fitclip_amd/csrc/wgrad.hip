// Weight-gradient GEMM ("TN") and the reductions around it, exact fp32 on the fp32-input matrix cores.
//
//   C[N1, N2] = beta * C + alpha * sum_m A[m, N1] * B[m, N2]        A = dY [M, N1], B = X [M, N2]   (both row-major)
//
// i.e. dW[out, in] = dY^T . X for a torch Linear (`F.linear(x, W, b)`), the backward half of the KD training step
// (reference: autograd through aligner/encoder/slip.py:364-385, driven by aligner/teacher_student.py:99-140).
// The reduction runs over the ROW index of both operands, so neither is K-contiguous: tiles are staged row-major
// [16 m-rows][BN columns] by LDS-DMA and the MFMA operands (one fp32 per lane: A[i = lane & 15][k = lane >> 4]) are
// fetched with ds_read_b32 - one m-row per 16 lanes.  ds_read_b32 banks over 32 dwords per 32-lane half, and the two
// m-rows of a half are a multiple of 32 dwords apart, so odd rows are stored with their 16-float column groups swapped
// (col ^ 16): conflict-free.  The swizzle is applied to the per-lane SOURCE address (the DMA destination is lane-linear).
//
// M is split over `splits` workgroups per output tile (the output has far fewer 256x256 tiles than the chip has CUs);
// each writes its partial tile to a scratch buffer and `reduce_partials` sums them in a FIXED order: results are
// deterministic (no float atomics).  Rows past the end of a split read a zero page, so no tail handling in the loop.
#include "common.h"

#include <algorithm>

namespace fc {

namespace {

#ifndef FITCLIP_TN_BK
#define FITCLIP_TN_BK 16
#endif
constexpr int TN_BK = FITCLIP_TN_BK;  // m-rows per LDS stage

struct TnArgs {
  const float* A;
  const float* B;
  float* P;            // [splits, N1, N2] partials
  const float* zeros;  // >= 256 floats of zeros (device)
  int M, N1, N2, lda, ldb;
  int rows_per_split;  // multiple of TN_BK
  int a_skip;          // > 0: A row m is stored at row m + m / a_skip + 1 (token rows of a ViT stream without the CLS rows)
};

template <int BN1, int BN2, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64) gemm_tn_kernel(const TnArgs g) {
  constexpr int NW = WM * WN;
  constexpr int TM = BN1 / WM, TNN = BN2 / WN;
  constexpr int FM = TM / 16, FN = TNN / 16;
  constexpr int STAGE_F = TN_BK * (BN1 + BN2);  // floats per stage
  constexpr int PIECES = STAGE_F / 256;         // 1 KiB LDS-DMA pieces per stage
  constexpr int LPW = PIECES / NW;
  static_assert(PIECES % NW == 0 && BN1 % 64 == 0 && BN2 % 64 == 0, "tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles2 = (g.N2 + BN2 - 1) / BN2;
  const int tiles = ((g.N1 + BN1 - 1) / BN1) * tiles2;
  const int split = blockIdx.x / tiles, tile = blockIdx.x - split * tiles;
  const int n1_0 = (tile / tiles2) * BN1, n2_0 = (tile % tiles2) * BN2;
  const int m_begin = split * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);

  // ---- per-lane staging plan: piece p covers floats [256 p, 256 p + 256) of the stage image [A rows | B rows]
  int prow[LPW], pcol[LPW];
  bool isb[LPW];
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const int o = (wave + i * NW) * 256 + lane * 4;
    isb[i] = o >= TN_BK * BN1;
    const int oo = isb[i] ? o - TN_BK * BN1 : o;
    const int bn = isb[i] ? BN2 : BN1;
    prow[i] = oo / bn;
    const int pc = oo - prow[i] * bn;
    const int lc = pc ^ ((prow[i] & 1) << 4);  // logical column held at this physical position
    const int n0 = isb[i] ? n2_0 : n1_0, nmax = isb[i] ? g.N2 : g.N1;
    pcol[i] = min(n0 + lc, nmax - 4);          // columns past the edge re-read the last chunk (never stored)
  }
  auto stage_load = [&](int stage, int m0) {
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int m = m0 + prow[i];
      const float* src;
      if (m < m_end) {
        if (isb[i]) {
          src = g.B + (size_t)m * g.ldb + pcol[i];
        } else {
          const size_t r = g.a_skip > 0 ? (size_t)m + m / g.a_skip + 1 : (size_t)m;
          src = g.A + r * g.lda + pcol[i];
        }
      } else {
        src = g.zeros + lane * 4;  // a zero A row kills the product; B rows are zeroed as well
      }
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + stage * STAGE_F + (wave + i * NW) * 256),
                                       16, 0, 0);
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r = lane & 15, q = lane >> 4;
  const int nsteps = (m_end - m_begin + TN_BK - 1) / TN_BK;
  if (nsteps > 0) stage_load(0, m_begin);
  for (int st = 0; st < nsteps; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < nsteps) stage_load((st + 1) & 1, m_begin + (st + 1) * TN_BK);
    const float* as = lds + (st & 1) * STAGE_F;
    const float* bs = as + TN_BK * BN1;
#pragma unroll
    for (int ks = 0; ks < TN_BK / 4; ++ks) {
      const int row = ks * 4 + q;              // m-row of this lane's k-slot
      const int sw = (row & 1) << 4;
      float af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = as[row * BN1 + ((wm * TM + i * 16 + r) ^ sw)];
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = bs[row * BN2 + ((wn * TNN + j * 16 + r) ^ sw)];
      // X as the MFMA A-operand, dY as the B-operand: a lane then holds 4 CONSECUTIVE n2 of one n1 (16-byte stores)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
  }
  float* out = g.P + (size_t)split * g.N1 * g.N2;
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int n1 = n1_0 + wm * TM + i * 16 + r;
    if (n1 >= g.N1) continue;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n2 = n2_0 + wn * TNN + j * 16 + 4 * q;
      if (n2 < g.N2) *reinterpret_cast<f32x4*>(out + (size_t)n1 * g.N2 + n2) = acc[i][j];
    }
  }
}

// C[r, c] = beta * C[r, c] + alpha * sum_s P[s][r * cols + c]      (fixed summation order: deterministic)
// (P and C may alias when C is plane 0 of P: every element is read before it is written, by the same thread)
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* P, int splits, size_t plane, int cols,
                                                              float* C, int ldc, float alpha, float beta) {
  const size_t n4 = plane / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 s = *reinterpret_cast<const f32x4*>(P + i * 4);
    for (int k = 1; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(P + (size_t)k * plane + i * 4);
    const size_t e = i * 4, row = e / cols, col = e - row * cols;
    f32x4* dst = reinterpret_cast<f32x4*>(C + row * ldc + col);
    f32x4 v = s * alpha;
    if (beta != 0.f) v += *dst * beta;
    *dst = v;
  }
}

// Column sums of X [rows, cols] (T): stage 1 writes P[chunk][cols] over row chunks; reduce_partials finishes.
// Thread = 4 consecutive columns; a block covers 256 columns x its row chunk with 4 waves on interleaved rows.
template <typename T>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const T* __restrict__ X, long ldx, int rows, int cols,
                                                             int rows_per_chunk, float* __restrict__ P) {
  __shared__ f32x4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strips = (cols + 255) / 256;
  const int chunk = blockIdx.x / strips, strip = blockIdx.x - chunk * strips;
  const int c = strip * 256 + lane * 4;
  const int r0 = chunk * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    for (int row = r0 + wave; row < r1; row += 4) {
      if constexpr (sizeof(T) == 4) {
        s += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(X) + (size_t)row * ldx + c);
      } else {
        const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16*>(X) + (size_t)row * ldx + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += static_cast<float>(v[e]);
      }
    }
  }
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < cols)
    *reinterpret_cast<f32x4*>(P + (size_t)chunk * cols + c) = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

struct TnPlan {
  bool big;
  int splits, rows_per_split, tiles;
};
TnPlan plan_tn(int M, int N1, int N2) {
  TnPlan p{};
  p.big = N1 >= 256 && N2 >= 256;
  const int bn = p.big ? 256 : 128;
  p.tiles = ((N1 + bn - 1) / bn) * ((N2 + bn - 1) / bn);
  const int max_splits = std::max(1, M / (4 * TN_BK));
  p.splits = std::max(1, std::min({max_splits, 64, (device_cus() + p.tiles / 2) / p.tiles}));
  p.rows_per_split = ((M + p.splits - 1) / p.splits + TN_BK - 1) / TN_BK * TN_BK;
  p.splits = (M + p.rows_per_split - 1) / p.rows_per_split;
  return p;
}

template <int BN1, int BN2, int WM, int WN>
int launch_tn_variant(const TnArgs& a, int blocks, hipStream_t st) {
  constexpr int lds = 2 * TN_BK * (BN1 + BN2) * 4;
  auto kern = gemm_tn_kernel<BN1, BN2, WM, WN>;
  if (lds > 64 * 1024 && raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "gemm_tn: cannot raise dynamic LDS to %d bytes", lds);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(WM * WN * 64), lds, st, a);
  FC_CHECK_LAUNCH("gemm_tn");
  return FC_OK;
}

}  // namespace

size_t gemm_tn_scratch_bytes(int M, int N1, int N2) {
  if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
  const TnPlan p = plan_tn(M, N1, N2);
  return (size_t)p.splits * N1 * N2 * sizeof(float);
}

int launch_reduce_partials(const float* P, int splits, int rows, int cols, float* C, int ldc, float alpha, float beta,
                           hipStream_t st) {
  const size_t plane = (size_t)rows * cols;
  if (plane == 0) return FC_OK;
  if (cols % 4 || ldc % 4 || (((uintptr_t)P | (uintptr_t)C) & 15)) return fail(FC_EINVAL, "reduce_partials: alignment");
  const size_t blocks = std::min<size_t>(2048, (plane / 4 + 255) / 256);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)std::max<size_t>(1, blocks)), dim3(256), 0, st, P, splits,
                     plane, cols, C, ldc, alpha, beta);
  FC_CHECK_LAUNCH("reduce_partials");
  return FC_OK;
}

int launch_gemm_tn(const float* A, const float* B, int M, int N1, int N2, int lda, int ldb, int a_skip, float alpha,
                   float beta, float* C, int ldc, float* scratch, size_t scratch_bytes, const float* zeros,
                   hipStream_t st) {
  if (N1 <= 0 || N2 <= 0) return FC_OK;
  if (M <= 0) return fail(FC_EINVAL, "gemm_tn: M=%d", M);
  if (N1 % 4 || N2 % 4 || lda % 4 || ldb % 4 || ldc % 4 || lda < N1 || ldb < N2 || ldc < N2)
    return fail(FC_EINVAL, "gemm_tn: N1=%d N2=%d lda=%d ldb=%d ldc=%d must be multiples of 4 and cover the tile", N1,
                N2, lda, ldb, ldc);
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)scratch | (uintptr_t)zeros) & 15 || !zeros || !scratch)
    return fail(FC_EINVAL, "gemm_tn: unaligned / missing operand");
  const TnPlan p = plan_tn(M, N1, N2);
  if (scratch_bytes < (size_t)p.splits * N1 * N2 * sizeof(float))
    return fail(FC_ENOMEM, "gemm_tn: scratch needs %zu bytes", (size_t)p.splits * N1 * N2 * sizeof(float));
  TnArgs a{};
  a.A = A; a.B = B; a.P = scratch; a.zeros = zeros;
  a.M = M; a.N1 = N1; a.N2 = N2; a.lda = lda; a.ldb = ldb; a.rows_per_split = p.rows_per_split; a.a_skip = a_skip;
  const int blocks = p.tiles * p.splits;
  const int rc = p.big ? launch_tn_variant<256, 256, 2, 4>(a, blocks, st) : launch_tn_variant<128, 128, 2, 2>(a, blocks, st);
  if (rc != FC_OK) return rc;
  return launch_reduce_partials(scratch, p.splits, N1, N2, C, ldc, alpha, beta, st);
}

size_t colsum_scratch_bytes(int rows, int cols) {
  const int chunks = std::max(1, std::min(256, (rows + 63) / 64));
  return (size_t)chunks * cols * sizeof(float);
}

// out[c] = beta * out[c] + sum_r X[r, c]
int launch_colsum(const void* X, int kind, long ldx, int rows, int cols, float* out, float beta, float* scratch,
                  size_t scratch_bytes, hipStream_t st) {
  if (cols <= 0) return FC_OK;
  if (rows <= 0) return fail(FC_EINVAL, "colsum: rows=%d", rows);
  if (cols % 4 || ldx % 4 || (((uintptr_t)X | (uintptr_t)out | (uintptr_t)scratch) & 15))
    return fail(FC_EINVAL, "colsum: alignment");
  const int chunks = std::max(1, std::min(256, (rows + 63) / 64));
  const int rpc = (rows + chunks - 1) / chunks;
  const int used = (rows + rpc - 1) / rpc;
  if (scratch_bytes < (size_t)used * cols * sizeof(float)) return fail(FC_ENOMEM, "colsum: scratch too small");
  const int strips = (cols + 255) / 256;
  if (kind == PREC_BF16)
    hipLaunchKernelGGL(colsum_partial_kernel<bf16>, dim3(used * strips), dim3(256), 0, st, (const bf16*)X, ldx, rows, cols, rpc, scratch);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(used * strips), dim3(256), 0, st, (const float*)X, ldx, rows, cols, rpc, scratch);
  FC_CHECK_LAUNCH("colsum");
  return launch_reduce_partials(scratch, used, 1, cols, out, cols, 1.f, beta, st);
}

}  // namespace fc

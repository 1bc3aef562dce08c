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
  float* CS;           // null, or [splits, N1]: per-split column sums of A (the bias gradient next to the weight gradient)
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
  // (numbering the workgroups XCD-major, so that the sharers of an A / B slab sit behind one L2, measured nothing: the kernel
  // is not waiting for its operands: with neither LDS-DMA nor operand reads in the loop it is 4 % faster, tools/tn_bench.py)
  const int n1_0 = (tile / tiles2) * BN1, n2_0 = (tile % tiles2) * BN2;
  const int m_begin = split * g.rows_per_split;
  const int m_end = min(g.M, m_begin + g.rows_per_split);

  // ---- per-lane staging plan: piece p covers floats [256 p, 256 p + 256) of the stage image [A rows | B rows].  The steps
  // are staged strictly in order, so every piece keeps a running 32-bit byte offset from its operand's row m_begin (uniform
  // 64-bit base), the rows it has left before m_end, and - for an A operand stored with skipped rows - m mod a_skip: a step
  // costs a handful of adds and selects per piece (the first version recomputed row, division and a branch per piece).
  static_assert(BN1 == BN2 && (PIECES / 2) % NW == 0, "the first LPW / 2 pieces of a wave are A rows, the others B rows");
  constexpr int LPA = LPW / 2;
  const int skip = g.a_skip;
  const size_t a_row0 = skip > 0 ? (size_t)m_begin + m_begin / skip + 1 : (size_t)m_begin;
  const char* a_base = reinterpret_cast<const char*>(g.A + a_row0 * g.lda);
  const char* b_base = reinterpret_cast<const char*>(g.B + (size_t)m_begin * g.ldb);
  const int step_q = skip > 0 ? TN_BK / skip : 0, step_r = skip > 0 ? TN_BK % skip : 0;
  unsigned poff[LPW];
  int pleft[LPA], prem[LPA];   // (A piece i and B piece LPA + i cover the same m-row)
#pragma unroll
  for (int i = 0; i < LPW; ++i) {
    const bool isb = i >= LPA;
    const int oo = (wave + (i - (isb ? LPA : 0)) * NW) * 256 + lane * 4;
    const int prow = oo / BN1;
    const int pc = oo - prow * BN1;
    const int lc = pc ^ ((prow & 1) << 4);  // logical column held at this physical position
    const int col = min((isb ? n2_0 : n1_0) + lc, (isb ? g.N2 : g.N1) - 4);  // columns past the edge re-read the last chunk (never stored)
    if (!isb) pleft[i] = m_end - (m_begin + prow);
    if (isb) {
      poff[i] = (unsigned)prow * (unsigned)(g.ldb * 4) + (unsigned)col * 4u;
    } else {
      int rows = prow;
      if (skip > 0) {
        const int rem = m_begin % skip + prow;  // (one exact division per piece, once)
        rows += rem / skip;
        prem[i] = rem % skip;
      } else {
        prem[i] = 0;
      }
      poff[i] = (unsigned)rows * (unsigned)(g.lda * 4) + (unsigned)col * 4u;
    }
  }
  auto stage_next = [&](int stage) {   // the next TN_BK rows -> `stage`
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const bool isb = i >= LPA;
      const char* real = (isb ? b_base : a_base) + poff[i];
      const char* src = pleft[isb ? i - LPA : i] > 0 ? real : reinterpret_cast<const char*>(g.zeros + lane * 4);  // a zero row kills the product
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + stage * STAGE_F + (wave + i * NW) * 256),
                                       16, 0, 0);
      if (isb) pleft[i - LPA] -= TN_BK;   // (after both pieces of the row)
      if (isb) {
        poff[i] += (unsigned)(TN_BK * g.ldb * 4);
      } else {
        int adv = TN_BK + step_q;
        if (skip > 0) {
          prem[i] += step_r;
          const bool wrap = prem[i] >= skip;
          prem[i] -= wrap ? skip : 0;
          adv += wrap ? 1 : 0;
        }
        poff[i] += (unsigned)adv * (unsigned)(g.lda * 4);
      }
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r = lane & 15, q = lane >> 4;
  const int nsteps = (m_end - m_begin + TN_BK - 1) / TN_BK;
  // Software pipeline (the hand-over of gemm_pipelined_kernel): the LDS-DMA runs TWO steps ahead of the MFMAs, and the
  // hand-over to step st + 1 - wait for its DMA (issued a whole step ago), barrier, DMA of step st + 2 into the stage
  // everybody has just finished reading, first operand reads of step st + 1 - sits in FRONT of the last MFMA group of step
  // st, which covers it.  (First version: wait + barrier + DMA burst + operand reads at the top of every step with the
  // matrix pipe idle - 0.80-0.84 of peak.)  Every accumulator still sees its products in the same order: same bits.
  constexpr int KS = TN_BK / 4;   // MFMA groups (k-slots of 4 m-rows) per step
  static_assert(KS % 2 == 0, "the operand double buffer alternates per group");
  float af[2][FM], bf[2][FN];
  // column sums of A (dY): the workgroups of the first n2 tile hold every A element of their slab in registers once - their
  // wn = 0 waves add them up (per lane: the m-rows of its k-slot, in step order; the four k-slots at the end)
  const bool do_cs = g.CS != nullptr && tile % tiles2 == 0 && wn == 0;
  float cs[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) cs[i] = 0.f;
  // Operand addresses: FOUR byte offsets per lane (even / odd 16-column groups of A and of B) + immediates.  The bank swizzle
  // flips bit 4 of the column on odd m-rows (q odd; 4 ks is even), and bit 4 of (tile column + 16 i + r) is the parity of i:
  // the flip is + 16 columns for even i and - 16 for odd i, so column group i sits at (i even ? even : odd base) + 64 (i & ~1) bytes.
  // A stage is a power of two bytes and LDS starts at 0: the other stage is one XOR away.
  constexpr int STAGE_B = STAGE_F * 4;
  static_assert((STAGE_B & (STAGE_B - 1)) == 0 && TM % 32 == 0 && TNN % 32 == 0, "stage toggle / swizzle parity");
  const int swz = (q & 1) << 4;
  // (the odd base includes group 1's 16 columns, so that it is never negative: odd i adds 64 (i - 1) bytes)
  int rd_a[2] = {(q * BN1 + wm * TM + r + swz) * 4, (q * BN1 + wm * TM + r + 16 - swz) * 4};
  int rd_b[2] = {(TN_BK * BN1 + q * BN2 + wn * TNN + r + swz) * 4, (TN_BK * BN1 + q * BN2 + wn * TNN + r + 16 - swz) * 4};
  auto toggle_stage = [&]() {
#pragma unroll
    for (int e = 0; e < 2; ++e) rd_a[e] ^= STAGE_B, rd_b[e] ^= STAGE_B;
  };
  auto read_operands = [&](int ks, float (&a)[FM], float (&b)[FN]) {   // k-slot ks of the stage the offsets point into
#pragma unroll
    for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const float*>(smem + rd_a[i & 1] + ks * 4 * BN1 * 4 + (i & ~1) * 64);
#pragma unroll
    for (int j = 0; j < FN; ++j) b[j] = *reinterpret_cast<const float*>(smem + rd_b[j & 1] + ks * 4 * BN2 * 4 + (j & ~1) * 64);
  };
  if (nsteps > 0) {
    stage_next(0);
    if (nsteps > 1) {
      stage_next(1);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPW) : "memory");   // step 0 has landed; step 1 may be in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_operands(0, af[0], bf[0]);
  }
  for (int st = 0; st < nsteps; ++st) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      constexpr int kReads = FM + FN;
      const int cur = ks & 1;                  // (KS is even: the parity carries over the step boundary)
      if (ks + 1 < KS) {
        read_operands(ks + 1, af[cur ^ 1], bf[cur ^ 1]);
      } else {
        if (st + 1 < nsteps) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every read of this stage has returned ...
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // ... and this wave's pieces of step st + 1 have landed
          __builtin_amdgcn_s_barrier();                       // ... for every wave
          asm volatile("" ::: "memory");
          if (st + 2 < nsteps) stage_next(st & 1);
        }
        // UNCONDITIONALLY (after the last step they are never used): inside the branch the reads would sit in a basic block of
        // their own in front of this group's first MFMA, and hipcc's lgkmcnt(0) for its operands would wait for them
        toggle_stage();
        read_operands(0, af[cur ^ 1], bf[cur ^ 1]);
      }
      // X as the MFMA A-operand, dY as the B-operand: a lane then holds 4 CONSECUTIVE n2 of one n1 (16-byte stores)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[cur][j], af[cur][i], acc[i][j], 0, 0, 0);
      if (do_cs) {
#pragma unroll
        for (int i = 0; i < FM; ++i) cs[i] += af[cur][i];
      }
      // issue order: ONE MFMA, then the operand reads of the next group, then the other MFMAs (hipcc's lgkmcnt(0) for this
      // group's operands sits in front of the first MFMA: with the reads in front of it that wait would cover them too)
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, FM * FN - 1, 0);
    }
  }
  if (do_cs) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      float v = cs[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n1 = n1_0 + wm * TM + i * 16 + r;
      if (q == 0 && n1 < g.N1) g.CS[(size_t)split * g.N1 + n1] = v;
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
    // (eight planes are requested before the first of them is added: a column sum comes here as up to 256 planes of a few KB,
    // and one dependent load per plane was a 256-deep latency chain; the additions keep their order)
    f32x4 s = *reinterpret_cast<const f32x4*>(P + i * 4);
    int k = 1;
    for (; k + 8 <= splits; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(P + (size_t)(k + u) * plane + i * 4);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; ++k) s += *reinterpret_cast<const f32x4*>(P + (size_t)k * plane + i * 4);
    const size_t e = i * 4, row = e / cols, col = e - row * cols;
    f32x4* dst = reinterpret_cast<f32x4*>(C + row * ldc + col);
    f32x4 v = s * alpha;
    if (beta != 0.f) v += *dst * beta;
    *dst = v;
  }
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
  return (size_t)p.splits * N1 * (N2 + 4) * sizeof(float);   // (+ the column-sum partials, 16-byte aligned behind the tiles)
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

// colsum != nullptr: colsum[n1] = beta * colsum[n1] + sum_m A[m, n1] from the same pass over A (dY: the bias gradient)
int launch_gemm_tn(const float* A, const float* B, int M, int N1, int N2, int lda, int ldb, int a_skip, float alpha,
                   float beta, float* C, int ldc, float* scratch, size_t scratch_bytes, const float* zeros,
                   hipStream_t st, float* colsum) {
  if (N1 <= 0 || N2 <= 0) return FC_OK;
  if (M <= 0) return fail(FC_EINVAL, "gemm_tn: M=%d", M);
  if (N1 % 4 || N2 % 4 || lda % 4 || ldb % 4 || ldc % 4 || lda < N1 || ldb < N2 || ldc < N2)
    return fail(FC_EINVAL, "gemm_tn: N1=%d N2=%d lda=%d ldb=%d ldc=%d must be multiples of 4 and cover the tile", N1,
                N2, lda, ldb, ldc);
  if (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)scratch | (uintptr_t)zeros) & 15 || !zeros || !scratch)
    return fail(FC_EINVAL, "gemm_tn: unaligned / missing operand");
  const TnPlan p = plan_tn(M, N1, N2);
  const size_t need = (size_t)p.splits * N1 * (N2 + (colsum ? 4 : 0)) * sizeof(float);
  if (scratch_bytes < need) return fail(FC_ENOMEM, "gemm_tn: scratch needs %zu bytes", need);
  // the kernel walks A and B through running 32-bit byte offsets from a split's first row (a_skip rows included): a split whose
  // rows span 4 GiB would wrap silently (few splits over very many rows: N1 = N2 = 4096 is one split per tile)
  {
    const size_t span_rows = (size_t)p.rows_per_split + 2 * TN_BK;
    const size_t a_rows = a_skip > 0 ? span_rows + span_rows / (size_t)a_skip + 2 : span_rows;
    if (a_rows * (size_t)lda * sizeof(float) >= (1ull << 32) || span_rows * (size_t)ldb * sizeof(float) >= (1ull << 32))
      return fail(FC_EINVAL, "gemm_tn: %d rows per split x lda=%d / ldb=%d exceed the 4 GiB of the kernel's 32-bit row offsets "
                             "(M=%d in %d split(s)): split the rows over several calls with beta = 1", p.rows_per_split, lda, ldb, M, p.splits);
  }
  if (colsum && ((uintptr_t)colsum & 15)) return fail(FC_EINVAL, "gemm_tn: unaligned column-sum output");
  TnArgs a{};
  a.A = A; a.B = B; a.P = scratch; a.zeros = zeros;
  a.CS = colsum ? scratch + (size_t)p.splits * N1 * N2 : nullptr;
  a.M = M; a.N1 = N1; a.N2 = N2; a.lda = lda; a.ldb = ldb; a.rows_per_split = p.rows_per_split; a.a_skip = a_skip;
  const int blocks = p.tiles * p.splits;
  const int rc = p.big ? launch_tn_variant<256, 256, 2, 4>(a, blocks, st) : launch_tn_variant<128, 128, 2, 2>(a, blocks, st);
  if (rc != FC_OK) return rc;
  if (colsum) {
    const int rc2 = launch_reduce_partials(a.CS, p.splits, 1, N1, colsum, N1, 1.f, beta, st);
    if (rc2 != FC_OK) return rc2;
  }
  return launch_reduce_partials(scratch, p.splits, N1, N2, C, ldc, alpha, beta, st);
}

}  // namespace fc

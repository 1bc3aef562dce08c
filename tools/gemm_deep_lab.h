// LAB ONLY (tools/gemm_lab.hip), NEGATIVE RESULT, not part of the library: bf16 GEMM with a four-deep K pipeline.
// Correct (bit-identical rows to the shipped kernels) but 15-20 % SLOWER than gemm_pipelined_kernel
// (profiles/r01_lab23_deep.log: 0.96-0.99 vs 1.15-1.23 PF/s).  Its own ablations say why: with the loads removed it
// reaches 1.25 PF/s (pipelined kernel: 1.41), i.e. a hand-over per 32 MFMAs instead of per 64 costs 12 %; and the loads
// still cost 25 % although they are spread one piece per MFMA group and requested three steps ahead -- the price of a
// 1 KiB LDS-DMA piece (~65 cycles of matrix-pipe idle per SIMD) is paid per piece, not per burst or per miss.
//
// gemm_pipelined_kernel stages 64-wide K-tiles in two 64 KiB LDS buffers: the K-tile needed two steps ahead can only be
// requested after the barrier that frees its buffer, i.e. ONE K-step before it is consumed, and its 8 LDS-DMA pieces
// per wave either go out as a burst (both waves of a SIMD stuck in the issue queue together) or spread over the step
// at the price of their latency slack.  Ablations put the cost of the loads at 20-27 % of the kernel.
//
// Here the same 128 KiB hold FOUR stages of 32-wide K-tiles (64-byte rows).  A K-step is 32 MFMAs per wave in four
// groups of eight; after each group the wave issues ONE piece (16 rows x 64 B) of the K-tile THREE steps ahead, so the
// DMA issue is spread evenly and every piece has 2.25-3 steps to land.  One raw barrier per (half-length) K-step.
//
//   * stage s (32 KiB): piece p = rows 16 p .. 16 p + 15 of the tile (A: p < 16, W: p >= 16), 1 KiB, lane-linear as the
//     DMA writes it: lane l -> row l >> 2, 16-byte slot l & 3.  The slot holds K-chunk (l & 3) ^ g(row), g(row) =
//     (-(row >> 2)) & 3, applied to the SOURCE address: with it the four 16-lane groups of a ds_read_b128 fragment read
//     (lane (r, q): row r, chunk q) touch sixteen different 16-byte slots of the 256-byte bank row.
//   * a piece is exactly one 16-row MFMA operand tile.
//   * K order: step j reads the 32-block 2 * ((j / 2 + rot) % (K / 64)) + (j & 1) with rot = (n0 / 256) % (K / 64):
//     the same accumulation order as gemm_kernel / gemm_pipelined_kernel, so all three give bit-identical rows.
//   * VM operations of a wave, in issue order (vmcnt retires in order):
//       ... [K-tile j+1: 4] [K-tile j+2: 4] [K-tile j+3: pieces 0-2] | hand-over of step j: K-tile j+1 must have landed
//     -> "at most 7 outstanding".  After an interior tile its 16 epilogue stores sit between the next tile's K-tiles 2
//     and 3: steps 0 and 1 of the next tile allow 7 + 16; at the end of the last tile the pieces that are no longer
//     issued are subtracted.
#pragma once
#include "gemm_kernel.h"

namespace fc {
namespace {

__device__ __forceinline__ void wait_vmcnt_rt(int n) {  // n wave-uniform; rounds DOWN to an available immediate
  if (n >= 23) wait_vmcnt<23>();
  else if (n >= 7) wait_vmcnt<7>();
  else if (n >= 4) wait_vmcnt<4>();
  else wait_vmcnt<0>();
}

template <int EPI, int ABL = 0>
__global__ void __launch_bounds__(512) gemm_deep_kernel(const GemmArgs g) {
  using T = bf16;
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
  constexpr int ROWK = 64;                    // bytes of K per staged row (32 bf16)
  constexpr int NSTG = 4;
  constexpr int STAGE = (BM + BN) * ROWK;     // 32 KiB
  constexpr int TM = BM / WM, TN = BN / WN, FM = TM / 16, FN = TN / 16;  // 128 x 64 per wave: 8 x 4 MFMA tiles
  constexpr int NG = FM / 2;                  // MFMA groups per K-step (2 row tiles x FN column tiles each)
  constexpr int ROWP = TN * 2, CPR = ROWP / 16, PATCH = 16 * ROWP, RPI = 64 / CPR, IPP = 16 / RPI;
  constexpr int OFF_STG = NSTG * STAGE;       // 8 output patches of 2 KiB
  constexpr int OFF_BIAS = OFF_STG + NW * PATCH;
  constexpr int NST = FM * IPP;               // 16 store instructions per wave per interior tile
  static_assert(EPI == EPI_BIAS_T || EPI == EPI_GELU_T, "epilogue");
  static_assert(NG == 4 && FN == 4 && NST == 16, "geometry assumed by the vmcnt bookkeeping");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- tile schedule: XCD x (= blockIdx & 7) owns a contiguous range of M-panels and walks its tiles N-fastest
  const int tilesN = (g.N + BN - 1) / BN, tilesM = (g.M + BM - 1) / BM;
  const int G = gridDim.x, xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
  const int nblk = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  const int pq = tilesM / 8, pr = tilesM % 8;
  const int mp0 = xcd < pr ? xcd * (pq + 1) : pr * (pq + 1) + (xcd - pr) * pq;
  const int npanel = pq + (xcd < pr ? 1 : 0);
  const int t_end = npanel * tilesN;
  int t = pos;
  if (t >= t_end) return;

  const int nk = g.K / 32;   // K-steps per tile; K % 64 == 0 (checked by the launcher)
  const int nk64 = g.K / 64;
  // ---- per-lane staging sources (32-bit byte offsets from the two base pointers; operands < 4 GiB)
  const int prow = lane >> 2;                                    // row inside a 16-row piece
  const unsigned pchunk = (unsigned)(((lane & 3) ^ ((-(prow >> 2)) & 3)) << 4);
  unsigned offA[2], offB[2];
  int rot = 0;
  auto tile_sources = [&](int tile, int& m0, int& n0) {
    const int tm = mp0 + tile / tilesN, tn = tile % tilesN;
    m0 = tm * BM;
    n0 = tn * BN;
    rot = (n0 >> 8) % nk64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = wave + i * NW;  // piece 0..15 of the A part / of the W part
      offA[i] = (unsigned)min((ABL == 2 ? 0 : m0) + p * 16 + prow, g.M - 1) * (unsigned)(g.lda * 2) + pchunk;
      offB[i] = (unsigned)min((ABL == 2 ? 0 : n0) + p * 16 + prow, g.N - 1) * (unsigned)(g.ldw * 2) + pchunk;
    }
  };
  // piece idx (0, 1: activation rows; 2, 3: weight rows) of K-step j of the tile whose sources are loaded
  auto stage_piece = [&](int stage, int j, auto IDX) {
    constexpr int idx = decltype(IDX)::value;
    int k64 = (j >> 1) + rot;
    if (k64 >= nk64) k64 -= nk64;
    const unsigned kb = (unsigned)(k64 * 128 + (j & 1) * 64);   // byte offset of the 32-block inside a row
    char* dst = smem + stage * STAGE + wave * 1024;
    if constexpr (idx < 2)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g.A) + (offA[idx] + kb)),
          (__attribute__((address_space(3))) void*)(dst + idx * NW * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g.W) + (offB[idx - 2] + kb)),
          (__attribute__((address_space(3))) void*)(dst + BM * ROWK + (idx - 2) * NW * 1024), 16, 0, 0);
  };
  auto bias_load = [&](int buf, int n0) {  // BN floats -> LDS by one LDS-DMA of wave 0
    if (wave == 0) {
      const float* p = g.bias + min(n0 + lane * 4, g.N - 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + OFF_BIAS + buf * 1024), 16, 0,
                                       0);
    }
  };

  // ---- fragment addresses: lane (r, q) reads row r, K-chunk q of a 16-row piece
  const int r = lane & 15, q = lane >> 4;
  const int frag = r * ROWK + ((q ^ ((-(r >> 2)) & 3)) << 4);
  const int a_base = wm * (TM / 16) * 1024 + frag;              // + i * 1024 per row tile
  const int b_base = BM * ROWK + wn * (TN / 16) * 1024 + frag;  // + j * 1024 per column tile

  int m0, n0;
  tile_sources(t, m0, n0);
  bias_load(0, n0);
#pragma unroll
  for (int j = 0; j < 3; ++j)
    static_for<4>([&](auto I) { stage_piece(j, j, I); });
  wait_vmcnt<8>();   // K-tile 0 (and the bias slice) have landed; K-tiles 1, 2 may be in flight
  block_barrier();
  bf16x8 wb[FN], wnext[FN], xp[2][2];
#pragma unroll
  for (int j = 0; j < FN; ++j) wb[j] = *reinterpret_cast<const bf16x8*>(smem + b_base + j * 1024);
#pragma unroll
  for (int a = 0; a < 2; ++a) xp[0][a] = *reinterpret_cast<const bf16x8*>(smem + a_base + a * 1024);

  int gstep = 0;              // global K-step counter of this workgroup (stage = gstep & 3)
  int it = 0;                 // tile iteration (bias buffer = it & 1)
  bool prev_counted = false;  // the previous tile issued exactly NST stores
  for (;;) {
    f32x4 acc[FM][FN];
    {
      const float* biasb = reinterpret_cast<const float*>(smem + OFF_BIAS + (it & 1) * 1024) + wn * TN + 4 * q;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(biasb + j * 16);
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = b;
      }
    }
    const int cm0 = m0, cn0 = n0;
    const int tnext = t + nblk;
    const bool has_next = tnext < t_end;

    for (int kt = 0; kt < nk; ++kt, ++gstep) {
      const char* st = smem + (gstep & 3) * STAGE;
      // what this step requests: K-tile kt + 3 of this tile, or K-tile kt + 3 - nk of the next one
      const bool fetch = kt + 3 < nk || has_next;
      const int fj = kt + 3 < nk ? kt + 3 : kt + 3 - nk;
      const int fstage = (gstep + 3) & 3;
      static_for<NG>([&](auto U) {
        constexpr int u = decltype(U)::value;
        if constexpr (u + 1 < NG) {
#pragma unroll
          for (int a = 0; a < 2; ++a)
            xp[(u + 1) & 1][a] = *reinterpret_cast<const bf16x8*>(st + a_base + (2 * (u + 1) + a) * 1024);
        } else {
          if (kt + 1 < nk || has_next) {
            // hand-over, in front of the last MFMA group: every LDS read of this stage has been issued; once they
            // have returned and K-tile kt+1 has landed (this wave's pieces: counted wait; everyone's: barrier) the next
            // step may start, and the stage read one step ago becomes the target of the next piece.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int allowed = 7;  // [K-tile +2: 4][K-tile +3: pieces 0..2]
            if (!has_next) allowed = kt + 3 < nk ? 7 : (kt + 2 < nk ? 4 : 0);
            if (prev_counted && kt < 2) allowed += NST;  // the previous tile's stores are younger than K-tile kt+1
            wait_vmcnt_rt(allowed);
            block_barrier();
            // first fragments of the next step (the weight fragments into wnext: this group still multiplies with wb)
            const char* nx = smem + ((gstep + 1) & 3) * STAGE;
#pragma unroll
            for (int j = 0; j < FN; ++j) wnext[j] = *reinterpret_cast<const bf16x8*>(nx + b_base + j * 1024);
#pragma unroll
            for (int a = 0; a < 2; ++a) xp[0][a] = *reinterpret_cast<const bf16x8*>(nx + a_base + a * 1024);
          }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[2 * u + a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xp[u & 1][a], acc[2 * u + a][j], 0, 0, 0);
        if constexpr (u + 1 == NG) {
#pragma unroll
          for (int j = 0; j < FN; ++j) wb[j] = wnext[j];
        }
        if (ABL != 1) {
          if (fetch) {
            if (u == 0 && kt + 3 == nk) {  // first request for the next tile: switch the sources, fetch its bias slice
              tile_sources(tnext, m0, n0);
              bias_load((it + 1) & 1, n0);
            }
            stage_piece(fstage, fj, U);
          }
        }
        constexpr int kReads = (u + 1 < NG) ? 2 : FN + 2;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * FN - 1, 0);
      });
    }

    const bool interior = cm0 + BM <= g.M && cn0 + BN <= g.N;
    if constexpr (ABL == 3) {
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (keep == 123.456f) reinterpret_cast<float*>(g.C)[0] = keep;
      prev_counted = false;
    } else {
      char* stg = smem + OFF_STG + wave * PATCH;
      char* wr = stg + r * ROWP + ((q & 1) << 3);
      const int rrow = lane / CPR, rch = lane % CPR;
      T* cbase = reinterpret_cast<T*>(g.C) + (size_t)(cm0 + wm * TM + rrow) * g.ldc + cn0 + wn * TN + rch * 8;
      const bool col_ok = cn0 + wn * TN + rch * 8 < g.N;
#pragma unroll
      for (int i = 0; i < FM; ++i) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          f32x4 v = acc[i][j];
          if constexpr (EPI == EPI_GELU_T) v = quick_gelu_fast4(v);
          bf16x4 o;
          o[0] = static_cast<bf16>(v[0]); o[1] = static_cast<bf16>(v[1]);
          o[2] = static_cast<bf16>(v[2]); o[3] = static_cast<bf16>(v[3]);
          *reinterpret_cast<bf16x4*>(wr + (((j * 2 + (q >> 1)) ^ (r & (CPR - 1))) << 4)) = o;
        }
#pragma unroll
        for (int h = 0; h < IPP; ++h) {
          const int row = h * RPI + rrow;
          const bf16x8 val = *reinterpret_cast<const bf16x8*>(stg + row * ROWP + ((rch ^ (row & (CPR - 1))) << 4));
          T* p = cbase + (size_t)(i * 16 + h * RPI) * g.ldc;
          if (interior || (cm0 + wm * TM + i * 16 + row < g.M && col_ok))
            __builtin_nontemporal_store(val, reinterpret_cast<bf16x8*>(p));
        }
      }
      prev_counted = interior;
    }
    if (!has_next) break;
    ++it;
    t = tnext;
  }
}

}  // namespace
}  // namespace fc

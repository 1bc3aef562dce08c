// LAB ONLY (tools/gemm_lab; not part of the library): bf16 MFMA GEMM with a FOUR-stage LDS ring of 32-column K-steps.
// RESULT (profiles/r03_gemm_lab_ring.log, M = 100864): correct, but 5 - 8 % SLOWER than gemm_pipelined_kernel on all four block
// shapes (QKV 0.42-0.45 vs 0.47 of the bf16 peak, c_fc 0.44 vs 0.48), also with every load hitting L2 (0.44-0.47): a K-step of
// 32 bf16 columns reads HALF lines, so the same bytes cost twice the L1 requests per FLOP - what the deeper prefetch wins on
// miss latency, the request rate loses.  (The three-plane split kernel, whose K-step reads 96 of a line's 128 bytes and
// feeds six MFMA products per loaded value, is where this structure pays: gemm_split3.h.)
//
// bf16 MFMA GEMM with a FOUR-stage LDS ring (gfx950).
//
//   C[M,N] (bf16) = epilogue(A[M,K] . W[N,K]^T + bias)      A, W bf16, both K-contiguous
//
// Same tile, wave layout, MFMA shape (v_mfma_f32_16x16x32_bf16), bias-initialised accumulators and LDS-transposed 128-byte
// store epilogue as gemm_pipelined_kernel (gemm_kernel.h); what differs is how the operands reach LDS.  That kernel stages
// K-steps of 64 columns in TWO stages of 64 KiB, so a K-step's LDS-DMA pieces can be issued at most one step (~1 us) before
// they are needed; rocprofv3 on the split-fp32 kernel of the same structure (tools/split3_lab) showed
// what that costs: the L1's pending-miss queue fills (TCP_PENDING_STALL_CYCLES ~ a quarter of the launch), the TA stalls,
// and a wave stuck on a DMA instruction issues no MFMAs.  Here a K-step is 32 columns (one MFMA k-extent, 64-byte tile
// rows, 32 KiB per stage) and FOUR stages form a ring: the pieces of step k + 3 are issued during step k, one or two at a
// time behind MFMA groups, and have almost three K-steps to land.  The hand-over to the next step (wait, barrier, first
// fragment reads) sits in front of the last MFMA group and costs nothing measurable once those reads are in the same
// basic block as the MFMAs (gemm_kernel.h).  The four stages leave no LDS for output patches: the epilogue borrows the stage
// the tile's last K-step has just released (the DMA that would refill it - step 3 of the next tile - follows the barrier
// that ends the epilogue).
//
// LDS image of a stage: [512 tile rows][64 bytes = 4 chunks of 8 bf16]; chunk q of row r is stored at chunk q ^ (r & 8 ? 3 : 0)
// (with 64-byte rows the sixteen lanes of a ds_read_b128 group cover row residues mod 4 four times each: the XOR sends the
// four to different 16-byte bank slots), applied on the SOURCE side of the LDS-DMA and on the fragment reads.
#pragma once
#include "gemm_kernel.h"

namespace fc {
namespace {

// ABL (tools/gemm_lab only): 0 = real kernel; 1 = no global loads inside the K loop; 2 = every workgroup stages tile (0, 0);
// 3 = no epilogue
template <int EPI, int ABL = 0>
__global__ void __launch_bounds__(512) gemm_ring_kernel(const GemmArgs g) {
  using T = bf16;
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4, NW = 8;
  constexpr int TM = BM / WM, TN = BN / WN;        // 128 x 64 per wave
  constexpr int FM = TM / 16, FN = TN / 16;        // 8 x 4 MFMA tiles of 16 x 16
  constexpr int BKE = 32;                          // K columns per step = one v_mfma_f32_16x16x32_bf16
  constexpr int ROWR = BKE * 2;                    // 64 bytes per tile row and stage
  constexpr int STAGE = (BM + BN) * ROWR;          // 32 KiB
  constexpr int NSTAGE = 4;
  constexpr int LPA = BM / 16 / NW, LPB = BN / 16 / NW, LPW = LPA + LPB;  // 1 KiB pieces (16 rows) per wave and stage: 2 + 2
  constexpr int OFF_BIAS = NSTAGE * STAGE;         // 2 x 1 KiB behind the stages
  constexpr int PATCHB = STAGE / NW;               // 4 KiB of the released stage per wave during the epilogue
  constexpr int ROWP = TN * 2;                     // 128 bytes per row of a wave's output patch
  constexpr int CPR = ROWP / 16;                   // 8 chunks per patch row
  constexpr int RPI = 64 / CPR;                    // 8 output rows per store instruction
  constexpr int IPP = 16 / RPI;                    // 2 store instructions per 16-row pass
  constexpr int NST = FM * IPP;                    // 16 store instructions per wave and interior tile
  constexpr int NG = FM / 2;                       // 4 MFMA groups per K-step: two row tiles x FN column tiles each
  static_assert(EPI == EPI_BIAS_T || EPI == EPI_GELU_T, "epilogue");
  static_assert(16 * ROWP <= PATCHB && NST + 2 * LPW < 64, "patch / counted vmcnt");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- tile schedule (gemm_pipelined_kernel's): XCD x (= blockIdx & 7) owns a contiguous range of M-panels (optionally
  // only 1 / nsplit of the N range); its workgroups stride through that range in N-fastest order
  const int tilesN = (g.N + BN - 1) / BN;
  const int tilesM = (g.M + BM - 1) / BM;
  const int G = gridDim.x, xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
  const int nblk = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  const int ngrp = (g.nsplit > 1 && 8 % g.nsplit == 0 && tilesN % g.nsplit == 0 && G == 8 * (G >> 3)) ? g.nsplit : 1;
  const int grp = xcd % ngrp, xi = xcd / ngrp, nx = 8 / ngrp;
  const int pq = tilesM / nx, pr = tilesM % nx;
  const int mp0 = xi < pr ? xi * (pq + 1) : pr * (pq + 1) + (xi - pr) * pq;
  const int npanel = pq + (xi < pr ? 1 : 0);
  const int tnn = tilesN / ngrp, tn0 = grp * tnn;
  const int t_end = npanel * tnn;
  int t = pos;
  if (t >= t_end) return;

  const int nk = g.K / BKE;  // K-steps (even, >= 8: checked on the host)
  const unsigned lda_b = (unsigned)g.lda * 2u, ldw_b = (unsigned)g.ldw * 2u;
  // K-steps of a tile are visited in a rotated order that only depends on the column tile (two steps = one 128-byte line
  // apart for neighbouring column tiles): a row's result does not depend on the rows around it
  int rot = 0;
  unsigned offA[LPA], offB[LPB];
  const int prow = lane >> 2, pch = lane & 3;  // a piece = 16 rows x 4 chunks, lane-linear in LDS
  auto tile_sources = [&](int tile, int& m0, int& n0) {
    const int tm = mp0 + tile / tnn, tn = tn0 + tile % tnn;
    m0 = tm * BM;
    n0 = tn * BN;
    rot = (tn * (g.nblock > 0 ? g.nblock - 1 : 2)) % nk;  // g.nblock - 1: lab override of the stride
#pragma unroll
    for (int i = 0; i < LPA; ++i) {
      const int row = (wave + i * NW) * 16 + prow;
      const int gr = min((ABL == 2 ? 0 : m0) + row, g.M - 1);
      offA[i] = (unsigned)gr * lda_b + (unsigned)((pch ^ ((row & 8) ? 3 : 0)) << 4);
    }
#pragma unroll
    for (int i = 0; i < LPB; ++i) {
      const int row = (wave + i * NW) * 16 + prow;
      const int gr = min((ABL == 2 ? 0 : n0) + row, g.N - 1);
      offB[i] = (unsigned)gr * ldw_b + (unsigned)((pch ^ ((row & 8) ? 3 : 0)) << 4);
    }
  };
  auto stage_piece = [&](int stage_off, int kt, auto IDX) {  // piece IDX (0 .. LPA-1 activations, then weights) of a K-step
    constexpr int idx = decltype(IDX)::value;
    kt += rot;
    if (kt >= nk) kt -= nk;
    char* dst = smem + stage_off + wave * 1024;
    if constexpr (idx < LPA)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g.A) + (offA[idx] + (unsigned)kt * ROWR)),
          (__attribute__((address_space(3))) void*)(dst + idx * NW * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g.W) + (offB[idx - LPA] + (unsigned)kt * ROWR)),
          (__attribute__((address_space(3))) void*)(dst + BM * ROWR + (idx - LPA) * NW * 1024), 16, 0, 0);
  };
  auto stage_load = [&](int stage_off, int kt) {
    static_for<LPW>([&](auto I) { stage_piece(stage_off, kt, I); });
  };
  auto bias_load = [&](int buf, int n0) {  // BN floats -> LDS by one LDS-DMA of wave 0
    if (wave == 0) {
      const float* p = g.bias + min(n0 + lane * 4, g.N - 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + OFF_BIAS + buf * 1024), 16, 0, 0);
    }
  };

  // fragments: lane (r = lane & 15, q = lane >> 4) reads the 16 bytes k = 8q .. 8q+7 of tile row r
  const int r = lane & 15, q = lane >> 4;
  const int qs = (q ^ ((r & 8) ? 3 : 0)) << 4;
  const int a_base = (wm * TM + r) * ROWR + qs;
  const int b_base = (BM + wn * TN + r) * ROWR + qs;
  auto read_w = [&](const char* st, bf16x8 (&w)[FN]) {
#pragma unroll
    for (int j = 0; j < FN; ++j) w[j] = *reinterpret_cast<const bf16x8*>(st + b_base + j * 16 * ROWR);
  };
  auto read_a = [&](const char* st, int pair, bf16x8 (&a)[2]) {
#pragma unroll
    for (int e = 0; e < 2; ++e) a[e] = *reinterpret_cast<const bf16x8*>(st + a_base + (2 * pair + e) * 16 * ROWR);
  };

  int m0, n0;
  tile_sources(t, m0, n0);
  bias_load(0, n0);
  // the ring: byte offsets of the stage being multiplied and of the three after it (wave-uniform scalars)
  int s0 = 0, s1 = STAGE, s2 = 2 * STAGE, s3 = 3 * STAGE;
  stage_load(s0, 0);
  stage_load(s1, 1);
  stage_load(s2, 2);
  stage_load(s3, 3);
  bf16x8 wf[2][FN], af[2][2];
  wait_vmcnt<3 * LPW>();  // the bias slice and K-step 0 of the first tile have landed
  block_barrier();
  read_w(smem + s0, wf[0]);
  read_a(smem + s0, 0, af[0]);
  int it = 0;
  bool prev_counted = false;  // the previous tile issued exactly NST stores between its prefetches and K-step 3 of this one

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

    // one K-step; PAR = kt & 1 = weight-fragment register set of this step (nk is even)
    auto kstep = [&](int kt, auto PAR) {
      constexpr int par = decltype(PAR)::value;
      const char* st = smem + s0;
      static_for<NG>([&](auto U) {
        constexpr int u = decltype(U)::value;
        if constexpr (u + 1 < NG) {
          read_a(st, u + 1, af[(u + 1) & 1]);
        } else {
          // hand-over in front of the LAST group: every LDS read of this stage has been issued; once they have returned the
          // stage may be refilled (with K-step kt + 4)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          // K-step kt + 1 must have landed.  Younger in this wave's queue: the pieces of steps kt + 2 and kt + 3 as far as those
          // steps exist, and - in the first two steps of a tile that follows a fully stored one - the NST epilogue stores
          const bool more2 = kt + 2 < nk || has_next, more3 = kt + 3 < nk || has_next;
          if (kt < 2 && prev_counted) wait_vmcnt<NST + 2 * LPW>();
          else if (more3) wait_vmcnt<2 * LPW>();
          else if (more2) wait_vmcnt<LPW>();
          else wait_vmcnt<0>();
          block_barrier();
          if (ABL != 1 && has_next && kt + 4 == nk) {  // the next tile's staging offsets, before its first pieces (step nk - 3)
            tile_sources(tnext, m0, n0);
            bias_load((it + 1) & 1, n0);
          }
          // the first fragments of the next K-step, unconditionally (same basic block as this group's MFMAs)
          const char* nx = smem + s1;
          read_w(nx, wf[par ^ 1]);
          read_a(nx, 0, af[0]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[2 * u + e][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[par][j], af[u & 1][e], acc[2 * u + e][j], 0, 0, 0);
        // issue order inside a group: ONE MFMA, then the LDS reads of the next group, then the other MFMAs
        constexpr int kReads = (u + 1 < NG) ? 2 : 2 + FN;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * FN - 1, 0);
        if constexpr (ABL != 1 && u < 2) {
          // two pieces of K-step kt + 3 (of this tile, or step 0 .. 2 of the next one) into the stage released at the hand-over
          // of step kt - 1; step 3 of a tile always arrives whole (prologue / behind the epilogue)
          if (kt >= 1 && (kt + 3 < nk || has_next)) {
            const int lk = kt + 3 < nk ? kt + 3 : kt + 3 - nk;
            stage_piece(s3, lk, std::integral_constant<int, 2 * u>{});
            stage_piece(s3, lk, std::integral_constant<int, 2 * u + 1>{});
          }
        }
      });
      const int released = s0;
      s0 = s1;
      s1 = s2;
      s2 = s3;
      s3 = released;
    };
    for (int kt = 0; kt < nk; kt += 2) {
      kstep(kt, std::integral_constant<int, 0>{});
      kstep(kt + 1, std::integral_constant<int, 1>{});
    }
    // af[0] / wf[0] now hold the first fragments of the next tile; s3 is the stage the last K-step released

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
      // bf16 outputs: 16 rows x 64 columns of the wave tile go through the wave's patch (16 x 128 bytes, chunk ^ (row & 7)) and
      // leave as 8 rows x 128 contiguous bytes per store instruction, non-temporal (gemm_pipelined_kernel's epilogue)
      char* stg = smem + s3 + wave * PATCHB;
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
        for (int hh = 0; hh < IPP; ++hh) {
          const int row = hh * RPI + rrow;
          const bf16x8 val = *reinterpret_cast<const bf16x8*>(stg + row * ROWP + ((rch ^ (row & (CPR - 1))) << 4));
          T* p = cbase + (size_t)(i * 16 + hh * RPI) * g.ldc;
          if (interior || (cm0 + wm * TM + i * 16 + row < g.M && col_ok))
            __builtin_nontemporal_store(val, reinterpret_cast<bf16x8*>(p));
        }
      }
      prev_counted = interior;
    }
    if (!has_next) break;
    // the patch area becomes a stage again: once every wave is through with its patch, K-step 3 of the next tile goes there
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    block_barrier();
    if (ABL != 1) stage_load(s3, 3);
    ++it;
    t = tnext;
  }
}

}  // namespace
}  // namespace fc

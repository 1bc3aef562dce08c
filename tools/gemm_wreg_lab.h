// LAB ONLY (tools/gemm_lab; negative result): gemm_pipelined_kernel<bf16> with the WEIGHT operand taken out
// of LDS - packed in fragment order in HBM / L2 (`pack_w_frag_kernel` below) and loaded straight into registers, one K-tile
// ahead.  A 128 x 64 wave tile reads 24 KiB of fragments per K-tile from LDS next to 8 KiB of LDS-DMA writes per wave: at full
// MFMA rate that is ALL the LDS cycles there are.  Without the weight: 16 KiB of reads and 4 KiB of writes - and with no loads
// at all in the K loop this kernel reaches 0.52-0.70 of the bf16 peak where the LDS version reaches 0.53.
// Results are correct (profiles/r03_gemm_lab_wreg.log) - but a K-tile of prefetch distance needs two register sets of 32 VGPRs
// next to 128 accumulators, and the kernel has 256: hipcc spills 100 registers and it runs at 0.15 of peak.  With ONE set per
// sub-step (half a K-tile of distance, no spills) the loads do not land in time: 0.41-0.45 against 0.46-0.50.
#pragma once
#include "gemm_kernel.h"

namespace fc {

namespace {

// W [N, K] bf16 row-major -> fragment order: ((((tn * nk + kt) * 4 + wn) * 2 + s) * 4 + j) * 64 + lane  x 16 bytes, lane (r, q)
// = row tn * 256 + wn * 64 + j * 16 + r, k = kt * 64 + s * 32 + 8 q .. + 7
// (tile of BNc columns = WNc wave columns of 64: FN = 4 fragments each)
__global__ void pack_w_frag_kernel(const bf16* __restrict__ W, bf16x8* __restrict__ out, int N, int K, int ldw, int BNc = 256) {
  const int nk = K / 64, WNc = BNc / 64;
  const size_t total = (size_t)((N + BNc - 1) / BNc) * nk * WNc * 2 * 4 * 64;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), j = (int)((i >> 6) & 3), s = (int)((i >> 8) & 1);
    size_t rest = i >> 9;
    const int wn = (int)(rest % WNc);
    rest /= WNc;
    const int kt = (int)(rest % nk), tn = (int)(rest / nk);
    const int row = tn * BNc + wn * 64 + j * 16 + (lane & 15), k = kt * 64 + s * 32 + (lane >> 4) * 8;
    bf16x8 v = {};
    if (row < N) v = *reinterpret_cast<const bf16x8*>(W + (size_t)row * ldw + k);
    out[i] = v;
  }
}

template <int BM, int BN, int WM, int WN, int EPI, int ABL = 0, int ROT = 1, int SCHED = 0>
__global__ void __launch_bounds__(WM * WN * 64) gemm_wreg_kernel(const GemmArgs g) {
  using T = bf16;
  constexpr int NW = WM * WN;
  constexpr int BKE = ROWB / (int)sizeof(T);
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int FM = TM / 16, FN = TN / 16;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int RG = (BM + BN) / 8;
  constexpr int LPW = RG / NW;
  constexpr bool kOutF32 = sizeof(T) == 4 || EPI == EPI_RESID_F32;  // (the residual stream is fp32 in the bf16 mode too)
  constexpr bool kStaged = !kOutF32;                  // bf16 outputs: LDS-transposed, 16-byte full-line stores
  using TOUT = std::conditional_t<kOutF32, float, T>;
  constexpr int ROWP = TN * 2;                        // bytes per row of a wave's output patch (bf16)
  constexpr int CPR = ROWP / 16;                      // 16-byte chunks per patch row
  constexpr int PATCH = 16 * ROWP;                    // one 16-row pass of the wave tile
  constexpr int RPI = 64 / CPR;                       // output rows per store instruction (8 x 128 B or 4 x 256 B)
  constexpr int IPP = 16 / RPI;                       // store instructions per pass
  constexpr int OFF_STG = 2 * STAGE;                  // NW patches
  constexpr int OFF_BIAS = OFF_STG + NW * 2048;       // 2 x 1 KiB   (f32 outputs: a 16-row x 128-byte patch per wave as well)
  constexpr int NST = kStaged ? FM * IPP : FM * FN;   // store instructions per wave per interior tile
  static_assert(RG % NW == 0 && BN <= 256 && (!kStaged || TN == 64 || TN == 128), "tile");
  static_assert(kStaged ? NW * PATCH <= NW * 2048 : (TN % 32 == 0 && FN % 2 == 0), "output patch");
  static_assert(EPI == EPI_BIAS_T || EPI == EPI_GELU_T, "epilogue");
  // the counted wait behind the epilogue stores needs LPW + NST to fit the 6-bit vmcnt; tilings with more stores per wave
  // (4 waves of 128x128) wait for everything at the first hand-over of the next tile instead
  constexpr bool kCounted = LPW + NST < 64;
  using FragT = typename Frag<T>::type;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- tile schedule.  XCD x (= blockIdx & 7) owns a contiguous range of M-panels; its workgroups stride through the
  // tiles of that range in N-BLOCK-major order: for each block of `nblock` N-tiles, every M-panel of the range
  // (nblock = 0: the whole N range, i.e. plain N-fastest order).  Tuning knobs, measured in profiles/r01_lab9/11:
  // smaller blocks / an N-split over XCD groups change the L2 re-fetch volume by up to -20 % but not the in-situ time
  // (the 4.7 MB c_fc weight and the ~6 live activation panels never fit a 4 MiB L2 together), so both default to off.
  const int tilesN = (g.N + BN - 1) / BN;
  const int tilesM = (g.M + BM - 1) / BM;
  const int G = gridDim.x, xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
  const int nblk = (G >> 3) + (xcd < (G & 7) ? 1 : 0);
  // N-split: the 8 XCDs form `ngrp` groups; a group only ever touches its own 1/ngrp of the N range, so only that part
  // of W competes for its L2s (c_fc: 2 x 2.4 MB instead of 4.7 MB per 4 MiB L2); the activation panels are then read
  // by ngrp XCDs instead of one.
  const int ngrp = (g.nsplit > 1 && 8 % g.nsplit == 0 && tilesN % g.nsplit == 0 && G == 8 * (G >> 3)) ? g.nsplit : 1;
  const int grp = xcd % ngrp, xi = xcd / ngrp, nx = 8 / ngrp;
  const int pq = tilesM / nx, pr = tilesM % nx;
  const int mp0 = xi < pr ? xi * (pq + 1) : pr * (pq + 1) + (xi - pr) * pq;  // first M-panel of this XCD
  const int npanel = pq + (xi < pr ? 1 : 0);
  const int tnn = tilesN / ngrp, tn0 = grp * tnn;                          // N-tile range of this XCD's group
  const int nbw = (g.nblock > 0 && g.nblock < tnn) ? g.nblock : tnn;
  const int t_begin = 0, t_end = npanel * tnn;  // local tile index inside the XCD's range
  auto tile_coords = [&](int k, int& tm, int& tn) {
    const int per_block = npanel * nbw;
    const int nb = k / per_block, rem = k - nb * per_block;
    const int wb = min(nbw, tnn - nb * nbw);
    tm = mp0 + rem / wb;
    tn = tn0 + nb * nbw + rem % wb;
  };
  int t = t_begin + pos;
  if (t >= t_end) return;

  // per-lane staging sources as 32-bit byte offsets from two (scalar) base pointers: the weight is < 4 GiB, the activations
  // are addressed from the first row of the current tile
  constexpr int LPA = BM / 8 / NW, LPB = BN / 8 / NW;  // LDS-DMA instructions per wave per stage for A / W
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must divide over the waves");
  const int rin = lane >> 3, pc = lane & 7;
  // (row >> 1) & 7 of a staged row only depends on (wave, rin): row = (wave + i * NW) * 8 + rin and NW * 4 = 0 mod 8
  static_assert((NW * 4) % 8 == 0, "swizzle term must not depend on i");
  const unsigned swz = (unsigned)((pc ^ ((wave * 4 + (rin >> 1)) & 7)) << 4);
  unsigned offA[LPA], offB[LPB];
  const char* a_tile = reinterpret_cast<const char*>(g.A);  // 64-bit base of the current tile's first activation row (scalar)
  const int nk = g.K / BKE;
  // K-tiles of a tile are visited in the rotated order rot, rot+1, ..., nk-1, 0, ..., rot-1 with rot = (first column / 256) mod nk:
  // the workgroups that share an activation panel then read different K-slices (different L2 channels) at any moment
  // instead of hammering the same lines in lockstep (+5..12 % on the K = 768 shapes).  rot only depends on the
  // N-tile, so a row's result still does not depend on the batch around it; gemm_kernel uses the same order.
  int rot = 0;
  auto tile_sources = [&](int tile, int& m0, int& n0) {
    int tm, tn;
    tile_coords(tile, tm, tn);
    m0 = tm * BM;
    n0 = tn * BN;
    if constexpr (ROT == 1) rot = (n0 >> 8) % nk;  // a function of the output column block only
    // activation rows: a 64-bit tile base + 32-bit offsets inside the tile (the 4w-wide MLP rows of a 2048-frame fp32 pass are
    // 5 GB; the weight stays below 4 GiB)
    const int mb = ABL == 2 ? 0 : m0;
    a_tile = reinterpret_cast<const char*>(g.A) + (size_t)mb * ((size_t)g.lda * sizeof(T));
#pragma unroll
    for (int i = 0; i < LPA; ++i) {
      const int row = min((wave + i * NW) * 8 + rin, g.M - 1 - mb);
      offA[i] = (unsigned)row * (unsigned)(g.lda * (int)sizeof(T)) + swz;
    }
#pragma unroll
    for (int i = 0; i < LPB; ++i) {
      const int row = (wave + i * NW) * 8 + rin;
      const int gr = min((ABL == 2 ? 0 : n0) + row, g.N - 1);
      offB[i] = (unsigned)gr * (unsigned)(g.ldw * (int)sizeof(T)) + swz;
    }
  };
  auto stage_load = [&](int stage, int kt) {
    kt += rot;
    if (kt >= nk) kt -= nk;
    char* dst = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < LPA; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(a_tile + (offA[i] + (unsigned)kt * ROWB)),
          (__attribute__((address_space(3))) void*)(dst + i * NW * 1024), 16, 0, 0);
  };
  // One LDS-DMA piece of a stage (idx < LPA: activation rows, else weight rows): lets the K loop spread the pieces of a
  // K-tile over its MFMA groups instead of issuing them in one burst.
  auto stage_piece = [&](int stage, int kt, auto IDX) {
    constexpr int idx = decltype(IDX)::value;
    kt += rot;
    if (kt >= nk) kt -= nk;
    char* dst = smem + stage * STAGE + wave * 1024;
    if constexpr (idx < LPA)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(a_tile + (offA[idx] + (unsigned)kt * ROWB)),
          (__attribute__((address_space(3))) void*)(dst + idx * NW * 1024), 16, 0, 0);
  };
  // SCHED > 0: the LDS-DMA pieces of the K-tile needed two steps ahead are not issued as one burst of LPW pieces inside
  // the hand-over; piece idx goes to slot piece_slot(SCHED, idx): -1 = still in the hand-over, u >= 0 = after MFMA
  // group u of the NEXT K-step (the stage it lands in was released by the hand-over barrier that precedes that step).
  // Each piece blocks its wave's issue port for ~100 cycles, and right after the barrier the two waves of a SIMD would
  // both be in that burst, with nobody feeding the matrix pipe.
  static_assert(SCHED == 0 || LPW == 8 || LPW == 6, "piece schedules are written for 8 pieces per wave");
  auto handover_pieces = [&](int stage, int kt) {
    static_for<LPW>([&](auto I) {
      if constexpr (piece_slot(SCHED, decltype(I)::value) < 0) stage_piece(stage, kt, I);
    });
  };
  auto bias_load = [&](int buf, int n0) {  // BN floats -> LDS by one LDS-DMA of wave 0 (part of that tile's first load)
    if (wave == 0) {
      const float* p = g.bias + min(n0 + lane * 4, g.N - 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + OFF_BIAS + buf * 1024), 16, 0,
                                       0);
    }
  };

  const int r = lane & 15, q = lane >> 4, f = (r >> 1) & 7;
  const int a_base = (wm * TM + r) * ROWB;
  const int b_base = BM * ROWB + (wn * TN + r) * ROWB;

  // ---- the weight never goes through LDS: g.aux holds it PACKED in fragment order - per (column tile of 256, K-tile of 64,
  // wave column wn): 2 sub-steps x FN fragments of 64 lanes x 16 bytes - and a wave loads the fragments of the NEXT K-tile
  // a whole K-tile ahead: sub-step 0 of the next K-step behind group 0, sub-step 1 behind group GPS (4 coalesced 1 KiB loads
  // each; K-steps unrolled in pairs so that the register set is a compile-time index)
  constexpr int LPD = LPA;      // LDS-DMA pieces per wave and stage that are actually issued (activations only)
  FragT wreg[2][2][FN];  // [K-step parity][sub-step][column tile]
  const char* wp = reinterpret_cast<const char*>(g.aux);
  auto w_load = [&](auto PAR, auto SUB, int kt, int n0w, int rotw) {
    constexpr int par = decltype(PAR)::value, sub = decltype(SUB)::value;
    kt += rotw;
    if (kt >= nk) kt -= nk;
    const char* src = wp + ((size_t)((n0w / BN) * nk + kt) * WN + wn) * (2 * FN * 1024) + lane * 16 + sub * FN * 1024;
#pragma unroll
    for (int j = 0; j < FN; ++j) wreg[par][sub][j] = *reinterpret_cast<const FragT*>(src + j * 1024);
  };
  int m0, n0;
  tile_sources(t, m0, n0);
  bias_load(0, n0);
  stage_load(0, 0);
  w_load(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, 0, n0, rot);
  w_load(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, 0, n0, rot);
  stage_load(1, 1);
  constexpr int NG = FM;        // MFMA groups per K-tile: 2 k-substeps x FM/2 row-tile pairs, 2*FN MFMAs each
  constexpr int GPS = FM / 2;   // groups per k-substep
  static_assert(FM % 2 == 0, "row tiles are consumed in pairs");
  int foff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) foff[s] = ((((sizeof(T) == 2) ? (4 * s + q) : (q + 4 * s)) ^ f) * 16);
  FragT xp[2][2];
  wait_vmcnt<LPD>();   // K-tile 0 of the first tile and its weight fragments have landed (K-tile 1 may still be in flight)
  block_barrier();
#pragma unroll
  for (int a = 0; a < 2; ++a) xp[0][a] = *reinterpret_cast<const FragT*>(smem + a_base + a * 16 * ROWB + foff[0]);
  int gbase = 0;            // global K-step counter at the start of the current tile (stage = step & 1)
  int it = 0;               // tile iteration (bias buffer = it & 1)
  bool prev_counted = false;  // the previous tile issued exactly NST stores after its prefetches

  for (;;) {
    // the accumulators start from the bias slice of this tile (in LDS since the hand-over that published K-tile 0)
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

    const int crot = rot;  // (tile_sources moves n0 / rot on to the next tile two K-steps before this one ends)
    auto kstep = [&](auto PAR, const int kt) {
      constexpr int par = decltype(PAR)::value;
      const int sidx = (gbase + kt) & 1;
      const char* st = smem + sidx * STAGE;
      const bool last = kt == nk - 1;
      static_for<NG>([&](auto U) {
        constexpr int u = decltype(U)::value;
        constexpr int s = u / GPS, p = u % GPS;
        if constexpr (u + 1 < NG) {
          // fragments of the next group (same K-tile) are requested before this group's MFMAs are issued
          constexpr int s1 = (u + 1) / GPS, p1 = (u + 1) % GPS;
#pragma unroll
          for (int a = 0; a < 2; ++a)
            xp[(u + 1) & 1][a] = *reinterpret_cast<const FragT*>(st + a_base + (2 * p1 + a) * 16 * ROWB + foff[s1]);
        } else {
         if (!last || has_next) {
          // hand-over to the next K-step, placed BEFORE the last MFMA group so that the barrier, the next LDS-DMA issue
          // and the first fragment reads of the next K-tile are covered by MFMAs.  Every LDS read of this stage has
          // been issued; once they have returned the stage may be overwritten by the other waves' DMA.
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if constexpr (kCounted) {
            if (kt == 0 && prev_counted) wait_vmcnt<NST>(); else wait_vmcnt<0>();
          } else {
            wait_vmcnt<0>();
          }
          block_barrier();
          if (ABL != 1) {
            if (kt + 2 < nk) {
              handover_pieces(sidx, kt + 2);
            } else if (has_next) {
              if (kt + 2 == nk) {
                tile_sources(tnext, m0, n0);
                bias_load((it + 1) & 1, n0);
                handover_pieces(sidx, 0);
              } else {
                stage_load(sidx, 1);  // always a burst: it has to be older than the epilogue stores (counted vmcnt)
              }
            }
          }
         }
          // the first fragments of the next K-step, UNCONDITIONALLY (after the last step of the last tile they are never
          // used): inside the branch above they would sit in a basic block of their own, in FRONT of this group's first MFMA
          // (sched_group_barrier cannot order across blocks), and hipcc's lgkmcnt(0) for that MFMA's operands would wait
          // for their whole LDS latency with the matrix pipe idle - once per K-step
          const char* nx = smem + (sidx ^ 1) * STAGE;
#pragma unroll
          for (int a = 0; a < 2; ++a) xp[0][a] = *reinterpret_cast<const FragT*>(nx + a_base + a * 16 * ROWB + foff[0]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int j = 0; j < FN; ++j) mma<T>(wreg[par][s][j], xp[u & 1][a], acc[2 * p + a][j]);
        if constexpr (u == 0 || u == GPS) {
          // the weight fragments of the NEXT K-step (of this tile, or K-tile 0 of the next tile) into the other register set
          if (ABL != 1 && (!last || has_next))
            w_load(std::integral_constant<int, par ^ 1>{}, std::integral_constant<int, u / GPS>{}, last ? 0 : kt + 1,
                   last ? n0 : cn0, last ? rot : crot);
        }
        if constexpr (SCHED > 0 && ABL != 1 && u + 1 < NG) {
          constexpr bool any = piece_slot(SCHED, 0) == u || piece_slot(SCHED, 1) == u || piece_slot(SCHED, 2) == u ||
                               piece_slot(SCHED, 3) == u || piece_slot(SCHED, 4) == u || piece_slot(SCHED, 5) == u ||
                               piece_slot(SCHED, 6) == u || piece_slot(SCHED, 7) == u;
          if constexpr (any) {
            // K-tile kt+1 (or K-tile 0 of the next output tile) into the stage the previous K-step has released
            if (kt > 0 && (kt + 1 < nk || has_next)) {
              const int lk = kt + 1 < nk ? kt + 1 : 0;
              static_for<LPW>([&](auto I) {
                if constexpr (piece_slot(SCHED, decltype(I)::value) == u) stage_piece(sidx ^ 1, lk, I);
              });
            }
          }
        }
        // pin the issue order hipcc would otherwise undo (it sinks the reads next to their first use): first the LDS
        // reads of the NEXT group, then this group's MFMAs
        // Issue order inside a group: ONE MFMA first, then the LDS reads of the next group, then the other MFMAs.
        // hipcc's s_waitcnt for this group's operands is an lgkmcnt(0) placed before the first MFMA; with the reads
        // in front of it that wait would also cover the reads just issued (a full LDS latency per group).
        constexpr int kMfmaPerGroup = 2 * FN * (sizeof(T) == 2 ? 1 : 4);
        constexpr int kReads = 2;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, kMfmaPerGroup - 1, 0);
      });
    };
    for (int kt = 0; kt < nk; kt += 2) {  // nk is even (checked on the host): the register set is kt & 1
      kstep(std::integral_constant<int, 0>{}, kt);
      kstep(std::integral_constant<int, 1>{}, kt + 1);
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
      if constexpr (kStaged) {
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
            if (interior || (cm0 + wm * TM + i * 16 + row < g.M && col_ok)) {
              // non-temporal: the tile is not read again by this kernel; a plain store write-allocates in L2 and evicts
              // the operand panels the other workgroups of the XCD are sharing (measured: operand re-fetch -35 %,
              // kernel +6..18 % on the N >= 2304 shapes)
              __builtin_nontemporal_store(val, reinterpret_cast<bf16x8*>(p));
            }
          }
        }
      } else {
        // f32 outputs: a lane holds 4 floats of ONE row per 16x16 fragment, so a direct store instruction would write
        // 16 rows x 64 bytes (half lines; WRITE_SIZE counted 1.43x the bytes).  Two column-adjacent fragments go through
        // a private 16-row x 128-byte LDS patch (chunk ^ (row & 7) swizzle: conflict-free both ways) and leave as
        // 8 rows x 128 contiguous bytes per store instruction - the same number of store instructions (FM * FN).
        char* stg = smem + OFF_STG + wave * 2048;
        const int rrow = lane >> 3, rch = lane & 7;
        // EPI_RESID_F32 (C += acc + bias: the residual stream updated in place, so the LayerNorm behind the projection reads
        // ONE fp32 row instead of row + delta and writes no row back): every lane adds the 16 bytes of C it is about to
        // overwrite.  They are requested RWIN row-tiles ahead (RWIN * FN loads in flight per lane: the fragment registers
        // of the K loop are free here), whole lines per instruction like the stores, and non-temporal like them: the 1 GB
        // stream must not push the operand panels out of L2.
        constexpr int RWIN = 4;
        f32x4 xres[EPI == EPI_RESID_F32 ? RWIN : 1][FN / 2][2];
        auto resid_load = [&](int i, f32x4 (&dst)[FN / 2][2]) {
#pragma unroll
          for (int jj = 0; jj < FN / 2; ++jj)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int mo = cm0 + wm * TM + i * 16 + h * 8 + rrow, no = cn0 + wn * TN + jj * 32 + rch * 4;
              dst[jj][h] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (interior || (mo < g.M && no < g.N))
                dst[jj][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const TOUT*>(g.C) + (size_t)mo * g.ldc + no));
            }
        };
        if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
          for (int i = 0; i < RWIN && i < FM; ++i) resid_load(i, xres[i]);
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int m = cm0 + wm * TM + i * 16 + r;
#pragma unroll
          for (int jj = 0; jj < FN / 2; ++jj) {
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
              const int j = 2 * jj + jh;
              const int n = cn0 + wn * TN + j * 16 + 4 * q;
              f32x4 v = acc[i][j];
              if constexpr (EPI == EPI_GELU_T) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quick_gelu_exact(v[e]);
              }
              if constexpr (EPI == EPI_DGELU_T) {
                // (these loads make hipcc drain vmcnt before the stores; the counted wait of the next tile stays valid,
                // it only asks for "at most NST operations still in flight")
                if (interior || (m < g.M && n < g.N)) {
                  const f32x4 pre = load4<T>(reinterpret_cast<const T*>(g.aux) + (size_t)m * g.ldc + n);
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] *= quick_gelu_grad(pre[e]);
                }
              }
              *reinterpret_cast<f32x4*>(stg + r * 128 + (((jh * 4 + q) ^ (r & 7)) << 4)) = v;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int row = h * 8 + rrow;
              f32x4 val = *reinterpret_cast<const f32x4*>(stg + row * 128 + ((rch ^ (row & 7)) << 4));
              if constexpr (EPI == EPI_RESID_F32) val = xres[i % RWIN][jj][h] + val;
              const int mo = cm0 + wm * TM + i * 16 + row, no = cn0 + wn * TN + jj * 32 + rch * 4;
              if (interior || (mo < g.M && no < g.N)) {
                f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<TOUT*>(g.C) + (size_t)mo * g.ldc + no);
                if constexpr (ABL == 4) *dst = val;  // lab: write-back instead of non-temporal stores
                else __builtin_nontemporal_store(val, dst);
              }
            }
          }
          if constexpr (EPI == EPI_RESID_F32) {
            if (i + RWIN < FM) resid_load(i + RWIN, xres[i % RWIN]);
          }
        }
      }
      prev_counted = interior;
    }
    if (!has_next) break;
    gbase += nk;
    ++it;
    t = tnext;
  }
}

}  // namespace
}  // namespace fc

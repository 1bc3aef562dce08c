// Split-fp32 GEMM over TWO-plane fp16 operands (gfx950): fp32 results from the fp16 matrix cores with THREE products per fp32
// product (gemm_split3.h needs six bf16 products and 6 bytes per value; here 3 and 4).
//
//   C[M,N] = epilogue((A[M,K] . W[N,K]^T) / s)      A, W: "x2" rows (common.h): every 32 fp32 columns one 128-byte line
//                                                   [h1 x32 | h2 x32] of fp16; s = the weight tensor's power-of-two scale
//
//   x = h1 + 2^-11 h2,  s w = g1 + g2      =>      x (s w) = h1 g1 + h1 g2 + h2 (2^-11 g1)     (+ 2^-22: the dropped h2 g2)
//
// Kernel: 256 x 256 output tile per workgroup of 8 waves (2 x 4; wave tile 128 x 64 = 8 x 4 MFMA tiles of 16 x 16), persistent
// (one workgroup per CU), v_mfma_f32_16x16x32_f16.  Round 5 shipped this kernel on the 32x32x16 shape (tools/gemm_split2_m32.h,
// kept for the lab): it ran at the chip's power limit, and at that limit the chip holds a higher clock on the 16x16x32 shape at
// equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7) - the bare MFMA + LDS-read loop of this kernel reaches
// 0.70 - 0.71 of the fp16 peak where the 32x32x16 loop reached 0.61 - 0.62, the whole kernel 7 - 9 % less time on the four block
// shapes (tools/x2k_lab, docs/rounds/round6.md).  One MFMA covers the whole K-depth of a 128-byte line (32 columns):
//   * a K-step = one line of every operand row = ONE k-slice; its 96 MFMAs per wave are issued in 8 groups (one 16-row tile each)
//     of 12: g1 h1 x 4 column tiles, g2 h1 x 4, (2^-11 g1) h2 x 4 - consecutive MFMAs hit different accumulators;
//   * fragments: lane (r = lane & 15, q = lane >> 4) reads the 16 bytes k = 8 q .. 8 q + 7 of plane p of tile row r: logical chunk
//     4 p + q (conflict-free under the image's chunk swizzle: the 16 lanes of a ds_read_b128 group touch 16 distinct bank slots);
//   * the weight fragments of a K-step (g1, g2 and the scaled copy of g1 for four column tiles: 48 registers) stay in registers
//     for all 8 groups; they are replaced IN the last group, plane by plane, each as soon as its four MFMAs have been issued (the
//     hand-over barrier in front of that group has published the next K-step); activations: two planes of the current and of the
//     next row tile (16 registers) next to 128 accumulators;
//   * LDS: two stages per operand, [activation rows, stage 0 | 1 | weight rows, stage 0 | 1], 128 bytes per tile row with the
//     chunk swizzle of gemm_pipelined_kernel (physical chunk pc of row r holds logical chunk pc ^ (r >> 1 & 7), applied on the
//     source side of the LDS-DMA), plus a 2 KiB output patch per wave and the bias slices: 146 KiB;
//   * LDS-DMA through buffer descriptors rebuilt per tile from scalars (rows beyond M / N fail the range check instead of being
//     clamped per lane), the pieces of K-step kt + 1 issued behind the first MFMA groups of K-step kt, the hand-over (counted
//     vmcnt, raw s_barrier) in front of the LAST group, the next tile's first two K-steps requested before the epilogue stores;
//   * epilogues: a lane of a 16 x 16 tile holds four consecutive columns of one row, two column tiles are one 128-byte line of
//     16 rows = the wave's patch: all lanes write, all lanes read back 8 rows x 128 contiguous bytes per store instruction:
//     fp32 rows, fp32 rows added to C in place (the residual stream), or exact QuickGELU + x2 rows.
// An output element sees one chain of 16x16x32 products in an order that only depends on its column tile: rows are bit-invariant
// to the batch, the tile height and the launch geometry.
#pragma once
#include "gemm_kernel.h"

namespace fc {
namespace {

// issue slot of LDS-DMA piece idx (0..lpa-1 activation rows, then weight rows) of a wave: -1 = in the hand-over (a whole K-step
// before its data is needed), u >= 0 = behind MFMA group u of the NEXT K-step (the stage was released by the hand-over barrier in
// front of that step; the piece must land before the last group of the same step waits for it, so only the first half is used)
constexpr int x2_piece_slot(int spread, int idx, int lpa = 4) {
  switch (spread) {
    case 1: return idx < lpa ? -1 : 0;
    case 2: return idx / 2 - 1;
    case 3: return idx / 2;
    case 4: return idx < lpa ? -1 : (idx - lpa) / 2;
    case 5: return idx / 3;               // 3, 3, 2 behind groups 0, 1, 2
    case 6: return idx / 4;               // 4 behind each of groups 0, 1
    default: return -1;
  }
}

// ABL (tools/x2k_lab only): 0 = real kernel; 1 = no global loads inside the K loop; 3 = no epilogue; 6 = no loads, no waits, no
// epilogue (MFMA + LDS reads only); 7 = the real kernel with conflict-free (and WRONG) patch writes of the x2 epilogue; 8 = two products per line (g1 h1 + g2 h2: the issue
// pattern of a plain 16-bit GEMM over 64-column lines - sizes what this kernel's structure would give the bf16 mode); 4 = the real
// kernel with s_memtime stamps around the hand-over wait, the hand-over barrier, the K loop and the epilogue, summed per wave into
// g.aux (8 x uint64 per wave; a diagnostic build: the stamps cost time themselves).
// RW: residual / positional-embedding rows requested RW patches (16 rows x 32 columns) ahead (EPI_RESID3_F32, EPI_PATCH_F32).
// PF: activation fragments requested PF row tiles ahead (2: three register sets, 256-row tiles only)
// PRIO (lab): 1 = waves 4..7 (the later-dispatched partner on every SIMD) at s_setprio 2 for the whole kernel; 2 = the two halves
// take the higher priority in alternate K-steps; 3 = waves 0..3 at s_setprio 2
// GW (lab): 0 = QuickGELU as two pair chains (quick_gelu_f32x4), 1 = every step on all four values (quick_gelu_f32x4_wide),
// 2 = 1 + the planes through the mixed-precision fma (split2_mix; shipped: c_fc at the bench size 4.442 -> 4.407 ms)
template <int EPI, int ABL = 0, int SPREAD = 0, int RW = 4, int RR = 0, int BMT = 256, int PF = 1, int PRIO = 0, int GW = 2>
__global__ void __launch_bounds__(512) gemm_split2_kernel(const GemmArgs g) {
  constexpr int BM = BMT, BN = 256, WM = 2, WN = 4, NW = 8;
  constexpr int TM = BM / WM, TN = BN / WN;        // 128 (64) x 64 per wave
  constexpr int FM = TM / 16, FN = TN / 16;        // 8 (4) x 4 MFMA tiles of 16 x 16
  constexpr int NG = FM;                           // MFMA groups per K-step: one per row tile
  // LDS: [activation rows, stage 0 | stage 1 | weight rows, stage 0 | stage 1 | patches | bias]: the two stages of an operand lie
  // BM (BN) x 128 bytes apart, so ONE address register per operand and plane reaches every fragment of both stages through the
  // 16-bit immediate of ds_read_b128 (stages 64 KiB apart took one register per row tile and stage: 24 address registers)
  constexpr int STG_A = BM * ROWB, STG_W = BN * ROWB;   // 32768 (16384), 32768
  constexpr int OFF_W = 2 * STG_A;
  constexpr int STAGE = STG_A + STG_W;             // bytes per stage: 65536 (49152)
  constexpr int LPA = BM / 8 / NW, LPB = BN / 8 / NW, LPW = LPA + LPB;  // LDS-DMA pieces per wave and stage: 4 (2) + 4
  static_assert(BM == 256 || BM == 192 || BM == 128, "tile height");
  static_assert(FN == 4 && NG % 2 == 0, "wave tile");
  static_assert(PF == 1 || (PF == 2 && NG == 8), "prefetch distance");
  constexpr int NAF = PF + 1;                      // activation fragment sets: row tile i lives in set i % NAF
  constexpr int OFF_STG = 2 * STAGE;               // 8 patches of 2 KiB
  constexpr int OFF_BIAS = OFF_STG + NW * 2048;    // 2 x 1 KiB
  constexpr bool kOutX2 = EPI == EPI_GELU_X2;
  constexpr bool kResid = EPI == EPI_RESID3_F32;   // C += acc + bias (fp32, in place)
  constexpr bool kPatch = EPI == EPI_PATCH_F32;    // C[m + m / P + 1] = acc + pos[m % P + 1] (patch embedding into the token stream)
  constexpr bool kAdd = kResid || kPatch;          // the epilogue adds rows it fetches RW patches ahead
  constexpr int NP = FM * (FN / 2);                // patches (16 rows x 32 columns) per wave and tile: 16 (8)
  constexpr int NST = NP * 2;                      // store instructions per wave and interior tile: 32 (16)
  static_assert(EPI == EPI_BIAS_F32 || EPI == EPI_GELU_X2 || EPI == EPI_RESID3_F32 || EPI == EPI_PATCH_F32, "epilogue");
  static_assert(LPW + NST < 64, "the counted wait behind the epilogue stores must fit the 6-bit vmcnt");

  constexpr int ABLK = ABL == 7 || ABL == 8 ? 0 : ABL;   // (ABL 7 / 8 are the real kernel but for the patch-write addresses / the products)
  constexpr bool kTwo = ABL == 8;            // lab: TWO products per line, g1 h1 + g2 h2 - what a plain 16-bit GEMM over 64-column lines issues
  constexpr int NPROD = kTwo ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- tile schedule (gemm_split3_kernel's): XCD x (= blockIdx & 7) owns a contiguous range of M-panels (optionally only 1 / nsplit
  // of the N range: the weights of its column tiles then stay in its L2); its workgroups stride through that range in N-fastest order
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
  const int qd_ = G >> 3, rd_ = G & 7;
  const int wk = (xcd < rd_ ? xcd * (qd_ + 1) : rd_ * (qd_ + 1) + (xcd - rd_) * qd_) + pos;   // XCD-major workgroup number
  const int t_end = RR ? tilesM * tilesN : npanel * tnn;
  int t = RR ? wk : pos;
  if (t >= t_end) return;

  const int nk = g.K / X2_GROUP;                   // K-steps: one 128-byte line of every operand row each (even, >= 4)
  const unsigned lda_b = (unsigned)g.lda * 2u, ldw_b = (unsigned)g.ldw * 2u;  // row strides in bytes (lda / ldw count fp16)
  const float w_s = g.wscale[0], w_inv = g.wscale[1];
  // LDS-DMA sources through two buffer descriptors (activation rows / weight rows of the CURRENT tile, rebuilt per tile from
  // scalars): a lane's address is [descriptor base] + [its row and swizzled chunk: ONE 32-bit register per operand, the same for
  // every tile] + [piece: 64 rows, a scalar] + [K-step: a scalar], and rows beyond M / N fail the descriptor's range check (their
  // lanes fetch nothing) instead of being clamped lane by lane.  With 64-bit global addresses hipcc kept a register PAIR per
  // activation piece and a register per weight piece alive across the K loop: 12 registers, here 2.
  int rot = 0;
  unsigned offA0, offB0;
  {
    const int rin = lane >> 3, pc = lane & 7;
    const unsigned swz = (unsigned)((pc ^ ((wave * 4 + (rin >> 1)) & 7)) << 4);
    offA0 = (unsigned)(wave * 8 + rin) * lda_b + swz;
    offB0 = (unsigned)(wave * 8 + rin) * ldw_b + swz;
  }
  __amdgpu_buffer_rsrc_t rsA, rsB;
  auto tile_sources = [&](int tile, int& m0, int& n0) {
    const int tm = RR ? tile / tilesN : mp0 + tile / tnn, tn = RR ? tile % tilesN : tn0 + tile % tnn;
    m0 = tm * BM;
    n0 = tn * BN;
    rot = (tn * (g.nblock > 0 ? g.nblock - 1 : 1)) % nk;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.A)) + (size_t)m0 * lda_b, 0,
                                            (int)((unsigned)min(BM, g.M - m0) * lda_b), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(g.W)) + (size_t)n0 * ldw_b, 0,
                                            (int)((unsigned)min(BN, g.N - n0) * ldw_b), 0x00020000);
  };
  // One LDS-DMA piece: 64 lanes x 16 bytes -> 1 KiB of LDS at lds_addr (M0) + 16 lane.  Spelled as inline asm: through the builtin
  // (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc orders every piece behind the LDS READS in flight (an `s_waitcnt lgkmcnt`
  // in front of each one - the fragment reads of the next MFMA group had just been issued), which no piece needs: a stage is
  // refilled only after the hand-over barrier that follows its last read.  Nothing else in this kernel uses M0, and the kernel
  // counts its vector-memory queue by hand (wait_vmcnt) - hipcc's own counted waits (epilogue loads / stores) only err on the
  // safe side when they do not know of the pieces.
  const unsigned lds0 = (unsigned)(uintptr_t)smem;
  auto dma_piece = [&](__amdgpu_buffer_rsrc_t rs, unsigned lds_addr, unsigned voff, int soff) {
    // ("m0" in the clobber list draws hipcc's reserved-register warning - that the register is ours to set is the point; the
    // build passes -Wno-inline-asm)
    // (one wait state between the SALU write of M0 and the instruction that reads it; the scalars the load reads must not come from
    // a VALU write less than 5 wait states upstream - hipcc pads nothing inside or in front of an asm string: build.py audits the ISA)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
  };
  auto stage_piece = [&](int stage, int kt, auto IDX) {  // piece IDX (0..LPA-1 activations, then weights) of K-step kt -> stage
    constexpr int idx = decltype(IDX)::value;
    kt += rot;
    if (kt >= nk) kt -= nk;
    const unsigned dst = lds0 + wave * 1024;
    if constexpr (idx < LPA)
      dma_piece(rsA, dst + stage * STG_A + idx * NW * 1024, offA0 + (unsigned)(idx * NW * 8) * lda_b, kt * X2_GROUP_BYTES);
    else
      dma_piece(rsB, dst + OFF_W + stage * STG_W + (idx - LPA) * NW * 1024, offB0 + (unsigned)((idx - LPA) * NW * 8) * ldw_b, kt * X2_GROUP_BYTES);
  };
  auto stage_load = [&](int stage, int kt) {
    static_for<LPW>([&](auto I) { stage_piece(stage, kt, I); });
  };
  __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.bias), 0, g.N * 4, 0x00020000);
  auto bias_load = [&](int buf, int n0) {  // BN floats -> LDS by one LDS-DMA of wave 0 (older than that tile's first K-step)
    if (wave == 0) dma_piece(rsBias, lds0 + OFF_BIAS + buf * 1024, (unsigned)(n0 + lane * 4) * 4u, 0);   // (columns beyond N: nothing fetched)
  };

  // fragment addresses: lane (r = lane & 15, q = lane >> 4) reads the 16 bytes k = 8 q .. 8 q + 7 of plane p of tile row r:
  // logical chunk 4 p + q of a row whose swizzle key is (row >> 1) & 7 = (r >> 1) & 7 (row tiles start at multiples of 16)
  int foff[2];
  int a_base, b_base;
  {
    const int r = lane & 15, q = lane >> 4, f = (r >> 1) & 7;
    foff[0] = (q ^ f) << 4;
    foff[1] = ((4 + q) ^ f) << 4;
    a_base = (wm * TM + r) * ROWB;
    b_base = OFF_W + (wn * TN + r) * ROWB;
  }
  auto read_w = [&](int stage, int p, f16x8 (&w)[FN]) {  // plane p of the four column tiles
#pragma unroll
    for (int j = 0; j < FN; ++j) w[j] = *reinterpret_cast<const f16x8*>(smem + b_base + foff[p] + (stage * STG_W + j * 16 * ROWB));
  };
  auto read_a = [&](int stage, int i, f16x8 (&a)[2]) {   // both planes of row tile i
#pragma unroll
    for (int p = 0; p < 2; ++p) a[p] = *reinterpret_cast<const f16x8*>(smem + a_base + foff[p] + (stage * STG_A + i * 16 * ROWB));
  };

  if constexpr (PRIO == 1) { if (wave >= 4) __builtin_amdgcn_s_setprio(2); }
  if constexpr (PRIO == 3) { if (wave < 4) __builtin_amdgcn_s_setprio(2); }
  int m0, n0;
  tile_sources(t, m0, n0);
  bias_load(0, n0);
  stage_load(0, 0);
  stage_load(1, 1);
  f16x8 g1[FN], g2[FN], gs[FN];   // weight planes of the current K-step; gs = 2^-11 g1
  f16x8 af[NAF][2];               // [row tile % NAF][plane]
  wait_vmcnt<LPW>();              // the bias slice and K-step 0 of the first tile have landed
  block_barrier();
  read_w(0, 0, g1);
  read_w(0, 1, g2);
  read_a(0, 0, af[0]);
  int it = 0;                 // tile iteration (bias buffer = it & 1)
  bool prev_counted = false;  // the previous tile issued exactly NST stores behind its prefetches
  unsigned long long st_data = 0, st_bar = 0, st_k = 0, st_epi = 0, st_mark = 0, st_tiles = 0, st_g4 = 0, st_half = 0, st_g0 = 0, st_first = 0;   // (ABL == 4)
  auto stamp = [] { return (unsigned long long)__builtin_amdgcn_s_memtime(); };

  for (;;) {
    f32x4 acc[FM][FN];
    {
      // the accumulators start from s * bias: register t of lane (c, q) of column tile j is column 16 j + 4 q + t of its row
      const float* biasb = reinterpret_cast<const float*>(smem + OFF_BIAS + (it & 1) * 1024) + wn * TN + 4 * (lane >> 4);
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(biasb + j * 16) * w_s;
#pragma unroll
        for (int i = 0; i < FM; ++i) acc[i][j] = b;
      }
    }
    const int cm0 = m0, cn0 = n0;
    const int tnext = t + (RR ? G : nblk);
    const bool has_next = tnext < t_end;

    // one K-step; PAR = kt & 1 = its stage (nk is even, so every tile starts in stage 0); FIRST = K-step 0 of a tile, whose
    // successor is already in flight (requested in front of the previous tile's epilogue stores, or by the prologue): the body is
    // instantiated once more for it, so that the steady-state groups are ONE basic block each (a uniform branch around the LDS-DMA
    // pieces cut every group in two, and hipcc's schedule - the LDS reads of the next group early - stopped at the cut)
    auto kstep = [&](int kt, auto PAR, auto FIRST) {
      constexpr int par = decltype(PAR)::value;
      constexpr bool first = decltype(FIRST)::value;
      const bool last = kt == nk - 1;
      if constexpr (PRIO == 2) {
        if ((wave >= 4) == (par == 1)) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
      }
      static_for<NG>([&](auto U) {
        constexpr int u = decltype(U)::value;
        constexpr bool tail = u + 1 == NG;
        if constexpr (ABLK == 4 && u == 0 && !first) st_g0 = stamp();
        if constexpr (ABLK == 4 && u == NG / 2 && !first) {
          st_g4 = stamp();
          st_first += st_g4 - st_g0;       // groups 0 .. NG / 2 - 1 of this wave
        }
        if constexpr (!tail) {
          // the planes of the row tile PF groups ahead are requested before this group's MFMAs (PF = 2: the first group of a
          // K-step requests two row tiles - the hand-over in front of the last group has only published row tile 0)
          if constexpr (PF == 1) read_a(par, u + 1, af[(u + 1) % NAF]);
          else {
            if constexpr (u == 0) read_a(par, 1, af[1]);
            if constexpr (u + 2 < NG) read_a(par, u + 2, af[(u + 2) % NAF]);
          }
        } else {
          if (!last || has_next) {
            // hand-over to the next K-step in front of the LAST group: every LDS read of this stage has been issued; once they
            // have returned the stage may be refilled (with K-step kt + 2)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            unsigned long long s0 = 0, s1 = 0;
            if constexpr (ABLK == 4) s0 = stamp();
            if constexpr (ABLK == 4 && !first) st_half += s0 - st_g4;   // groups NG / 2 .. NG - 2 of this wave
            if constexpr (ABLK < 5) {
              if (kt == 0 && prev_counted) wait_vmcnt<NST>(); else wait_vmcnt<0>();
            }
            if constexpr (ABLK == 4) s1 = stamp();
            block_barrier();
            if constexpr (ABLK == 4) {
              const unsigned long long s2 = stamp();
              st_data += s1 - s0;
              st_bar += s2 - s1;
            }
            if (ABLK != 1 && ABLK < 5) {
              if (kt + 2 < nk) {
                static_for<LPW>([&](auto I) {
                  if constexpr (x2_piece_slot(SPREAD, decltype(I)::value, LPA) < 0) stage_piece(par, kt + 2, I);
                });
              } else if (has_next) {
                if (kt + 2 == nk) {
                  tile_sources(tnext, m0, n0);
                  bias_load((it + 1) & 1, n0);
                  static_for<LPW>([&](auto I) {
                    if constexpr (x2_piece_slot(SPREAD, decltype(I)::value, LPA) < 0) stage_piece(par, 0, I);
                  });
                } else {
                  stage_load(par, 1);  // always a burst: it has to be older than the epilogue stores (counted vmcnt)
                }
              }
            }
          }
          // the first activation fragments of the next K-step, UNCONDITIONALLY (after the last step of the last tile they are
          // never used): a branch around them would put them in a basic block of their own in front of this group's first MFMA
          read_a(par ^ 1, 0, af[0]);
        }
        if constexpr (u == 0) {
          // the third weight operand of this K-step: 2^-11 g1 (exact for g1 >= 2^-3; below, its error is 2^-25 absolute on a term
          // that is 2^-11 of the product)
          if constexpr (!kTwo) {
#pragma unroll
            for (int j = 0; j < FN; ++j) gs[j] = g1[j] * static_cast<_Float16>(1.f / X2_RESID_SCALE);
          }
        }
        // products in issue order: g1 h1, g2 h1, (2^-11 g1) h2 - consecutive MFMAs hit different accumulators
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g1[j], af[u % NAF][0], acc[u][j], 0, 0, 0);
        if constexpr (tail) read_w(par ^ 1, 0, g1);   // plane by plane: the next K-step's weights take the registers just released
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(g2[j], af[u % NAF][kTwo ? 1 : 0], acc[u][j], 0, 0, 0);
        if constexpr (tail) read_w(par ^ 1, 1, g2);
        if constexpr (!kTwo) {
#pragma unroll
          for (int j = 0; j < FN; ++j) acc[u][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gs[j], af[u % NAF][1], acc[u][j], 0, 0, 0);
        }
        if constexpr (SPREAD > 0 && ABLK != 1 && ABLK < 5 && !tail && !first) {
          // K-step kt + 1 (or K-step 0 of the next tile) into the stage the previous hand-over released - UNCONDITIONALLY: behind
          // the last K-step of a workgroup's last tile the pieces fetch K-step 0 of that tile once more (valid addresses, a stage
          // nobody reads; drained before the kernel ends)
          const int lk = kt + 1 < nk ? kt + 1 : 0;
          static_for<LPW>([&](auto I) {
            if constexpr (x2_piece_slot(SPREAD, decltype(I)::value, LPA) == u) stage_piece(par ^ 1, lk, I);
          });
        }
        // issue order inside a group: ONE MFMA, then the LDS reads of the next group, then the other MFMAs (hipcc would otherwise
        // sink the reads next to their first use, and its wait for this group's operands would cover them); in the last group the
        // weight planes of the next K-step follow the four MFMAs that last used their registers
        if constexpr (!tail) {
          constexpr int kReads = PF == 1 ? 2 : (u == 0 ? 4 : u + 2 < NG ? 2 : 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if constexpr (kReads > 0) __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, NPROD * FN - 1, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, FN, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, FN, 0);
          if constexpr (!kTwo) __builtin_amdgcn_sched_group_barrier(0x008, FN, 0);
        }
      });
    };
    if constexpr (ABLK == 4) st_mark = stamp();
    kstep(0, std::integral_constant<int, 0>{}, std::true_type{});
    kstep(1, std::integral_constant<int, 1>{}, std::false_type{});
    for (int kt = 2; kt < nk; kt += 2) {
      kstep(kt, std::integral_constant<int, 0>{}, std::false_type{});
      kstep(kt + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }
    // g1 / g2 / af[0] now hold the first fragments of the next tile
    if constexpr (ABLK == 4) {
      const unsigned long long now = stamp();
      st_k += now - st_mark;
      st_mark = now;
    }

    const bool interior = cm0 + BM <= g.M && cn0 + BN <= g.N;
    if constexpr (ABLK == 3 || ABLK >= 6) {
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) keep += acc[i][j][e];
      if (keep == 123.456f) reinterpret_cast<float*>(g.C)[0] = keep;
      prev_counted = false;
    } else {
      // The tile's position is decided ONCE: an interior tile (the hot case) runs a branch-free body - with a uniform branch
      // around every store hipcc had put each LDS read, its wait and its store in a basic block of their own, the round trip
      // through the patch exposed 32 times per tile; in straight-line code the next patch's arithmetic fills it
      auto epilogue = [&](auto INTERIOR) {
        constexpr bool inter = decltype(INTERIOR)::value;
        char* stg = smem + OFF_STG + wave * 2048;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int c = lane_e & 15, q = lane_e >> 4;   // (epilogue-local copies)
        // patch: 16 rows x 128 bytes; 16-byte chunk L of row c at chunk L ^ key(c), key(c) = (c & 7) ^ (c >> 3): conflict-free both ways
        const int key = (c & 7) ^ (c >> 3);
        const int rrow = lane_e >> 3, rch = lane_e & 7;
        // LDS operations of a wave execute in order; the compiler must neither reuse nor move them across the write -> read and the
        // read -> next write points of a patch
        auto patch_fence = [] { asm volatile("" ::: "memory"); };
        // (the interior body is one basic block of 16 patches: without a scheduling fence per patch hipcc computes several patches
        // ahead and spills ~90 registers)
        auto patch_end = [&] { patch_fence(); if constexpr (!kOutX2) __builtin_amdgcn_sched_barrier(0); };
        const int rd_off[2] = {rrow * 128 + ((rch ^ rrow) << 4), (8 + rrow) * 128 + ((rch ^ rrow ^ 1) << 4)};
        if constexpr (!kOutX2) {
          // EPI_RESID3_F32: every lane adds the 16 bytes of C it is about to overwrite (the residual stream, updated in place);
          // they are requested RW patches ahead, whole lines per instruction and non-temporal, like the stores.
          // EPI_PATCH_F32 (the patch embedding: row m = (image, patch) of the im2col matrix goes to token row m + m / P + 1, behind
          // its image's class token): the same machinery fetches the positional-embedding row (m % P) + 1 of g.aux [P + 1, N]
          // (L2-resident) instead.  m / P through the float reciprocal, exact for m < 2^23 after one correction step.
          constexpr int RWIN = RW;
          f32x4 xres[kAdd ? RWIN : 1][2];
          const float inv_p = kPatch ? 1.f / (float)g.P : 0.f;
          auto token_row = [&](int mo, int& rem) {   // (image index, patch index) of im2col row mo
            int q = (int)((float)mo * inv_p);
            rem = mo - q * g.P;
            if (rem < 0) { rem += g.P; --q; }
            if (rem >= g.P) { rem -= g.P; ++q; }
            return q;
          };
          auto resid_load = [&](int pt, f32x4 (&dst)[2]) {
            const int i = pt / (FN / 2), jp = pt % (FN / 2);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              const int mo = cm0 + wm * TM + i * 16 + s * 8 + rrow, no = cn0 + wn * TN + jp * 32 + rch * 4;
              dst[s] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (inter || (mo < g.M && no < g.N)) {
                if constexpr (kPatch) {
                  int rem;
                  token_row(mo, rem);
                  dst[s] = *reinterpret_cast<const f32x4*>(g.aux + (size_t)(rem + 1) * g.N + no);
                } else {
                  dst[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(g.C) + (size_t)mo * g.ldc + no));
                }
              }
            }
          };
          if constexpr (kAdd) {
#pragma unroll
            for (int pt = 0; pt < RWIN && pt < NP; ++pt) resid_load(pt, xres[pt]);
          }
          char* wr = stg + c * 128;
#pragma unroll
          for (int i = 0; i < FM; ++i) {
#pragma unroll
            for (int jp = 0; jp < FN / 2; ++jp) {
              const int pt = i * (FN / 2) + jp;
#pragma unroll
              for (int jj = 0; jj < 2; ++jj)
                *reinterpret_cast<f32x4*>(wr + (((4 * jj + q) ^ key) << 4)) = acc[i][2 * jp + jj] * w_inv;
              patch_fence();
#pragma unroll
              for (int s = 0; s < 2; ++s) {
                f32x4 val = *reinterpret_cast<const f32x4*>(stg + rd_off[s]);
                if constexpr (kAdd) val = xres[pt % RWIN][s] + val;
                const int mo = cm0 + wm * TM + i * 16 + s * 8 + rrow, no = cn0 + wn * TN + jp * 32 + rch * 4;
                if (inter || (mo < g.M && no < g.N)) {
                  size_t out_row = (size_t)mo;
                  if constexpr (kPatch) {
                    int rem;
                    out_row = (size_t)mo + (size_t)token_row(mo, rem) + 1;
                  }
                  f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(g.C) + out_row * g.ldc + no);
                  __builtin_nontemporal_store(val, dst);
                }
              }
              if constexpr (kAdd) {
                if (pt + RWIN < NP) resid_load(pt + RWIN, xres[pt % RWIN]);
              }
              patch_end();
            }
          }
        } else {
          // x2 outputs (the next GEMM's activation operand): exact QuickGELU, then the two fp16 planes; two column tiles are ONE
          // 128-byte line [h1 x32 | h2 x32] of 16 rows: the four values of lane (c, q) in column tile jj are the 8 bytes at
          // 32 jj + 8 q of either plane = half (q & 1) of chunk 2 jj + (q >> 1) (+ 4 for the second plane).  The planes of patch
          // pt + 1 are computed between the writes and the reads of patch pt.
          const size_t ldc_b = (size_t)g.ldc * 2;
          char* cbase = reinterpret_cast<char*>(g.C) + (size_t)(cm0 + wm * TM + rrow) * ldc_b + rch * 16;
          // (the 16 lanes of a ds_write_b64 group hold ONE 8-byte half of 16 rows, and rows c, c ^ 9 share a chunk: a two-way conflict,
          // 16 LDS cycles per patch.  ABL 7, lab only: the half also taken from the row's parity - conflict-free, WRONG data - sizes
          // what a conflict-free layout could gain: docs/rounds/round6.md)
          char* wr = stg + c * 128 + ((q & 1) ^ (ABL == 7 ? (c & 1) : 0)) * 8;
          const int qh = q >> 1;
          // range test on the PLANES: h1 = fp16(v) is an infinity exactly when v does not fit fp16 (|v| >= 65520) and a NaN when v
          // is one, and either turns `bad` into a NaN for good (x * 0 + bad, two packed instructions per four values; a running
          // maximum of |v| cost three and dropped NaNs)
          f16x4 bad = {0, 0, 0, 0};
          const f16x4 zero4 = {0, 0, 0, 0};
          f16x4 h1[2][2], h2[2][2];   // [patch parity][column tile]
          auto planes = [&](auto PT) {
            constexpr int pt = decltype(PT)::value, i = pt / (FN / 2), jp = pt % (FN / 2);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              f32x4 v = acc[i][2 * jp + jj] * w_inv;
              if constexpr (GW) v = quick_gelu_f32x4_wide(v);  // (packed pairs, the two halves interleaved; the bits of quick_gelu_exact)
              else v = quick_gelu_f32x4(v);
              if constexpr (GW == 2) split2_mix(v, h1[pt & 1][jj], h2[pt & 1][jj]); else split2(v, h1[pt & 1][jj], h2[pt & 1][jj]);
              bad = __builtin_elementwise_fma(h1[pt & 1][jj], zero4, bad);
            }
            asm volatile("" : "+v"(bad));   // (accumulated HERE: hipcc otherwise keeps the planes alive for one reduction at the end)
          };
          planes(std::integral_constant<int, 0>{});
          static_for<NP>([&](auto PT) {
            constexpr int pt = decltype(PT)::value, i = pt / (FN / 2), jp = pt % (FN / 2);
            const int group = (cn0 + wn * TN + jp * 32) / X2_GROUP;                       // wave-uniform
            char* tile_base = cbase + (size_t)(i * 16) * ldc_b + (size_t)group * X2_GROUP_BYTES;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
              *reinterpret_cast<f16x4*>(wr + (((2 * jj + qh) ^ key) << 4)) = h1[pt & 1][jj];
              *reinterpret_cast<f16x4*>(wr + (((4 + 2 * jj + qh) ^ key) << 4)) = h2[pt & 1][jj];
            }
            patch_fence();
            if constexpr (pt + 1 < NP) planes(std::integral_constant<int, pt + 1>{});
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              const f16x8 val = *reinterpret_cast<const f16x8*>(stg + rd_off[s]);
              const int mo = cm0 + wm * TM + i * 16 + s * 8 + rrow;
              if (inter || (mo < g.M && group * X2_GROUP < g.N))
                __builtin_nontemporal_store(val, reinterpret_cast<f16x8*>(tile_base + (size_t)(s * 8) * ldc_b));
            }
            patch_end();
          });
          if (g.sat_flag) {
            const float b0 = (float)bad[0] + (float)bad[1], b1 = (float)bad[2] + (float)bad[3];
            if (b0 != b0 || b1 != b1) atomicOr(g.sat_flag, 1);
          }
        }
      };
      if (interior) epilogue(std::true_type{}); else epilogue(std::false_type{});
      if constexpr (ABLK == 4) {
        st_epi += stamp() - st_mark;
        ++st_tiles;
      }
      prev_counted = interior;
    }
    if (!has_next) break;
    ++it;
    t = tnext;
  }
  if constexpr (ABLK == 4) {
    if (lane == 0 && g.aux) {
      unsigned long long* d = reinterpret_cast<unsigned long long*>(const_cast<float*>(g.aux)) + ((size_t)blockIdx.x * NW + wave) * 8;
      d[0] = st_data; d[1] = st_bar; d[2] = st_k; d[3] = st_epi; d[4] = st_tiles; d[5] = (unsigned long long)nk; d[6] = st_first; d[7] = st_half;
    }
  }
  wait_vmcnt<0>();   // (the unread pieces behind the last K-step: no LDS-DMA may be in flight when the workgroup's LDS is released)
}

}  // namespace
}  // namespace fc

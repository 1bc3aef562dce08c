// MFMA GEMM with fused epilogues for the transformer blocks of the CLIP towers (gfx950).
//
//   C[M,N] = epilogue(A[M,K] . W[N,K]^T)          A, W both K-contiguous (W = torch Linear weight as stored)
//
// One kernel template covers both arithmetic modes of the library:
//   * T = bf16 : v_mfma_f32_16x16x32_bf16, fp32 accumulate             (throughput path)
//   * T = f32  : v_mfma_f32_16x16x4_f32, bit-exact fp32 fma chain      (parity path)
// Both use the same LDS image: a tile row is 128 bytes of K (64 bf16 / 32 f32) cut in eight 16-byte chunks.
//
// Data movement (guide: cdna_hip_programming.md section 5):
//   * global -> LDS with `global_load_lds_dwordx4` (no VGPR round trip).  One wave-instruction fills 8 rows x 128 B.
//     The LDS destination of that instruction is lane-linear, so the bank swizzle is applied to the per-lane SOURCE
//     address: physical chunk pc of row r holds logical chunk  pc ^ ((r >> 1) & 7)  (rule 21: same involution on the
//     read side).  With it every 16-lane group of a ds_read_b128 fragment read touches 16 distinct 16-byte bank
//     slots.
//   * two LDS stages; the loads of K-tile t+1 are issued right after the barrier that publishes tile t and fly
//     underneath the MFMAs of tile t (one barrier per K-tile).
//   * fragments: lane (r = lane & 15, q = lane >> 4) reads 16 bytes of row r.  bf16: logical chunk 4s+q = k 8q..8q+7
//     of k-substep s (the MFMA operand layout).  f32: logical chunk q+4s; element t of the four floats feeds MFMA
//     t, i.e. the MFMA k-slot q of step (s,t) is k = 16s+4q+t.  A and W use the same permutation, so the dot product
//     is complete and every k is used once.
//   * the MFMA is issued with W as the A-operand and the activations as the B-operand, so a lane ends up holding
//     FOUR CONSECUTIVE output columns of one output row: the epilogue is one 8-byte (bf16) or 16-byte (f32) access
//     per 16x16 tile instead of four scalar ones.
//   * blockIdx -> tile mapping is XCD-aware (T1, bijective form): the blocks that land on one XCD walk the N-tiles
//     of consecutive M-panels, so the activation panel is fetched from HBM once and re-read from that XCD's L2.
#pragma once
#include "common.h"
#include <type_traits>

namespace fc {

namespace {

constexpr int ROWB = 128;  // bytes of K per LDS tile row

// bf16 outputs: x * sigmoid(1.702 x) with the hardware exp2 / rcp (1 ulp each, far below bf16 resolution): 5 VALU
// instructions per element.  (`__frcp_rn` expands to the 12-instruction IEEE division sequence; with it the epilogue
// of the c_fc GEMM cost 20 % of the launch.)
__device__ __forceinline__ float quick_gelu_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930156f * x));  // 1.702 * log2(e)
}
// Four accumulator values at once: the multiplies / add as packed-fp32 instructions (v_pk_mul_f32 / v_pk_add_f32, two
// elements per issue), only exp2 and rcp stay per element.
__device__ __forceinline__ f32x4 quick_gelu_fast4(f32x4 x) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 lo = {x[0], x[1]}, hi = {x[2], x[3]};
  const f32x2 c = {-2.4554669595930156f, -2.4554669595930156f}, one = {1.f, 1.f};
  const f32x2 tl = lo * c, th = hi * c;
  f32x2 el = {__builtin_amdgcn_exp2f(tl[0]), __builtin_amdgcn_exp2f(tl[1])};
  f32x2 eh = {__builtin_amdgcn_exp2f(th[0]), __builtin_amdgcn_exp2f(th[1])};
  el = el + one;
  eh = eh + one;
  f32x2 rl = {__builtin_amdgcn_rcpf(el[0]), __builtin_amdgcn_rcpf(el[1])};
  f32x2 rh = {__builtin_amdgcn_rcpf(eh[0]), __builtin_amdgcn_rcpf(eh[1])};
  rl = lo * rl;
  rh = hi * rh;
  return f32x4{rl[0], rl[1], rh[0], rh[1]};
}
__device__ __forceinline__ float quick_gelu_exact(float x) { return quick_gelu_f32(x); }      // common.h
__device__ __forceinline__ float quick_gelu_grad(float x) { return quick_gelu_grad_f32(x); }
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 load4<bf16>(const bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{static_cast<float>(v[0]), static_cast<float>(v[1]), static_cast<float>(v[2]), static_cast<float>(v[3])};
}

template <typename T> struct Frag;
template <> struct Frag<bf16> { using type = bf16x8; };
template <> struct Frag<float> { using type = f32x4; };

template <typename T>
__device__ __forceinline__ void mma(const typename Frag<T>::type& w, const typename Frag<T>::type& x, f32x4& acc);

template <>
__device__ __forceinline__ void mma<bf16>(const bf16x8& w, const bf16x8& x, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma<float>(const f32x4& w, const f32x4& x, f32x4& acc) {
#pragma unroll
  for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t], x[t], acc, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ void store4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void store4<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const f32x4& v) {
  bf16x4 o;
  o[0] = static_cast<bf16>(v[0]); o[1] = static_cast<bf16>(v[1]);
  o[2] = static_cast<bf16>(v[2]); o[3] = static_cast<bf16>(v[3]);
  *reinterpret_cast<bf16x4*>(p) = o;
}

template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& body) {
  if constexpr (I < N) {
    body(std::integral_constant<int, I>{});
    static_for<N, I + 1>(body);
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void block_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// wave-uniform q in 1..MAXQ: s_waitcnt vmcnt(BASE + PER * q) (the counter is an immediate: one compare + branch per value)
template <int BASE, int PER, int MAXQ>
__device__ __forceinline__ void wait_vmcnt_steps(int q) {
  static_for<MAXQ>([&](auto I) {
    constexpr int n = BASE + PER * (decltype(I)::value + 1);
    if (q == decltype(I)::value + 1) wait_vmcnt<(n < 64 ? n : 63)>();
  });
}

// ABL (tools/gemm_lab only): 0 = real kernel; 1 = no global loads inside the K loop; 2 = every block stages tile (0,0);
// 3 = no epilogue stores; 4 = (f32, pipelined) write-back instead of non-temporal epilogue stores.
// NSTG > 2: a ring of NSTG LDS stages with the LDS-DMA NSTG - 1 K-tiles ahead (counted vmcnt, one raw barrier per K-step) - for
// SMALL tiles on small problems: the text tower of a 32-caption eval batch is 2464 rows; its N = 512 GEMMs are 80 tiles of
// 128 x 128 for 256 CUs, and a 64 x 64 tile's K-step is 0.4 us of MFMA issue - with two stages every K-step waits for the
// latency of its own loads.  Same K order and MFMA chain per output element as every other kernel of this file.
template <typename T, int BM, int BN, int WM, int WN, int EPI, int ABL = 0, int NSTG = 2>
__global__ void __launch_bounds__(WM * WN * 64) gemm_kernel(const GemmArgs g) {
  constexpr int NW = WM * WN;
  constexpr int BKE = ROWB / (int)sizeof(T);  // K elements per tile row
  constexpr int TM = BM / WM, TN = BN / WN;   // wave tile
  constexpr int FM = TM / 16, FN = TN / 16;   // 16x16 fragments per wave tile
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int RG = (BM + BN) / 8;  // 8-row groups (one glds wave-instruction each)
  constexpr int LPW = RG / NW;       // glds per wave per stage
  static_assert(RG % NW == 0, "row groups must divide over the waves");
  static_assert(BM % 16 == 0 && BN % 16 == 0, "tile");
  // EPI_RANKS_I32 (scoring without the score matrix): the workgroup owns the row block blockIdx % tilesM and walks g.ctw column
  // tiles of it.  FIRST it multiplies its rows with the rows' own TARGET columns (the weight tile gathers row t_m of W for
  // local row m) and keeps the diagonal - s[m, t_m], by the same MFMA chain in the same K order as any other element of the
  // product, i.e. with the bits fc_similarity would have stored - in LDS; then, tile by tile, every lane compares its
  // alpha * acc with its row's reference and counts.  No [M, N] store.
  constexpr bool kRanks = EPI == EPI_RANKS_I32;
  static_assert(!kRanks || (sizeof(T) == 4 && BM == BN && WM == WN), "ranks epilogue: square fp32 tile, diagonal inside the diagonal waves");
  using FragT = typename Frag<T>::type;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ float diag_s[kRanks ? BM : 1];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware, bijective block -> tile map
  const int tilesN = (g.N + BN - 1) / BN;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, qd = nwg >> 3, rd = nwg & 7;
  const int t = (xcd < rd ? xcd * (qd + 1) : rd * (qd + 1) + (xcd - rd) * qd) + (orig >> 3);
  const int tilesM = (g.M + BM - 1) / BM;
  const int m0 = kRanks ? (t % tilesM) * BM : (t / tilesN) * BM;
  const int ct0 = kRanks ? (t / tilesM) * g.ctw : t % tilesN;               // first column tile
  const int npass = kRanks ? min(g.ctw, tilesN - ct0) : 1;                  // column tiles of this workgroup
  const int r = lane & 15, q = lane >> 4, f = (r >> 1) & 7;
  int cnt[kRanks ? FM : 1] = {};

  for (int pass = kRanks ? -1 : 0; pass < npass; ++pass) {   // (ranks: pass -1 = the rows' own target columns)
  const int n0 = (ct0 + max(pass, 0)) * BN;
  // ---- per-lane staging sources
  const char* src[LPW];
  // patch gather (EPI_PATCH_F32, f32, g.gR > 0): the A row of an (image, patch) pair is not contiguous - its 16-byte
  // chunk `lchunk` of K-tile kt holds 4 consecutive pixels of one patch row: k = 32 kt + 4 lchunk = (c, py, px)
  constexpr bool kGather = EPI == EPI_PATCH_F32 && sizeof(T) == 4;
  constexpr int LPA_G = BM / 8 / NW;  // the first LPA_G pieces of a wave are activation rows
  int lchunk[kGather ? LPA_G : 1];
  {
    const int rin = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int row = (wave + i * NW) * 8 + rin;  // row in the stacked [A tile ; W tile] image
      const int chunk = pc ^ ((row >> 1) & 7);
      if (row < BM) {
        const int gr = min((ABL == 2 ? 0 : m0) + row, g.M - 1);
        if constexpr (kGather) {
          if (g.gR > 0) {
            const int G = g.gR / g.gP;
            const int img = gr / g.P, pidx = gr - img * g.P, gy = pidx / G, gx = pidx - gy * G;
            src[i] = reinterpret_cast<const char*>(g.A) +
                     (((size_t)img * 3 * g.gR + (size_t)gy * g.gP) * g.gR + (size_t)gx * g.gP) * sizeof(float);
            if (i < LPA_G) lchunk[i] = chunk;
            continue;
          }
        }
        src[i] = reinterpret_cast<const char*>(g.A) + ((size_t)gr * g.lda) * sizeof(T) + chunk * 16;
      } else {
        int gr = min((ABL == 2 ? 0 : n0) + row - BM, g.N - 1);
        if constexpr (kRanks) {
          if (pass < 0) {  // the target column of local row (row - BM)
            const int m = min(m0 + row - BM, g.M - 1);
            gr = min(max(g.targets ? g.targets[m] : m + g.tgt_off, 0), g.N - 1);
          }
        }
        src[i] = reinterpret_cast<const char*>(g.W) + ((size_t)gr * g.ldw) * sizeof(T) + chunk * 16;
      }
    }
  }
  auto stage_load = [&](int stage, int kt) {
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const char* p = src[i] + (size_t)kt * ROWB;
      if constexpr (kGather) {
        if (g.gR > 0 && i < LPA_G) {
          const int k = kt * BKE + lchunk[i] * 4, pp = g.gP * g.gP;
          const int c = k / pp, rem = k - c * pp, py = rem / g.gP, px = rem - py * g.gP;
          p = src[i] + (((size_t)c * g.gR + py) * g.gR + px) * sizeof(float);
        }
      }
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)p,
          (__attribute__((address_space(3))) void*)(smem + stage * STAGE + (wave + i * NW) * 1024),
          16, 0, 0);
    }
  };

  // The two store epilogues and the residual update start the accumulators from the bias (as the pipelined kernel does, so
  // both kernels produce bit-identical results and a row's value does not depend on which one the batch size selects).
  constexpr bool kBiasInit = EPI == EPI_BIAS_T || EPI == EPI_GELU_T || EPI == EPI_RESID_F32;
  f32x4 acc[FM][FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if constexpr (kBiasInit) b = *reinterpret_cast<const f32x4*>(g.bias + min(n0 + wn * TN + j * 16 + 4 * q, g.N - 4));
#pragma unroll
    for (int i = 0; i < FM; ++i) acc[i][j] = b;
  }

  const int a_base = (wm * TM + r) * ROWB;
  const int b_base = BM * ROWB + (wn * TN + r) * ROWB;

  const int nk = g.K / BKE;
  // the two block-GEMM epilogues use the rotated K order of gemm_pipelined_kernel (bit-identical results); scoring
  // (EPI_STORE_F32 / EPI_RANKS_I32) and patch embedding keep the natural order, so a score does not depend on its column position
  const int rot = kBiasInit ? (n0 >> 8) % nk : 0;
  auto krot = [&](int kt) { return kt + rot >= nk ? kt + rot - nk : kt + rot; };
  if constexpr (kRanks) __syncthreads();  // (every wave has left the previous pass's last stage)
  static_assert(NSTG >= 2 && (NSTG == 2 || (NSTG - 2) * LPW < 64), "ring depth");
  if constexpr (NSTG == 2) {
    stage_load(0, krot(0));
  } else {
#pragma unroll
    for (int p = 0; p < NSTG - 1; ++p)
      if (p < nk) stage_load(p, krot(p));
  }
  for (int kt = 0; kt < nk; ++kt) {
    if constexpr (NSTG == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (ABL != 1 && kt + 1 < nk) stage_load((kt + 1) & 1, krot(kt + 1));
    } else {
      // K-tile kt has landed when at most the K-tiles behind it are in flight: NSTG - 2 of them, fewer at the end of the row
      const int behind = min(NSTG - 2, nk - 1 - kt);
      if (behind > 0) wait_vmcnt_steps<0, LPW, NSTG - 2>(behind); else wait_vmcnt<0>();
      block_barrier();  // ... for every wave; and every wave has finished reading the stage of K-tile kt - 1
      if (ABL != 1 && kt + NSTG - 1 < nk) stage_load((kt + NSTG - 1) % NSTG, krot(kt + NSTG - 1));
    }
    const char* st = smem + (kt % NSTG) * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = (sizeof(T) == 2) ? (4 * s + q) : (q + 4 * s);
      const int off = (c ^ f) * 16;
      FragT xa[FM], wb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) xa[i] = *reinterpret_cast<const FragT*>(st + a_base + i * 16 * ROWB + off);
#pragma unroll
      for (int j = 0; j < FN; ++j) wb[j] = *reinterpret_cast<const FragT*>(st + b_base + j * 16 * ROWB + off);
      if constexpr (sizeof(T) == 4) {  // k-slot outermost: consecutive MFMAs hit different accumulators (see the pipelined kernel)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j][t], xa[i][t], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) mma<T>(wb[j], xa[i], acc[i][j]);
      }
    }
  }

  if constexpr (kRanks) {
    if (pass < 0) {
      // local row m = wm TM + 16 i + r meets its own column in wave wn == wm, fragment j == i, lane group q == r / 4, element r % 4
      if (wm == wn && q == (r >> 2)) {
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const f32x4 v = acc[i][i] * g.alpha;
          diag_s[wm * TM + i * 16 + r] = (r & 3) == 0 ? v[0] : (r & 3) == 1 ? v[1] : (r & 3) == 2 ? v[2] : v[3];
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int ml = wm * TM + i * 16 + r, m = min(m0 + ml, g.M - 1);
        const float ref = diag_s[ml];
        const int tcol = min(max(g.targets ? g.targets[m] : m + g.tgt_off, 0), g.N - 1);
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const f32x4 v = acc[i][j] * g.alpha;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int n = n0 + wn * TN + j * 16 + 4 * q + e;
            cnt[i] += (n < g.N) && ((v[e] > ref) || (v[e] == ref && n < tcol));
          }
        }
      }
    }
    continue;
  }

  if constexpr (ABL == 3) {
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (keep == 123.456f) reinterpret_cast<float*>(g.C)[0] = keep;
    return;
  }
  // ---- epilogue: lane holds C[m = .. + r][n = .. + 4q .. 4q+3] for every (i, j)
#pragma unroll
  for (int i = 0; i < FM; ++i) {
    const int m = m0 + wm * TM + i * 16 + r;
    if (m >= g.M) continue;
    size_t orow = (size_t)m;
    int prow = 0;
    if constexpr (EPI == EPI_PATCH_F32) {
      const int img = m / g.P;
      prow = m - img * g.P + 1;
      orow = (size_t)img * (g.P + 1) + prow;
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int n = n0 + wn * TN + j * 16 + 4 * q;
      if (n >= g.N) continue;  // N is a multiple of 4 (checked on the host)
      f32x4 v = acc[i][j];
      if constexpr (EPI == EPI_STORE_F32) {
        v *= g.alpha;
        if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + n);
        store4<float>(reinterpret_cast<float*>(g.C) + orow * g.ldc + n, v);
      } else if constexpr (EPI == EPI_PATCH_F32) {
        v += *reinterpret_cast<const f32x4*>(g.aux + (size_t)prow * g.N + n);
        store4<float>(reinterpret_cast<float*>(g.C) + orow * g.ldc + n, v);
      } else if constexpr (EPI == EPI_RESID_F32) {
        float* p = reinterpret_cast<float*>(g.C) + orow * g.ldc + n;
        v = *reinterpret_cast<const f32x4*>(p) + v;  // (the accumulator started from the bias)
        store4<float>(p, v);
      } else if constexpr (EPI == EPI_DGELU_T) {
        const f32x4 pre = load4<T>(reinterpret_cast<const T*>(g.aux) + orow * g.ldc + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= quick_gelu_grad(pre[e]);
        store4<T>(reinterpret_cast<T*>(g.C) + orow * g.ldc + n, v);
      } else if constexpr (!kRanks) {
        if constexpr (EPI == EPI_GELU_T) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (sizeof(T) == 2) ? quick_gelu_fast(v[e]) : quick_gelu_exact(v[e]);
        }
        store4<T>(reinterpret_cast<T*>(g.C) + orow * g.ldc + n, v);
      }
    }
  }
  }  // pass
  if constexpr (kRanks) {
    // a row's count sits in the four lane groups of two waves (wn = 0, 1): fold the groups, then one integer atomic per wave
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int c = cnt[i];
      c += __shfl_xor(c, 16, 64);
      c += __shfl_xor(c, 32, 64);
      const int m = m0 + wm * TM + i * 16 + r;
      if (q == 0 && m < g.M && c) atomicAdd(reinterpret_cast<int*>(g.C) + m, c);
    }
  }
}


// ====================================================================================== persistent, pipelined GEMM
// Same tile, LDS image, swizzle and MFMA shapes as gemm_kernel, restructured around what the first profiles showed
// (profiles/r01_v1_*): with K = 768 an output tile only lives for 12 K-steps, so the un-overlapped prologue (first
// HBM/L2 round trip) and above all the epilogue (8-byte stores in 32-byte row pieces: 43 % of the kernel at
// N = 2304) dominated.  Here
//   * the grid is one workgroup per CU and every workgroup walks its own list of tiles (XCD-contiguous, so the 32
//     workgroups of an XCD always work on 32 consecutive tiles);
//   * the K-tiles of the NEXT output tile are already streaming into LDS while the current tile finishes: tile t+1's
//     K-tile 0 is issued during the last K-step of tile t, its K-tile 1 right after the barrier that ends tile t,
//     i.e. BEFORE the epilogue stores.  vmcnt retires in order, so the first two K-steps of tile t+1 wait with a
//     COUNTED `s_waitcnt vmcnt(stores [+ loads])`: the epilogue stores of tile t drain underneath the MFMAs of tile
//     t+1 instead of being waited for;
//   * bf16 outputs are transposed through a private 2 KiB LDS patch per wave, so every store instruction writes
//     8 rows x 128 contiguous bytes with 16 bytes per lane (4x fewer, full-line stores); the bias slice of the tile
//     arrives by LDS-DMA with the first K-tile (an ordinary global load here would make hipcc drain vmcnt to 0);
//   * barriers are raw `s_barrier`s with hand-placed waits (a `__syncthreads()` would add `vmcnt(0)`).
// Supported epilogues: EPI_BIAS_T, EPI_GELU_T (the four big GEMMs of a transformer block).  Needs K/BKE >= 3.
// Issue slot of LDS-DMA piece idx (0..3 activation rows, 4..7 weight rows) under schedule `sched`; see the kernel.
constexpr int piece_slot(int sched, int idx) {
  switch (sched) {
    case 1: return idx / 4;                          // group 0: A, group 1: W
    case 2: return idx < 4 ? -1 : 0;                 // A in the hand-over, W after group 0
    case 3: return idx < 4 ? -1 : 1;                 // A in the hand-over, W after group 1
    case 4: return idx < 4 ? -1 : (idx - 4) / 2;     // A in the hand-over, W over groups 0, 1
    case 5: return idx < 4 ? 0 : 1 + (idx - 4) / 2;  // A after group 0, W over groups 1, 2
    case 6: return idx < 2 ? -1 : (idx - 2) / 2;     // 2 in the hand-over, 2 after each of groups 0..2
    case 7: return idx / 2;                          // 2 after each of groups 0..3
    case 8: return idx < 4 ? 1 : 0;                  // group 0: W, group 1: A
    case 9: return 0;                                // all 8 after group 0
    case 10: return idx < 4 ? 2 : 0;                 // group 0: W, group 2: A
    case 11: return idx < 4 ? 2 + idx / 2 : (idx - 4) / 2;  // W over groups 0, 1; A over groups 2, 3
    case 12: return idx < 4 ? 0 : -1;                // W in the hand-over, A after group 0
    case 13: return idx < 4 ? 2 : 1;                 // group 1: W, group 2: A
    default: return -1;                              // everything in the hand-over
  }
}

// Issue slot of piece idx inside a K-step of `ng` MFMA groups: the schedules above name groups 0..2 of a full tile's 8; a
// lower tile (HT, below) has 2, 4 or 6 groups and the hand-over sits in front of its last one, so slots are clamped to ng - 2.
constexpr int piece_slot_in(int sched, int idx, int ng) {
  const int s = piece_slot(sched, idx);
  // (a piece issued after group u has the groups u + 1 .. ng - 2 to land before the hand-over waits for it: a K-step of 2 or 4
  // groups has no such window - everything goes out in the hand-over, one whole K-step ahead)
  return (s < 0 || ng < 6) ? -1 : (s > ng - 2 ? ng - 2 : s);
}

// The K loop is software pipelined: the MFMAs of a K-tile are issued in groups of 2*FN (one pair of 16-row fragments x
// the wave's FN column fragments, per k-substep); the LDS reads of group
// u+1 are issued before the MFMAs of group u (pinned with sched_group_barrier: hipcc otherwise sinks the reads next
// to their first use), and the hand-over to the next K-tile (wait for its DMA, barrier, issue the DMA two tiles ahead,
// first fragment reads) sits in front of the LAST group of the current tile, so neither LDS latency nor the barrier
// leaves the matrix pipe idle.  The accumulators start from the bias slice (no bias registers in the epilogue).
//
// HT > 0 (round 4; fp32): a TAIL of lower tiles.  fp32 is MFMA-bound on every CU, so with 256-row tiles a launch pays for whole
// rounds of tiles over the CUs: out_proj of a 128-frame call (the reference's eval batch: 32 clips x 4 frames) has 297 tiles
// for 256 CUs = 2 rounds for 1.16 rounds of work.  The host (gemm.hip) cuts the rows in a HEAD of g.hp 256-row panels whose
// tiles fill whole rounds (walked in the panel-sharing order below) and a TAIL, the remaining rows, walked as tiles of
// HT x 64 rows (HT = 1, 2, 3; XCD-major round robin, N fastest) by the same workgroups in the same launch: a tile of HT units
// stages HT x 64 activation rows per K-step, each wave row takes 32 HT of them and the K-step issues 2 HT MFMA groups
// instead of 8 - its cost is proportional to its height.  The tail's K-tiles are already streaming in while the last head
// tile finishes.  An output element sees its K-tiles in the rotated order of its COLUMN tile and its products in the same
// MFMA chain whatever the tile height: results are bit-identical to the fixed-tile kernels.
template <typename T, int BM, int BN, int WM, int WN, int EPI, int ABL = 0, int ROT = 1, int SCHED = 0, int HT = 0>
__global__ void __launch_bounds__(WM * WN * 64) gemm_pipelined_kernel(const GemmArgs g) {
  constexpr int NW = WM * WN;
  constexpr int BKE = ROWB / (int)sizeof(T);
  constexpr int TN = BN / WN;
  constexpr int FN = TN / 16;
  constexpr int HQF = BM / 64;                        // 64-row units of a full tile; a wave row covers 32 rows of each
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
  constexpr int SPQ = 2 * (kStaged ? IPP : FN);       // store instructions per wave per 64-row unit of an interior tile
  constexpr int NST = HQF * SPQ;                      // ... per full tile
  static_assert(WM == 2 && BM % 64 == 0, "a tile is cut in 64-row units, 32 rows of each per wave row");
  static_assert(RG % NW == 0 && BN <= 256 && (!kStaged || TN == 64 || TN == 128), "tile");
  static_assert(kStaged ? NW * PATCH <= NW * 2048 : (TN % 32 == 0 && FN % 2 == 0), "output patch");
  static_assert(EPI == EPI_BIAS_T || EPI == EPI_GELU_T || EPI == EPI_RESID_F32 ||
                ((EPI == EPI_DGELU_T || EPI == EPI_PATCH_F32) && sizeof(T) == 4), "epilogue");
  constexpr bool BAL = HT > 0;
  static_assert(!BAL || (NW == 8 && BM == 256 && HT < HQF), "tail tiles: 8 waves, one LDS-DMA piece per wave and 64-row unit");
  // EPI_PATCH_F32 (fp32, round 4): the patch embedding.  A is the FRAMES tensor f32 [n, 3, gR, gR]: row m = (image, patch), column
  // k = (channel, py, px); the LDS-DMA loader gathers the 16-byte pieces (4 pixels of a patch row) itself, as gemm_kernel does:
  // a lane's piece of K-tile kt sits at [tile's first image] + [its (image, patch, chunk) offset, 32 bits] + [a K-tile term that
  // only depends on kt: channel and first patch row].  Patch sizes 8, 16, 32 (a K-tile of 32 floats is 4, 2 or 1 patch rows and
  // K-tiles per channel is a power of two).  No bias, natural K order (the bits of gemm_kernel's patch epilogue); the epilogue
  // adds the positional embedding and skips the CLS row of every image.
  constexpr bool kPatch = EPI == EPI_PATCH_F32;
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
  const int tilesM = BAL ? g.hp : (g.M + BM - 1) / BM;   // panels of the head
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
  // fp32 (kRR): tiles are dealt ROUND ROBIN - in round c, workgroup wk (XCD-major number) takes tile c G + wk of the N-fastest
  // order, so the 32 workgroups of an XCD work on 32 consecutive tiles (they share ~32 / tilesN activation panels in their L2)
  // and every workgroup gets the same number of tiles to within one.  (The panel-range partition above quantises PER XCD: 85
  // panels of out_proj are 11 panels = 33 tiles for 32 workgroups on five of the XCDs - a second round for one tile.  bf16
  // keeps it: there the chip is power- and bandwidth-limited and a partial round costs nothing measurable.)
  const bool kRR = sizeof(T) == 4 && ABL != 2 && g.order != 1;   // (g.order: 0 / 2 = round robin, 1 = XCD panel ranges)
  const int qd_ = G >> 3, rd_ = G & 7;
  const int wk = (xcd < rd_ ? xcd * (qd_ + 1) : rd_ * (qd_ + 1) + (xcd - rd_) * qd_) + pos;
  // this workgroup's tiles, numbered 0 .. nh + ntw - 1: nh head tiles, then ntw tail tiles (tail tile numbers wk, wk + G, ...)
  const int t_end = npanel * tnn;
  const int head_tiles = tilesM * tilesN;
  const int nh = kRR ? (wk < head_tiles ? (head_tiles - wk + G - 1) / G : 0)
                     : (pos < t_end ? (t_end - pos + nblk - 1) / nblk : 0);
  int ntw = 0;
  if constexpr (BAL) {
    const int rows = g.M - g.hp * BM;
    const int ntail = rows > 0 ? ((rows + 64 * HT - 1) / (64 * HT)) * tilesN : 0;
    ntw = wk < ntail ? (ntail - wk + G - 1) / G : 0;
  }
  const int cur_end = nh + ntw;
  if (cur_end == 0) return;
  // tile number `c` of this workgroup: first row, first column, height in 64-row units
  auto tile_at = [&](int c, int& m0, int& n0, int& hq) {
    if (!BAL || c < nh) {
      if (kRR) {
        const int lt = wk + c * G;
        const int tm = lt / tilesN;
        m0 = tm * BM;
        n0 = (lt - tm * tilesN) * BN;
      } else {
        const int lt = pos + c * nblk;
        const int per_block = npanel * nbw;
        const int nb = lt / per_block, rem = lt - nb * per_block;
        const int wb = min(nbw, tnn - nb * nbw);
        m0 = (mp0 + rem / wb) * BM;
        n0 = (tn0 + nb * nbw + rem % wb) * BN;
      }
      hq = HQF;
    } else {
      const int tt = wk + (c - nh) * G;
      const int rt = tt / tilesN;
      m0 = g.hp * BM + rt * (64 * HT);
      n0 = (tt - rt * tilesN) * BN;
      hq = HT;
    }
  };

  // per-lane staging sources as 32-bit byte offsets from two (scalar) base pointers: the weight is < 4 GiB, the activations
  // are addressed from the first row of the current tile
  constexpr int LPA = BM / 8 / NW, LPB = BN / 8 / NW;  // LDS-DMA instructions per wave per stage for A / W
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile rows must divide over the waves");
  // (row >> 1) & 7 of a staged row only depends on (wave, rin): row = (wave + i * NW) * 8 + rin and NW * 4 = 0 mod 8
  static_assert((NW * 4) % 8 == 0, "swizzle term must not depend on i");
  unsigned offA[LPA], offB[LPB];
  const char* a_tile = reinterpret_cast<const char*>(g.A);  // 64-bit base of the current tile's first activation row (scalar)
  const int nk = g.K / BKE;
  // K-tiles of a tile are visited in the rotated order rot, rot+1, ..., nk-1, 0, ..., rot-1 with rot = (first column / 256) mod nk:
  // the workgroups that share an activation panel then read different K-slices (different L2 channels) at any moment
  // instead of hammering the same lines in lockstep (+5..12 % on the K = 768 shapes).  rot only depends on the
  // N-tile, so a row's result still does not depend on the batch around it; gemm_kernel uses the same order.
  int rot = 0;
  int hq_ld = HQF;   // height (64-row units) of the tile whose K-tiles are being staged, and the fragment base of a wave in it
  const int r = lane & 15, q = lane >> 4, f = (r >> 1) & 7;
  int a_base_ld = (wm * 32 * HQF + r) * ROWB;
  int cur = 0;       // number of the tile whose K-tiles are being staged
  auto tile_sources = [&](int c, int& m0, int& n0) {
    cur = c;
    tile_at(c, m0, n0, hq_ld);
    // (the staging rows are rebuilt from an opaque copy of the lane id: what is only needed here, once per tile, must not
    // stay in registers - or in scratch - across the K loops)
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    const int rin = lane_s >> 3, pc = lane_s & 7;
    const unsigned swz = (unsigned)((pc ^ ((wave * 4 + (rin >> 1)) & 7)) << 4);
    a_base_ld = (wm * 32 * hq_ld + (lane_s & 15)) * ROWB;
    if constexpr (ROT == 1 && !kPatch) rot = (n0 >> 8) % nk;  // a function of the output column block only
    // activation rows: a 64-bit tile base + 32-bit offsets inside the tile (the 4w-wide MLP rows of a 2048-frame fp32 pass are
    // 5 GB; the weight stays below 4 GiB)
    const int mb = ABL == 2 ? 0 : m0;
    if constexpr (kPatch) {
      const int img0 = mb / g.P, Gp = g.gR / g.gP;
      const int lch4 = (int)(swz >> 4) * 4;                       // first float of this lane's logical chunk inside a K-tile
      const int py_off = lch4 / g.gP, px = lch4 - py_off * g.gP;
      a_tile = reinterpret_cast<const char*>(g.A) + (size_t)img0 * 3 * g.gR * g.gR * sizeof(float);
#pragma unroll
      for (int i = 0; i < LPA; ++i) {
        const int gm = mb + min((wave + i * NW) * 8 + rin, g.M - 1 - mb);
        const int img = gm / g.P, pidx = gm - img * g.P, gy = pidx / Gp, gx = pidx - gy * Gp;
        offA[i] = (unsigned)((((img - img0) * 3 * g.gR + gy * g.gP + py_off) * g.gR + gx * g.gP + px) * (int)sizeof(float));
      }
    } else {
      a_tile = reinterpret_cast<const char*>(g.A) + (size_t)mb * ((size_t)g.lda * sizeof(T));
#pragma unroll
      for (int i = 0; i < LPA; ++i) {
        const int row = min((wave + i * NW) * 8 + rin, g.M - 1 - mb);
        offA[i] = (unsigned)row * (unsigned)(g.lda * (int)sizeof(T)) + swz;
      }
    }
#pragma unroll
    for (int i = 0; i < LPB; ++i) {
      const int row = (wave + i * NW) * 8 + rin;
      const int gr = min((ABL == 2 ? 0 : n0) + row, g.N - 1);
      offB[i] = (unsigned)gr * (unsigned)(g.ldw * (int)sizeof(T)) + swz;
    }
  };
  // byte offset of K-tile kt inside an activation row (patch gather: channel c = kt / (K-tiles per channel), first patch row)
  const int kpc_shift = kPatch ? 31 - __builtin_clz((unsigned)max(1, g.gP * g.gP / BKE)) : 0;
  auto a_koff = [&](int kt) -> unsigned {
    if constexpr (kPatch) {
      const int c = kt >> kpc_shift, py0 = (kt - (c << kpc_shift)) * (BKE / g.gP);
      return (unsigned)((c * g.gR + py0) * g.gR * (int)sizeof(float));
    } else {
      return (unsigned)kt * ROWB;
    }
  };
  // activation piece i of a wave covers rows (wave + i NW) 8 ..: unit i NW / 8 of the tile; a lower tile need not stage it.
  // SKIP is a property of the CODE that issues the piece, not of the tile: the full-height body stages every piece
  // unconditionally (no compare + branch per piece in its K loop; measured neutral either way) - also when the K-tiles belong to
  // the first tail tile (rows past its height are clamped to valid addresses and never read); only the tail body and the
  // prologue leave the unused pieces out.
  auto a_piece_staged = [&](int i, auto SKIP) { return !decltype(SKIP)::value || (i * NW) / 8 < hq_ld; };
  auto stage_load = [&](int stage, int kt, auto SKIP) {
    kt += rot;
    if (kt >= nk) kt -= nk;
    char* dst = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < LPA; ++i)
      if (a_piece_staged(i, SKIP))
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a_tile + (offA[i] + a_koff(kt))),
            (__attribute__((address_space(3))) void*)(dst + i * NW * 1024), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < LPB; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g.W) + (offB[i] + (unsigned)kt * ROWB)),
          (__attribute__((address_space(3))) void*)(dst + BM * ROWB + i * NW * 1024), 16, 0, 0);
  };
  // One LDS-DMA piece of a stage (idx < LPA: activation rows, else weight rows): lets the K loop spread the pieces of a
  // K-tile over its MFMA groups instead of issuing them in one burst.
  auto stage_piece = [&](int stage, int kt, auto IDX, auto SKIP) {
    constexpr int idx = decltype(IDX)::value;
    kt += rot;
    if (kt >= nk) kt -= nk;
    char* dst = smem + stage * STAGE + wave * 1024;
    if constexpr (idx < LPA) {
      if (a_piece_staged(idx, SKIP))
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a_tile + (offA[idx] + a_koff(kt))),
            (__attribute__((address_space(3))) void*)(dst + idx * NW * 1024), 16, 0, 0);
    } else {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(g.W) + (offB[idx - LPA] + (unsigned)kt * ROWB)),
          (__attribute__((address_space(3))) void*)(dst + BM * ROWB + (idx - LPA) * NW * 1024), 16, 0, 0);
    }
  };
  // SCHED > 0: the LDS-DMA pieces of the K-tile needed two steps ahead are not issued as one burst of LPW pieces inside
  // the hand-over; piece idx goes to slot piece_slot(SCHED, idx): -1 = still in the hand-over, u >= 0 = after MFMA
  // group u of the NEXT K-step (the stage it lands in was released by the hand-over barrier that precedes that step).
  // Each piece blocks its wave's issue port for ~100 cycles, and right after the barrier the two waves of a SIMD would
  // both be in that burst, with nobody feeding the matrix pipe.
  static_assert(SCHED == 0 || LPW == 8, "piece schedules are written for 8 pieces per wave");
  auto bias_load = [&](int buf, int n0) {  // BN floats -> LDS by one LDS-DMA of wave 0 (part of that tile's first load)
    if (!kPatch && wave == 0) {
      const float* p = g.bias + min(n0 + lane * 4, g.N - 4);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(smem + OFF_BIAS + buf * 1024), 16, 0,
                                       0);
    }
  };

  const int b_base = BM * ROWB + (wn * TN + r) * ROWB;

  int m0, n0;
  tile_sources(0, m0, n0);
  bias_load(0, n0);
  stage_load(0, 0, std::bool_constant<BAL>{});
  stage_load(1, 1, std::bool_constant<BAL>{});
  int foff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) foff[s] = ((((sizeof(T) == 2) ? (4 * s + q) : (q + 4 * s)) ^ f) * 16);
  FragT wb[2][FN], xp[2][2];
  // K-tile 0 of the first tile has landed (K-tile 1 may still be in flight: LPB + one piece per staged unit)
  if constexpr (BAL) wait_vmcnt_steps<LPB, 1, HQF>(hq_ld); else wait_vmcnt<LPW>();  // (a workgroup may own tail tiles only)
  block_barrier();
#pragma unroll
  for (int j = 0; j < FN; ++j) wb[0][j] = *reinterpret_cast<const FragT*>(smem + b_base + j * 16 * ROWB + foff[0]);
#pragma unroll
  for (int a = 0; a < 2; ++a) xp[0][a] = *reinterpret_cast<const FragT*>(smem + a_base_ld + a * 16 * ROWB + foff[0]);
  int gbase = 0;            // global K-step counter at the start of the current tile (stage = step & 1)
  int it = 0;               // tile iteration (bias buffer = it & 1)
  int prev_counted = 0;     // > 0: the previous tile issued exactly prev_counted * SPQ stores after its prefetches

  // One output tile of HQ 64-row units: accumulators from the bias, the K loop, the epilogue.  Returns false after the last tile.
  auto run_tile = [&](auto HQC) -> bool {
    constexpr int HQ = decltype(HQC)::value;
    constexpr int FMq = 2 * HQ;     // 16-row fragments per wave
    constexpr int NG = FMq;         // MFMA groups per K-tile: 2 k-substeps x HQ row-fragment pairs, 2*FN MFMAs each
    constexpr int GPS = HQ;         // groups per k-substep
    constexpr int TMq = 32 * HQ;    // rows per wave row
    constexpr std::bool_constant<(BAL && HQ < HQF)> kSkip{};   // (see a_piece_staged)
    // the accumulators start from the bias slice of this tile (in LDS since the hand-over that published K-tile 0)
    f32x4 acc[FMq][FN];
    {
      const float* biasb = reinterpret_cast<const float*>(smem + OFF_BIAS + (it & 1) * 1024) + wn * TN + 4 * q;
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const f32x4 b = kPatch ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(biasb + j * 16);
#pragma unroll
        for (int i = 0; i < FMq; ++i) acc[i][j] = b;
      }
    }

    const int cm0 = m0, cn0 = n0;
    // (lane-derived addresses are rebuilt from an opaque copy of the lane id at every tile: hipcc would otherwise hoist the
    // address arithmetic of the K loop AND of the epilogue in front of the tile loop and keep all of it alive across both)
    int lane_k = lane;
    asm volatile("" : "+v"(lane_k));
    const int a_base = (wm * TMq + (lane_k & 15)) * ROWB;
    const int tnext = cur + 1;
    const bool has_next = tnext < cur_end;

    for (int kt = 0; kt < nk; ++kt) {
      const int sidx = (gbase + kt) & 1;
      const char* st = smem + sidx * STAGE;
      const bool last = kt == nk - 1;
      static_for<NG>([&](auto U) {
        constexpr int u = decltype(U)::value;
        constexpr int s = u / GPS, p = u % GPS;
        if constexpr (u + 1 < NG) {
          // fragments of the next group (same K-tile) are requested before this group's MFMAs are issued
          constexpr int s1 = (u + 1) / GPS, p1 = (u + 1) % GPS;
          if (p1 == 0) {
#pragma unroll
            for (int j = 0; j < FN; ++j) wb[s1 & 1][j] = *reinterpret_cast<const FragT*>(st + b_base + j * 16 * ROWB + foff[s1]);
          }
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
            if (kt == 0 && prev_counted) {
              if constexpr (BAL) wait_vmcnt_steps<0, SPQ, HQF>(prev_counted); else wait_vmcnt<NST>();
            } else {
              wait_vmcnt<0>();
            }
          } else {
            wait_vmcnt<0>();
          }
          block_barrier();
          if (ABL != 1) {
            if (kt + 2 < nk) {
              static_for<LPW>([&](auto I) {
                if constexpr (piece_slot_in(SCHED, decltype(I)::value, NG) < 0) stage_piece(sidx, kt + 2, I, kSkip);
              });
            } else if (has_next) {
              if (kt + 2 == nk) {
                tile_sources(tnext, m0, n0);
                bias_load((it + 1) & 1, n0);
                // (the next tile may be of another height: its own hand-over share is decided by ITS group count, but
                // what matters here is only that every piece of its K-tile 0 is issued exactly once: the pieces that
                // do not go out here follow after this tile's groups in the last K-step, same predicate)
                static_for<LPW>([&](auto I) {
                  if constexpr (piece_slot_in(SCHED, decltype(I)::value, NG) < 0) stage_piece(sidx, 0, I, kSkip);
                });
              } else {
                stage_load(sidx, 1, kSkip);  // always a burst: it has to be older than the epilogue stores (counted vmcnt)
              }
            }
          }
         }
          // the first fragments of the next K-step, UNCONDITIONALLY (after the last step of the last tile they are never
          // used): inside the branch above they would sit in a basic block of their own, in FRONT of this group's first MFMA
          // (sched_group_barrier cannot order across blocks), and hipcc's lgkmcnt(0) for that MFMA's operands would wait
          // for their whole LDS latency with the matrix pipe idle - once per K-step
          const char* nx = smem + (sidx ^ 1) * STAGE;
          const int a_nx = (BAL && last) ? a_base_ld : a_base;   // (the next K-tile may belong to the next tile)
#pragma unroll
          for (int j = 0; j < FN; ++j) wb[0][j] = *reinterpret_cast<const FragT*>(nx + b_base + j * 16 * ROWB + foff[0]);
#pragma unroll
          for (int a = 0; a < 2; ++a) xp[0][a] = *reinterpret_cast<const FragT*>(nx + a_nx + a * 16 * ROWB + foff[0]);
        }
        if constexpr (sizeof(T) == 4) {
          // fp32: a fragment pair is FOUR chained v_mfma_f32_16x16x4_f32 on one accumulator (40-cycle dependent latency
          // against a 32-cycle issue interval).  Issue them k-slot by k-slot across the group's 2 * FN accumulators, so
          // consecutive MFMAs are independent and ONE wave can keep the pipe full while its SIMD partner waits.  Each
          // accumulator still sees its products in the same order: results are bit-identical.
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int j = 0; j < FN; ++j)
                acc[2 * p + a][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s & 1][j][t], xp[u & 1][a][t], acc[2 * p + a][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < FN; ++j) mma<T>(wb[s & 1][j], xp[u & 1][a], acc[2 * p + a][j]);
        }
        if constexpr (SCHED > 0 && ABL != 1 && u + 1 < NG) {
          constexpr bool any = piece_slot_in(SCHED, 0, NG) == u || piece_slot_in(SCHED, 1, NG) == u || piece_slot_in(SCHED, 2, NG) == u ||
                               piece_slot_in(SCHED, 3, NG) == u || piece_slot_in(SCHED, 4, NG) == u || piece_slot_in(SCHED, 5, NG) == u ||
                               piece_slot_in(SCHED, 6, NG) == u || piece_slot_in(SCHED, 7, NG) == u;
          if constexpr (any) {
            // K-tile kt+1 (or K-tile 0 of the next output tile) into the stage the previous K-step has released
            if (kt > 0 && (kt + 1 < nk || has_next)) {
              const int lk = kt + 1 < nk ? kt + 1 : 0;
              static_for<LPW>([&](auto I) {
                if constexpr (piece_slot_in(SCHED, decltype(I)::value, NG) == u) stage_piece(sidx ^ 1, lk, I, kSkip);
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
        constexpr int kReads = (u + 1 < NG) ? ((u + 1) % GPS == 0 ? FN : 0) + 2 : FN + 2;
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, kReads, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, kMfmaPerGroup - 1, 0);
      });
    }
    const bool interior = cm0 + 64 * HQ <= g.M && cn0 + BN <= g.N;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int r = lane_e & 15, q = lane_e >> 4;   // (epilogue-local copies, see above)
    if constexpr (ABL == 3) {
      float keep = 0.f;
#pragma unroll
      for (int i = 0; i < FMq; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) keep += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (keep == 123.456f) reinterpret_cast<float*>(g.C)[0] = keep;
      prev_counted = 0;
    } else {
      if constexpr (kStaged) {
        char* stg = smem + OFF_STG + wave * PATCH;
        char* wr = stg + r * ROWP + ((q & 1) << 3);
        const int rrow = lane_e / CPR, rch = lane_e % CPR;
        T* cbase = reinterpret_cast<T*>(g.C) + (size_t)(cm0 + wm * TMq + rrow) * g.ldc + cn0 + wn * TN + rch * 8;
        const bool col_ok = cn0 + wn * TN + rch * 8 < g.N;
#pragma unroll
        for (int i = 0; i < FMq; ++i) {
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
            if (interior || (cm0 + wm * TMq + i * 16 + row < g.M && col_ok)) {
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
        const int rrow = lane_e >> 3, rch = lane_e & 7;
        // EPI_RESID_F32 (C += acc + bias: the residual stream updated in place, so the LayerNorm behind the projection reads
        // ONE fp32 row instead of row + delta and writes no row back): every lane adds the 16 bytes of C it is about to
        // overwrite.  They are requested RWIN row-tiles ahead (RWIN * FN loads in flight per lane: the fragment registers
        // of the K loop are free here), whole lines per instruction like the stores, and non-temporal like them: the 1 GB
        // stream must not push the operand panels out of L2.
        constexpr int RWIN = FMq < 4 ? FMq : 4;
        f32x4 xres[EPI == EPI_RESID_F32 ? RWIN : 1][FN / 2][2];
        auto resid_load = [&](int i, f32x4 (&dst)[FN / 2][2]) {
#pragma unroll
          for (int jj = 0; jj < FN / 2; ++jj)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int mo = cm0 + wm * TMq + i * 16 + h * 8 + rrow, no = cn0 + wn * TN + jj * 32 + rch * 4;
              dst[jj][h] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (interior || (mo < g.M && no < g.N))
                dst[jj][h] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(reinterpret_cast<const TOUT*>(g.C) + (size_t)mo * g.ldc + no));
            }
        };
        if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
          for (int i = 0; i < RWIN; ++i) resid_load(i, xres[i]);
        }
#pragma unroll
        for (int i = 0; i < FMq; ++i) {
          const int m = cm0 + wm * TMq + i * 16 + r;
#pragma unroll
          for (int jj = 0; jj < FN / 2; ++jj) {
#pragma unroll
            for (int jh = 0; jh < 2; ++jh) {
              const int j = 2 * jj + jh;
              const int n = cn0 + wn * TN + j * 16 + 4 * q;
              f32x4 v = acc[i][j];
#ifdef FITCLIP_LAB
              // (tools/ only; the switch is GemmArgs::P, unused by these epilogues - api.hip.  1: the epilogue correction of an
              // algebraic LayerNorm fusion, rstd (acc - mean g) + c, on stand-in vectors out of the bias slice in LDS - what it
              // would COST; 2: no QuickGELU; 3: the plain QuickGELU, no compensated exponent)
              if constexpr (EPI == EPI_BIAS_T || EPI == EPI_GELU_T) {
                if (g.P == 1) {
                  const float* sl = reinterpret_cast<const float*>(smem + OFF_BIAS + (it & 1) * 1024);
                  const float mu = sl[(i * 16 + r) & 255] * 1e-3f, rs = 1.f + sl[(i * 16 + r + 64) & 255] * 1e-3f;
                  const f32x4 gv = *reinterpret_cast<const f32x4*>(sl + wn * TN + j * 16 + 4 * q);
                  const f32x4 cv = *reinterpret_cast<const f32x4*>(sl + ((wn * TN + j * 16 + 4 * q + 128) & 255));
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rs, __builtin_fmaf(-mu, gv[e], v[e]), cv[e]);
                }
              }
              if constexpr (EPI == EPI_GELU_T) {
                if (g.P == 3) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v[e] * -2.4554669857025146f));
                } else if (g.P != 2) {
                  v = quick_gelu_f32x4(v);
                }
              }
#else
              if constexpr (EPI == EPI_GELU_T) {
                v = quick_gelu_f32x4(v);  // (packed pairs; the bits of quick_gelu_exact)
              }
#endif
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
              const int mo = cm0 + wm * TMq + i * 16 + row, no = cn0 + wn * TN + jj * 32 + rch * 4;
              if (interior || (mo < g.M && no < g.N)) {
                size_t orow = (size_t)mo;
                if constexpr (kPatch) {  // token row of (image, patch): the image's CLS row is skipped; + positional embedding
                  const int img = mo / g.P, prow = mo - img * g.P + 1;
                  orow = (size_t)img * (g.P + 1) + prow;
                  val += *reinterpret_cast<const f32x4*>(g.aux + (size_t)prow * g.N + no);
                }
                f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<TOUT*>(g.C) + orow * g.ldc + no);
                if constexpr (ABL == 4) *dst = val;  // lab: write-back instead of non-temporal stores
                else __builtin_nontemporal_store(val, dst);
              }
            }
          }
          if constexpr (EPI == EPI_RESID_F32) {
            if (i + RWIN < FMq) resid_load(i + RWIN, xres[i % RWIN]);
          }
        }
      }
      prev_counted = interior ? HQ : 0;
    }
    if (!has_next) return false;
    gbase += nk;
    ++it;
    return true;
  };

  // the head tiles, then the tail tiles: two loops, so that each body's address arithmetic is hoisted in front of ITS loop only
  bool more = true;
  if (!BAL || hq_ld == HQF) {
    do more = run_tile(std::integral_constant<int, HQF>{});
    while (more && (!BAL || hq_ld == HQF));
  }
  if constexpr (BAL) {
    while (more) more = run_tile(std::integral_constant<int, HT>{});
  }
}

}  // namespace
}  // namespace fc

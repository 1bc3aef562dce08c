// Fused multi-head self-attention for the CLIP towers: softmax(q k^T / sqrt(64) [+ causal mask]) v, head dim 64.
// Reads q | k | v straight out of the fused-QKV GEMM output [rows, 3D] (no transposes in HBM) and writes [rows, D].
// Reference semantics: nn.MultiheadAttention inside ResidualAttentionBlock (aligner/encoder/slip.py:364-380), additive
// causal mask of the text tower (slip.py:454-460).
//
// Sequences are short (197 visual tokens, 77 text tokens), so K and V of one (sequence, head) pair fit in LDS and the
// softmax is single pass: no online rescaling, every score of a query row is in registers at once.
//
// bf16 kernels (throughput path): `attn_bf16_v2_kernel` (<= 224 tokens) and `attn_bf16_flash_kernel` (longer sequences), below.
// Common to all MFMA kernels of this file:
//   * S^T = K.Q^T is computed (keys on the accumulator rows, queries on the lanes, "swapped QK^T"): the softmax of a
//     query is then a per-lane reduction over registers plus two cross-lane steps, and the exponentiated accumulator
//     registers ARE an operand of P.V (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand");
//   * each wave owns 16-query tiles; padded keys are masked to -inf, padded V rows are zero.
//
// f32 kernel (parity path): plain VALU, one thread per query row, K/V broadcast from LDS, online softmax in fp32.
#include "common.h"

#include <algorithm>

namespace fc {

namespace {

constexpr float kNegInf = -__builtin_inff();
// 1 / sqrt(64) and log2 e in one factor on the Q fragments of the fp32 kernels that compute their softmax in the log2 domain
constexpr float kQScaleLog2 = 0.125f * 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------------------------
// bf16 kernel (second generation; what the first profile asked for: the first kernel - V transposed through registers into
// LDS, four waves - spent its time in exposed staging latency, 2-byte output stores and an unbalanced 13-tiles-over-4-waves
// split, not in MFMAs):
//   * K and V both arrive by LDS-DMA (`global_load_lds_dwordx4`, no VGPR round trip, issued once per workgroup and
//     overlapped with the Q fragment loads); both stay ROW-major [key][64 d] in LDS.  K uses the 16-byte chunk swizzle
//     of the GEMM (conflict-free ds_read_b128 of the QK^T A-operand); V uses a 32-byte granule swizzle
//     (granule ^ ((key >> 1) & 3)) that makes the transposing read below conflict-free.  The swizzles are applied to
//     the per-lane SOURCE address (the DMA destination is lane-linear).
//   * P.V is computed as O^T = V^T . P^T: the V^T A-operand (8 keys of one d per lane) is fetched with
//     `ds_read_b64_tr_b16`, the hardware transposing LDS read (cdna_hip_programming.md T10): lane 4q+p of a 16-lane
//     group supplies the address of key row q, d columns 4p..4p+3, and lane i receives column i of the four rows.
//     The P^T B-operand is the exponentiated S^T accumulator itself (keys already on the k axis).  Queries stay on
//     the lanes, so the softmax denominator is lane-local and each lane ends with 4 consecutive d of one query.
//   * the 16x64 output tile is transposed through a private 2 KiB LDS patch and stored as 8 rows x 128 B per
//     instruction; exp is a bare v_exp_f32 (scores pre-scaled by log2(e)/8).
//   * NWAVES waves per workgroup, each owning q-tiles w, w + NWAVES, ... (7 waves for 13 visual q-tiles: 93 % balance).
template <int NKT, int NWAVES, bool CAUSAL, int NFULL>
__global__ void __launch_bounds__(NWAVES * 64) attn_bf16_v2_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                  int S, int heads) {
  constexpr int NK = NKT * 16;
  constexpr int OFF_V = NK * 128;
  constexpr int OFF_O = 2 * NK * 128;  // NWAVES x 2 KiB output patches
  constexpr int MAXQ = 2;              // q-tiles per wave held in registers (Q fragments are preloaded)
  constexpr int NVMIN = (NK / 8) / NWAVES;  // V pieces EVERY wave issues (some waves issue one more)
  static_assert(NKT % 2 == 0, "P.V consumes key tiles in pairs");
  static_assert(NFULL >= 0 && NFULL <= NKT, "NFULL = key tiles known to be unmasked at compile time");
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)seq * S * ld + h * 64;
  const bf16* Kg = base + D;
  const bf16* Vg = base + 2 * D;
  const int r = lane & 15, q4 = lane >> 4, f = (r >> 1) & 7;
  const int nqt = (S + 15) >> 4;

  // ---- Q fragments of this wave's q-tiles (B-operand of S^T = K.Q^T: lane holds 8 d of query r), requested FIRST:
  // vmcnt retires in order, so Q, then K, then V become available one after the other.
  bf16x8 qf[MAXQ][2];
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int qt = wave + i * NWAVES;
    const int qrow = min(qt * 16 + r, S - 1);
    // Spelled as asm: hipcc drains vmcnt to 0 before the first use of a register loaded while LDS-DMA ("flat") pieces
    // are in flight, which would make the first softmax wait for V as well.  The registers are handed to the compiler
    // by the counted wait below (its "+v" operands), never before.
#pragma unroll
    for (int s = 0; s < 2; ++s)
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qf[i][s]) : "v"(base + (long)qrow * ld + (4 * s + q4) * 8) : "memory");
  }
  // ---- stage K, then V, by LDS-DMA: one wave-instruction = 8 key rows x 128 B.  Fully unrolled with a compile-time
  // count of unconditional pieces, so that hipcc's own wait for the Q registers stays a counted vmcnt.
  {
    const int rin = lane >> 3, pc = lane & 7;
    constexpr int NPIECE = (NK / 8 + NWAVES - 1) / NWAVES;
#pragma unroll
    for (int isv = 0; isv < 2; ++isv) {
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) {
        const int grp = wave + j * NWAVES;
        if (j < NVMIN || grp < NK / 8) {
          const int row = grp * 8 + rin;
          const int srow = min(row, S - 1);  // padded keys read a valid row; they are masked / multiplied by P = 0
          const int c = isv ? ((((pc >> 1) ^ ((row >> 1) & 3)) << 1) | (pc & 1)) : (pc ^ ((row >> 1) & 7));
          const bf16* src = (isv ? Vg : Kg) + (long)srow * ld + c * 8;
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)src,
              (__attribute__((address_space(3))) void*)(smem + (isv ? OFF_V : 0) + grp * 1024), 16, 0, 0);
        }
      }
    }
  }
  // Q and K have landed once at most the V pieces are outstanding; V is only needed after the first softmax.
  static_assert(MAXQ == 2, "the wait below names every Q fragment");
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[1][0]), "+v"(qf[1][1]) : "n"(NVMIN) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const float kScale = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
  char* patch = smem + OFF_O + wave * 2048;
#pragma unroll
  for (int i = 0; i < MAXQ; ++i) {
    const int qt = wave + i * NWAVES;
    const bool active = qt < nqt;  // wave-uniform
    const int query = qt * 16 + r;
    f32x4 sT[NKT];
    float inv = 0.f;
    if (active) {
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const bool dead = (t >= NFULL && t * 16 >= S) || (CAUSAL && t > qt);  // every key of the tile is masked
        if (!dead) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(smem + (t * 16 + r) * 128 + (((4 * s + q4) ^ f) << 4));
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[i][s], acc, 0, 0, 0);
          }
        }
        sT[t] = acc;
      }
      // mask (only tiles that can contain masked keys) and row maximum of the RAW scores (the scale is positive)
      float mx = kNegInf;
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        if (CAUSAL || t >= NFULL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int key = t * 16 + 4 * q4 + e;
            if (key >= S || (CAUSAL && key > query)) sT[t][e] = kNegInf;
          }
        }
        // v_max3_f32 spelled out: `fmaxf` on raw MFMA results makes hipcc canonicalise every input first (IEEE mode),
        // three times the instructions
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(sT[t][0]), "v"(sT[t][1]));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(sT[t][2]), "v"(sT[t][3]));
      }
      mx = max_over_lane_groups(mx);
      // p = exp2(s * c - mx * c): one packed fma per two scores, bare v_exp_f32, packed running sum
      const f32x2 c2 = {kScale, kScale}, nm2 = {-mx * kScale, -mx * kScale};
      f32x2 sum2 = {0.f, 0.f};
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        f32x2 a = {sT[t][0], sT[t][1]}, b = {sT[t][2], sT[t][3]};
        a = a * c2 + nm2;
        b = b * c2 + nm2;
        a = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        b = f32x2{__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
        sum2 += a;
        sum2 += b;
        sT[t] = f32x4{a[0], a[1], b[0], b[1]};
      }
      inv = __builtin_amdgcn_rcpf(sum_over_lane_groups(sum2[0] + sum2[1]));
    }
    if (i == 0) {  // V has landed (every wave passes here exactly once, active or not)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    if (!active) continue;

    // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // transposing-read addresses: lane 4q+p of its 16-lane group -> key row (.. + q), d columns 16n + 4p .. +3
    const int tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
    for (int ks = 0; ks < NKT / 2; ++ks) {
      if (CAUSAL && 2 * ks > qt) continue;
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = static_cast<bf16>(sT[2 * ks][e]);
        pf[4 + e] = static_cast<bf16>(sT[2 * ks + 1][e]);
      }
      const int key0 = ks * 32 + 4 * q4 + tq;   // rows of the first read (k-slots j = 0..3), +16 for the second
      const int sw0 = (key0 >> 1) & 3, sw1 = ((key0 + 16) >> 1) & 3;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(smem + OFF_V + key0 * 128 + ((n ^ sw0) << 5) + tp * 8));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(smem + OFF_V + (key0 + 16) * 128 + ((n ^ sw1) << 5) + tp * 8));
        const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[n], 0, 0, 0);
      }
    }
    // o[n][e] = O(query r, d = 16n + 4 q4 + e): pack, transpose through the wave's LDS patch, store full 128-B rows
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      bf16x4 pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) pk[e] = static_cast<bf16>(o[n][e] * inv);
      *reinterpret_cast<bf16x4*>(patch + r * 128 + (((n * 4 + q4) ^ ((r & 7) << 1)) << 3)) = pk;
    }
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int row = hh * 8 + (lane >> 3), ch = lane & 7;
      const bf16x8 val = *reinterpret_cast<const bf16x8*>(patch + row * 128 + (((2 * ch) ^ ((row & 7) << 1)) << 3));
      const int qo = qt * 16 + row;
      if (qo < S) *reinterpret_cast<bf16x8*>(out + ((long)seq * S + qo) * D + h * 64 + ch * 8) = val;
    }
  }
}

// ---- long sequences (S > 224: ViT-L/14 has 257 tokens, ViT-L/14@336 577): K/V do not fit LDS next to a second
// workgroup, so they stream through a 2-deep ring of 64-key tiles with an online softmax (running max / sum per query,
// accumulator rescaled when the max moves).  Same operand layouts, swizzles and MFMA shapes as attn_bf16_v2_kernel; one
// 16-query tile per wave, NW waves per workgroup, grid = sequences x heads x query blocks.  Non-causal only (CLIP's
// text context is 77).  Per K/V tile: wait for its DMA, barrier, issue the next tile's DMA (lands under this tile's
// math), S^T = K.Q^T, mask, softmax update, O^T += V^T.P^T.
template <int NW>
__global__ void __launch_bounds__(NW * 64) attn_bf16_flash_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                                 int S, int heads, int nqb) {
  constexpr int TK = 64;                 // keys per tile
  constexpr int STAGE = 2 * TK * 128;    // K tile + V tile
  constexpr int OFF_O = 2 * STAGE;
  constexpr int NPIECE = (2 * TK / 8 + NW - 1) / NW;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qb = blockIdx.x % nqb, sh = blockIdx.x / nqb;
  const int seq = sh / heads, h = sh - seq * heads;
  const int D = heads * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)seq * S * ld + h * 64;
  const bf16* Kg = base + D;
  const bf16* Vg = base + 2 * D;
  const int r = lane & 15, q4 = lane >> 4, f = (r >> 1) & 7;
  const int nqt = (S + 15) >> 4, nkt = (S + TK - 1) / TK;
  const int qt = qb * NW + wave;
  const bool active = qt < nqt;  // wave-uniform; inactive waves still stage tiles and take every barrier

  bf16x8 qf[2];
  {
    const int qrow = min(qt * 16 + r, S - 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(base + (long)qrow * ld + (4 * s + q4) * 8);
  }
  auto stage_tile = [&](int kt) {
    const int rin = lane >> 3, pc = lane & 7;
    char* dst = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int grp = wave + j * NW;  // 0..7: K pieces, 8..15: V pieces
      if (grp < 2 * TK / 8) {
        const bool isv = grp >= TK / 8;
        const int row = (isv ? grp - TK / 8 : grp) * 8 + rin;        // key inside the tile
        const int srow = min(kt * TK + row, S - 1);                   // padded keys read a valid row (masked below)
        const int c = isv ? ((((pc >> 1) ^ ((row >> 1) & 3)) << 1) | (pc & 1)) : (pc ^ ((row >> 1) & 7));
        const bf16* src = (isv ? Vg : Kg) + (long)srow * ld + c * 8;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + grp * 1024), 16, 0, 0);
      }
    }
  };
  stage_tile(0);

  const float kScale = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
  const f32x2 c2 = {kScale, kScale};
  float m = kNegInf, l = 0.f;  // running max of the raw scores / running sum, for query r (replicated over q4)
  f32x4 o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tq = (lane >> 2) & 3, tp = lane & 3;

  for (int kt = 0; kt < nkt; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of tile kt
    __syncthreads();                                   // everyone's pieces; and tile kt-1 is no longer being read
    if (kt + 1 < nkt) stage_tile(kt + 1);
    if (!active) continue;
    const char* Ks = smem + (kt & 1) * STAGE;
    const char* Vs = Ks + TK * 128;
    f32x4 sT[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + r) * 128 + (((4 * s + q4) ^ f) << 4));
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc, 0, 0, 0);
      }
      sT[t] = acc;
    }
    float mx = m;
    const bool tail = (kt + 1) * TK > S;  // uniform: only the last tile can hold padded keys
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (tail) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kt * TK + t * 16 + 4 * q4 + e >= S) sT[t][e] = kNegInf;
      }
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(sT[t][0]), "v"(sT[t][1]));
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(sT[t][2]), "v"(sT[t][3]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));  // >= m, finite from the first tile on (key 0 is never masked)
    const float alpha = __builtin_amdgcn_exp2f((m - mx) * kScale);  // exp2(-inf) = 0 on the first tile
    m = mx;
    const f32x2 nm2 = {-mx * kScale, -mx * kScale};
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f32x2 a = {sT[t][0], sT[t][1]}, b = {sT[t][2], sT[t][3]};
      a = a * c2 + nm2;
      b = b * c2 + nm2;
      a = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
      b = f32x2{__builtin_amdgcn_exp2f(b[0]), __builtin_amdgcn_exp2f(b[1])};
      sum2 += a;
      sum2 += b;
      sT[t] = f32x4{a[0], a[1], b[0], b[1]};
    }
    float sum = sum2[0] + sum2[1];
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    l = l * alpha + sum;
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] *= alpha;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = static_cast<bf16>(sT[2 * ks][e]);
        pf[4 + e] = static_cast<bf16>(sT[2 * ks + 1][e]);
      }
      const int key0 = ks * 32 + 4 * q4 + tq;
      const int sw0 = (key0 >> 1) & 3, sw1 = ((key0 + 16) >> 1) & 3;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(Vs + key0 * 128 + ((n ^ sw0) << 5) + tp * 8));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(Vs + (key0 + 16) * 128 + ((n ^ sw1) << 5) + tp * 8));
        const bf16x8 vf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[n], 0, 0, 0);
      }
    }
  }
  if (!active) return;
  const float inv = __builtin_amdgcn_rcpf(l);
  char* patch = smem + OFF_O + wave * 2048;
#pragma unroll
  for (int n = 0; n < 4; ++n) {
    bf16x4 pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = static_cast<bf16>(o[n][e] * inv);
    *reinterpret_cast<bf16x4*>(patch + r * 128 + (((n * 4 + q4) ^ ((r & 7) << 1)) << 3)) = pk;
  }
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int row = hh * 8 + (lane >> 3), ch = lane & 7;
    const bf16x8 val = *reinterpret_cast<const bf16x8*>(patch + row * 128 + (((2 * ch) ^ ((row & 7) << 1)) << 3));
    const int qo = qt * 16 + row;
    if (qo < S) *reinterpret_cast<bf16x8*>(out + ((long)seq * S + qo) * D + h * 64 + ch * 8) = val;
  }
}

// ---- exact-fp32 attention on the fp32-input matrix cores (`v_mfma_f32_16x16x4_f32`: fp32 products, fp32 accumulate),
// for sequences of up to NKT * 16 tokens; same structure as attn_bf16_v2_kernel with fp32 operands:
//   * K and V stay row-major [key][64 floats] (256-byte rows) in LDS, filled by LDS-DMA (a piece = 4 rows); the 16-byte
//     chunk index of a row is XORed with (key & 15) on the source side.
//   * S^T = K.Q^T: lane (r, g) reads K[key r][16 c + 4 g .. +3] with ONE ds_read_b128 per chunk c and feeds the four
//     floats to four MFMAs (k index g <-> d = 16 c + 4 g + e); the Q fragments are read from global the same way, so the
//     k permutation is the same on both operands.  Conflict-free under the XOR above.
//   * P.V as O^T = V^T.P^T: the S^T accumulator register e of key tile t is the B operand of MFMA e (k index g <-> key
//     16 t + 4 g + e); its A operand V[that key][16 n + r] is one ds_read_b32 (conflict-free as well).
//   * softmax in registers with the accurate expf; each lane ends with 4 consecutive d of one query: float4 stores.
// One 16-query tile per wave, NKT waves.
template <int NKT, int NW, bool CAUSAL>
__global__ void __launch_bounds__(NW * 64) attn_f32_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                int S, int heads) {
  constexpr int NK = NKT * 16;
  constexpr int OFF_V = NK * 256;
  constexpr int NPIECE = (NK / 4 + NW - 1) / NW;  // 4-row pieces per wave, for K and again for V
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const long ld = 3L * D;
  const float* base = qkv + (long)seq * S * ld + h * 64;
  const int r = lane & 15, g = lane >> 4;
  const int nqt = (S + 15) >> 4;
  {  // stage K, then V: lane l of a piece -> row l >> 4, chunk l & 15
    const int prow = lane >> 4, pch = lane & 15;
#pragma unroll
    for (int isv = 0; isv < 2; ++isv) {
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) {
        const int piece = wave + j * NW;
        if (piece < NK / 4) {
          const int row = piece * 4 + prow;
          const int srow = min(row, S - 1);  // padded keys read a valid row; they are masked / multiplied by P = 0
          const float* src = base + (isv ? 2 * D : D) + (long)srow * ld + ((pch ^ (row & 15)) << 2);
          __builtin_amdgcn_global_load_lds(
              (const __attribute__((address_space(1))) void*)src,
              (__attribute__((address_space(3))) void*)(smem + (isv ? OFF_V : 0) + piece * 1024), 16, 0, 0);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  constexpr int QPW = (NKT + NW - 1) / NW;  // 16-query tiles per wave (1 when NW == NKT)
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) {
  const int qt = wave + qi * NW;
  if (qt >= nqt) break;
  // Q fragments: lane (r, g) holds Q[query r][16 c + 4 g .. +3], pre-scaled by 1/sqrt(64) (exact: a power of two)
  f32x4 qf[4];
  {
    const float* qrow = base + (long)min(qt * 16 + r, S - 1) * ld + 4 * g;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const f32x4*>(qrow + 16 * c) * 0.125f;
  }
  const int query = qt * 16 + r;
  f32x4 sT[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (t * 16 < S && (!CAUSAL || t <= qt)) {
      const char* krow = smem + (t * 16 + r) * 256;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 kf = *reinterpret_cast<const f32x4*>(krow + (((4 * c + g) ^ r) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[c][e], acc, 0, 0, 0);
      }
    }
    sT[t] = acc;
  }
  float mx = kNegInf;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = t * 16 + 4 * g + e;
      if (key >= S || (CAUSAL && key > query)) sT[t][e] = kNegInf;
      mx = fmaxf(mx, sT[t][e]);
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float p = exp_neg_f32(sT[t][e] - mx);  // common.h: compensated v_exp_f32, exact-fp32 grade
      sT[t][e] = p;
      sum += p;
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.f / sum;

  f32x4 o[4];
#pragma unroll
  for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    if (t * 16 >= S || (CAUSAL && t > qt)) continue;  // every P of the tile is zero
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int key = t * 16 + 4 * g + e;               // key & 15 = 4 g + e
      const char* vrow = smem + OFF_V + key * 256 + (r & 3) * 4;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float vf = *reinterpret_cast<const float*>(vrow + (((4 * n + (r >> 2)) ^ (4 * g + e)) << 4));
        o[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, sT[t][e], o[n], 0, 0, 0);
      }
    }
  }
  // o[n][e] = O(query r, d = 16 n + 4 g + e)
  if (query < S) {
    float* orow = out + ((long)seq * S + query) * D + h * 64 + 4 * g;
#pragma unroll
    for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(orow + 16 * n) = o[n] * inv;
  }
  }  // q-tile loop
}

// ---- exact-fp32 attention for 97..224 tokens (the ViT's 197), second generation.  attn_f32_mfma_kernel keeps K and V of a
// whole (sequence, head) in LDS: 100 KB -> ONE workgroup per CU, so the staging latency of every workgroup is exposed and
// 13 query tiles on 14 waves leave one SIMD with 4 tiles and three with 3 (measured 53 % of the MFMA issue time).  Here the
// keys STREAM through LDS in blocks of 64 (4 key tiles, K + V = 32 KB), double-buffered: block b + 1 is on its way (LDS-DMA)
// while block b is multiplied, one barrier per block, and at 66 KB two workgroups of 8 waves share a CU (16 waves: four
// per SIMD).  Each wave keeps Q, the running maximum, sum and O^T of its (up to two) query tiles in registers
// across the blocks (online softmax: later blocks rescale by exp(m_old - m_new)); operand layouts, swizzles and MFMA
// order are those of attn_f32_mfma_kernel.
// ABL (tools/attn_lab_f32.hip only): 1 = no exponentials, 2 = V operand not read from LDS, 3 = K operand not read from
// LDS, 4 = no staging, 5 = no P.V products, 6 = no S products.  (Per-wave s_setprio to break the phase lock of the
// waves of a SIMD: measured +1 %, not kept.)
// X3: `out` is written as x3 rows [rows, 4 D bf16 positions] of the fp32 result (split-fp32 mode: out_proj's A operand).
template <int NW, int ABL = 0, bool X3 = false>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4)))  // <= 128 VGPRs: two workgroups per CU
attn_f32_blocks_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S, int heads) {
  constexpr int BT = 4, BK = BT * 16;          // key tiles / keys per block
  constexpr int OFF_V = BK * 256, VPIECE = 1024 + 64, BUF = OFF_V + (BK / 4) * VPIECE;
  constexpr int NPIECE = (2 * BK / 4 + NW - 1) / NW;   // 1 KiB pieces (4 rows of K or V) per wave and block
  constexpr int QPW = 2;                        // query tiles per wave (13 tiles on 8 waves)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const long ld = 3L * D;
  const float* base = qkv + (long)seq * S * ld + h * 64;
  const int r = lane & 15, g = lane >> 4;
  const int nqt = (S + 15) >> 4;
  const int nblk = (S + BK - 1) / BK;
  // query tiles of this wave: `wave` and `NW + wave` (13 tiles on 8 waves: waves 0..4 carry two).  Handing the second
  // tiles to different waves in alternate workgroups, so that two workgroups on a CU load different SIMDs, measured
  // nothing (tools/attn_lab_f32 history); what matters is 8 waves = two per SIMD from every workgroup (7 waves: +13 %).
  const int qtile[2] = {wave, NW + wave};

  // K: 1 KiB pieces of 4 keys, chunk c of key k stored at chunk c ^ (k & 15) (b128 fragment reads of 16 keys x one chunk
  // are conflict-free).  V: pieces of 4 keys at a stride of 1024 + 64 bytes, rows unpermuted: the four lane groups of a
  // P.V operand read (keys 4 g + e: row e of four consecutive pieces) land in four different quarter-banks, and the whole
  // address is one lane-dependent base plus compile-time offsets.
  // The source of a piece is a wave-uniform address (item base + blk * BK rows + 32 rows for the wave's second K / V piece) plus
  // a per-lane offset that is the same in every block (row = 4 wave + lane / 16 and its chunk swizzle; rows 32 apart share the
  // swizzle): two offsets (K, V) per block and a piece costs one 64-bit add + one global_load_lds - address arithmetic
  // between MFMAs is not free on the fp32 pipe (see the softmax below).  Only the block that straddles S clamps its rows.
  static_assert(NW * NPIECE == 2 * BK / 4 && NW * 4 == BK / 2, "piece j of a wave: K rows 4 wave .., + 32, then V rows likewise");
  auto stage = [&](int blk) {  // keys [blk * BK, blk * BK + BK): lane l of a 4-row piece -> row l >> 4, chunk l & 15
    char* dst = smem + (blk & 1) * BUF;
    if (blk * BK + BK <= S) {  // wave-uniform
      int lane_s = lane;
      asm volatile("" : "+v"(lane_s));  // (opaque: a dozen instructions per block instead of four registers through the kernel)
      const int srow0 = wave * 4 + (lane_s >> 4);
      const unsigned koff0 = (unsigned)((D + srow0 * (int)ld + (((lane_s & 15) ^ (srow0 & 15)) << 2)) * 4);
      const unsigned voff0 = (unsigned)((2 * D + srow0 * (int)ld + ((lane_s & 15) << 2)) * 4);
      const char* bbase = reinterpret_cast<const char*>(base + (long)blk * BK * ld);  // wave-uniform
      long half = 32 * 4 * ld;  // bytes from a wave's first K / V piece to its second
      asm volatile("" : "+s"(half));  // (a scalar add on the base, not a second pair of per-lane offsets)
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) {
        const int isv = j >> 1, piece = wave + (j & 1) * NW;  // pieces wave, wave + 8 of K, then of V
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bbase + (j & 1) * half + (isv ? voff0 : koff0)),
                                         (__attribute__((address_space(3))) void*)(dst + (isv ? OFF_V + piece * VPIECE : piece * 1024)),
                                         16, 0, 0);
      }
    } else {
      const int prow = lane >> 4, pch = lane & 15;
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) {
        const int isv = j >> 1, piece = wave + (j & 1) * NW;
        const int row = piece * 4 + prow;
        const int srow = min(blk * BK + row, S - 1);  // padded keys read a valid row; they are masked
        const float* src = base + (isv ? 2 * D : D) + (long)srow * ld + ((isv ? pch : pch ^ (row & 15)) << 2);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + (isv ? OFF_V + piece * VPIECE : piece * 1024)),
                                         16, 0, 0);
      }
    }
  };

  if (ABL != 4) stage(0);

  // lane-dependent parts of the LDS addresses: K fragment (S^T A-operand) row r, chunk (4 c + g) ^ r;
  // V element (P.V A-operand) key 4 g + e of the tile, float 16 n + r: piece g of the tile, row e
  int koff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) koff[c] = r * 256 + (((4 * c + g) ^ r) << 4);
  const int voff = OFF_V + g * VPIECE + r * 4;

  f32x4 o[QPW][4], qf[QPW][4];
  float mrun[QPW], lrun[QPW];
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) {
    mrun[qi] = kNegInf;
    lrun[qi] = 0.f;
    const float* qrow = base + (long)min(qtile[qi] * 16 + r, S - 1) * ld + 4 * g;
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[qi][c] = *reinterpret_cast<const f32x4*>(qrow + 16 * c) * kQScaleLog2;
#pragma unroll
    for (int n = 0; n < 4; ++n) o[qi][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int blk = 0; blk < nblk; ++blk) {
    if (ABL != 4) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of block blk (issued a whole block ago)
      __syncthreads();  // block blk is complete, and nobody reads the other buffer any more
      if (blk + 1 < nblk) stage(blk + 1);
    }
    const char* kv = smem + (blk & 1) * BUF;
    const char* kfrag[4] = {kv + koff[0], kv + koff[1], kv + koff[2], kv + koff[3]};  // one add per block, not per read
    const char* vfrag = kv + voff;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      const int qt = qtile[qi];
      if (qt < nqt) {
        f32x4 sT[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          if (blk * BK + t * 16 < S) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const f32x4 kf = ABL == 3 ? qf[qi][(c + t) & 3] : *reinterpret_cast<const f32x4*>(kfrag[c] + t * 4096);
              if (ABL == 6) { acc += kf * qf[qi][c]; continue; }
#pragma unroll
              for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[qi][c][e], acc, 0, 0, 0);
            }
          }
          sT[t] = acc;
        }
        // softmax in the log2 domain (the scores carry log2 e / 8 from the Q fragments): p = 2^(s - max) is ONE v_exp_f32 on an
        // exact difference.  VALU instructions do not hide under fp32 MFMAs (tools/mfma_f32_chain_lab: both run on the SIMD's
        // fp32 lanes, every v_* between two MFMAs adds its 2 - 7 cycles to the 32 of the MFMA), so what counts here is their
        // NUMBER: v_max3, packed subtract / add / multiply, no compensated exponential (its job - the rounding of s log2 e - is
        // done by the scale of Q), masked scores are -inf and 2^-inf = 0 without a select.
        float mx = mrun[qi];
#pragma unroll
        for (int t = 0; t < BT; ++t) {
          if (blk * BK + t * 16 + 16 > S) {  // wave-uniform: only the tile that straddles S (and the ones past it) mask
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (blk * BK + t * 16 + 4 * g + e >= S) sT[t][e] = kNegInf;
          }
          mx = fmaxf(fmaxf(mx, sT[t][0]), sT[t][1]);   // v_max3_f32
          mx = fmaxf(fmaxf(mx, sT[t][2]), sT[t][3]);
        }
        mx = max_over_lane_groups(mx);            // finite from the first block on (key 0 is never masked)
        const float alpha = __builtin_amdgcn_exp2f(mrun[qi] - mx);  // 2^-inf = 0 on the first block
        mrun[qi] = mx;
        f32x2 sum2 = {0.f, 0.f};
#pragma unroll
        for (int t = 0; t < BT; ++t) {
          const f32x2 lo = f32x2{sT[t][0], sT[t][1]} - mx, hi = f32x2{sT[t][2], sT[t][3]} - mx;  // v_pk_add_f32
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            sT[t][e] = ABL == 1 ? lo[e] : __builtin_amdgcn_exp2f(lo[e]);
            sT[t][2 + e] = ABL == 1 ? hi[e] : __builtin_amdgcn_exp2f(hi[e]);
          }
          sum2 += f32x2{sT[t][0], sT[t][1]};
          sum2 += f32x2{sT[t][2], sT[t][3]};
        }
        lrun[qi] = lrun[qi] * alpha + (sum2[0] + sum2[1]);  // per lane group; the four groups are added at the end
#pragma unroll
        for (int n = 0; n < 4; ++n) o[qi][n] *= alpha;
#pragma unroll
        for (int t = 0; t < BT; ++t) {
          if (blk * BK + t * 16 >= S) continue;  // every P of the tile is zero
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
              const float vf = ABL == 2 ? qf[qi][n][e] : *reinterpret_cast<const float*>(vfrag + t * 4 * VPIECE + e * 256 + n * 64);
              if (ABL == 5) { o[qi][n][e] += vf * sT[t][e]; continue; }
              o[qi][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, sT[t][e], o[qi][n], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  if constexpr (X3) {
    // x3 rows (common.h): the 64 columns of a head are four 128-byte lines [p1 | p2 | p3 | unused] per query row.  Each wave
    // stages the planes of a 16-row x 32-column patch (two lines per row: 16 x 192 bytes, rows 208 bytes apart against
    // bank conflicts) in its own 4 KiB of the (now idle) K/V buffers and writes them out as WHOLE lines, 16 bytes per lane:
    // chunk x = 16 row + 8 line + c lands at byte 16 c of that line (c = 6, 7: zeros - a partly written line costs a
    // read-modify-write at the memory side).
    __syncthreads();  // every wave is done reading K / V
    char* stg = smem + wave * 4096;
    constexpr int ROWS = 208;
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      const float inv = 1.f / sum_over_lane_groups(lrun[qi]);
      const int q0 = qtile[qi] * 16;
      if (q0 >= S) continue;  // wave-uniform
#pragma unroll
      for (int gs = 0; gs < 2; ++gs) {
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          bf16x4 p1, p2, p3;
          split3(o[qi][2 * gs + nh] * inv, p1, p2, p3);
          char* w0 = stg + r * ROWS + nh * 96 + g * 8;  // lane (r, g): columns 4 g .. 4 g + 3 of 16-column block 2 gs + nh
          *reinterpret_cast<bf16x4*>(w0) = p1;
          *reinterpret_cast<bf16x4*>(w0 + 32) = p2;
          *reinterpret_cast<bf16x4*>(w0 + 64) = p3;
        }
        char* obase = reinterpret_cast<char*>(out) + ((long)seq * S + q0) * ((long)D * 8) + (long)(h * 4 + gs * 2) * X3_GROUP_BYTES;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = it * 64 + lane, row = idx >> 4, line = (idx >> 3) & 1, c = idx & 7;  // 16 chunks of 16 bytes per row
          bf16x8 val = {};
          if (c < 6) val = *reinterpret_cast<const bf16x8*>(stg + row * ROWS + (line * 6 + c) * 16);
          if (q0 + row < S) *reinterpret_cast<bf16x8*>(obase + (long)row * ((long)D * 8) + line * X3_GROUP_BYTES + c * 16) = val;
        }
      }
    }
  } else {
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));  // (opaque: the output addresses are computed here, not kept in registers through the blocks)
      const int query = qtile[qi] * 16 + (lane_o & 15);
      const float inv = 1.f / sum_over_lane_groups(lrun[qi]);
      if (query < S) {
        float* orow = out + ((long)seq * S + query) * D + h * 64 + 4 * (lane_o >> 4);
#pragma unroll
        for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(orow + 16 * n) = o[qi][n] * inv;
      }
    }
  }
}

// f32 parity kernel: thread per query (256 queries per workgroup), K/V rows broadcast from LDS in chunks of `kc` keys
// (any sequence length; one chunk up to 256 keys), online softmax in key order.
template <bool CAUSAL>
__global__ void __launch_bounds__(256) attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S,
                                                       int heads, int kc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = reinterpret_cast<float*>(smem);
  float* Vs = Ks + kc * 64;
  const int tid = threadIdx.x;
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const long ld = 3L * D;
  const float* base = qkv + (long)seq * S * ld + h * 64;
  const int query = blockIdx.y * 256 + tid;
  const bool valid = query < S;
  float qv[64], o[64];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const f32x4 t = valid ? *reinterpret_cast<const f32x4*>(base + query * ld + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qv[c * 4 + e] = t[e] * 0.125f;
      o[c * 4 + e] = 0.f;
    }
  }
  float m = kNegInf, l = 0.f;
  const int kend = CAUSAL ? query + 1 : S;
  for (int k0 = 0; k0 < S; k0 += kc) {
    const int kn = min(kc, S - k0);
    __syncthreads();  // the previous chunk is no longer being read
    for (int idx = tid; idx < kn * 16; idx += 256) {
      const int key = idx >> 4, c = (idx & 15) * 4;
      *reinterpret_cast<f32x4*>(Ks + key * 64 + c) = *reinterpret_cast<const f32x4*>(base + D + (k0 + key) * ld + c);
      *reinterpret_cast<f32x4*>(Vs + key * 64 + c) = *reinterpret_cast<const f32x4*>(base + 2 * D + (k0 + key) * ld + c);
    }
    __syncthreads();
    const int stop = valid ? min(kn, kend - k0) : 0;
    for (int key = 0; key < stop; ++key) {
      const float* kr = Ks + key * 64;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + c * 4);
        s0 = fmaf(qv[c * 4 + 0], kv[0], s0);
        s1 = fmaf(qv[c * 4 + 1], kv[1], s1);
        s2 = fmaf(qv[c * 4 + 2], kv[2], s2);
        s3 = fmaf(qv[c * 4 + 3], kv[3], s3);
      }
      const float sc = (s0 + s1) + (s2 + s3);
      const float mn = fmaxf(m, sc);
      const float a = expf(m - mn);  // exp(-inf) = 0 on the first key
      const float p = expf(sc - mn);
      l = l * a + p;
      m = mn;
      const float* vr = Vs + key * 64;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + c * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[c * 4 + e] = fmaf(p, vv[e], o[c * 4 + e] * a);
      }
    }
  }
  if (!valid) return;
  const float inv = 1.f / l;
  float* orow = out + ((long)seq * S + query) * D + h * 64;
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = o[c * 4 + e] * inv;
    *reinterpret_cast<f32x4*>(orow + c * 4) = t;
  }
}

template <int NKT, int NWAVES, bool CAUSAL, int NFULL>
int launch_bf16_v2_variant(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t st) {
  constexpr int lds = 2 * NKT * 16 * 128 + NWAVES * 2048;
  if (lds > 64 * 1024 &&
      raise_dynamic_lds((const void*)attn_bf16_v2_kernel<NKT, NWAVES, CAUSAL, NFULL>, lds) != hipSuccess)
    return fail(FC_ELAUNCH, "attention(bf16): cannot raise dynamic LDS");
  hipLaunchKernelGGL((attn_bf16_v2_kernel<NKT, NWAVES, CAUSAL, NFULL>), dim3(n_seq * heads), dim3(NWAVES * 64), lds, st,
                     (const bf16*)qkv, (bf16*)out, S, heads);
  FC_CHECK_LAUNCH("attention(bf16 v2)");
  return FC_OK;
}

template <int NKT, int NWAVES>
int launch_bf16_v2(const void* qkv, void* out, int n_seq, int S, int heads, int causal, hipStream_t st) {
  if ((S + 15) / 16 > 2 * NWAVES) return fail(FC_EINVAL, "attention(bf16): %d query tiles exceed the wave plan", (S + 15) / 16);
  if (causal) return launch_bf16_v2_variant<NKT, NWAVES, true, 0>(qkv, out, n_seq, S, heads, st);
  // sequences that need this key-tile count for real (the ViT: 197 tokens in 14 tiles) only mask the last two tiles
  if (S > (NKT - 2) * 16) return launch_bf16_v2_variant<NKT, NWAVES, false, NKT - 2>(qkv, out, n_seq, S, heads, st);
  return launch_bf16_v2_variant<NKT, NWAVES, false, 0>(qkv, out, n_seq, S, heads, st);
}

template <int NW>
int launch_bf16_flash_nw(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t st) {
  constexpr int lds = 2 * 2 * 64 * 128 + NW * 2048;
  const int nqb = ((S + 15) / 16 + NW - 1) / NW;
  const long blocks = (long)n_seq * heads * nqb;
  if (blocks > 0x7fffffffL) return fail(FC_EINVAL, "attention(bf16): grid of %ld workgroups", blocks);
  hipLaunchKernelGGL((attn_bf16_flash_kernel<NW>), dim3((unsigned)blocks), dim3(NW * 64), lds, st, (const bf16*)qkv,
                     (bf16*)out, S, heads, nqb);
  FC_CHECK_LAUNCH("attention(bf16 flash)");
  return FC_OK;
}

// waves per workgroup = 16-query tiles per query block: the choice that wastes the fewest padded tiles (257 tokens =
// 17 tiles -> 3 blocks of 6; 577 tokens = 37 tiles -> 5 blocks of 8 would pad 3, 7 blocks of 6 pads 5, 10 of 4 pads 3)
int launch_bf16_flash(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t st) {
  const int nqt = (S + 15) / 16;
  int best = 8, waste = 1 << 30;
  for (int nw : {8, 6, 4}) {
    const int w = (nqt + nw - 1) / nw * nw - nqt;
    if (w < waste) { waste = w; best = nw; }
  }
  if (best == 8) return launch_bf16_flash_nw<8>(qkv, out, n_seq, S, heads, st);
  if (best == 6) return launch_bf16_flash_nw<6>(qkv, out, n_seq, S, heads, st);
  return launch_bf16_flash_nw<4>(qkv, out, n_seq, S, heads, st);
}

template <int NKT, int NW, bool CAUSAL>
int launch_f32_mfma(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t st) {
  constexpr int lds = 2 * NKT * 16 * 256;
  if (lds > 64 * 1024 && raise_dynamic_lds((const void*)attn_f32_mfma_kernel<NKT, NW, CAUSAL>, lds) != hipSuccess)
    return fail(FC_ELAUNCH, "attention(f32 mfma): cannot raise dynamic LDS");
  hipLaunchKernelGGL((attn_f32_mfma_kernel<NKT, NW, CAUSAL>), dim3(n_seq * heads), dim3(NW * 64), lds, st,
                     (const float*)qkv, (float*)out, S, heads);
  FC_CHECK_LAUNCH("attention(f32 mfma)");
  return FC_OK;
}

int launch_f32_blocks(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t st, bool x3 = false) {
  constexpr int NW = 8, lds = 2 * (64 * 256 + 16 * (1024 + 64));  // two 33 KiB buffers (64 keys of K and V): two workgroups per CU
  if (x3)
    hipLaunchKernelGGL((attn_f32_blocks_kernel<NW, 0, true>), dim3(n_seq * heads), dim3(NW * 64), lds, st,
                       (const float*)qkv, (float*)out, S, heads);
  else
    hipLaunchKernelGGL((attn_f32_blocks_kernel<NW>), dim3(n_seq * heads), dim3(NW * 64), lds, st, (const float*)qkv,
                       (float*)out, S, heads);
  FC_CHECK_LAUNCH("attention(f32 blocks)");
  return FC_OK;
}

}  // namespace

// fp32 attention whose output is written as x3 rows [n_seq * S, 4 * heads * 64 bf16 positions] (split-fp32 mode).  Only the
// streaming-block kernel writes them directly (the ViT's 197 tokens); `attention_x3_supported` tells the caller when to
// run the plain fp32 kernel + launch_split3_rows instead.
bool attention_x3_supported(int S, int causal) { return !causal && S > 112 && S <= 224; }
int launch_attention_x3(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t stream) {
  if (n_seq <= 0) return FC_OK;
  if (!attention_x3_supported(S, 0) || heads <= 0) return fail(FC_EINVAL, "attention(x3): S=%d heads=%d", S, heads);
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 127)) return fail(FC_EINVAL, "attention(x3): unaligned operand");
  return launch_f32_blocks(qkv, out, n_seq, S, heads, stream, true);
}

int launch_attention(int precision, const void* qkv, void* out, int n_seq, int S, int heads, int causal,
                     hipStream_t stream) {
  if (n_seq <= 0) return FC_OK;
  if (S <= 0 || heads <= 0) return fail(FC_EINVAL, "attention: S=%d heads=%d", S, heads);
  if (((uintptr_t)qkv | (uintptr_t)out) & 15) return fail(FC_EINVAL, "attention: unaligned operand");
  if (precision == PREC_BF16) {
    if (S <= 32) return launch_bf16_v2<2, 2>(qkv, out, n_seq, S, heads, causal, stream);
    if (S <= 96) return launch_bf16_v2<6, 5>(qkv, out, n_seq, S, heads, causal, stream);
    // 8 waves: every workgroup puts two waves on each SIMD (7 waves for the 13 query tiles of the ViT left one SIMD with a
    // single wave per workgroup)
    if (S <= 224) return launch_bf16_v2<14, 8>(qkv, out, n_seq, S, heads, causal, stream);
    if (causal) return fail(FC_EINVAL, "attention(bf16): causal attention over %d > 224 tokens is not supported", S);
    return launch_bf16_flash(qkv, out, n_seq, S, heads, stream);
  }
  if (S <= 288) {  // K and V of one (sequence, head) fit LDS in fp32 up to 288 tokens (144 KiB)
    if (S <= 96)
      return causal ? launch_f32_mfma<6, 6, true>(qkv, out, n_seq, S, heads, stream)
                    : launch_f32_mfma<6, 6, false>(qkv, out, n_seq, S, heads, stream);
    if (S <= 224) {
      if (!causal) return launch_f32_blocks(qkv, out, n_seq, S, heads, stream);
      return launch_f32_mfma<14, 14, true>(qkv, out, n_seq, S, heads, stream);
    }
    return causal ? launch_f32_mfma<18, 8, true>(qkv, out, n_seq, S, heads, stream)
                  : launch_f32_mfma<18, 8, false>(qkv, out, n_seq, S, heads, stream);
  }
  const int kc = std::min(S, 256), lds = kc * 64 * 4 * 2;
  auto k0 = attn_f32_kernel<false>;
  auto k1 = attn_f32_kernel<true>;
  if (raise_dynamic_lds((const void*)k0, 128 * 1024) != hipSuccess || raise_dynamic_lds((const void*)k1, 128 * 1024) != hipSuccess)
    return fail(FC_ELAUNCH, "attention(f32): cannot raise dynamic LDS");
  const dim3 grid(n_seq * heads, (S + 255) / 256), block(256);
  if (causal)
    hipLaunchKernelGGL(k1, grid, block, lds, stream, (const float*)qkv, (float*)out, S, heads, kc);
  else
    hipLaunchKernelGGL(k0, grid, block, lds, stream, (const float*)qkv, (float*)out, S, heads, kc);
  FC_CHECK_LAUNCH("attention(f32)");
  return FC_OK;
}

}  // namespace fc

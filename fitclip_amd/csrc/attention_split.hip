// Split-fp32 attention for the ViT's 197 tokens (193..208: 13 key and query tiles): softmax(q k^T / 8) v with BOTH products
// on the bf16 matrix cores at fp32 accuracy - every fp32 operand as three bf16 planes (common.h, split3), every fp32 product
// as the six bf16 products p1q1 + p1q2 + p2q1 + p1q3 + p2q2 + p3q1 accumulated in fp32, the arithmetic of gemm_split3.h.
// Six bf16 MFMAs cover 32 k of a 16x16 tile in 6 x 16 cycles where `v_mfma_f32_16x16x4_f32` needs 8 x 32: 2.7x the matrix
// rate of attn_f32_blocks_kernel (attention.hip), whose place in the split-fp32 mode (fc_config.split_gemm) this kernel
// takes.  Reference semantics: nn.MultiheadAttention inside ResidualAttentionBlock (aligner/encoder/slip.py:364-380).
//
// Input: the fp32 output [rows, 3 D] of the fused QKV GEMM.  Output: x3 rows (common.h) - out_proj's A operand.
// One PERSISTENT workgroup of 8 waves per CU walks over the (sequence, head) pairs:
//   * K and V are read from HBM as fp32 INTO REGISTERS, split there, and stored as three bf16 planes each, row-major
//     [key][64 d] (128-byte rows, 208 rows): 6 x 26 KiB = 156 KiB of LDS, one workgroup per CU.  K rows carry the 16-byte
//     chunk swizzle of attn_bf16_v2_kernel (conflict-free ds_read_b128 of the S^T A operand), V rows its 32-byte granule
//     swizzle (conflict-free `ds_read_b64_tr_b16`, the transposing read that delivers the V^T A operand of
//     O^T = V^T P^T).  The loads of the next operand are in flight under the products of the current one.
//   * 13 query tiles on 8 waves: waves 0..4 carry two tiles and run them as ONE 32-query tile on
//     `v_mfma_f32_32x32x16_bf16` (every K / V fragment read feeds both), waves 5..7 one tile on `16x16x32`.
//   * S^T = K Q^T with the Q planes in registers; the softmax is single pass over the score tiles in registers, with the
//     compensated exponential of the fp32 kernel; the probabilities are split into planes right before they become the
//     B operand of P.V.
//   * two barriers per pass (K complete / V complete).  Measured and NOT kept: two phase groups of
//     waves, one interval apart, so that one wave of a SIMD multiplies while the other splits and exponentiates - as
//     separate instantiations the code (69 KB) no longer fits the 64 KB instruction cache two CUs share, as one body with
//     run-time roles hipcc spills the registers in-flight loads land in; both ran slower than this lockstep loop.
//   * each lane ends with 4 consecutive d of one query per 16-column group: split once more and stored as the [p1|p2|p3|0]
//     quarters of the group's 128-byte line (the four stores of a line leave the same wave back to back).
#include "common.h"

#include <algorithm>

namespace fc {

namespace {

constexpr float kNegInfS = -__builtin_inff();

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short s16x4;

// The canonical three-plane split (common.h, split3) of two values at a time, spelled for the packed instructions: one
// v_cvt_pk_bf16_f32 per plane, the plane widened back with a shift and a mask, one v_pk_add_f32 per residual - 9 VALU
// instructions per pair (hipcc's own vectoriser reaches that on some pairs of split3 and 14 on others).
struct planes32 { unsigned p1, p2, p3; };
__device__ __forceinline__ planes32 split3_pair(const f32x2 x) {
#pragma clang fp contract(off)
  unsigned p1, p2, p3;
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
  const f32x2 r1 = x - f32x2{__uint_as_float(p1 << 16), __uint_as_float(p1 & 0xffff0000u)};
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
  const f32x2 r2 = r1 - f32x2{__uint_as_float(p2 << 16), __uint_as_float(p2 & 0xffff0000u)};
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
  return planes32{p1, p2, p3};
}
__device__ __forceinline__ void split3x8(const f32x4& a, const f32x4& b, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
  const planes32 w = split3_pair(f32x2{a[0], a[1]}), x = split3_pair(f32x2{a[2], a[3]});
  const planes32 y = split3_pair(f32x2{b[0], b[1]}), z = split3_pair(f32x2{b[2], b[3]});
  p1 = __builtin_bit_cast(bf16x8, u32x4{w.p1, x.p1, y.p1, z.p1});
  p2 = __builtin_bit_cast(bf16x8, u32x4{w.p2, x.p2, y.p2, z.p2});
  p3 = __builtin_bit_cast(bf16x8, u32x4{w.p3, x.p3, y.p3, z.p3});
}
__device__ __forceinline__ void split3x4(const f32x4& a, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
  const planes32 w = split3_pair(f32x2{a[0], a[1]}), x = split3_pair(f32x2{a[2], a[3]});
  p1 = __builtin_bit_cast(bf16x4, u32x2{w.p1, x.p1});
  p2 = __builtin_bit_cast(bf16x4, u32x2{w.p2, x.p2});
  p3 = __builtin_bit_cast(bf16x4, u32x2{w.p3, x.p3});
}
constexpr int NKT = 13, NKR = NKT * 16;      // 16-key tiles / rows per plane
constexpr int PL = NKR * 128;                // bytes per plane
constexpr int OFF_V = 3 * PL;
constexpr int NW = 8, NT = NW * 64;
// Staging shares ((key, 8-d chunk) items of 64 lanes x N iterations per operand): the waves with one query tile (5..7) take
// 6 iterations each, three of the five two-tile waves 3 each (3 * 384 + 3 * 192 = 1728 >= 1664 items), two stage nothing -
// the two-tile waves carry twice the splits, exponentials and MFMAs of the others.
constexpr int NIT16 = 6, NIT32 = 3;
constexpr int ATTN_SPLIT_LDS = 6 * PL;
// the six products, smallest terms first: (plane of the LDS operand, plane of the register operand)
__device__ constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};

// ABL (tools/attn_split_lab.hip only): 1 = no S products, 2 = no P.V products, 3 = no exponentials, 4 = no staging of K / V,
// 10 = 1 + 2 + 3 (memory traffic, splits and barriers only), 11 = no output stores,
// 20 = s_memtime stamps around the barriers of wave 0 of workgroup 0, 30 / 31 = the waves that stage nothing are 0, 4 / 0, 2
// instead of 0, 1
template <int ABL>
struct Flags {
  static constexpr bool kNoS = ABL == 1 || ABL == 10, kNoPV = ABL == 2 || ABL == 10, kNoExp = ABL == 3 || ABL == 10,
                        kNoStage = ABL == 4, kNoStore = ABL == 11;
};

struct Ctx {
  const float* qkv;
  char* out;
  char* smem;
  long long* stamps;
  int S, heads, D, n_items;
  long ld;
  int* sat;    // x2 output: device flag for values beyond fp16's range (may be null)
  int tid, lane, wave;
  int stage0;  // first (key, chunk) staging item of this wave (wave-uniform), -1: this wave stages nothing

  __device__ __forceinline__ const float* item_base(int item) const {
    const int seq = item / heads, h = item - seq * heads;
    return qkv + (long)seq * S * ld + h * 64;
  }
  // K or V of one (sequence, head) as fp32 into registers: staging item = (key, 8-d chunk c).  Padded keys (and the items
  // past the last row: loaded, never stored) read a valid row; padded keys are masked in the softmax.
  template <int N>
  __device__ __forceinline__ void load_kv(const float* base, int isv, f32x4 (&raw)[N][2]) const {
    if (stage0 < 0) return;
    // (staging addresses are rebuilt from an opaque copy of the lane id at every call: hipcc would otherwise hoist them out of
    // the persistent loop over the (image, head) items and keep them - spilled - across the products)
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
#pragma unroll
    for (int it = 0; it < N; ++it) {
      const int item = stage0 + it * 64 + lane_s, key = item >> 3, c = item & 7;
      const float* src = base + (isv ? 2 * D : D) + (long)min(key, S - 1) * ld + c * 8;
      raw[it][0] = *reinterpret_cast<const f32x4*>(src);
      raw[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
  }
  // ... and from the registers as three bf16 planes into LDS
  template <int N>
  __device__ __forceinline__ void store_kv(int isv, const f32x4 (&raw)[N][2]) const {
    if (stage0 < 0) return;
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
#pragma unroll
    for (int it = 0; it < N; ++it) {
      const int item = stage0 + it * 64 + lane_s, key = item >> 3, c = item & 7;
      if (item < NKR * 8) {
        bf16x8 p1, p2, p3;
        split3x8(raw[it][0], raw[it][1], p1, p2, p3);
        const int pos = isv ? ((((c >> 1) ^ ((key >> 1) & 3)) << 5) | ((c & 1) << 4)) : ((c ^ ((key >> 1) & 7)) << 4);
        char* dst = smem + (isv ? OFF_V : 0) + key * 128 + pos;
        *reinterpret_cast<bf16x8*>(dst) = p1;
        *reinterpret_cast<bf16x8*>(dst + PL) = p2;
        *reinterpret_cast<bf16x8*>(dst + 2 * PL) = p3;
      }
    }
  }
  // workgroup barrier that also orders this wave's LDS accesses before it - and nothing else: global loads stay in flight
  __device__ __forceinline__ void barrier() const {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
};

// ---- one 16-query tile (`wave`) on v_mfma_f32_16x16x32_bf16: lane (r, g) = (lane & 15, lane >> 4) holds query r;
// sT[t][e] = score(key 16 t + 4 g + e, query r); k-slot (g, j) of P.V step ks <-> key 32 ks + 16 (j >> 2) + 4 g + (j & 3)
template <int ABL>
struct Tile16 {
  using F = Flags<ABL>;
  static constexpr int NITW = NIT16;
  f32x4 qraw[2][2], sT[NKT], o[4];
  bf16x8 qf[3][2];
  float inv;
  int r, g, f, qtile;

  __device__ __forceinline__ void init(const Ctx& c) {
    r = c.lane & 15, g = c.lane >> 4, f = (r >> 1) & 7;
    qtile = c.wave;
  }
  __device__ __forceinline__ void load_q(const Ctx& c, const float* base) {
    const float* qrow = base + (long)min(qtile * 16 + r, c.S - 1) * c.ld + 8 * g;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qraw[s][0] = *reinterpret_cast<const f32x4*>(qrow + 32 * s);
      qraw[s][1] = *reinterpret_cast<const f32x4*>(qrow + 32 * s + 4);
    }
  }
  __device__ __forceinline__ void split_q() {  // pre-scaled by 1 / sqrt(64) (exact)
#pragma unroll
    for (int s = 0; s < 2; ++s) split3x8(qraw[s][0] * 0.125f, qraw[s][1] * 0.125f, qf[0][s], qf[1][s], qf[2][s]);
  }
  __device__ __forceinline__ void scores(const Ctx& c) {
    const char* kbase = c.smem + r * 128;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 kf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          kf[p] = *reinterpret_cast<const bf16x8*>(kbase + p * PL + t * 2048 + (((4 * s + g) ^ f) << 4));
        if (F::kNoS) {
          acc[0] += (float)kf[0][0] + (float)kf[1][1] + (float)kf[2][2];
          continue;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[PA[k]], qf[PB[k]][s], acc, 0, 0, 0);
      }
      sT[t] = acc;
    }
  }
  __device__ __forceinline__ void softmax(const Ctx& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)  // only the last tile has masked keys (192 < S <= 208)
      if ((NKT - 1) * 16 + 4 * g + e >= c.S) sT[NKT - 1][e] = kNegInfS;
    float mx = sT[0][0];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      // v_max3_f32 spelled out: `fmaxf` on raw MFMA results makes hipcc canonicalise every input first (IEEE mode)
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(sT[t][0]), "v"(sT[t][1]));
      asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(sT[t][2]), "v"(sT[t][3]));
    }
    mx = max_over_lane_groups(mx);  // finite: key 0 is never masked
    const f32x2 m2 = {mx, mx};
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT - 1; ++t) {
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x2 x = f32x2{sT[t][2 * hh], sT[t][2 * hh + 1]} - m2;
        const f32x2 pr = F::kNoExp ? x : exp_neg_finite_pair(x);
        sT[t][2 * hh] = pr[0];
        sT[t][2 * hh + 1] = pr[1];
        sum2 += pr;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // the guarded exponential: exp(-inf) = 0
      const float pr = F::kNoExp ? 0.f : exp_neg_f32(sT[NKT - 1][e] - mx);
      sT[NKT - 1][e] = pr;
      sum2[e & 1] += pr;
    }
    inv = 1.f / sum_over_lane_groups(sum2[0] + sum2[1]);
  }
  // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]
  __device__ __forceinline__ void pv(const Ctx& c) {
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = (c.lane >> 2) & 3, tp = c.lane & 3;
    const int vsw = (2 * g + (tq >> 1)) & 3;  // ((key >> 1) & 3) of every row this lane addresses
    int voff[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) voff[n] = OFF_V + (4 * g + tq) * 128 + ((n ^ vsw) << 5) + tp * 8;
#pragma unroll
    for (int ks = 0; ks < (NKT + 1) / 2; ++ks) {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const bool pair = 2 * ks + 1 < NKT;  // the last step has one key tile: its upper k-slots carry P = 0 ...
      bf16x8 pp[3];
      split3x8(sT[2 * ks], pair ? sT[pair ? 2 * ks + 1 : 0] : zero4, pp[0], pp[1], pp[2]);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        bf16x8 vf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const char* va = c.smem + voff[n] + ks * 4096 + p * PL;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(  // ... against (finite) V rows that exist
              (__attribute__((address_space(3))) s16x4*)(va + (pair ? 2048 : 0)));
          vf[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        if (F::kNoPV) {
          o[n][0] += (float)vf[0][0] + (float)vf[1][1] + (float)vf[2][2] + (float)pp[0][0] + (float)pp[1][1] + (float)pp[2][2];
          continue;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[PA[k]], pp[PB[k]], o[n], 0, 0, 0);
      }
    }
  }
  // o[n][e] = O(query r, d = 16 n + 4 g + e): the planes of four consecutive columns, 8 bytes per quarter of the line
  // X2: two fp16 planes instead (common.h: a head is two 128-byte lines [h1 x32 | h2 x32]); returns max |value| of this lane
  template <bool X2>
  __device__ __forceinline__ float store_out(const Ctx& c, int item) const {
    const int seq = item / c.heads, h = item - seq * c.heads;
    const int query = qtile * 16 + r;
    float amax = 0.f;
    if (query < c.S && (!F::kNoStore || inv == 123.f)) {
      if constexpr (X2) {
        char* line = c.out + ((long)seq * c.S + query) * ((long)c.D * 4) + (long)(h * 2) * X2_GROUP_BYTES + g * 8;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const f32x4 v = o[n] * inv;
          amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
          f16x4 h1, h2;
          split2(v, h1, h2);
          char* dst = line + (n >> 1) * X2_GROUP_BYTES + (n & 1) * 32;
          *reinterpret_cast<f16x4*>(dst) = h1;
          *reinterpret_cast<f16x4*>(dst + 64) = h2;
        }
      } else {
        char* line = c.out + ((long)seq * c.S + query) * ((long)c.D * 8) + (long)(h * 4) * X3_GROUP_BYTES + g * 8;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          bf16x4 p1, p2, p3;
          split3x4(o[n] * inv, p1, p2, p3);
          *reinterpret_cast<bf16x4*>(line + n * X3_GROUP_BYTES) = p1;
          *reinterpret_cast<bf16x4*>(line + n * X3_GROUP_BYTES + 32) = p2;
          *reinterpret_cast<bf16x4*>(line + n * X3_GROUP_BYTES + 64) = p3;
          *reinterpret_cast<bf16x4*>(line + n * X3_GROUP_BYTES + 96) = bf16x4{};
        }
      }
    }
    return amax;
  }
};

// ---- two query tiles (`wave`, `wave + 8`) as ONE 32-query tile on v_mfma_f32_32x32x16_bf16 (the same FLOPs per cycle as
// 16x16x32, half the MFMA issue slots and LDS fragment reads).  Lane (c, hh) = (lane & 31, lane >> 5): query c of the pair
// (tile `wave` for c < 16, `wave + 8` above); the S^T accumulator of key tile T holds keys 32 T + 8 b + 4 hh + j in
// register 4 b + j, and registers 8 i .. 8 i + 7 are the B operand of the P.V step over keys 32 T + 16 i .. + 15
// (k-slot (hh, j8) <-> key 32 T + 16 i + 8 (j8 >> 2) + 4 hh + (j8 & 3)).
template <int ABL>
struct Tile32 {
  using F = Flags<ABL>;
  static constexpr int NT32 = (NKR + 31) / 32;  // 7 key tiles of 32; rows 208 .. 223 do not exist (masked, clamped reads)
  static constexpr int NITW = NIT32;
  f32x4 qraw[4][2];
  f32x16 sT[NT32], o[2];
  bf16x8 qf[3][4];
  float inv;
  int c, hh, query;

  __device__ __forceinline__ void init(const Ctx& x) {
    c = x.lane & 31, hh = x.lane >> 5;
    query = (x.wave + (c >> 4) * NW) * 16 + (c & 15);
  }
  __device__ __forceinline__ void load_q(const Ctx& x, const float* base) {
    const float* qrow = base + (long)min(query, x.S - 1) * x.ld + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qraw[ks][0] = *reinterpret_cast<const f32x4*>(qrow + 16 * ks);
      qraw[ks][1] = *reinterpret_cast<const f32x4*>(qrow + 16 * ks + 4);
    }
  }
  __device__ __forceinline__ void split_q() {  // pre-scaled by 1 / sqrt(64) (exact)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) split3x8(qraw[ks][0] * 0.125f, qraw[ks][1] * 0.125f, qf[0][ks], qf[1][ks], qf[2][ks]);
  }
  __device__ __forceinline__ void scores(const Ctx& x) {
    // K fragment addresses (A operand of S^T: key 32 T + c, d = 16 ks + 8 hh .. + 7 = chunk 2 ks + hh, swizzled)
    const int kswz = (c >> 1) & 7;             // ((32 T + c) >> 1) & 7
    const int key6 = min(192 + c, NKR - 1);    // the last tile: rows past 207 read row 207 (masked)
    const int kswz6 = (key6 >> 1) & 7;
    int koff[4], koff6[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      koff[ks] = c * 128 + (((2 * ks + hh) ^ kswz) << 4);
      koff6[ks] = key6 * 128 + (((2 * ks + hh) ^ kswz6) << 4);
    }
#pragma unroll
    for (int T = 0; T < NT32; ++T) {
      f32x16 acc = {};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        bf16x8 kf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
          kf[p] = *reinterpret_cast<const bf16x8*>(x.smem + (T == NT32 - 1 ? koff6[ks] : koff[ks] + T * 4096) + p * PL);
        if (F::kNoS) {
          acc[0] += (float)kf[0][0] + (float)kf[1][1] + (float)kf[2][2];
          continue;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[PA[k]], qf[PB[k]][ks], acc, 0, 0, 0);
      }
      sT[T] = acc;
    }
  }
  __device__ __forceinline__ void softmax(const Ctx& x) {
#pragma unroll
    for (int e = 0; e < 8; ++e)  // the last tile holds keys 192 + 8 b + 4 hh + j; its registers 8 .. 15 are keys >= 208
      if (192 + 8 * (e >> 2) + 4 * hh + (e & 3) >= x.S) sT[NT32 - 1][e] = kNegInfS;
    float mx = sT[0][0];
#pragma unroll
    for (int T = 0; T < NT32; ++T)
#pragma unroll
      for (int e = 0; e < (T == NT32 - 1 ? 8 : 16); e += 2)
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx) : "v"(mx), "v"(sT[T][e]), "v"(sT[T][e + 1]));
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));  // finite: key 0 is never masked
    }
    const f32x2 m2 = {mx, mx};
    f32x2 sum2 = {0.f, 0.f};
#pragma unroll
    for (int T = 0; T < NT32 - 1; ++T) {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2 xx = f32x2{sT[T][e], sT[T][e + 1]} - m2;
        const f32x2 pr = F::kNoExp ? xx : exp_neg_finite_pair(xx);
        sT[T][e] = pr[0];
        sT[T][e + 1] = pr[1];
        sum2 += pr;
      }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {  // the guarded exponential (exp(-inf) = 0); keys 208 .. 223 are P = 0 outright
      const float pr = e < 8 && !F::kNoExp ? exp_neg_f32(sT[NT32 - 1][e] - mx) : 0.f;
      sT[NT32 - 1][e] = pr;
      sum2[e & 1] += pr;
    }
    float sum = sum2[0] + sum2[1];
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
      sum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    inv = 1.f / sum;
  }
  // O^T[d][query]: o[m][4 b + j] = O(query c, d = 32 m + 8 b + 4 hh + j)
  __device__ __forceinline__ void pv(const Ctx& x) {
    // V fragment addresses (A operand through the transposing read): lane 4 tq + tp of a 16-lane group supplies key row
    // (.. + tq), d columns 32 m + 16 (c >> 4) + 4 tp .. + 3
    const int tq = (x.lane >> 2) & 3, tp = x.lane & 3;
    const int vsw = (2 * hh + (tq >> 1)) & 3;  // ((key >> 1) & 3) of every row this lane addresses
    int voff[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) voff[m] = OFF_V + (4 * hh + tq) * 128 + (((2 * m + (c >> 4)) ^ vsw) << 5) + tp * 8;
    o[0] = f32x16{};
    o[1] = f32x16{};
#pragma unroll
    for (int st = 0; st < NKT; ++st) {  // 13 steps of 16 keys
      const int T = st >> 1, i = st & 1;
      bf16x8 pp[3];
      split3x8(f32x4{sT[T][8 * i], sT[T][8 * i + 1], sT[T][8 * i + 2], sT[T][8 * i + 3]},
               f32x4{sT[T][8 * i + 4], sT[T][8 * i + 5], sT[T][8 * i + 6], sT[T][8 * i + 7]}, pp[0], pp[1], pp[2]);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        bf16x8 vf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const char* va = x.smem + voff[m] + st * 2048 + p * PL;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + 1024));
          vf[p] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        if (F::kNoPV) {
          o[m][0] += (float)vf[0][0] + (float)vf[1][1] + (float)vf[2][2] + (float)pp[0][0] + (float)pp[1][1] + (float)pp[2][2];
          continue;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) o[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[PA[k]], pp[PB[k]], o[m], 0, 0, 0);
      }
    }
  }
  // o[m][4 b4 + j] = O(query c, d = 32 m + 8 b4 + 4 hh + j): of the 16 columns of group n = 2 m + (b4 >> 1) this lane holds
  // 4 hh .. + 3 (b4 even) and 8 + 4 hh .. + 3 (b4 odd), its partner lane (c, 1 - hh) the other two quads.  One
  // v_permlane32_swap per register hands the lower lane columns 0 .. 7 and the upper lane 8 .. 15 of every plane: 16-byte
  // stores, half the store instructions (the address unit works per 128-byte line a store touches - 32 per instruction
  // here - and the 8-byte version kept the waves waiting on it).
  // X2: two fp16 planes (a 32-column half of the head is one line: o[m] IS line m): the same swap hands the lower lane columns
  // 16 (n & 1) .. + 7 and the upper lane the next eight; returns max |value| of this lane
  template <bool X2>
  __device__ __forceinline__ float store_out(const Ctx& x, int item) const {
    const int seq = item / x.heads, h = item - seq * x.heads;
    if (F::kNoStore && inv != 123.f) return 0.f;
    const bool live = query < x.S;  // (the swaps need every lane)
    float amax = 0.f;
    if constexpr (X2) {
      char* line = x.out + ((long)seq * x.S + min(query, x.S - 1)) * ((long)x.D * 4) + (long)(h * 2) * X2_GROUP_BYTES + hh * 16;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const int m = n >> 1, be = 2 * (n & 1);
        const f32x4 ve = f32x4{o[m][4 * be], o[m][4 * be + 1], o[m][4 * be + 2], o[m][4 * be + 3]} * inv;
        const f32x4 vo = f32x4{o[m][4 * be + 4], o[m][4 * be + 5], o[m][4 * be + 6], o[m][4 * be + 7]} * inv;
#pragma unroll
        for (int q = 0; q < 4; ++q) amax = fmaxf(amax, fmaxf(fabsf(ve[q]), fabsf(vo[q])));
        f16x4 e[2], od[2];
        split2(ve, e[0], e[1]);
        split2(vo, od[0], od[1]);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const u32x2 ue = __builtin_bit_cast(u32x2, e[p]), uo = __builtin_bit_cast(u32x2, od[p]);
          const auto s0 = __builtin_amdgcn_permlane32_swap(ue[0], uo[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(ue[1], uo[1], false, false);
          if (live) *reinterpret_cast<u32x4*>(line + m * X2_GROUP_BYTES + (n & 1) * 32 + p * 64) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      }
      return live ? amax : 0.f;
    } else {
    char* line = x.out + ((long)seq * x.S + min(query, x.S - 1)) * ((long)x.D * 8) + (long)(h * 4) * X3_GROUP_BYTES + hh * 16;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int m = n >> 1, be = 2 * (n & 1);
      bf16x4 e[3], od[3];
      split3x4(f32x4{o[m][4 * be], o[m][4 * be + 1], o[m][4 * be + 2], o[m][4 * be + 3]} * inv, e[0], e[1], e[2]);
      split3x4(f32x4{o[m][4 * be + 4], o[m][4 * be + 5], o[m][4 * be + 6], o[m][4 * be + 7]} * inv, od[0], od[1], od[2]);
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        const u32x2 ue = __builtin_bit_cast(u32x2, e[p]), uo = __builtin_bit_cast(u32x2, od[p]);
        const auto s0 = __builtin_amdgcn_permlane32_swap(ue[0], uo[0], false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(ue[1], uo[1], false, false);
        // lower lanes: (own quad, partner's quad) of the even b4 = columns 0 .. 7; upper lanes: columns 8 .. 15
        if (live) *reinterpret_cast<u32x4*>(line + n * X3_GROUP_BYTES + p * 32) = u32x4{s0[0], s1[0], s0[1], s1[1]};
      }
      if (live) *reinterpret_cast<u32x4*>(line + n * X3_GROUP_BYTES + 96) = u32x4{0u, 0u, 0u, 0u};
    }
    return 0.f;
    }
  }
};

// The persistent loop of one wave: one pass per (sequence, head).  Global loads of the NEXT operand are in flight under
// the products of the current one: V(i) is requested when the S products of item i start and lands in LDS behind its
// softmax; K(i + 1) and Q(i + 1) are requested when the P.V products of item i start, and K(i + 1) replaces K(i) behind
// them.  Barrier A: K(i) is complete and nobody reads V(i - 1) any more; barrier B: V(i) is complete and nobody reads K(i)
// any more.
template <int ABL, class Tile, bool X2>
__device__ __forceinline__ void run_wave(const Ctx& c) {
  using F = Flags<ABL>;
  Tile t;
  t.init(c);
  float amax = 0.f;
  f32x4 raw[Tile::NITW][2];
  int nstamp = 0;
  auto bar = [&]() {
    const bool on = ABL == 20 && blockIdx.x == 0 && c.wave == 0 && c.lane == 0;
    if (on) c.stamps[nstamp] = __builtin_amdgcn_s_memtime();
    c.barrier();
    if (on) c.stamps[nstamp + 1] = __builtin_amdgcn_s_memtime();
    nstamp += 2;
  };
  int item = blockIdx.x;
  const float* base = c.item_base(item);
  t.load_q(c, base);
  if (!F::kNoStage) {
    c.load_kv(base, 0, raw);
    c.store_kv(0, raw);
  }
  for (;;) {
    t.split_q();
    bar();  // A
    if (!F::kNoStage) c.load_kv(base, 1, raw);
    t.scores(c);
    __builtin_amdgcn_sched_barrier(0);
    t.softmax(c);
    __builtin_amdgcn_sched_barrier(0);
    if (!F::kNoStage) c.store_kv(1, raw);
    bar();  // B
    const int next = item + gridDim.x;
    const bool has_next = next < c.n_items;  // workgroup-uniform
    if (has_next) {
      base = c.item_base(next);
      t.load_q(c, base);
      if (!F::kNoStage) c.load_kv(base, 0, raw);
    }
    __builtin_amdgcn_sched_barrier(0);
    t.pv(c);
    __builtin_amdgcn_sched_barrier(0);
    if (has_next && !F::kNoStage) c.store_kv(0, raw);  // K(i + 1) over K(i): every wave is past barrier B
    amax = fmaxf(amax, t.template store_out<X2>(c, item));
    if (!has_next) break;
    item = next;
  }
  if (X2 && c.sat && !(amax <= 65504.f)) atomicOr(c.sat, 1);
}

template <int ABL = 0, bool X2 = false>
__global__ void __launch_bounds__(NT) attn_split_kernel(const float* __restrict__ qkv, char* __restrict__ out, int S, int heads,
                                                        int n_items, long long* __restrict__ stamps, int* sat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx c;
  c.qkv = qkv, c.out = out, c.smem = smem, c.stamps = stamps, c.sat = sat;
  c.S = S, c.heads = heads, c.D = heads * 64, c.n_items = n_items;
  c.ld = 3L * c.D;
  c.tid = threadIdx.x, c.lane = c.tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  if (blockIdx.x >= n_items) return;  // (the launcher never asks for more workgroups than items)
  {
    const int idle0 = 0, idle1 = ABL == 30 ? 4 : ABL == 31 ? 2 : 1;  // (measured: 0, 1 and 0, 2 equal, 0, 4 1.5 % slower)
    if (c.wave >= 5) {
      c.stage0 = (c.wave - 5) * (NIT16 * 64);
    } else if (c.wave == idle0 || c.wave == idle1) {
      c.stage0 = -1;
    } else {
      int slot = 0;
#pragma unroll
      for (int w = 0; w < 5; ++w)
        if (w < c.wave && w != idle0 && w != idle1) ++slot;
      c.stage0 = 3 * NIT16 * 64 + slot * (NIT32 * 64);
    }
  }
  if (c.wave + NW < NKT)  // wave-uniform: waves 0 .. 4 carry two query tiles (w, w + 8)
    run_wave<ABL, Tile32<ABL>, X2>(c);
  else
    run_wave<ABL, Tile16<ABL>, X2>(c);
}

}  // namespace

bool attention_split_supported(int S, int causal) { return !causal && S > 192 && S <= 208; }

// qkv: f32 [n_seq * S, 3 * heads * 64]; out: x3 rows [n_seq * S, 4 * heads * 64 bf16 positions], or - out_kind KIND_X2 - x2 rows
// [n_seq * S, 2 * heads * 64 fp16 positions] (sat_flag: common.h)
int launch_attention_split(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t stream, int out_kind, int* sat_flag) {
  if (n_seq <= 0) return FC_OK;
  if (!attention_split_supported(S, 0) || heads <= 0) return fail(FC_EINVAL, "attention(split): S=%d heads=%d", S, heads);
  if (out_kind != KIND_X3 && out_kind != KIND_X2) return fail(FC_EINVAL, "attention(split): output kind %d", out_kind);
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 127)) return fail(FC_EINVAL, "attention(split): unaligned operand");
  const long items = (long)n_seq * heads;
  if (items > 0x7fffffffL) return fail(FC_EINVAL, "attention(split): %ld (sequence, head) pairs", items);
  // (per launch: raise_dynamic_lds memoises per (device, kernel), and the CU count belongs to the current device as well)
  const void* kern = out_kind == KIND_X2 ? (const void*)attn_split_kernel<0, true> : (const void*)attn_split_kernel<0, false>;
  if (raise_dynamic_lds(kern, ATTN_SPLIT_LDS) != hipSuccess)  // the six planes: one workgroup per CU
    return fail(FC_ELAUNCH, "attention(split): cannot raise dynamic LDS");
  const int cus = device_cus();
  const dim3 grid((unsigned)std::min<long>(items, cus));
  if (out_kind == KIND_X2)
    hipLaunchKernelGGL((attn_split_kernel<0, true>), grid, dim3(NT), ATTN_SPLIT_LDS, stream, (const float*)qkv, (char*)out, S, heads,
                       (int)items, (long long*)nullptr, sat_flag);
  else
    hipLaunchKernelGGL((attn_split_kernel<0, false>), grid, dim3(NT), ATTN_SPLIT_LDS, stream, (const float*)qkv, (char*)out, S, heads,
                       (int)items, (long long*)nullptr, (int*)nullptr);
  FC_CHECK_LAUNCH("attention(split)");
  return FC_OK;
}

}  // namespace fc

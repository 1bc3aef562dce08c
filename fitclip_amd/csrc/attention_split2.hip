// Split-fp32 attention on TWO fp16 planes (gfx950): softmax(q k^T / 8) v for the ViT's 197 tokens (193..208) with both products
// as THREE fp16 products per fp32 product - the attention of the fc_config.split_gemm = 2 mode (gemm_split2.h), taking the place
// of attn_split_kernel's six bf16 products there.  Reference semantics: nn.MultiheadAttention inside ResidualAttentionBlock
// (aligner/encoder/slip.py:364-380).
//
// Both operands of either product are ACTIVATIONS, so there is no tensor to scale ahead of time: every value is held as
// x = h1 + 2^-11 h2 with h1 = fp16(x), h2 = fp16((x - h1) 2^11) (common.h: exact to 2^-23 |x| down to 2^-14, 2^-36 absolute
// below), and a product is recovered from
//     h1 g1  +  2^-11 (h1 g2 + h2 g1)                (the dropped h2 g2 is 2^-22 of the product)
// with the two cross terms in a SECOND accumulator that is folded in once per score tile / output tile.  The accumulators here
// are small (16 registers per 32-key score tile, 32 per 32-column output half), unlike the GEMM's, so the second set fits.
// Half the MFMAs of the six-product kernel, two thirds of its LDS planes (4 x 26 KiB), 6 instead of 9 VALU instructions per
// pair in every split.
//
// Structure: attn_split_kernel's (attention_split.hip) - one PERSISTENT workgroup of 8 waves per CU walks over the (sequence,
// head) pairs; K and V go from HBM as fp32 through registers into LDS planes ([key][64 d], 128-byte rows: K with the 16-byte
// chunk swizzle, V with the 32-byte granule swizzle for `ds_read_b64_tr_b16`); waves 0..4 carry two query tiles as ONE 32-query
// tile on v_mfma_f32_32x32x16_f16, waves 5..7 one tile on v_mfma_f32_16x16x32_f16; single-pass softmax in registers with the
// compensated exponential; the loads of the next operand in flight under the products of the current one, two barriers per pass.
// Differences: the two-tile waves split their probabilities into planes ONCE, in place of the scores (same registers), and run
// P.V one 32-column half of the head at a time - main + cross accumulator of one half = the 32 registers the six-product kernel
// needs for its single accumulator of both - storing each half (exactly one 128-byte x2 line per query) as soon as it is done.
// Output: x2 rows (out_proj's A operand).  Every value that enters a plane is checked against fp16's range (q, k, v on the way in,
// the output on the way out): beyond 65504 the device-side flag is raised (common.h).
#include "common.h"

#include <algorithm>

namespace fc {

namespace {

constexpr float kNegInfH = -__builtin_inff();

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short s16x4;

constexpr float kCross = 1.f / X2_RESID_SCALE;   // 2^-11
// 1 / sqrt(64) and log2 e in one factor on Q before it is split: the scores come out in the log2 domain and a probability is ONE
// v_exp_f32 of an exact difference (the compensated exponential of rounds 3-4 spent 7 packed operations per pair on the rounding of
// s log2 e, which the scale of Q now carries; vector instructions are what bounds this kernel next to its HBM traffic)
constexpr float kQLog2 = 0.125f * 1.4426950408889634f;

// two values -> (h1 pair, h2 pair) as packed words: v_cvt_pk_f16_f32, two conversions back, one packed subtract, one packed
// multiply, v_cvt_pk_f16_f32 - 6 VALU instructions per pair
struct planes2 { unsigned h1, h2; };
__device__ __forceinline__ planes2 split2_pair(const f32x2 x) {
#pragma clang fp contract(off)
  const f16x2 a = __builtin_convertvector(x, f16x2);
  const f32x2 r = (x - __builtin_convertvector(a, f32x2)) * X2_RESID_SCALE;
  const f16x2 b = __builtin_convertvector(r, f16x2);
  return planes2{__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)};
}
__device__ __forceinline__ void split2x8(const f32x4& a, const f32x4& b, f16x8& h1, f16x8& h2) {
  const planes2 w = split2_pair(f32x2{a[0], a[1]}), x = split2_pair(f32x2{a[2], a[3]});
  const planes2 y = split2_pair(f32x2{b[0], b[1]}), z = split2_pair(f32x2{b[2], b[3]});
  h1 = __builtin_bit_cast(f16x8, u32x4{w.h1, x.h1, y.h1, z.h1});
  h2 = __builtin_bit_cast(f16x8, u32x4{w.h2, x.h2, y.h2, z.h2});
}
// running max |x| over eight values (v_max3_f32 with |.| source modifiers: one instruction per two values).  A maximum drops NaN
// operands: this test catches finite values beyond fp16's range and infinities; a NaN among q, k, v reaches the frame's embedding
// through the softmax and the projections and is caught by the output scan of fc_encode_image (api.hip: nonfinite_flag_kernel)
__device__ __forceinline__ void amax8(float& m, const f32x4& a, const f32x4& b) {
  asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a[0]), "v"(a[1]));
  asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a[2]), "v"(a[3]));
  asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(b[0]), "v"(b[1]));
  asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(b[2]), "v"(b[3]));
}

constexpr int NKT = 13, NKR = NKT * 16;      // 16-key tiles / rows per plane
constexpr int PL = NKR * 128;                // bytes per plane
constexpr int OFF_V = 2 * PL;
constexpr int NW = 8, NT = NW * 64;
// staging shares as in attn_split_kernel: the one-tile waves take 6 iterations of 64 (key, 8-d chunk) items each, three of the
// two-tile waves 3 each, two stage nothing
constexpr int NIT16 = 6, NIT32 = 3;
constexpr int ATTN_SPLIT2_LDS = 4 * PL;      // 104 KiB

// ABL (tools/attn_split_lab.hip only): 1 = no S products, 2 = no P.V products, 3 = no exponentials, 4 = no staging, 11 = no stores
template <int ABL>
struct Flags2 {
  static constexpr bool kNoS = ABL == 1 || ABL == 10, kNoPV = ABL == 2 || ABL == 10, kNoExp = ABL == 3 || ABL == 10,
                        kNoStage = ABL == 4, kNoStore = ABL == 11;
};

struct Ctx2 {
  const float* qkv;
  char* out;
  char* smem;
  int* sat;
  int S, heads, D, n_items;
  long ld;
  int tid, lane, wave;
  int stage0;  // first (key, chunk) staging item of this wave (wave-uniform), -1: this wave stages nothing
  float amax;  // max |value| this lane has put into a plane (or written) so far

  __device__ __forceinline__ const float* item_base(int item) const {
    const int seq = item / heads, h = item - seq * heads;
    return qkv + (long)seq * S * ld + h * 64;
  }
  template <int N>
  __device__ __forceinline__ void load_kv(const float* base, int isv, f32x4 (&raw)[N][2]) const {
    if (stage0 < 0) return;
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
  template <int N>
  __device__ __forceinline__ void store_kv(int isv, const f32x4 (&raw)[N][2]) {
    if (stage0 < 0) return;
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
#pragma unroll
    for (int it = 0; it < N; ++it) {
      const int item = stage0 + it * 64 + lane_s, key = item >> 3, c = item & 7;
      if (item < NKR * 8) {
        f16x8 h1, h2;
        amax8(amax, raw[it][0], raw[it][1]);
        split2x8(raw[it][0], raw[it][1], h1, h2);
        const int pos = isv ? ((((c >> 1) ^ ((key >> 1) & 3)) << 5) | ((c & 1) << 4)) : ((c ^ ((key >> 1) & 7)) << 4);
        char* dst = smem + (isv ? OFF_V : 0) + key * 128 + pos;
        *reinterpret_cast<f16x8*>(dst) = h1;
        *reinterpret_cast<f16x8*>(dst + PL) = h2;
      }
    }
  }
  __device__ __forceinline__ void barrier() const {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
};

// ---- one 16-query tile (`wave`) on v_mfma_f32_16x16x32_f16: lane (r, g) = (lane & 15, lane >> 4) holds query r;
// sT[t][e] = score(key 16 t + 4 g + e, query r); k-slot (g, j) of P.V step ks <-> key 32 ks + 16 (j >> 2) + 4 g + (j & 3)
template <int ABL>
struct Tile16h {
  using F = Flags2<ABL>;
  static constexpr int NITW = NIT16;
  f32x4 qraw[2][2], sT[NKT], o[4];
  f16x8 qf[2][2];
  float inv;
  int r, g, f, qtile;

  __device__ __forceinline__ void init(const Ctx2& c) {
    r = c.lane & 15, g = c.lane >> 4, f = (r >> 1) & 7;
    qtile = c.wave;
  }
  __device__ __forceinline__ void load_q(const Ctx2& c, const float* base) {
    const float* qrow = base + (long)min(qtile * 16 + r, c.S - 1) * c.ld + 8 * g;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      qraw[s][0] = *reinterpret_cast<const f32x4*>(qrow + 32 * s);
      qraw[s][1] = *reinterpret_cast<const f32x4*>(qrow + 32 * s + 4);
    }
  }
  __device__ __forceinline__ void split_q(Ctx2& c) {  // pre-scaled by log2 e / sqrt(64): the softmax runs in the log2 domain
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      amax8(c.amax, qraw[s][0], qraw[s][1]);
      split2x8(qraw[s][0] * kQLog2, qraw[s][1] * kQLog2, qf[0][s], qf[1][s]);
    }
  }
  __device__ __forceinline__ void scores(const Ctx2& c) {
    const char* kbase = c.smem + r * 128;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f16x8 kf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          kf[p] = *reinterpret_cast<const f16x8*>(kbase + p * PL + t * 2048 + (((4 * s + g) ^ f) << 4));
        if (F::kNoS) {
          acc[0] += (float)kf[0][0] + (float)kf[1][1];
          continue;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[0][s], acc, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[0], qf[1][s], acx, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[1], qf[0][s], acx, 0, 0, 0);
      }
      sT[t] = acx * kCross + acc;
    }
  }
  __device__ __forceinline__ void softmax(const Ctx2& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e)  // only the last tile has masked keys (192 < S <= 208)
      if ((NKT - 1) * 16 + 4 * g + e >= c.S) sT[NKT - 1][e] = kNegInfH;
    float mx = sT[0][0];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
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
        const f32x2 pr = F::kNoExp ? x : f32x2{__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
        sT[t][2 * hh] = pr[0];
        sT[t][2 * hh + 1] = pr[1];
        sum2 += pr;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // (masked keys: 2^-inf = 0)
      const float pr = F::kNoExp ? 0.f : __builtin_amdgcn_exp2f(sT[NKT - 1][e] - mx);
      sT[NKT - 1][e] = pr;
      sum2[e & 1] += pr;
    }
    inv = 1.f / sum_over_lane_groups(sum2[0] + sum2[1]);
  }
  // O^T[d][query] = sum_key V^T[d][key] P^T[key][query]; then the x2 rows of this tile
  __device__ __forceinline__ void pv_and_store(Ctx2& c, int item) {
    f32x4 ox[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = ox[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int tq = (c.lane >> 2) & 3, tp = c.lane & 3;
    const int vsw = (2 * g + (tq >> 1)) & 3;  // ((key >> 1) & 3) of every row this lane addresses
    int voff[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) voff[n] = OFF_V + (4 * g + tq) * 128 + ((n ^ vsw) << 5) + tp * 8;
#pragma unroll
    for (int ks = 0; ks < (NKT + 1) / 2; ++ks) {
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
      const bool pair = 2 * ks + 1 < NKT;  // the last step has one key tile: its upper k-slots carry P = 0 ...
      f16x8 pp[2];
      split2x8(sT[2 * ks], pair ? sT[pair ? 2 * ks + 1 : 0] : zero4, pp[0], pp[1]);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        f16x8 vf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const char* va = c.smem + voff[n] + ks * 4096 + p * PL;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(  // ... against (finite) V rows that exist
              (__attribute__((address_space(3))) s16x4*)(va + (pair ? 2048 : 0)));
          vf[p] = __builtin_bit_cast(f16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        if (F::kNoPV) {
          o[n][0] += (float)vf[0][0] + (float)vf[1][1] + (float)pp[0][0] + (float)pp[1][1];
          continue;
        }
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[0], pp[0], o[n], 0, 0, 0);
        ox[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[0], pp[1], ox[n], 0, 0, 0);
        ox[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[1], pp[0], ox[n], 0, 0, 0);
      }
    }
    // o[n][e] = O(query r, d = 16 n + 4 g + e): a head is two 128-byte lines [h1 x32 | h2 x32]; 8 bytes per plane and n
    const int seq = item / c.heads, h = item - seq * c.heads;
    const int query = qtile * 16 + r;
    if (query < c.S && (!F::kNoStore || inv == 123.f)) {
      char* line = c.out + ((long)seq * c.S + query) * ((long)c.D * 4) + (long)(h * 2) * X2_GROUP_BYTES + g * 8;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 v = (ox[n] * kCross + o[n]) * inv;
        c.amax = fmaxf(fmaxf(c.amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        f16x4 h1, h2;
        split2(v, h1, h2);
        char* dst = line + (n >> 1) * X2_GROUP_BYTES + (n & 1) * 32;
        *reinterpret_cast<f16x4*>(dst) = h1;
        *reinterpret_cast<f16x4*>(dst + 64) = h2;
      }
    }
  }
};

// ---- two query tiles (`wave`, `wave + 8`) as ONE 32-query tile on v_mfma_f32_32x32x16_f16.  Lane (c, hh) = (lane & 31,
// lane >> 5): query c of the pair (tile `wave` for c < 16, `wave + 8` above); the S^T accumulator of key tile T holds keys
// 32 T + 8 b + 4 hh + j in register 4 b + j, and registers 8 i .. 8 i + 7 are the B operand of the P.V step over keys
// 32 T + 16 i .. + 15 (k-slot (hh, j8) <-> key 32 T + 16 i + 8 (j8 >> 2) + 4 hh + (j8 & 3)).
template <int ABL>
struct Tile32h {
  using F = Flags2<ABL>;
  static constexpr int NT32 = (NKR + 31) / 32;  // 7 key tiles of 32; rows 208 .. 223 do not exist (masked, clamped reads)
  static constexpr int NITW = NIT32;
  f32x4 qraw[4][2];
  f32x16 sT[NT32];
  f16x8 qf[2][4];
  f16x8 pl[NT32][2][2];   // the probabilities as planes [key tile][16-key half][plane]: written once, in place of sT
  float inv;
  int c, hh, query;

  __device__ __forceinline__ void init(const Ctx2& x) {
    c = x.lane & 31, hh = x.lane >> 5;
    query = (x.wave + (c >> 4) * NW) * 16 + (c & 15);
  }
  __device__ __forceinline__ void load_q(const Ctx2& x, const float* base) {
    const float* qrow = base + (long)min(query, x.S - 1) * x.ld + 8 * hh;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qraw[ks][0] = *reinterpret_cast<const f32x4*>(qrow + 16 * ks);
      qraw[ks][1] = *reinterpret_cast<const f32x4*>(qrow + 16 * ks + 4);
    }
  }
  __device__ __forceinline__ void split_q(Ctx2& x) {  // pre-scaled by log2 e / sqrt(64): the softmax runs in the log2 domain
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      amax8(x.amax, qraw[ks][0], qraw[ks][1]);
      split2x8(qraw[ks][0] * kQLog2, qraw[ks][1] * kQLog2, qf[0][ks], qf[1][ks]);
    }
  }
  __device__ __forceinline__ void scores(const Ctx2& x) {
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
      f32x16 acc = {}, acx = {};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        f16x8 kf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
          kf[p] = *reinterpret_cast<const f16x8*>(x.smem + (T == NT32 - 1 ? koff6[ks] : koff[ks] + T * 4096) + p * PL);
        if (F::kNoS) {
          acc[0] += (float)kf[0][0] + (float)kf[1][1];
          continue;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[0][ks], acc, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[0], qf[1][ks], acx, 0, 0, 0);
        acx = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[1], qf[0][ks], acx, 0, 0, 0);
      }
      sT[T] = acx * kCross + acc;
    }
  }
  __device__ __forceinline__ void softmax(const Ctx2& x) {
#pragma unroll
    for (int e = 0; e < 8; ++e)  // the last tile holds keys 192 + 8 b + 4 hh + j; its registers 8 .. 15 are keys >= 208
      if (192 + 8 * (e >> 2) + 4 * hh + (e & 3) >= x.S) sT[NT32 - 1][e] = kNegInfH;
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
        const f32x2 pr = F::kNoExp ? xx : f32x2{__builtin_amdgcn_exp2f(xx[0]), __builtin_amdgcn_exp2f(xx[1])};
        sT[T][e] = pr[0];
        sT[T][e + 1] = pr[1];
        sum2 += pr;
      }
      // ... and this tile's probabilities become planes right away (the scores are dead: same registers)
#pragma unroll
      for (int i = 0; i < 2; ++i)
        split2x8(f32x4{sT[T][8 * i], sT[T][8 * i + 1], sT[T][8 * i + 2], sT[T][8 * i + 3]},
                 f32x4{sT[T][8 * i + 4], sT[T][8 * i + 5], sT[T][8 * i + 6], sT[T][8 * i + 7]}, pl[T][i][0], pl[T][i][1]);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {  // (masked keys: 2^-inf = 0); keys 208 .. 223 are P = 0 outright
      const float pr = e < 8 && !F::kNoExp ? __builtin_amdgcn_exp2f(sT[NT32 - 1][e] - mx) : 0.f;
      sT[NT32 - 1][e] = pr;
      sum2[e & 1] += pr;
    }
    split2x8(f32x4{sT[NT32 - 1][0], sT[NT32 - 1][1], sT[NT32 - 1][2], sT[NT32 - 1][3]},
             f32x4{sT[NT32 - 1][4], sT[NT32 - 1][5], sT[NT32 - 1][6], sT[NT32 - 1][7]}, pl[NT32 - 1][0][0], pl[NT32 - 1][0][1]);
    float sum = sum2[0] + sum2[1];
    {
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
      sum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    inv = 1.f / sum;
  }
  // O^T[d][query], one 32-column half m of the head at a time: om[4 b4 + j] = O(query c, d = 32 m + 8 b4 + 4 hh + j); half m IS
  // line m of the head's x2 rows: of its 32 columns this lane holds the quads 4 hh .. + 3 of every 8-column block, its partner lane
  // (c, 1 - hh) the others; one v_permlane32_swap per register hands the lower lane columns 16 (n & 1) .. + 7 and the upper lane
  // the next eight of every plane: 16-byte stores.
  __device__ __forceinline__ void pv_and_store(Ctx2& x, int item) {
    const int tq = (x.lane >> 2) & 3, tp = x.lane & 3;
    const int vsw = (2 * hh + (tq >> 1)) & 3;  // ((key >> 1) & 3) of every row this lane addresses
    const int seq = item / x.heads, h = item - seq * x.heads;
    const bool live = query < x.S;             // (the swaps need every lane)
    char* line = x.out + ((long)seq * x.S + min(query, x.S - 1)) * ((long)x.D * 4) + (long)(h * 2) * X2_GROUP_BYTES + hh * 16;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int voff = OFF_V + (4 * hh + tq) * 128 + (((2 * m + (c >> 4)) ^ vsw) << 5) + tp * 8;
      f32x16 om = {}, ox = {};
#pragma unroll
      for (int st = 0; st < NKT; ++st) {  // 13 steps of 16 keys
        const int T = st >> 1, i = st & 1;
        f16x8 vf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const char* va = x.smem + voff + st * 2048 + p * PL;
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(va + 1024));
          vf[p] = __builtin_bit_cast(f16x8, __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        if (F::kNoPV) {
          om[0] += (float)vf[0][0] + (float)vf[1][1] + (float)pl[T][i][0][0] + (float)pl[T][i][1][1];
          continue;
        }
        om = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], pl[T][i][0], om, 0, 0, 0);
        ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[0], pl[T][i][1], ox, 0, 0, 0);
        ox = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[1], pl[T][i][0], ox, 0, 0, 0);
      }
      if (F::kNoStore && inv != 123.f) continue;
      om = (ox * kCross + om) * inv;
#pragma unroll
      for (int n = 0; n < 2; ++n) {   // the two 16-column halves of line m
        const f32x4 ve = f32x4{om[8 * n], om[8 * n + 1], om[8 * n + 2], om[8 * n + 3]};
        const f32x4 vo = f32x4{om[8 * n + 4], om[8 * n + 5], om[8 * n + 6], om[8 * n + 7]};
        if (live) amax8(x.amax, ve, vo);
        f16x4 e[2], od[2];
        split2(ve, e[0], e[1]);
        split2(vo, od[0], od[1]);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const u32x2 ue = __builtin_bit_cast(u32x2, e[p]), uo = __builtin_bit_cast(u32x2, od[p]);
          const auto s0 = __builtin_amdgcn_permlane32_swap(ue[0], uo[0], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(ue[1], uo[1], false, false);
          if (live) *reinterpret_cast<u32x4*>(line + m * X2_GROUP_BYTES + n * 32 + p * 64) = u32x4{s0[0], s1[0], s0[1], s1[1]};
        }
      }
    }
  }
};

// The persistent loop of one wave (attn_split_kernel's): barrier A: K(i) is complete and nobody reads V(i - 1) any more;
// barrier B: V(i) is complete and nobody reads K(i) any more.
template <int ABL, class Tile>
__device__ __forceinline__ void run_wave2(Ctx2& c) {
  using F = Flags2<ABL>;
  Tile t;
  t.init(c);
  f32x4 raw[Tile::NITW][2];
  int item = blockIdx.x;
  const float* base = c.item_base(item);
  t.load_q(c, base);
  if (!F::kNoStage) {
    c.load_kv(base, 0, raw);
    c.store_kv(0, raw);
  }
  for (;;) {
    t.split_q(c);
    c.barrier();  // A
    if (!F::kNoStage) c.load_kv(base, 1, raw);
    t.scores(c);
    __builtin_amdgcn_sched_barrier(0);
    t.softmax(c);
    __builtin_amdgcn_sched_barrier(0);
    if (!F::kNoStage) c.store_kv(1, raw);
    c.barrier();  // B
    const int next = item + gridDim.x;
    const bool has_next = next < c.n_items;  // workgroup-uniform
    if (has_next) {
      base = c.item_base(next);
      t.load_q(c, base);
      if (!F::kNoStage) c.load_kv(base, 0, raw);
    }
    __builtin_amdgcn_sched_barrier(0);
    t.pv_and_store(c, item);
    __builtin_amdgcn_sched_barrier(0);
    if (has_next && !F::kNoStage) c.store_kv(0, raw);  // K(i + 1) over K(i): every wave is past barrier B
    if (!has_next) break;
    item = next;
  }
  if (c.sat && !(c.amax <= 65504.f)) atomicOr(c.sat, 1);
}

template <int ABL = 0>
__global__ void __launch_bounds__(NT) attn_split2_kernel(const float* __restrict__ qkv, char* __restrict__ out, int S, int heads,
                                                         int n_items, int* sat) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Ctx2 c;
  c.qkv = qkv, c.out = out, c.smem = smem, c.sat = sat;
  c.S = S, c.heads = heads, c.D = heads * 64, c.n_items = n_items;
  c.ld = 3L * c.D;
  c.tid = threadIdx.x, c.lane = c.tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.amax = 0.f;
  if (blockIdx.x >= n_items) return;  // (the launcher never asks for more workgroups than items)
  if (c.wave >= 5) {
    c.stage0 = (c.wave - 5) * (NIT16 * 64);
  } else if (c.wave <= 1) {
    c.stage0 = -1;
  } else {
    c.stage0 = 3 * NIT16 * 64 + (c.wave - 2) * (NIT32 * 64);
  }
  if (c.wave + NW < NKT)  // wave-uniform: waves 0 .. 4 carry two query tiles (w, w + 8)
    run_wave2<ABL, Tile32h<ABL>>(c);
  else
    run_wave2<ABL, Tile16h<ABL>>(c);
}

}  // namespace

bool attention_split2_supported(int S, int causal) { return !causal && S > 192 && S <= 208; }

// qkv: f32 [n_seq * S, 3 * heads * 64]; out: x2 rows [n_seq * S, 2 * heads * 64 fp16 positions] (sat_flag: common.h)
int launch_attention_split2(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t stream, int* sat_flag) {
  if (n_seq <= 0) return FC_OK;
  if (!attention_split2_supported(S, 0) || heads <= 0) return fail(FC_EINVAL, "attention(split2): S=%d heads=%d", S, heads);
  if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 127)) return fail(FC_EINVAL, "attention(split2): unaligned operand");
  const long items = (long)n_seq * heads;
  if (items > 0x7fffffffL) return fail(FC_EINVAL, "attention(split2): %ld (sequence, head) pairs", items);
  if (raise_dynamic_lds((const void*)attn_split2_kernel<0>, ATTN_SPLIT2_LDS) != hipSuccess)  // four planes: one workgroup per CU
    return fail(FC_ELAUNCH, "attention(split2): cannot raise dynamic LDS");
  hipLaunchKernelGGL((attn_split2_kernel<0>), dim3((unsigned)std::min<long>(items, device_cus())), dim3(NT), ATTN_SPLIT2_LDS, stream,
                     (const float*)qkv, (char*)out, S, heads, (int)items, sat_flag);
  FC_CHECK_LAUNCH("attention(split2)");
  return FC_OK;
}

}  // namespace fc

// Backward of the fused multi-head self-attention (head dim 64), exact fp32 on the fp32-input matrix cores.
//
//   forward (attention.hip):  S = (Q / 8) K^T [+ causal mask],  P = softmax_rows(S),  O = P V
//   backward:                 dV = P^T dO,   dP = dO V^T,   dS = P * (dP - D),  D_q = sum_d dO[q,d] O[q,d],
//                             dQ = dS K / 8, dK = dS^T Q / 8
// (autograd of nn.MultiheadAttention inside ResidualAttentionBlock, aligner/encoder/slip.py:364-380, which the KD
// training step aligner/teacher_student.py:99-140 differentiates through.)
//
// One workgroup per (sequence, head), two phases over the same LDS, no atomics, deterministic:
//   phase 1  K and V in LDS (as in the forward kernel); each wave owns 16-query tiles: recomputes S^T and the softmax
//            statistics exactly as the forward does, then per key tile dP^T = V.dO^T, dS^T, and accumulates
//            dQ^T = K^T.dS^T with the dS^T accumulator registers as the MFMA B operand.  Row max / 1/sum / D of every
//            query go to a small LDS table.
//   phase 2  Q and dO replace K and V in LDS; each wave owns a 16-KEY tile (K, V fragments in registers) and walks the
//            query tiles: S, P (statistics from the table), dP, dS, and accumulates dV^T = dO^T.P and dK^T = Q^T.dS.
// Operand layouts, swizzles and the "accumulator as the next MFMA's operand" trick are those of attn_f32_mfma_kernel.
#include "common.h"


namespace fc {

namespace {

constexpr float kNegInfB = -__builtin_inff();

template <int NKT, int NW, bool CAUSAL>
__global__ void __launch_bounds__(NW * 64) attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ o,
                                                               const float* __restrict__ d_o, float* __restrict__ dqkv,
                                                               int S, int heads) {
  constexpr int NK = NKT * 16;
  constexpr int OFF_B = NK * 256;            // second tile (V, then dO)
  constexpr int OFF_STAT = 2 * NK * 256;     // [3][NK] floats: row max, 1 / row sum, D
  constexpr int NPIECE = (NK / 4 + NW - 1) / NW;
  constexpr int TPW = (NKT + NW - 1) / NW;   // tiles per wave in either phase
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* stat_m = reinterpret_cast<float*>(smem + OFF_STAT);
  float* stat_l = stat_m + NK;
  float* stat_d = stat_l + NK;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const long ld = 3L * D;
  const float* base = qkv + (long)seq * S * ld + h * 64;          // Q; K at + D, V at + 2 D
  const float* obase = o + (long)seq * S * D + h * 64;
  const float* dobase = d_o + (long)seq * S * D + h * 64;
  float* dbase = dqkv + (long)seq * S * ld + h * 64;
  const int r = lane & 15, g = lane >> 4;
  const int nt = (S + 15) >> 4;

  // rows of two [S, 64] matrices -> the two LDS tiles; lane l of a 4-row piece -> row l >> 4, physical chunk l & 15
  auto stage = [&](const float* a, long lda, const float* b, long ldb) {
    const int prow = lane >> 4, pch = lane & 15;
#pragma unroll
    for (int isb = 0; isb < 2; ++isb) {
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) {
        const int piece = wave + j * NW;
        if (piece < NK / 4) {
          const int row = piece * 4 + prow;
          const int srow = min(row, S - 1);  // padded rows read a valid row; they are masked below
          const float* src = (isb ? b + (long)srow * ldb : a + (long)srow * lda) + ((pch ^ (row & 15)) << 2);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(smem + (isb ? OFF_B : 0) + piece * 1024),
                                           16, 0, 0);
        }
      }
    }
  };

  // ------------------------------------------------------------------------------------------------ phase 1: dQ
  stage(base + D, ld, base + 2 * D, ld);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll 1
  for (int qi = 0; qi < TPW; ++qi) {
    const int qt = wave + qi * NW;
    if (qt >= nt) break;
    const int query = qt * 16 + r;
    const int qrow = min(query, S - 1);
    f32x4 qf[4], dof[4];
    float dsum = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qf[c] = *reinterpret_cast<const f32x4*>(base + (long)qrow * ld + 16 * c + 4 * g) * 0.125f;
      dof[c] = *reinterpret_cast<const f32x4*>(dobase + (long)qrow * D + 16 * c + 4 * g);
      const f32x4 of = *reinterpret_cast<const f32x4*>(obase + (long)qrow * D + 16 * c + 4 * g);
      dsum += (dof[c][0] * of[0] + dof[c][1] * of[1]) + (dof[c][2] * of[2] + dof[c][3] * of[3]);
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);   // D of query r, replicated over g
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
    float mx = kNegInfB;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = t * 16 + 4 * g + e;
        if (key >= S || (CAUSAL && key > query)) sT[t][e] = kNegInfB;
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
        const float p = exp_neg_f32(sT[t][e] - mx);  // the forward's function: identical P
        sT[t][e] = p;
        sum += p;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if (g == 0) {
      stat_m[qt * 16 + r] = mx;
      stat_l[qt * 16 + r] = inv;
      stat_d[qt * 16 + r] = dsum;
    }
    f32x4 dq[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) dq[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      if (t * 16 >= S || (CAUSAL && t > qt)) continue;  // every P of the tile is zero
      // dP^T[key][query] = sum_d V[key][d] dO[query][d]
      f32x4 dp = {0.f, 0.f, 0.f, 0.f};
      const char* vrow = smem + OFF_B + (t * 16 + r) * 256;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 vf = *reinterpret_cast<const f32x4*>(vrow + (((4 * c + g) ^ r) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[e], dof[c][e], dp, 0, 0, 0);
      }
      // dS^T = P^T * (dP^T - D); masked entries have P = 0
      f32x4 ds;
#pragma unroll
      for (int e = 0; e < 4; ++e) ds[e] = (sT[t][e] * inv) * (dp[e] - dsum);
      // dQ^T[d][query] += sum_key K[key][d] dS^T[key][query]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = t * 16 + 4 * g + e;  // key & 15 = 4 g + e
        const char* kr = smem + key * 256 + (r & 3) * 4;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const float kf = *reinterpret_cast<const float*>(kr + (((4 * n + (r >> 2)) ^ (4 * g + e)) << 4));
          dq[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf, ds[e], dq[n], 0, 0, 0);
        }
      }
    }
    if (query < S) {
      float* drow = dbase + (long)query * ld + 4 * g;
#pragma unroll
      for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(drow + 16 * n) = dq[n] * 0.125f;
    }
  }

  // --------------------------------------------------------------------------------------------- phase 2: dK, dV
  __syncthreads();  // every wave is done with K / V in LDS, and the statistics table is complete
  stage(base, ld, dobase, (long)D);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll 1
  for (int ki = 0; ki < TPW; ++ki) {
    const int kt = wave + ki * NW;
    if (kt >= nt) break;
    const int key = kt * 16 + r;
    const int krow = min(key, S - 1);
    f32x4 kf[4], vf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      kf[c] = *reinterpret_cast<const f32x4*>(base + D + (long)krow * ld + 16 * c + 4 * g);
      vf[c] = *reinterpret_cast<const f32x4*>(base + 2 * D + (long)krow * ld + 16 * c + 4 * g);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      dk[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      dv[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll 1
    for (int t = CAUSAL ? kt : 0; t < nt; ++t) {  // causal: queries before this key tile never see it
      // S[q][key] and dP[q][key] for q = 16 t + 4 g + e' (accumulator rows), key = lane & 15
      f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
      const char* qrow = smem + (t * 16 + r) * 256;
      const char* drow = smem + OFF_B + (t * 16 + r) * 256;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 qf = *reinterpret_cast<const f32x4*>(qrow + (((4 * c + g) ^ r) << 4));
        const f32x4 df = *reinterpret_cast<const f32x4*>(drow + (((4 * c + g) ^ r) << 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[e], kf[c][e], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(df[e], vf[c][e], dp, 0, 0, 0);
        }
      }
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(stat_m + t * 16 + 4 * g);
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(stat_l + t * 16 + 4 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(stat_d + t * 16 + 4 * g);
      f32x4 p, ds;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = t * 16 + 4 * g + e;
        const bool ok = q < S && key < S && (!CAUSAL || key <= q);
        p[e] = ok ? exp_neg_f32(s[e] * 0.125f - m4[e]) * l4[e] : 0.f;
        ds[e] = ok ? p[e] * (dp[e] - d4[e]) : 0.f;
      }
      // dV^T[d][key] += sum_q dO[q][d] P[q][key];   dK^T[d][key] += sum_q Q[q][d] dS[q][key]
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int q = t * 16 + 4 * g + e;  // q & 15 = 4 g + e
        const char* qr = smem + q * 256 + (r & 3) * 4;
        const char* dr = smem + OFF_B + q * 256 + (r & 3) * 4;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const int off = ((4 * n + (r >> 2)) ^ (4 * g + e)) << 4;
          const float qv = *reinterpret_cast<const float*>(qr + off);
          const float dov = *reinterpret_cast<const float*>(dr + off);
          dv[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(dov, p[e], dv[n], 0, 0, 0);
          dk[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(qv, ds[e], dk[n], 0, 0, 0);
        }
      }
    }
    if (key < S) {
      float* kout = dbase + D + (long)key * ld + 4 * g;
      float* vout = dbase + 2 * D + (long)key * ld + 4 * g;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        *reinterpret_cast<f32x4*>(kout + 16 * n) = dk[n] * 0.125f;
        *reinterpret_cast<f32x4*>(vout + 16 * n) = dv[n];
      }
    }
  }
}

template <int NKT, int NW, bool CAUSAL>
int launch_bwd_variant(const float* qkv, const float* o, const float* d_o, float* dqkv, int n_seq, int S, int heads,
                       hipStream_t st) {
  constexpr int lds = 2 * NKT * 16 * 256 + 3 * NKT * 16 * 4;
  auto kern = attn_bwd_f32_kernel<NKT, NW, CAUSAL>;
  if (lds > 64 * 1024 && raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "attention backward: cannot raise dynamic LDS");
  hipLaunchKernelGGL(kern, dim3(n_seq * heads), dim3(NW * 64), lds, st, qkv, o, d_o, dqkv, S, heads);
  FC_CHECK_LAUNCH("attention backward");
  return FC_OK;
}

}  // namespace

// qkv [n_seq * S, 3 D] (forward input), o [n_seq * S, D] (forward output), d_o [n_seq * S, D] -> dqkv [n_seq * S, 3 D]
int launch_attention_backward(int precision, const void* qkv, const void* o, const void* d_o, void* dqkv, int n_seq,
                              int S, int heads, int causal, hipStream_t stream) {
  if (n_seq <= 0) return FC_OK;
  if (precision != PREC_F32) return fail(FC_EINVAL, "attention backward: only the fp32 mode is implemented");
  if (S <= 0 || heads <= 0) return fail(FC_EINVAL, "attention backward: S=%d heads=%d", S, heads);
  if (((uintptr_t)qkv | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dqkv) & 15)
    return fail(FC_EINVAL, "attention backward: unaligned operand");
  const float *q = (const float*)qkv, *oo = (const float*)o, *dd = (const float*)d_o;
  float* dq = (float*)dqkv;
  if (S <= 32)
    return causal ? launch_bwd_variant<2, 2, true>(q, oo, dd, dq, n_seq, S, heads, stream)
                  : launch_bwd_variant<2, 2, false>(q, oo, dd, dq, n_seq, S, heads, stream);
  if (S <= 96)
    return causal ? launch_bwd_variant<6, 6, true>(q, oo, dd, dq, n_seq, S, heads, stream)
                  : launch_bwd_variant<6, 6, false>(q, oo, dd, dq, n_seq, S, heads, stream);
  if (S <= 224) {
    // (8 waves: 7 left one SIMD with a single wave)
    return causal ? launch_bwd_variant<14, 8, true>(q, oo, dd, dq, n_seq, S, heads, stream)
                  : launch_bwd_variant<14, 8, false>(q, oo, dd, dq, n_seq, S, heads, stream);
  }
  return fail(FC_EINVAL, "attention backward: sequences longer than 224 tokens are not supported (S=%d)", S);
}

}  // namespace fc

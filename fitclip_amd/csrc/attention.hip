// Fused multi-head self-attention for the CLIP towers: softmax(q k^T / sqrt(64) [+ causal mask]) v, head dim 64.
// Reads q | k | v straight out of the fused-QKV GEMM output [rows, 3D] (no transposes in HBM) and writes [rows, D].
// Reference semantics: nn.MultiheadAttention inside ResidualAttentionBlock (aligner/encoder/slip.py:364-380), additive
// causal mask of the text tower (slip.py:454-460).
//
// Sequences are short (197 visual tokens, 77 text tokens), so K and V of one (sequence, head) pair fit in LDS and the
// softmax is single pass: no online rescaling, every score of a query row is in registers at once.
//
// bf16 kernel (throughput path), one workgroup = 4 waves per (sequence, head):
//   * K tile in LDS as [key][64 d] rows of 128 B with the same 16-byte chunk swizzle as the GEMM (conflict-free
//     ds_read_b128 of the MFMA A-operand); V is stored TRANSPOSED [d][key] (two keys packed per ds_write_b32) so the
//     P.V B-operand (8 consecutive keys of one d) is two ds_read_b64.
//   * S^T = K.Q^T is computed (keys on the accumulator rows, queries on the lanes, "swapped QK^T"): the softmax of a
//     query is then a per-lane reduction over registers plus two cross-lane steps, and the exponentiated accumulator
//     registers ARE the A-operand of P.V after a bf16 pack (cdna_hip_programming.md section 3, "An accumulator tile as
//     the next MFMA's operand"): k-slot 8q+j of PV step ks maps to key 32ks + 16(j>>2) + 4q + (j&3), and the V^T
//     fragment is gathered with the same map.
//   * each wave owns 16-query tiles qt = wave, wave+4, ...; padded keys are masked to -inf, padded V rows are zero.
//
// f32 kernel (parity path): plain VALU, one thread per query row, K/V broadcast from LDS, online softmax in fp32.
#include "common.h"

namespace fc {

namespace {

constexpr float kNegInf = -__builtin_inff();

template <int NKT, bool CAUSAL>
__global__ void __launch_bounds__(256) attn_bf16_kernel(const bf16* __restrict__ qkv, bf16* __restrict__ out, int S,
                                                        int heads) {
  constexpr int NK = NKT * 16;      // padded key count (multiple of 32)
  constexpr int VS = (NK + 8) * 2;  // bytes per V^T row (pad keeps the ds_read_b64 pattern conflict-free)
  static_assert(NKT % 2 == 0, "P.V consumes key tiles in pairs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Ks = smem;
  char* Vt = smem + NK * 128;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const long ld = 3L * D;
  const bf16* base = qkv + (long)seq * S * ld + h * 64;
  const bf16* Qg = base;
  const bf16* Kg = base + D;
  const bf16* Vg = base + 2 * D;

  const bf16x8 zero8 = {};
  // ---- stage K (row-major, swizzled 16-byte chunks)
  for (int idx = tid; idx < NK * 8; idx += 256) {
    const int key = idx >> 3, c = idx & 7;
    bf16x8 v = zero8;
    if (key < S) v = *reinterpret_cast<const bf16x8*>(Kg + key * ld + c * 8);
    *reinterpret_cast<bf16x8*>(Ks + key * 128 + ((c ^ ((key >> 1) & 7)) << 4)) = v;
  }
  // ---- stage V transposed: item = (key pair p, d-chunk c)
  for (int idx = tid; idx < (NK / 2) * 8; idx += 256) {
    const int p = idx % (NK / 2), c = idx / (NK / 2);
    const int k0 = 2 * p, k1 = 2 * p + 1;
    bf16x8 v0 = zero8, v1 = zero8;
    if (k0 < S) v0 = *reinterpret_cast<const bf16x8*>(Vg + k0 * ld + c * 8);
    if (k1 < S) v1 = *reinterpret_cast<const bf16x8*>(Vg + k1 * ld + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      bf16x2 pr;
      pr[0] = v0[j];
      pr[1] = v1[j];
      *reinterpret_cast<bf16x2*>(Vt + (c * 8 + j) * VS + k0 * 2) = pr;
    }
  }
  __syncthreads();

  const int r = lane & 15, q4 = lane >> 4, f = (r >> 1) & 7;
  const int nqt = (S + 15) >> 4;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int query = qt * 16 + r;
    const int qrow = min(query, S - 1);
    bf16x8 qf[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(Qg + qrow * ld + (4 * s + q4) * 8);

    // S^T tiles: sT[t][reg] = score(key = 16t + 4*q4 + reg, query)
    f32x4 sT[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (!CAUSAL || t <= qt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + r) * 128 + (((4 * s + q4) ^ f) << 4));
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[s], acc, 0, 0, 0);
        }
      }
      sT[t] = acc;
    }
    // mask + softmax over keys (registers x tiles within the lane, then lanes l ^ 16, l ^ 32)
    float mx = kNegInf;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = t * 16 + 4 * q4 + e;
        const bool ok = key < S && (!CAUSAL || key <= query);
        const float v = ok ? sT[t][e] * 0.125f : kNegInf;
        sT[t][e] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = __expf(sT[t][e] - mx);
        sT[t][e] = p;
        sum += p;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);

    // O = P.V : A-operand = P (rows = queries on the lanes), B-operand = V^T fragment
    f32x4 o[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKT / 2; ++ks) {
      if (CAUSAL && 2 * ks > qt) continue;
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = static_cast<bf16>(sT[2 * ks][e]);
        pf[4 + e] = static_cast<bf16>(sT[2 * ks + 1][e]);
      }
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const char* vrow = Vt + (n * 16 + r) * VS + (ks * 32 + 4 * q4) * 2;
        const bf16x4 v0 = *reinterpret_cast<const bf16x4*>(vrow);
        const bf16x4 v1 = *reinterpret_cast<const bf16x4*>(vrow + 32);
        const bf16x8 vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
        o[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf, o[n], 0, 0, 0);
      }
    }
    // o[n][e] = O(query' = 16 qt + 4 q4 + e, d = 16 n + r); the row sums live on lane (query' & 15)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float inv = 1.f / __shfl(sum, 4 * q4 + e, 64);
      const int qo = qt * 16 + 4 * q4 + e;
      if (qo < S) {
        bf16* orow = out + ((long)seq * S + qo) * D + h * 64 + r;
#pragma unroll
        for (int n = 0; n < 4; ++n) orow[n * 16] = static_cast<bf16>(o[n][e] * inv);
      }
    }
  }
}

// f32 parity kernel: thread per query, K/V rows broadcast from LDS.
template <bool CAUSAL>
__global__ void __launch_bounds__(256) attn_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S,
                                                       int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = reinterpret_cast<float*>(smem);
  float* Vs = Ks + S * 64;
  const int tid = threadIdx.x;
  const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
  const int D = heads * 64;
  const long ld = 3L * D;
  const float* base = qkv + (long)seq * S * ld + h * 64;
  for (int idx = tid; idx < S * 16; idx += 256) {
    const int key = idx >> 4, c = (idx & 15) * 4;
    *reinterpret_cast<f32x4*>(Ks + key * 64 + c) = *reinterpret_cast<const f32x4*>(base + D + key * ld + c);
    *reinterpret_cast<f32x4*>(Vs + key * 64 + c) = *reinterpret_cast<const f32x4*>(base + 2 * D + key * ld + c);
  }
  __syncthreads();
  for (int query = tid; query < S; query += 256) {
    float qv[64], o[64];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(base + query * ld + c * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qv[c * 4 + e] = t[e] * 0.125f;
        o[c * 4 + e] = 0.f;
      }
    }
    float m = kNegInf, l = 0.f;
    const int kend = CAUSAL ? query + 1 : S;
    for (int key = 0; key < kend; ++key) {
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
      const float s = (s0 + s1) + (s2 + s3);
      const float mn = fmaxf(m, s);
      const float a = expf(m - mn);  // exp(-inf) = 0 on the first key
      const float p = expf(s - mn);
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
}

template <int NKT>
int launch_bf16(const void* qkv, void* out, int n_seq, int S, int heads, int causal, hipStream_t st) {
  constexpr int NK = NKT * 16;
  constexpr int lds = NK * 128 + 64 * (NK + 8) * 2;
  const dim3 grid(n_seq * heads), block(256);
  if (causal)
    hipLaunchKernelGGL((attn_bf16_kernel<NKT, true>), grid, block, lds, st, (const bf16*)qkv, (bf16*)out, S, heads);
  else
    hipLaunchKernelGGL((attn_bf16_kernel<NKT, false>), grid, block, lds, st, (const bf16*)qkv, (bf16*)out, S, heads);
  FC_CHECK_LAUNCH("attention(bf16)");
  return FC_OK;
}

}  // namespace

int launch_attention(int precision, const void* qkv, void* out, int n_seq, int S, int heads, int causal,
                     hipStream_t stream) {
  if (n_seq <= 0) return FC_OK;
  if (S <= 0 || heads <= 0) return fail(FC_EINVAL, "attention: S=%d heads=%d", S, heads);
  if (((uintptr_t)qkv | (uintptr_t)out) & 15) return fail(FC_EINVAL, "attention: unaligned operand");
  if (precision == PREC_BF16) {
    if (S <= 32) return launch_bf16<2>(qkv, out, n_seq, S, heads, causal, stream);
    if (S <= 96) return launch_bf16<6>(qkv, out, n_seq, S, heads, causal, stream);
    if (S <= 224) return launch_bf16<14>(qkv, out, n_seq, S, heads, causal, stream);
    return fail(FC_EINVAL, "attention(bf16): sequence length %d > 224 not supported", S);
  }
  const int lds = S * 64 * 4 * 2;
  if (lds > 160 * 1024) return fail(FC_EINVAL, "attention(f32): sequence length %d does not fit LDS", S);
  auto k0 = attn_f32_kernel<false>;
  auto k1 = attn_f32_kernel<true>;
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return fail(FC_ELAUNCH, "attention(f32): cannot raise dynamic LDS");
    configured = true;
  }
  const dim3 grid(n_seq * heads), block(256);
  if (causal)
    hipLaunchKernelGGL(k1, grid, block, lds, stream, (const float*)qkv, (float*)out, S, heads);
  else
    hipLaunchKernelGGL(k0, grid, block, lds, stream, (const float*)qkv, (float*)out, S, heads);
  FC_CHECK_LAUNCH("attention(f32)");
  return FC_OK;
}

}  // namespace fc

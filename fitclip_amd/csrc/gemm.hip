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
#include "common.h"

namespace fc {

namespace {

constexpr int ROWB = 128;  // bytes of K per LDS tile row

__device__ __forceinline__ float quick_gelu_fast(float x) { return x * __frcp_rn(1.f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_exact(float x) { return x / (1.f + expf(-1.702f * x)); }

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

template <typename T, int BM, int BN, int WM, int WN, int EPI>
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
  using FragT = typename Frag<T>::type;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware, bijective block -> tile map
  const int tilesN = (g.N + BN - 1) / BN;
  const int nwg = gridDim.x, orig = blockIdx.x;
  const int xcd = orig & 7, qd = nwg >> 3, rd = nwg & 7;
  const int t = (xcd < rd ? xcd * (qd + 1) : rd * (qd + 1) + (xcd - rd) * qd) + (orig >> 3);
  const int m0 = (t / tilesN) * BM, n0 = (t % tilesN) * BN;

  // ---- per-lane staging sources
  const char* src[LPW];
  {
    const int rin = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int row = (wave + i * NW) * 8 + rin;  // row in the stacked [A tile ; W tile] image
      const int chunk = pc ^ ((row >> 1) & 7);
      if (row < BM) {
        const int gr = min(m0 + row, g.M - 1);
        src[i] = reinterpret_cast<const char*>(g.A) + ((size_t)gr * g.lda) * sizeof(T) + chunk * 16;
      } else {
        const int gr = min(n0 + row - BM, g.N - 1);
        src[i] = reinterpret_cast<const char*>(g.W) + ((size_t)gr * g.ldw) * sizeof(T) + chunk * 16;
      }
    }
  }
  auto stage_load = [&](int stage, int kt) {
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(src[i] + (size_t)kt * ROWB),
          (__attribute__((address_space(3))) void*)(smem + stage * STAGE + (wave + i * NW) * 1024),
          16, 0, 0);
    }
  };

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int r = lane & 15, q = lane >> 4, f = (r >> 1) & 7;
  const int a_base = (wm * TM + r) * ROWB;
  const int b_base = BM * ROWB + (wn * TN + r) * ROWB;

  const int nk = g.K / BKE;
  stage_load(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage_load((kt + 1) & 1, kt + 1);
    const char* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int c = (sizeof(T) == 2) ? (4 * s + q) : (q + 4 * s);
      const int off = (c ^ f) * 16;
      FragT xa[FM], wb[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) xa[i] = *reinterpret_cast<const FragT*>(st + a_base + i * 16 * ROWB + off);
#pragma unroll
      for (int j = 0; j < FN; ++j) wb[j] = *reinterpret_cast<const FragT*>(st + b_base + j * 16 * ROWB + off);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) mma<T>(wb[j], xa[i], acc[i][j]);
    }
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
        v += *reinterpret_cast<const f32x4*>(g.bias + n);
        v += *reinterpret_cast<const f32x4*>(p);
        store4<float>(p, v);
      } else {
        v += *reinterpret_cast<const f32x4*>(g.bias + n);
        if constexpr (EPI == EPI_GELU_T) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (sizeof(T) == 2) ? quick_gelu_fast(v[e]) : quick_gelu_exact(v[e]);
        }
        store4<T>(reinterpret_cast<T*>(g.C) + orow * g.ldc + n, v);
      }
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN, int EPI>
int launch_one(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * (BM + BN) * ROWB;
  auto kern = gemm_kernel<T, BM, BN, WM, WN, EPI>;
  static bool configured = false;  // per instantiation; the attribute is per function, not per device state we mutate
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) !=
        hipSuccess)
      return fail(FC_ELAUNCH, "gemm: cannot raise dynamic LDS to %d bytes", lds);
    configured = true;
  }
  const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(WM * WN * 64), lds, stream, a);
  FC_CHECK_LAUNCH("gemm");
  return FC_OK;
}

template <typename T, int EPI>
int launch_tile(const GemmArgs& a, int tile, hipStream_t stream) {
  if (tile == 0) {
    const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
    tile = t256 >= 512 ? 2 : 1;
  }
  if (tile == 2) return launch_one<T, 256, 256, 2, 4, EPI>(a, stream);
  return launch_one<T, 128, 128, 2, 2, EPI>(a, stream);
}

template <typename T>
int launch_epi(int epi, const GemmArgs& a, int tile, hipStream_t stream) {
  switch (epi) {
    case EPI_BIAS_T: return launch_tile<T, EPI_BIAS_T>(a, tile, stream);
    case EPI_GELU_T: return launch_tile<T, EPI_GELU_T>(a, tile, stream);
    case EPI_RESID_F32: return launch_tile<T, EPI_RESID_F32>(a, tile, stream);
    case EPI_PATCH_F32: return launch_tile<T, EPI_PATCH_F32>(a, tile, stream);
    case EPI_STORE_F32: return launch_tile<T, EPI_STORE_F32>(a, tile, stream);
  }
  return fail(FC_EINVAL, "gemm: unknown epilogue %d", epi);
}

}  // namespace

int launch_gemm(int precision, int epilogue, const GemmArgs& a, int tile, hipStream_t stream) {
  const int esz = precision == PREC_BF16 ? 2 : 4;
  const int bke = ROWB / esz;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fail(FC_EINVAL, "gemm: empty problem %dx%dx%d", a.M, a.N, a.K);
  if (a.K % bke) return fail(FC_EINVAL, "gemm: K=%d must be a multiple of %d", a.K, bke);
  if (a.N % 4) return fail(FC_EINVAL, "gemm: N=%d must be a multiple of 4", a.N);
  if ((a.lda * esz) % 16 || (a.ldw * esz) % 16 || a.lda < a.K || a.ldw < a.K)
    return fail(FC_EINVAL, "gemm: lda=%d / ldw=%d must cover K and keep rows 16-byte aligned", a.lda, a.ldw);
  if (a.ldc % 4 || a.ldc < a.N) return fail(FC_EINVAL, "gemm: ldc=%d", a.ldc);
  if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.C) & 15) return fail(FC_EINVAL, "gemm: unaligned operand");
  if (epilogue != EPI_STORE_F32 && epilogue != EPI_PATCH_F32 && !a.bias) return fail(FC_EINVAL, "gemm: bias missing");
  if (epilogue == EPI_PATCH_F32 && (!a.aux || a.P <= 0)) return fail(FC_EINVAL, "gemm: patch epilogue needs pos/P");
  if (tile < 0 || tile > 2) return fail(FC_EINVAL, "gemm: tile=%d", tile);
  return precision == PREC_BF16 ? launch_epi<bf16>(epilogue, a, tile, stream)
                                : launch_epi<float>(epilogue, a, tile, stream);
}

}  // namespace fc

// Host-side dispatch of the MFMA GEMM (kernel: gemm_kernel.h).
#include <cstdlib>

#include "gemm_kernel.h"

#include <algorithm>

namespace fc {

namespace {

template <typename T, int BM, int BN, int WM, int WN, int EPI>
int launch_one(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = 2 * (BM + BN) * ROWB;
  auto kern = gemm_kernel<T, BM, BN, WM, WN, EPI>;
  if (raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "gemm: cannot raise dynamic LDS to %d bytes", lds);
  const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(WM * WN * 64), lds, stream, a);
  FC_CHECK_LAUNCH("gemm");
  return FC_OK;
}

int num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    cus = n;
  }
  return cus;
}

// persistent, pipelined kernel: one workgroup per CU walks the tiles (BM = 256; 128 for the tail launch below)
template <typename T, int EPI, int SCHED, int BM = 256>
int launch_pipelined_sched(const GemmArgs& a, hipStream_t stream) {
  constexpr int BN = 256, WM = 2, WN = 4;
  constexpr int lds = 2 * (BM + BN) * ROWB + WM * WN * 2048 + 2048;  // two stages + a 2 KiB output patch per wave + bias
  auto kern = gemm_pipelined_kernel<T, BM, BN, WM, WN, EPI, 0, 1, SCHED>;
  if (raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "gemm: cannot raise dynamic LDS to %d bytes", lds);
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(std::min(tiles, num_cus())), dim3(WM * WN * 64), lds, stream, a);
  FC_CHECK_LAUNCH("gemm(pipelined)");
  return FC_OK;
}

// LDS-DMA issue schedule (gemm_kernel.h: piece_slot).  Lab (tools/gemm_lab, profiles/r01_lab17_sched.log): the K = 768
// shapes gain 6-8 % from spreading all pieces over the first two MFMA groups of the next K-step (8), the K = 3072
// c_proj, which streams its activations from HBM, 4 % from requesting them early (2: activations in the hand-over,
// weights after group 0).  In situ (kernel traces of bench.py per schedule, tools/sched_ab.sh,
// profiles/r01_sched_in_situ.txt) only c_fc keeps the gain of 8 (433 vs 448 / 462 us for 2 / 0); QKV is indifferent
// between 2 and 8, out_proj and c_proj are fastest with 2.  Hence: 8 for the QuickGELU GEMM, 2 otherwise.
// FITCLIP_GEMM_SCHED=0|2|8 overrides (A/B runs); the result does not depend on the schedule.
template <typename T, int EPI>
int launch_pipelined_full(const GemmArgs& a, hipStream_t stream) {
  static const int forced = [] {
    const char* e = getenv("FITCLIP_GEMM_SCHED");
    return e ? atoi(e) : -1;
  }();
  const int sched = forced >= 0 ? forced : (EPI == EPI_GELU_T ? 8 : 2);
  switch (sched) {
    case 2: return launch_pipelined_sched<T, EPI, 2>(a, stream);
    case 8: return launch_pipelined_sched<T, EPI, 8>(a, stream);
    default: return launch_pipelined_sched<T, EPI, 0>(a, stream);
  }
}

// fp32 is MFMA-bound on every CU, so a partial last round of 256 x 256 tiles costs a whole round (api.hip plans the
// passes of a big batch around that; a small batch - the reference evaluates 32 clips = 128 frames at a time - cannot be
// planned).  Here the row panels that fill whole rounds go to one launch and the REST to a second launch of 128 x 256
// tiles when its half-size rounds are cheaper (c_proj / out_proj of 128 frames: 297 tiles = 2 rounds -> 591 half tiles =
// 3 half rounds; QKV of 385 frames: 11 -> 9 + 1.6 rounds).  Same K order per output element: results are bit-identical.
template <typename T, int EPI>
int launch_pipelined(const GemmArgs& a, hipStream_t stream) {
  if constexpr (sizeof(T) == 4) {
    static const bool no_tail = getenv("FITCLIP_GEMM_NO_TAIL") != nullptr;  // A/B switch
    const int cus = num_cus(), tilesN = (a.N + 255) / 256, panels = (a.M + 255) / 256;
    int g = cus, t = tilesN;
    while (t) { const int r = g % t; g = t; t = r; }  // gcd
    const int step = cus / g;                          // panels per whole number of rounds
    const int p0 = panels / step * step;
    const long rem_tiles = (long)(panels - p0) * tilesN;
    if (rem_tiles > 0 && !no_tail) {
      const int rem_rows = a.M - p0 * 256;
      const long half_tiles = (long)((rem_rows + 127) / 128) * tilesN;
      const double cost256 = (double)((rem_tiles + cus - 1) / cus);
      const double cost128 = 0.54 * (double)((half_tiles + cus - 1) / cus);  // a 128-row tile: half the work at ~92 % of the rate
      if (cost128 < cost256) {
        if (p0 > 0) {
          GemmArgs head = a;
          head.M = p0 * 256;
          const int rc = launch_pipelined_full<T, EPI>(head, stream);
          if (rc != FC_OK) return rc;
        }
        GemmArgs tail = a;
        const size_t r0 = (size_t)p0 * 256;
        tail.M = rem_rows;
        tail.A = static_cast<const char*>(a.A) + r0 * a.lda * sizeof(T);
        tail.C = static_cast<char*>(a.C) + r0 * a.ldc * sizeof(T);
        if constexpr (EPI == EPI_DGELU_T) tail.aux = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.aux) + r0 * a.ldc * sizeof(T));
        return launch_pipelined_sched<T, EPI, 0, 128>(tail, stream);
      }
    }
  }
  return launch_pipelined_full<T, EPI>(a, stream);
}

template <typename T>
bool pipelined_ok(const GemmArgs& a) {
  const int esz = (int)sizeof(T);
  return a.K / (ROWB / esz) >= 3 && a.N % 8 == 0 && (a.ldc * esz) % 16 == 0 && a.ldc % 4 == 0 && a.bias != nullptr &&
         ((uintptr_t)a.bias & 15) == 0 && (size_t)256 * a.lda * esz < (1ull << 32) &&
         (size_t)a.N * a.ldw * esz < (1ull << 32);
}

template <typename T>
int resolve_tile(int epi, const GemmArgs& a, int tile) {
  if (tile != 0) return tile;
  const bool has_pipelined = epi == EPI_BIAS_T || epi == EPI_GELU_T || epi == EPI_RESID_F32 || (epi == EPI_DGELU_T && sizeof(T) == 4);
  const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  if (has_pipelined && t256 >= 192 && pipelined_ok<T>(a)) return 3;
  return t256 >= 512 ? 2 : 1;
}

// tile: 0 = auto, 1 = 128x128 plain, 2 = 256x256 plain, 3 = 256x256 persistent + pipelined (BIAS_T / GELU_T only)
template <typename T, int EPI>
int launch_tile(const GemmArgs& a, int tile, hipStream_t stream) {
  constexpr bool kHasPipelined = EPI == EPI_BIAS_T || EPI == EPI_GELU_T || EPI == EPI_RESID_F32 || (EPI == EPI_DGELU_T && sizeof(T) == 4);
  tile = resolve_tile<T>(EPI, a, tile);
  if (tile == 3) {
    if constexpr (kHasPipelined) {
      if (!pipelined_ok<T>(a)) return fail(FC_EINVAL, "gemm: shape not supported by the pipelined kernel");
      return launch_pipelined<T, EPI>(a, stream);
    } else {
      return fail(FC_EINVAL, "gemm: the pipelined kernel has no epilogue %d", EPI);
    }
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
    case EPI_DGELU_T: return launch_tile<T, EPI_DGELU_T>(a, tile, stream);
    case EPI_BIAS_F32:
    case EPI_GELU_X3:
    case EPI_RESID3_F32: return fail(FC_EINVAL, "gemm: epilogue %d belongs to the three-plane split-fp32 GEMM (fc_gemm_split3)", epi);
  }
  return fail(FC_EINVAL, "gemm: unknown epilogue %d", epi);
}

}  // namespace

int gemm_resolved_tile(int precision, int epilogue, const GemmArgs& a, int tile) {
  return precision == PREC_BF16 ? resolve_tile<bf16>(epilogue, a, tile) : resolve_tile<float>(epilogue, a, tile);
}

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
  if (epilogue != EPI_STORE_F32 && epilogue != EPI_PATCH_F32 && epilogue != EPI_DGELU_T && !a.bias)
    return fail(FC_EINVAL, "gemm: bias missing");
  if (epilogue == EPI_DGELU_T && (!a.aux || ((uintptr_t)a.aux & 15))) return fail(FC_EINVAL, "gemm: dgelu epilogue needs aux");
  if (epilogue == EPI_PATCH_F32 && (!a.aux || a.P <= 0)) return fail(FC_EINVAL, "gemm: patch epilogue needs pos/P");
  if (a.gR > 0) {
    const int G = a.gP > 0 ? a.gR / a.gP : 0;
    if (epilogue != EPI_PATCH_F32 || precision != PREC_F32 || a.gP <= 0 || a.gP % 4 || a.gR % a.gP || a.gR % 4 ||
        G * G != a.P || a.K != 3 * a.gP * a.gP)
      return fail(FC_EINVAL, "gemm: patch gather needs the f32 patch-embed epilogue, patch %% 4 == 0 and K = 3 p^2");
  }
  if (tile < 0 || tile > 3) return fail(FC_EINVAL, "gemm: tile=%d", tile);
  return precision == PREC_BF16 ? launch_epi<bf16>(epilogue, a, tile, stream)
                                : launch_epi<float>(epilogue, a, tile, stream);
}

}  // namespace fc

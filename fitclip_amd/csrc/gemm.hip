// Host-side dispatch of the MFMA GEMM (kernel: gemm_kernel.h).
#include "gemm_kernel.h"

#include <algorithm>
#ifdef FITCLIP_LAB
#include <cstdlib>
#endif

namespace fc {

namespace {

template <typename T, int BM, int BN, int WM, int WN, int EPI, int NSTG = 2>
int launch_one(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = NSTG * (BM + BN) * ROWB;
  auto kern = gemm_kernel<T, BM, BN, WM, WN, EPI, 0, NSTG>;
  if (raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "gemm: cannot raise dynamic LDS to %d bytes", lds);
  const int tilesM = (a.M + BM - 1) / BM, tilesN = (a.N + BN - 1) / BN;
  hipLaunchKernelGGL(kern, dim3(tilesM * tilesN), dim3(WM * WN * 64), lds, stream, a);
  FC_CHECK_LAUNCH("gemm");
  return FC_OK;
}

// persistent, pipelined kernel: one workgroup per CU walks its tiles (HT > 0: head panels g.hp, then a tail of HT x 64-row tiles)
template <typename T, int EPI, int SCHED, int HT>
int launch_pipelined_ht(const GemmArgs& a, hipStream_t stream) {
  constexpr int BM = 256, BN = 256, WM = 2, WN = 4;
  constexpr int lds = 2 * (BM + BN) * ROWB + WM * WN * 2048 + 2048;  // two stages + a 2 KiB output patch per wave + bias
  auto kern = gemm_pipelined_kernel<T, BM, BN, WM, WN, EPI, 0, 1, SCHED, HT>;
  if (raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "gemm: cannot raise dynamic LDS to %d bytes", lds);
  const long tilesN = (a.N + BN - 1) / BN;
  long work = (long)((a.M + BM - 1) / BM) * tilesN;
  if constexpr (HT > 0) {
    const long rows = a.M - (long)a.hp * BM;
    work = std::max((long)a.hp * tilesN, (rows + 64 * HT - 1) / (64 * HT) * tilesN);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long>(work, device_cus())), dim3(WM * WN * 64), lds, stream, a);
  FC_CHECK_LAUNCH("gemm(pipelined)");
  return FC_OK;
}

// fp32 is MFMA-bound on every CU, so a partial last round of 256 x 256 tiles costs a whole round (a small batch - the
// reference evaluates 32 clips = 128 frames at a time - cannot be planned around that: out_proj / c_proj have 297 tiles for
// 256 CUs).  The rows are cut in a head of `hp` panels, whose tiles fill whole rounds, and a tail walked as tiles of
// ht x 64 rows by the same launch (gemm_kernel.h); (hp, ht) minimise the modelled time: rounds of head tiles + rounds of tail
// tiles x the relative cost of a tile of that height (proportional to the height, plus what a lower tile hides less well: its
// weight tile is staged for fewer rows, its prologue and epilogue are paid per tile).  Same K order per output element: results
// are bit-identical whatever the cut.  bf16 keeps whole 256-row tiles: there the chip is power- and bandwidth-limited and a
// partial round costs nothing measurable.
struct TailPlan { int hp, ht; };
// Relative cost of a tail tile of ht x 64 rows against a 256-row tile, fitted to forced cuts of the block shapes at 128 frames
// (tools/bench_gemm.py --tiles 4,5,6,7; M = 25 216): a tile takes ~1.7 us per K-step and 64-row unit, plus ~0.26 us per K-step
// and ~28 us per tile that do not shrink with its height (hand-over, weight staging, the exposed ends of the tail phase):
//   K = 768 (24 K-steps): 0.46 / 0.71 / 0.96 of a full tile for 1 / 2 / 3 units;  K = 3072: 0.33 / 0.58 / 0.83.
double tail_tile_cost(int ht, int nk) { return ht / 4.0 + 0.04 + 4.0 / std::max(nk, 1); }
TailPlan plan_tail(int M, int tilesN, int nk, int cus) {
  const int panels = (M + 255) / 256;
  const long full_rounds = (long)panels * tilesN / cus;
  TailPlan best{panels, 0};
  double best_cost = (double)(((long)panels * tilesN + cus - 1) / cus);
  for (long r = full_rounds; r >= std::max(0L, full_rounds - 1); --r) {
    const int hp = (int)std::min<long>(panels, r * cus / tilesN);
    const long rows = (long)M - (long)hp * 256;
    if (rows <= 0) continue;
    const double head = (double)(((long)hp * tilesN + cus - 1) / cus);
    for (int ht = 1; ht <= 3; ++ht) {
      const long tiles = (rows + 64 * ht - 1) / (64 * ht) * tilesN;
      const double c = head + (double)((tiles + cus - 1) / cus) * tail_tile_cost(ht, nk);
      if (c < best_cost - 1e-9) best_cost = c, best = TailPlan{hp, ht};
    }
  }
  return best;
}

// LDS-DMA issue schedule (gemm_kernel.h: piece_slot).  Lab (tools/gemm_lab, profiles/r01_lab17_sched.log): the K = 768
// shapes gain 6-8 % from spreading all pieces over the first two MFMA groups of the next K-step (8), the K = 3072
// c_proj, which streams its activations from HBM, 4 % from requesting them early (2: activations in the hand-over,
// weights after group 0).  In situ (kernel traces of bench.py per schedule, tools/sched_ab.sh,
// profiles/r01_sched_in_situ.txt) only c_fc keeps the gain of 8 (433 vs 448 / 462 us for 2 / 0); QKV is indifferent
// between 2 and 8, out_proj and c_proj are fastest with 2.  Hence: 8 for the QuickGELU GEMM, 2 otherwise.  The result does
// not depend on the schedule.
// forced_ht: -1 = the planned cut; 0 = whole 256-row tiles only; 1..3 = a tail of that height behind the largest head of whole
// rounds (tests: every instantiation against the plain kernels)
template <typename T, int EPI>
int launch_pipelined(const GemmArgs& a, int forced_ht, hipStream_t stream) {
  constexpr int SCHED = EPI == EPI_GELU_T ? 8 : 2;
  if constexpr (sizeof(T) == 4) {
    const int panels = (a.M + 255) / 256, tilesN = (a.N + 255) / 256, cus = device_cus();
    TailPlan p = plan_tail(a.M, tilesN, a.K / (ROWB / 4), cus);
    if (forced_ht == 0) p = TailPlan{panels, 0};
    if (forced_ht > 0) p = TailPlan{(int)std::min<long>(panels - 1, (long)panels * tilesN / cus * cus / tilesN), forced_ht};
    GemmArgs b = a;
    b.hp = p.hp;
#ifdef FITCLIP_LAB
    static const int lab_order = [] { const char* e = getenv("FITCLIP_LAB_GEMM_ORDER"); return e ? atoi(e) : 0; }();
    b.order = lab_order;
#endif
    switch (p.ht) {
      case 1: return launch_pipelined_ht<T, EPI, SCHED, 1>(b, stream);
      case 2: return launch_pipelined_ht<T, EPI, SCHED, 2>(b, stream);
      case 3: return launch_pipelined_ht<T, EPI, SCHED, 3>(b, stream);
      default: return launch_pipelined_ht<T, EPI, SCHED, 0>(b, stream);
    }
  }
  return launch_pipelined_ht<T, EPI, SCHED, 0>(a, stream);
}

template <typename T>
bool pipelined_ok(const GemmArgs& a, int epi = EPI_BIAS_T) {
  const int esz = (int)sizeof(T);
  if (epi == EPI_PATCH_F32) {  // the gathering loader: fp32 frames, patch size 8 / 16 / 32 (gemm_kernel.h, kPatch)
    if (esz != 4 || a.gR <= 0 || !(a.gP == 8 || a.gP == 16 || a.gP == 32) || a.bias != nullptr ||
        (size_t)6 * a.gR * a.gR * 4 >= (1ull << 31))
      return false;
  } else if (a.bias == nullptr || ((uintptr_t)a.bias & 15) != 0 || (size_t)256 * a.lda * esz >= (1ull << 32)) {
    return false;
  }
  return a.K / (ROWB / esz) >= 3 && a.N % 8 == 0 && (a.ldc * esz) % 16 == 0 && a.ldc % 4 == 0 &&
         (size_t)a.N * a.ldw * esz < (1ull << 32);
}

constexpr bool small_ring_epilogue(int epi) { return epi == EPI_BIAS_T || epi == EPI_GELU_T || epi == EPI_RESID_F32; }

template <typename T>
int resolve_tile(int epi, const GemmArgs& a, int tile) {
  if (tile != 0) return tile;
  const bool has_pipelined = epi == EPI_BIAS_T || epi == EPI_GELU_T || epi == EPI_RESID_F32 ||
                             ((epi == EPI_DGELU_T || epi == EPI_PATCH_F32) && sizeof(T) == 4);
  const long t256 = (long)((a.M + 255) / 256) * ((a.N + 255) / 256);
  if (has_pipelined && t256 >= 192 && pipelined_ok<T>(a, epi)) return 3;
  // small fp32 block GEMMs (the text tower of an eval batch: 32 captions = 2464 rows, the N = 512 projections 80 tiles of
  // 128 x 128 for 256 CUs): 64 x 64 tiles on a four-stage ring (gemm_kernel.h, NSTG) while the 128 x 128 tiles would not fill
  // 2.5 rounds.  tools/text_gemm_probe.py, us per launch 128 x 128 -> ring: 32 captions out_proj 39 -> 22, c_proj 136 -> 76, c_fc
  // 75 -> 55, QKV 43 -> 44; 128 captions out_proj 76 -> 53, c_proj 267 -> 199 (QKV / c_fc run on the pipelined kernel there)
  const long t128 = (long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  if (sizeof(T) == 4 && small_ring_epilogue(epi) && 2 * t128 < 5 * (long)device_cus()) return 8;
  return t256 >= 512 ? 2 : 1;
}

// tile: 0 = auto, 1 = 128x128 plain, 2 = 256x256 plain, 3 = 256x256 persistent + pipelined (the block epilogues); 4..7 = the
// pipelined kernel with its fp32 row cut forced: 4 = whole tiles only, 5..7 = a tail of 1..3 x 64-row tiles; 8 = 64 x 64 tiles on
// a four-stage ring (fp32 block epilogues, small problems)
template <typename T, int EPI>
int launch_tile(const GemmArgs& a, int tile, hipStream_t stream) {
  const int forced_ht = tile >= 4 && tile <= 7 ? tile - 4 : -1;
  if (tile >= 4 && tile <= 7) tile = 3;
  constexpr bool kHasPipelined = EPI == EPI_BIAS_T || EPI == EPI_GELU_T || EPI == EPI_RESID_F32 ||
                                 ((EPI == EPI_DGELU_T || EPI == EPI_PATCH_F32) && sizeof(T) == 4);
  tile = resolve_tile<T>(EPI, a, tile);
  if (tile == 3) {
    if constexpr (kHasPipelined) {
      if (!pipelined_ok<T>(a, EPI)) return fail(FC_EINVAL, "gemm: shape not supported by the pipelined kernel");
      return launch_pipelined<T, EPI>(a, forced_ht, stream);
    } else {
      return fail(FC_EINVAL, "gemm: the pipelined kernel has no epilogue %d", EPI);
    }
  }
  if (tile == 2) return launch_one<T, 256, 256, 2, 4, EPI>(a, stream);
  if (tile == 8) {
    if constexpr (sizeof(T) == 4 && small_ring_epilogue(EPI)) return launch_one<T, 64, 64, 2, 2, EPI, 4>(a, stream);
    else return fail(FC_EINVAL, "gemm: the 64 x 64 ring kernel serves the fp32 block epilogues");
  }
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
    case EPI_RANKS_I32: return fail(FC_EINVAL, "gemm: the ranks epilogue belongs to fc_similarity_ranks");
    case EPI_BIAS_F32:
    case EPI_GELU_X3:
    case EPI_RESID3_F32: return fail(FC_EINVAL, "gemm: epilogue %d belongs to the three-plane split-fp32 GEMM (fc_gemm_split3)", epi);
  }
  return fail(FC_EINVAL, "gemm: unknown epilogue %d", epi);
}

}  // namespace

void gemm_tail_plan(int M, int N, int K, int* head_panels, int* tail_units) {
  const TailPlan p = plan_tail(M, (N + 255) / 256, K / (ROWB / 4), device_cus());
  *head_panels = p.hp;
  *tail_units = p.ht;
}

// Ranks of the target columns of alpha * T @ V^T without the matrix (gemm_kernel.h, EPI_RANKS_I32): row blocks of 128 texts x
// groups of `ctw` column tiles, enough workgroups for two per CU; the integer counts of a row's column groups meet in `ranks`
// by atomicAdd (integers: the order does not matter), which is zeroed first on the same stream.
int launch_similarity_ranks(const float* T, const float* V, int nt, int nv, int dim, float alpha, int target_offset,
                            const int32_t* targets, int32_t* ranks, hipStream_t stream) {
  if (nt == 0) return FC_OK;
  if (nt < 0 || nv <= 0 || dim <= 0 || !T || !V || !ranks) return fail(FC_EINVAL, "similarity_ranks: bad argument");
  if (!targets && (target_offset < 0 || target_offset + nt > nv))   // (the same contract as fc_ranks)
    return fail(FC_EINVAL, "similarity_ranks: rows=%d cols=%d offset=%d", nt, nv, target_offset);
  if (dim % 32) return fail(FC_EINVAL, "similarity_ranks: dim=%d must be a multiple of 32", dim);
  if (((uintptr_t)T | (uintptr_t)V) & 15) return fail(FC_EINVAL, "similarity_ranks: unaligned operand");
  constexpr int B = 128;
  constexpr int lds = 2 * (B + B) * ROWB;
  auto kern = gemm_kernel<float, B, B, 2, 2, EPI_RANKS_I32>;
  if (raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "similarity_ranks: cannot raise dynamic LDS to %d bytes", lds);
  if (hipMemsetAsync(ranks, 0, (size_t)nt * sizeof(int32_t), stream) != hipSuccess)
    return fail(FC_ELAUNCH, "similarity_ranks: hipMemsetAsync failed");
  GemmArgs a{};
  a.A = T; a.W = V; a.C = ranks; a.alpha = alpha; a.M = nt; a.N = nv; a.K = dim; a.lda = dim; a.ldw = dim; a.ldc = 1;
  a.targets = targets; a.tgt_off = target_offset;
  const int tilesM = (nt + B - 1) / B, tilesN = (nv + B - 1) / B;
  const int splits = std::max(1, std::min(tilesN, (2 * device_cus() + tilesM - 1) / tilesM));
  a.ctw = (tilesN + splits - 1) / splits;
  hipLaunchKernelGGL(kern, dim3(tilesM * ((tilesN + a.ctw - 1) / a.ctw)), dim3(256), lds, stream, a);
  FC_CHECK_LAUNCH("similarity_ranks");
  return FC_OK;
}

int gemm_resolved_tile(int precision, int epilogue, const GemmArgs& a, int tile) {
  if (tile >= 4 && tile <= 7) tile = 3;
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
  if (tile < 0 || tile > 8) return fail(FC_EINVAL, "gemm: tile=%d", tile);
  return precision == PREC_BF16 ? launch_epi<bf16>(epilogue, a, tile, stream)
                                : launch_epi<float>(epilogue, a, tile, stream);
}

}  // namespace fc

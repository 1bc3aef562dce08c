// Host-side dispatch of the three-plane split-fp32 GEMM (kernel: gemm_split3.h) and the fp32 -> x3 row converter.
#include "gemm_split3.h"

#include <algorithm>
#ifdef FITCLIP_LAB
#include <cstdlib>
#endif

namespace fc {

namespace {

template <int EPI, int RW = 4, int NTA = 0, int RR = 0>
int launch_variant(const GemmArgs& a, hipStream_t stream) {
  constexpr int lds = 3 * 512 * 96 + 2048;  // three stages (the epilogue borrows the released one) + two bias slices
  // SPREAD = 1: the LDS-DMA pieces of a K-step are issued two at a time behind MFMA groups of the step before (lab: 4 - 8 % over
  // one burst of six per wave at the hand-over - with the bursts the L1's pending-miss queue fills, the TA stalls, and a wave
  // stuck on a DMA instruction issues no MFMAs: TCP_PENDING_STALL_CYCLES 24 % of the launch, tools/split3_lab + rocprofv3)
  auto kern = gemm_split3_kernel<EPI, 0, 1, RW, NTA, RR>;
  if (raise_dynamic_lds(reinterpret_cast<const void*>(kern), lds) != hipSuccess)
    return fail(FC_ELAUNCH, "gemm_split3: cannot raise dynamic LDS to %d bytes", lds);
  // as many workgroups per XCD as its BUSIEST XCD has tiles (the schedule gives every XCD a contiguous range of M-panels: with
  // min(tiles, CUs) workgroups a launch below one round left that XCD with a second one - gemm_split2.hip: x2_xcd_tiles)
  const int tilesM = (a.M + 255) / 256, tilesN = (a.N + 255) / 256;
  const int ngrp = (!RR && a.nsplit > 1 && 8 % a.nsplit == 0 && tilesN % a.nsplit == 0) ? a.nsplit : 1;
  const int xcd_tiles = RR ? (tilesM * tilesN + 7) / 8 : ((tilesM + 8 / ngrp - 1) / (8 / ngrp)) * (tilesN / ngrp);
  hipLaunchKernelGGL(kern, dim3(std::min(8 * xcd_tiles, device_cus())), dim3(512), lds, stream, a);
  FC_CHECK_LAUNCH("gemm_split3");
  return FC_OK;
}

template <int EPI, int RW = 4>
int launch_one(const GemmArgs& a, hipStream_t stream) {
#ifdef FITCLIP_LAB  // (tools/ only: libfitclip_hip_lab.so; the product library has no environment-dependent behaviour)
  static const int lab = [] { const char* e = getenv("FITCLIP_LAB_SPLIT3"); return e ? atoi(e) : 0; }();
  GemmArgs b = a;
  if (lab & 4) b.nsplit = 1;
  if (lab & 8) b.nsplit = 2;
  switch (lab & 3) {
    case 1: return launch_variant<EPI, RW, 2, 0>(b, stream);   // nt activations
    case 2: return launch_variant<EPI, RW, 0, 1>(b, stream);   // round-robin deal
    case 3: return launch_variant<EPI, RW, 2, 1>(b, stream);
    default: return launch_variant<EPI, RW, 0, 0>(b, stream);
  }
#else
  return launch_variant<EPI, RW, 0, 0>(a, stream);
#endif
}

// fp32 rows -> x3 rows: thread per (row, line of 16 columns, half): 8 values -> three 16-byte chunks (+ 16 zero bytes of the
// line's unused quarter, so that every line is written whole)
__global__ void __launch_bounds__(256) split3_rows_kernel(const float* __restrict__ in, long ld_in, char* __restrict__ out,
                                                          long ld_out_bytes, long rows, int K) {
  const long per_row = K / 8;  // 8-column pieces
  const long total = rows * per_row;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / per_row;
    const int piece = (int)(i - row * per_row), group = piece >> 1, half = piece & 1;
    const float* src = in + row * ld_in + piece * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
    bf16x4 a1, a2, a3, b1, b2, b3;
    split3(lo, a1, a2, a3);
    split3(hi, b1, b2, b3);
    char* dst = out + row * ld_out_bytes + (long)group * X3_GROUP_BYTES + half * 16;
    auto put = [&](char* p, const bf16x4& x, const bf16x4& y) {
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = x[e]; v[4 + e] = y[e]; }
      *reinterpret_cast<bf16x8*>(p) = v;
    };
    put(dst, a1, b1);
    put(dst + 32, a2, b2);
    put(dst + 64, a3, b3);
    *reinterpret_cast<f32x4*>(dst + 96) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

}  // namespace

bool gemm_split3_ok(const GemmArgs& a) {
  return a.M > 0 && a.N > 0 && a.K >= 64 && a.K % 32 == 0 && a.N % 32 == 0 && a.bias != nullptr &&
         ((uintptr_t)a.bias & 15) == 0 && (size_t)256 * a.lda * 2 < (1ull << 32) && (size_t)a.N * a.ldw * 2 < (1ull << 32);
}

int launch_gemm_split3(int epilogue, const GemmArgs& a, hipStream_t stream) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fail(FC_EINVAL, "gemm_split3: empty problem %dx%dx%d", a.M, a.N, a.K);
  if (a.K % 32 || a.K < 64) return fail(FC_EINVAL, "gemm_split3: K=%d must be a multiple of 32, at least 64", a.K);
  if (a.N % 32) return fail(FC_EINVAL, "gemm_split3: N=%d must be a multiple of 32", a.N);
  const long need = x3_row_elems(a.K);
  if (a.lda < need || a.ldw < need || a.lda % 64 || a.ldw % 64)
    return fail(FC_EINVAL, "gemm_split3: lda=%d / ldw=%d must cover the %ld bf16 positions of an x3 row in whole 128-byte lines",
                a.lda, a.ldw, need);
  if (((uintptr_t)a.A | (uintptr_t)a.W) & 127 || ((uintptr_t)a.C & 15))
    return fail(FC_EINVAL, "gemm_split3: operands must be 128-byte aligned (x3 rows are made of whole lines)");
  if (!a.bias || ((uintptr_t)a.bias & 15)) return fail(FC_EINVAL, "gemm_split3: bias missing or unaligned");
  if ((size_t)256 * a.lda * 2 >= (1ull << 32) || (size_t)a.N * a.ldw * 2 >= (1ull << 32))
    return fail(FC_EINVAL, "gemm_split3: the weight (or a 256-row tile of the activations) exceeds the 4 GiB of the kernel's 32-bit row offsets");
  GemmArgs b = a;
  // c_fc (12 column tiles): four XCD groups split the N range, so that only a quarter of the 19 MB x3 weight cycles through
  // each 4 MiB L2 (lab, 512 frames: 7.4 -> 3.6 GB fetched per launch, 2.43 -> 2.31 ms; the shapes with 3 or 9 column tiles
  // cannot be split over 2, 4 or 8 groups)
  if (b.nsplit == 0 && (b.N / 256) % 4 == 0 && b.N / 256 >= 8 && b.N % 256 == 0) b.nsplit = 4;
  switch (epilogue) {
    case EPI_BIAS_F32:
      if (a.ldc % 4 || a.ldc < a.N) return fail(FC_EINVAL, "gemm_split3: ldc=%d", a.ldc);
      return launch_one<EPI_BIAS_F32>(b, stream);
    case EPI_RESID3_F32:
      if (a.ldc % 4 || a.ldc < a.N) return fail(FC_EINVAL, "gemm_split3: ldc=%d", a.ldc);
      // two tiles of the stream in flight: windows of 2, 3 and 4 measure the same (709 / 708 / 707 pairs/s on the bench step),
      // and the wider ones spill - the first fragments of the next tile are live across the epilogue
      return launch_one<EPI_RESID3_F32, 2>(b, stream);
    case EPI_GELU_X3:
      if (a.ldc % 64 || a.ldc < x3_row_elems(a.N) || ((uintptr_t)a.C & 127))
        return fail(FC_EINVAL, "gemm_split3: the x3 output needs 128-byte aligned rows of >= 4 N bf16 (ldc=%d)", a.ldc);
      return launch_one<EPI_GELU_X3>(b, stream);
  }
  return fail(FC_EINVAL, "gemm_split3: epilogue %d", epilogue);
}

int launch_split3_rows(const float* in, long ld_in, void* out, long ld_out, long rows, int K, hipStream_t stream) {
  if (rows <= 0) return FC_OK;
  if (K % X3_GROUP || ld_in % 4 || ld_out % 64 || ld_out < x3_row_elems(K) || ((uintptr_t)in & 15) || ((uintptr_t)out & 127))
    return fail(FC_EINVAL, "split3_rows: K %% 16, alignment or row stride");
  const long total = rows * (K / 8);
  const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
  hipLaunchKernelGGL(split3_rows_kernel, dim3(blocks), dim3(256), 0, stream, in, ld_in, static_cast<char*>(out), ld_out * 2,
                     rows, K);
  FC_CHECK_LAUNCH("split3_rows");
  return FC_OK;
}

}  // namespace fc

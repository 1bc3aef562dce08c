// GEMM lab: standalone timing / ablation harness for the MFMA GEMM kernels (tuning aid; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fitclip_amd/csrc -I include -I tools tools/gemm_lab.hip -o tools/bin/gemm_lab
//   ./gemm_lab [M] [reps]
// For every (shape, variant): checks 8192 sampled outputs against a naive fp32 dot product of the same bf16 operands,
// then times `reps` back-to-back launches with hipEvents on random (never zero-filled) operands.
#include "gemm_kernel.h"

#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <functional>
#include <vector>

namespace fc {
void set_error(const std::string&) {}
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fprintf(stderr, "\n");
  return code;
}
}  // namespace fc

using namespace fc;

#define HIP_OK(x)                                                                   \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

__global__ void fill_kernel(bf16* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = ((x & 0xFFFF) + ((x >> 16) & 0xFFFF)) * (1.f / 65536.f) - 1.f;  // triangular in [-1, 1)
    p[i] = static_cast<bf16>(u * scale);
  }
}
__global__ void fill_f32_kernel(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
    p[i] = (x & 0xFFFF) * (1.f / 65536.f) - 0.5f;
  }
}
// sampled check: err[s] = |C[m,n] - (dot + bias)| for 8192 pseudo-random (m, n)
__global__ void check_kernel(const bf16* A, const bf16* W, const float* bias, const bf16* C, int M, int N, int K,
                             float* err) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = s * 0x9E3779B1u + 12345u;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
  const int m = (s < 64) ? (M - 1 - s % min(M, 64)) : (int)(x % (unsigned)M);
  const int n = (int)((x >> 7) % (unsigned)N);
  float acc = 0.f;
  for (int k = 0; k < K; ++k) acc += (float)A[(size_t)m * K + k] * (float)W[(size_t)n * K + k];
  acc += bias[n];
  err[s] = fabsf((float)C[(size_t)m * N + n] - acc) / (fabsf(acc) + 1.f);
}

struct Variant {
  const char* name;
  std::function<void(const GemmArgs&, hipStream_t)> launch;
};
template <int NBLOCK, int NSPLIT = 0, typename F>
std::function<void(const GemmArgs&, hipStream_t)> with_nblock(F f) {
  return [f](const GemmArgs& a, hipStream_t st) { GemmArgs b = a; b.nblock = NBLOCK; b.nsplit = NSPLIT; f(b, st); };
}

template <typename T, int BM, int BN, int WM, int WN, int EPI, int ABL>
void launch_plain(const GemmArgs& a, hipStream_t st) {
  constexpr int lds = 2 * (BM + BN) * ROWB;
  auto kern = gemm_kernel<T, BM, BN, WM, WN, EPI, ABL>;
  static bool configured = false;
  if (!configured) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), lds, st, a);
}

template <typename T, int BM, int BN, int WM, int WN, int EPI, int ABL, int ROT = 0, int SCHED = 0>
void launch_pipelined(const GemmArgs& a, hipStream_t st) {
  constexpr int lds = 2 * (BM + BN) * ROWB + (sizeof(T) == 2 ? WM * WN * 32 * (BN / WN) : 0) + 2048;
  auto kern = gemm_pipelined_kernel<T, BM, BN, WM, WN, EPI, ABL, ROT, SCHED>;
  static bool configured = false;
  if (!configured) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const int cap = 256 * (BM * BN >= 256 * 256 ? 1 : 2);
  hipLaunchKernelGGL(kern, dim3(tiles < cap ? tiles : cap), dim3(WM * WN * 64), lds, st, a);
}

// (the lab kernels whose results were negative - weights from registers, the four-stage BK = 32 kernel, the four-stage ring - were
// removed in round 5; their logs are profiles/r01_lab23_deep.log, r03_gemm_lab_wreg*.log, r03_gemm_lab_ring.log)

#include "gemm_lab_variants.inc"

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 50432;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  struct Shape { const char* name; int N, K; };
  const Shape shapes[] = {{"qkv", 2304, 768}, {"out_proj", 768, 768}, {"c_fc", 3072, 768}, {"c_proj", 768, 3072}};
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  bf16* flush = nullptr;
  if (argc > 3 && atoi(argv[3])) HIP_OK(hipMalloc(&flush, (size_t)640 << 20));
  std::vector<Variant> variants = make_variants();
  for (const Shape& sh : shapes) {
    bf16 *A, *W, *C;
    float *bias, *err;
    HIP_OK(hipMalloc(&A, (size_t)M * sh.K * 2));
    HIP_OK(hipMalloc(&W, (size_t)sh.N * sh.K * 2));
    HIP_OK(hipMalloc(&C, (size_t)M * sh.N * 2));
    HIP_OK(hipMalloc(&bias, sh.N * 4));
    HIP_OK(hipMalloc(&err, 8192 * 4));
    fill_kernel<<<2048, 256, 0, st>>>(A, (size_t)M * sh.K, 1u, 1.0f);
    fill_kernel<<<2048, 256, 0, st>>>(W, (size_t)sh.N * sh.K, 2u, 2.0f / sqrtf((float)sh.K));
    fill_f32_kernel<<<64, 256, 0, st>>>(bias, sh.N, 3u);
    GemmArgs a{};
    a.A = A; a.W = W; a.bias = bias; a.C = C; a.aux = nullptr; a.alpha = 1.f;
    a.M = M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldw = sh.K; a.ldc = sh.N; a.P = 0;
    for (const Variant& v : variants) {
      HIP_OK(hipMemsetAsync(C, 0, (size_t)M * sh.N * 2, st));
      v.launch(a, st);
      HIP_OK(hipGetLastError());
      check_kernel<<<32, 256, 0, st>>>(A, W, bias, C, M, sh.N, sh.K, err);
      std::vector<float> herr(8192);
      HIP_OK(hipMemcpyAsync(herr.data(), err, 8192 * 4, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      float worst = 0.f;
      for (float e : herr) worst = fmaxf(worst, e);
      for (int i = 0; i < 3; ++i) v.launch(a, st);
      float ms = 0.f;
      if (!flush) {
        HIP_OK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) v.launch(a, st);
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipStreamSynchronize(st));
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
      } else {  // cold caches: 640 MB are written between launches (evicts L2 and the 256 MiB Infinity Cache)
        for (int i = 0; i < reps; ++i) {
          fill_kernel<<<2048, 256, 0, st>>>(flush, (size_t)320 << 20, 7u + i, 1.0f);
          HIP_OK(hipEventRecord(e0, st));
          v.launch(a, st);
          HIP_OK(hipEventRecord(e1, st));
          HIP_OK(hipStreamSynchronize(st));
          float one = 0.f;
          HIP_OK(hipEventElapsedTime(&one, e0, e1));
          ms += one / reps;
        }
      }
      const double tf = 2.0 * M * sh.N * sh.K / (ms * 1e-3) / 1e12;
      printf("%-9s M=%d N=%d K=%d  %-28s %8.3f ms %8.1f TF/s  %.3f of peak  maxrelerr=%.2e %s\n", sh.name, M, sh.N,
             sh.K, v.name, ms, tf, tf / 2500.0, worst, worst < 2e-2f ? "ok" : "WRONG(expected for ablations)");
      fflush(stdout);
    }
    HIP_OK(hipFree(A)); HIP_OK(hipFree(W)); HIP_OK(hipFree(C)); HIP_OK(hipFree(bias)); HIP_OK(hipFree(err));
  }
  return 0;
}

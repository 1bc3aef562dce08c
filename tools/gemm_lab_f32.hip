// fp32 GEMM lab: ablations of the persistent pipelined kernel on the four block shapes in the exact-fp32 mode
// (v_mfma_f32_16x16x4_f32).  ABL: 1 = no LDS-DMA inside the K loop, 3 = no epilogue stores; SCHED = LDS-DMA issue schedule.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fitclip_amd/csrc -I include tools/gemm_lab_f32.hip -o tools/bin/gemm_lab_f32
//   tools/bin/gemm_lab_f32 [frames=1024] [reps=5]
#include "gemm_kernel.h"

#include <cstdarg>
#include <cstdlib>
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

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    p[i] = (((x & 0xFFFF) + ((x >> 16) & 0xFFFF)) * (1.f / 65536.f) - 1.f) * scale;
  }
}

template <int EPI, int ABL, int SCHED, int WM = 2, int WN = 4>
void launch(const GemmArgs& a, hipStream_t st) {
  constexpr int BM = 256, BN = 256;
  constexpr int lds = 2 * (BM + BN) * ROWB + WM * WN * 2048 + 2048;
  auto kern = gemm_pipelined_kernel<float, BM, BN, WM, WN, EPI, ABL, 1, SCHED>;
  HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(std::min(tiles, 256)), dim3(WM * WN * 64), lds, st, a);
}

struct Variant { const char* name; void (*fn)(const GemmArgs&, hipStream_t); };

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 5;
  const int M = frames * 197;
  struct Shape { const char* name; int N, K, epi; } shapes[] = {{"c_fc", 3072, 768, 1}, {"c_proj", 768, 3072, 0}, {"qkv", 2304, 768, 0}, {"out_proj", 768, 768, 0}};
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  for (auto& sh : shapes) {
    float *A, *W, *C, *bias;
    HIP_OK(hipMalloc(&A, (size_t)M * sh.K * 4)); HIP_OK(hipMalloc(&W, (size_t)sh.N * sh.K * 4));
    HIP_OK(hipMalloc(&C, (size_t)M * sh.N * 4)); HIP_OK(hipMalloc(&bias, sh.N * 4));
    fill_f32<<<2048, 256, 0, st>>>(A, (size_t)M * sh.K, 1u, 1.f);
    fill_f32<<<2048, 256, 0, st>>>(W, (size_t)sh.N * sh.K, 2u, 2.f / sqrtf((float)sh.K));
    fill_f32<<<64, 256, 0, st>>>(bias, sh.N, 3u, 1.f);
    GemmArgs a{};
    a.A = A; a.W = W; a.bias = bias; a.C = C; a.alpha = 1.f; a.M = M; a.N = sh.N; a.K = sh.K; a.lda = sh.K; a.ldw = sh.K; a.ldc = sh.N;
    std::vector<Variant> vs;
    if (sh.epi == 1) {
      vs = {{"sched8 (in use)", launch<1, 0, 8>}, {"sched2", launch<1, 0, 2>}, {"sched0 burst", launch<1, 0, 0>},
            {"sched8 ABL1 no-loads", launch<1, 1, 8>}, {"sched8 ABL3 no-stores", launch<1, 3, 8>}, {"sched8 ABL4 plain stores", launch<1, 4, 8>},
            {"4 waves 128x128 sched0", launch<1, 0, 0, 2, 2>}, {"sched8 again", launch<1, 0, 8>}};
    } else {
      vs = {{"sched2 (in use)", launch<0, 0, 2>}, {"sched8", launch<0, 0, 8>}, {"sched0 burst", launch<0, 0, 0>},
            {"sched2 ABL1 no-loads", launch<0, 1, 2>}, {"sched2 ABL3 no-stores", launch<0, 3, 2>}, {"sched2 ABL4 plain stores", launch<0, 4, 2>},
            {"4 waves 128x128 sched0", launch<0, 0, 0, 2, 2>}, {"sched2 again", launch<0, 0, 2>}};
    }
    for (auto& v : vs) {
      v.fn(a, st);
      HIP_OK(hipStreamSynchronize(st));
      hipEvent_t e0, e1;
      HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
      HIP_OK(hipEventRecord(e0, st));
      for (int r = 0; r < reps; ++r) v.fn(a, st);
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipEventSynchronize(e1));
      float ms;
      HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      ms /= reps;
      printf("%-9s M=%d N=%d K=%d  %-24s %8.1f us  %6.1f TF/s  (%.3f of 157.3)\n", sh.name, M, sh.N, sh.K, v.name, ms * 1e3,
             2.0 * M * sh.N * sh.K / (ms * 1e-3) / 1e12, 2.0 * M * sh.N * sh.K / (ms * 1e-3) / 1e12 / 157.3);
    }
    HIP_OK(hipFree(A)); HIP_OK(hipFree(W)); HIP_OK(hipFree(C)); HIP_OK(hipFree(bias));
  }
  return 0;
}

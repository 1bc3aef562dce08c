// Lab for the three-plane split-fp32 GEMM (fitclip_amd/csrc/gemm_split3.h): accuracy against a float64 dot product of the
// fp32 operands on sampled outputs, and timing on the four block shapes (random operands).  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fitclip_amd/csrc -I include tools/split3_lab.hip -o tools/bin/split3_lab
//   tools/bin/split3_lab [M] [reps]
#include "gemm_split3.hip"

#include <algorithm>
#include <cmath>
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
hipError_t raise_dynamic_lds(const void* kernel, int bytes) {
  return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}
}  // namespace fc
using namespace fc;

#define HIP_OK(x)                                                               \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(2);                                                                  \
    }                                                                           \
  } while (0)

__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 0x9E3779B1u + seed;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    const float u = ((x & 0xFFFF) + ((x >> 16) & 0xFFFF)) * (1.f / 65536.f) - 1.f;  // triangular in [-1, 1)
    p[i] = u * scale * (1.f + 1e-3f * (float)(x & 1023));                               // full 24-bit mantissas
  }
}
// sampled check against float64: out[s] = {ref, got}
__global__ void check(const float* A, const float* W, const float* bias, const void* C, int x3_out, long ldc, int M, int N, int K,
                      double* out) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned x = s * 0x9E3779B1u + 12345u;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
  const int m = (s < 64) ? (M - 1 - s % min(M, 64)) : (int)(x % (unsigned)M);
  const int n = (int)((x >> 7) % (unsigned)N);
  double acc = 0.0;
  for (int k = 0; k < K; ++k) acc += (double)A[(size_t)m * K + k] * (double)W[(size_t)n * K + k];
  acc += bias[n];
  double got;
  if (x3_out) {
    acc = acc / (1.0 + exp(-1.702 * acc));
    const bf16* line = reinterpret_cast<const bf16*>(C) + (size_t)m * ldc + (size_t)(n / 16) * 64 + (n % 16);
    got = (double)(float)line[0] + (double)(float)line[16] + (double)(float)line[32];
  } else {
    got = reinterpret_cast<const float*>(C)[(size_t)m * ldc + n];
  }
  out[2 * s] = acc;
  out[2 * s + 1] = got;
}

template <int EPI, int ABL, int SPREAD = 0>
void launch_variant(const GemmArgs& a, hipStream_t st) {
  constexpr int lds = 3 * 512 * 96 + 2048;
  auto kern = gemm_split3_kernel<EPI, ABL, SPREAD>;
  static bool configured = false;
  if (!configured) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles < 256 ? tiles : 256), dim3(512), lds, st, a);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 100864;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  struct Shape { const char* name; int N, K, x3; };
  const Shape shapes[] = {{"qkv", 2304, 768, 0}, {"out_proj", 768, 768, 0}, {"c_fc", 3072, 768, 1}, {"c_proj", 768, 3072, 0}};
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  for (const Shape& sh : shapes) {
    float *A, *W, *bias;
    void *A3, *W3, *C;
    double* chk;
    const long lda = x3_row_elems(sh.K), ldc = sh.x3 ? x3_row_elems(sh.N) : sh.N;
    const size_t cbytes = (size_t)M * ldc * (sh.x3 ? 2 : 4);
    HIP_OK(hipMalloc(&A, (size_t)M * sh.K * 4));
    HIP_OK(hipMalloc(&W, (size_t)sh.N * sh.K * 4));
    HIP_OK(hipMalloc(&A3, (size_t)M * lda * 2));
    HIP_OK(hipMalloc(&W3, (size_t)sh.N * lda * 2));
    HIP_OK(hipMalloc(&C, cbytes));
    HIP_OK(hipMalloc(&bias, sh.N * 4));
    HIP_OK(hipMalloc(&chk, 8192 * 2 * 8));
    fill_f32<<<2048, 256, 0, st>>>(A, (size_t)M * sh.K, 1u, 1.0f);
    fill_f32<<<2048, 256, 0, st>>>(W, (size_t)sh.N * sh.K, 2u, 2.0f / sqrtf((float)sh.K));
    fill_f32<<<64, 256, 0, st>>>(bias, sh.N, 3u, 0.5f);
    if (launch_split3_rows(A, sh.K, A3, lda, M, sh.K, st) || launch_split3_rows(W, sh.K, W3, lda, sh.N, sh.K, st)) return 3;
    GemmArgs a{};
    a.A = A3; a.W = W3; a.bias = bias; a.C = C; a.alpha = 1.f;
    a.M = M; a.N = sh.N; a.K = sh.K; a.lda = (int)lda; a.ldw = (int)lda; a.ldc = (int)ldc;
    struct V { const char* name; void (*fn)(const GemmArgs&, hipStream_t); int nsplit; int nblock = 0; int P = 0; int gR = 0; };
    std::vector<V> vs;
    if (sh.x3) {
      vs = {{"x3 gelu", launch_variant<EPI_GELU_X3, 0>, 0}, {"x3 gelu nsplit4", launch_variant<EPI_GELU_X3, 0>, 4},
            {"x3 gelu spread", launch_variant<EPI_GELU_X3, 0, 1>, 0}, {"x3 gelu spread nsplit4", launch_variant<EPI_GELU_X3, 0, 1>, 4},
            {"x3 gelu spread ABL2", launch_variant<EPI_GELU_X3, 2, 1>, 0},
            {"x3 gelu spread nsplit2", launch_variant<EPI_GELU_X3, 0, 1>, 2}, {"x3 gelu spread nsplit2 rot1", launch_variant<EPI_GELU_X3, 0, 1>, 2, 2},
            {"x3 gelu spread nsplit4 rot1", launch_variant<EPI_GELU_X3, 0, 1>, 4, 2},
            {"x3 gelu spread2 nsplit4", launch_variant<EPI_GELU_X3, 0, 2>, 4}, {"x3 gelu spread3 nsplit4", launch_variant<EPI_GELU_X3, 0, 3>, 4},
            {"x3 gelu rot0", launch_variant<EPI_GELU_X3, 0>, 0, 1}, {"x3 gelu rot1", launch_variant<EPI_GELU_X3, 0>, 0, 2},
            {"x3 gelu rot2", launch_variant<EPI_GELU_X3, 0>, 0, 3}, {"x3 gelu rot3", launch_variant<EPI_GELU_X3, 0>, 0, 4},
            {"x3 gelu rot1 nsplit4", launch_variant<EPI_GELU_X3, 0>, 4, 2}, {"x3 gelu rot2 nsplit4", launch_variant<EPI_GELU_X3, 0>, 4, 3},
            {"x3 gelu rot1 grp2", launch_variant<EPI_GELU_X3, 0>, 0, 2, 2}, {"x3 gelu rot1 grp3", launch_variant<EPI_GELU_X3, 0>, 0, 2, 3},
            {"x3 gelu rot2 grp3", launch_variant<EPI_GELU_X3, 0>, 0, 3, 3}, {"x3 gelu rot1 grp4", launch_variant<EPI_GELU_X3, 0>, 0, 2, 4},
            {"x3 gelu rot2 grp4 nsplit4", launch_variant<EPI_GELU_X3, 0>, 4, 3, 4},
            {"x3 gelu ABL2 same tile", launch_variant<EPI_GELU_X3, 2>, 0}, {"x3 gelu ABL3 no-store", launch_variant<EPI_GELU_X3, 3>, 0}, {"x3 gelu ABL1 no-loads", launch_variant<EPI_GELU_X3, 1>, 0},
            {"x3 gelu ABL5 noload nowait", launch_variant<EPI_GELU_X3, 5>, 0}, {"x3 gelu ABL6 mfma only", launch_variant<EPI_GELU_X3, 6>, 0}};
    } else {
      vs = {{"f32 out", launch_variant<EPI_BIAS_F32, 0>, 0}, {"f32 out nsplit4", launch_variant<EPI_BIAS_F32, 0>, 4},
            {"f32 out spread", launch_variant<EPI_BIAS_F32, 0, 1>, 0}, {"f32 out spread ABL2", launch_variant<EPI_BIAS_F32, 2, 1>, 0},
            {"f32 out spread2 (mid)", launch_variant<EPI_BIAS_F32, 0, 2>, 0}, {"f32 out spread3 (3+3)", launch_variant<EPI_BIAS_F32, 0, 3>, 0},
            {"f32 out rot0", launch_variant<EPI_BIAS_F32, 0>, 0, 1}, {"f32 out rot1", launch_variant<EPI_BIAS_F32, 0>, 0, 2},
            {"f32 out rot2", launch_variant<EPI_BIAS_F32, 0>, 0, 3}, {"f32 out rot3", launch_variant<EPI_BIAS_F32, 0>, 0, 4},
            {"f32 out rot1 grp3", launch_variant<EPI_BIAS_F32, 0>, 0, 2, 3}, {"f32 out rot2 grp3", launch_variant<EPI_BIAS_F32, 0>, 0, 3, 3},
            {"f32 out ABL2 same tile", launch_variant<EPI_BIAS_F32, 2>, 0}, {"f32 out ABL3 no-store", launch_variant<EPI_BIAS_F32, 3>, 0}, {"f32 out ABL1 no-loads", launch_variant<EPI_BIAS_F32, 1>, 0},
            {"f32 out ABL5 noload nowait", launch_variant<EPI_BIAS_F32, 5>, 0}, {"f32 out ABL6 mfma only", launch_variant<EPI_BIAS_F32, 6>, 0},
            {"f32 out ABL9 6+dependent order", launch_variant<EPI_BIAS_F32, 9>, 0}, {"f32 out ABL7 6+no barrier", launch_variant<EPI_BIAS_F32, 7>, 0}, {"f32 out ABL8 7+no lgkm wait", launch_variant<EPI_BIAS_F32, 8>, 0}};
    }
    // correctness first, then interleaved timing rounds (every variant once per round; min and median over the rounds)
    std::vector<double> errs;
    for (const V& v : vs) {
      GemmArgs b = a;
      b.nsplit = v.nsplit;
      b.nblock = v.nblock;
      b.P = v.P;
      b.gR = v.gR;
      HIP_OK(hipMemsetAsync(C, 0, cbytes, st));
      v.fn(b, st);
      HIP_OK(hipGetLastError());
      check<<<32, 256, 0, st>>>(A, W, bias, C, sh.x3, ldc, M, sh.N, sh.K, chk);
      std::vector<double> h(8192 * 2);
      HIP_OK(hipMemcpyAsync(h.data(), chk, h.size() * 8, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      double worst = 0, big = 0;
      for (int s = 0; s < 8192; ++s) { worst = fmax(worst, fabs(h[2 * s] - h[2 * s + 1])); big = fmax(big, fabs(h[2 * s])); }
      errs.push_back(worst / big);
    }
    const int rounds = argc > 3 ? atoi(argv[3]) : 5;
    std::vector<std::vector<float>> times(vs.size());
    for (int round = 0; round < rounds; ++round) {
      for (size_t vi = 0; vi < vs.size(); ++vi) {
        GemmArgs b = a;
        b.nsplit = vs[vi].nsplit;
        b.nblock = vs[vi].nblock;
        b.P = vs[vi].P;
        b.gR = vs[vi].gR;
        vs[vi].fn(b, st);
        HIP_OK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) vs[vi].fn(b, st);
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        times[vi].push_back(ms / reps);
      }
    }
    for (size_t vi = 0; vi < vs.size(); ++vi) {
      std::sort(times[vi].begin(), times[vi].end());
      const float best = times[vi].front(), med = times[vi][times[vi].size() / 2];
      const double tf = 2.0 * M * sh.N * sh.K / (med * 1e-3) / 1e12;
      printf("%-9s M=%d N=%d K=%d  %-26s min %7.3f med %7.3f ms  %6.1f TF/s fp32-eq (%6.1f bf16 = %.3f of peak)  err %.2e %s\n",
             sh.name, M, sh.N, sh.K, vs[vi].name, best, med, tf, 6 * tf, 6 * tf / 2500.0, errs[vi],
             errs[vi] < 1e-5 ? "ok" : "WRONG(abl)");
    }
    fflush(stdout);
    HIP_OK(hipFree(A)); HIP_OK(hipFree(W)); HIP_OK(hipFree(A3)); HIP_OK(hipFree(W3)); HIP_OK(hipFree(C));
    HIP_OK(hipFree(bias)); HIP_OK(hipFree(chk));
  }
  return 0;
}

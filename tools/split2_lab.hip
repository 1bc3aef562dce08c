// Lab for the two-plane fp16 split-fp32 GEMM (fitclip_amd/csrc/gemm_split2.h): accuracy against a float64 dot product of the
// fp32 operands on sampled outputs (next to the three-plane bf16 kernel, gemm_split3.h, on the same operands), timing on the
// four block shapes, ablations, and what the fp16 matrix cores do with subnormal inputs.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fitclip_amd/csrc -I include tools/split2_lab.hip -o tools/bin/split2_lab
//   tools/bin/split2_lab [M] [reps] [rounds] [activation scale] [weight scale] [only variants whose name contains this]
#include "gemm_split2.hip"
#include "gemm_split3.hip"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
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
int device_cus() { return 256; }
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
// sampled check against float64: out[s] = {ref, got}.  kind: 0 = fp32 rows, 2 = x2 rows (QuickGELU applied), 3 = x3 rows (same)
__global__ void check(const float* A, const float* W, const float* bias, const void* C, int kind, long ldc, int M, int N, int K,
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
  if (kind == 3) {
    acc = acc / (1.0 + exp(-1.702 * acc));
    const bf16* line = reinterpret_cast<const bf16*>(C) + (size_t)m * ldc + (size_t)(n / 16) * 64 + (n % 16);
    got = (double)(float)line[0] + (double)(float)line[16] + (double)(float)line[32];
  } else if (kind == 2) {
    acc = acc / (1.0 + exp(-1.702 * acc));
    const _Float16* line = reinterpret_cast<const _Float16*>(C) + (size_t)m * ldc + (size_t)(n / 32) * 64 + (n % 32);
    got = (double)(float)line[0] + (double)(float)line[32] / 2048.0;
  } else {
    got = reinterpret_cast<const float*>(C)[(size_t)m * ldc + n];
  }
  out[2 * s] = acc;
  out[2 * s + 1] = got;
}

// what v_mfma_f32_32x32x16_f16 does with subnormal fp16 inputs: D = A(all rows = a) x B(all = b), k = 16 products
__global__ void denorm_probe(float a, float b, float* out) {
  f16x8 va, vb;
  for (int e = 0; e < 8; ++e) { va[e] = static_cast<_Float16>(a); vb[e] = static_cast<_Float16>(b); }
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(va, vb, acc, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = acc[0];
    out[1] = (float)static_cast<_Float16>(a);
    const f16x8 vs = va * static_cast<_Float16>(1.f / 2048.f);
    out[2] = (float)vs[0];
  }
}

template <int EPI, int ABL, int SPREAD = 0, int RW = 2, int RR = 0>
void launch_x2(const GemmArgs& a, hipStream_t st) {
  auto kern = gemm_split2_kernel<EPI, ABL, SPREAD, RW, RR>;
  static bool configured = false;
  if (!configured) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kSplit2Lds));
    configured = true;
  }
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles < 256 ? tiles : 256), dim3(512), kSplit2Lds, st, a);
}
template <int EPI>
void launch_x3(const GemmArgs& a, hipStream_t st) {  // the shipped three-plane kernel (SPREAD = 1)
  constexpr int lds = 3 * 512 * 96 + 2048;
  auto kern = gemm_split3_kernel<EPI, 0, 1, (EPI == EPI_RESID3_F32 ? 2 : 4)>;
  static bool configured = false;
  if (!configured) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    configured = true;
  }
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles < 256 ? tiles : 256), dim3(512), lds, st, a);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 151296;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  const int rounds = argc > 3 ? atoi(argv[3]) : 5;
  const float a_scale = argc > 4 ? (float)atof(argv[4]) : 1.0f;
  const float w_mul = argc > 5 ? (float)atof(argv[5]) : 1.0f;
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  {
    float* d;
    HIP_OK(hipMalloc(&d, 64));
    const float cases[][2] = {{1.f, 1.f}, {3.0e-5f, 1024.f}, {6.0e-8f, 16384.f}, {3.0e-5f, 3.0e-5f}, {1.0e-3f, 1.f}};
    for (auto& c : cases) {
      denorm_probe<<<1, 64, 0, st>>>(c[0], c[1], d);
      float h[3];
      HIP_OK(hipMemcpyAsync(h, d, 12, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      const float a16 = h[1], b16 = (float)(_Float16)c[1];
      printf("denorm probe: a=%g (fp16 %g) b=%g: mfma sum of 16 products = %.9g, expected %.9g;  a * 2^-11 as v_pk_mul_f16 = %g (exact %g)\n",
             c[0], a16, c[1], h[0], 16.0 * (double)a16 * (double)b16, h[2], a16 / 2048.0);
    }
    HIP_OK(hipFree(d));
  }
  struct Shape { const char* name; int N, K, epi; };   // epi: 0 = fp32 rows out, 1 = QuickGELU + plane rows out, 2 = residual update
  const Shape shapes[] = {{"qkv", 2304, 768, 0}, {"out_proj", 768, 768, 2}, {"c_fc", 3072, 768, 1}, {"c_proj", 768, 3072, 2}};
  for (const Shape& sh : shapes) {
    float *A, *W, *bias, *scale2;
    void *A2, *W2, *A3, *W3, *C;
    double* chk;
    int* flag;
    const long lda2 = x2_row_elems(sh.K), lda3 = x3_row_elems(sh.K);
    const long ldc2 = sh.epi == 1 ? x2_row_elems(sh.N) : sh.N, ldc3 = sh.epi == 1 ? x3_row_elems(sh.N) : sh.N;
    const size_t cbytes = (size_t)M * ldc3 * (sh.epi == 1 ? 2 : 4);   // (the x3 rows are the larger ones)
    HIP_OK(hipMalloc(&A, (size_t)M * sh.K * 4));
    HIP_OK(hipMalloc(&W, (size_t)sh.N * sh.K * 4));
    HIP_OK(hipMalloc(&A2, (size_t)M * lda2 * 2));
    HIP_OK(hipMalloc(&W2, (size_t)sh.N * lda2 * 2));
    HIP_OK(hipMalloc(&A3, (size_t)M * lda3 * 2));
    HIP_OK(hipMalloc(&W3, (size_t)sh.N * lda3 * 2));
    HIP_OK(hipMalloc(&C, cbytes));
    HIP_OK(hipMalloc(&bias, sh.N * 4));
    HIP_OK(hipMalloc(&scale2, 8));
    HIP_OK(hipMalloc(&flag, 4));
    HIP_OK(hipMalloc(&chk, 8192 * 2 * 8));
    HIP_OK(hipMemsetAsync(flag, 0, 4, st));
    fill_f32<<<2048, 256, 0, st>>>(A, (size_t)M * sh.K, 1u, a_scale);
    fill_f32<<<2048, 256, 0, st>>>(W, (size_t)sh.N * sh.K, 2u, w_mul * 2.0f / sqrtf((float)sh.K));
    fill_f32<<<64, 256, 0, st>>>(bias, sh.N, 3u, 0.5f);
    if (launch_split2_rows(A, sh.K, A2, lda2, M, sh.K, flag, st) || launch_split2_weight(W, sh.K, W2, lda2, sh.N, sh.K, scale2, st)) return 3;
    if (launch_split3_rows(A, sh.K, A3, lda3, M, sh.K, st) || launch_split3_rows(W, sh.K, W3, lda3, sh.N, sh.K, st)) return 3;
    float hs[2];
    int hflag;
    HIP_OK(hipMemcpyAsync(hs, scale2, 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    printf("%s: weight scale 2^%d (1/s = %g), activation saturation flag %d\n", sh.name, (int)log2f(hs[0]), hs[1], hflag);
    GemmArgs a{};
    a.bias = bias; a.C = C; a.alpha = 1.f;
    a.M = M; a.N = sh.N; a.K = sh.K;
    struct V { const char* name; void (*fn)(const GemmArgs&, hipStream_t); int x3; int nsplit; int nblock = 0; };
    std::vector<V> vs;
    if (sh.epi == 1) {
      vs = {{"x3 shipped (nsplit4)", launch_x3<EPI_GELU_X3>, 1, 4},
            {"x2 gelu", launch_x2<EPI_GELU_X2, 0, 0>, 0, 0}, {"x2 gelu nsplit4", launch_x2<EPI_GELU_X2, 0, 0>, 0, 4},
            {"x2 gelu spread1 nsplit4", launch_x2<EPI_GELU_X2, 0, 1>, 0, 4}, {"x2 gelu spread2 nsplit4", launch_x2<EPI_GELU_X2, 0, 2>, 0, 4},
            {"x2 gelu spread3 nsplit4", launch_x2<EPI_GELU_X2, 0, 3>, 0, 4}, {"x2 gelu spread4 nsplit4", launch_x2<EPI_GELU_X2, 0, 4>, 0, 4},
            {"x2 gelu spread2", launch_x2<EPI_GELU_X2, 0, 2>, 0, 0}, {"x2 gelu spread2 nsplit2", launch_x2<EPI_GELU_X2, 0, 2>, 0, 2},
            {"x2 gelu spread2 rr", launch_x2<EPI_GELU_X2, 0, 2, 2, 1>, 0, 0},
            {"x2 gelu spread2 nsplit4 rot0", launch_x2<EPI_GELU_X2, 0, 2>, 0, 4, 1}, {"x2 gelu spread2 nsplit4 rot2", launch_x2<EPI_GELU_X2, 0, 2>, 0, 4, 3},
            {"x2 gelu ABL2 same tile", launch_x2<EPI_GELU_X2, 2, 2>, 0, 4}, {"x2 gelu ABL3 no-store", launch_x2<EPI_GELU_X2, 3, 2>, 0, 4},
            {"x2 gelu ABL1 no-loads", launch_x2<EPI_GELU_X2, 1, 2>, 0, 4}, {"x2 gelu ABL6 mfma only", launch_x2<EPI_GELU_X2, 6, 2>, 0, 4}};
    } else if (sh.epi == 2) {
      vs = {{"x3 shipped", launch_x3<EPI_RESID3_F32>, 1, 0},
            {"x2 resid", launch_x2<EPI_RESID3_F32, 0, 0>, 0, 0}, {"x2 resid spread1", launch_x2<EPI_RESID3_F32, 0, 1>, 0, 0},
            {"x2 resid spread2", launch_x2<EPI_RESID3_F32, 0, 2>, 0, 0}, {"x2 resid spread3", launch_x2<EPI_RESID3_F32, 0, 3>, 0, 0},
            {"x2 resid spread4", launch_x2<EPI_RESID3_F32, 0, 4>, 0, 0},
            {"x2 resid spread2 rw4", launch_x2<EPI_RESID3_F32, 0, 2, 4>, 0, 0}, {"x2 resid spread2 rr", launch_x2<EPI_RESID3_F32, 0, 2, 2, 1>, 0, 0},
            {"x2 resid spread2 rot0", launch_x2<EPI_RESID3_F32, 0, 2>, 0, 0, 1}, {"x2 resid spread2 rot2", launch_x2<EPI_RESID3_F32, 0, 2>, 0, 0, 3},
            {"x2 bias-only spread2", launch_x2<EPI_BIAS_F32, 0, 2>, 0, 0},
            {"x2 resid ABL2 same tile", launch_x2<EPI_RESID3_F32, 2, 2>, 0, 0}, {"x2 resid ABL3 no-store", launch_x2<EPI_RESID3_F32, 3, 2>, 0, 0},
            {"x2 resid ABL1 no-loads", launch_x2<EPI_RESID3_F32, 1, 2>, 0, 0}, {"x2 resid ABL6 mfma only", launch_x2<EPI_RESID3_F32, 6, 2>, 0, 0}};
    } else {
      vs = {{"x3 shipped", launch_x3<EPI_BIAS_F32>, 1, 0},
            {"x2 f32", launch_x2<EPI_BIAS_F32, 0, 0>, 0, 0}, {"x2 f32 spread1", launch_x2<EPI_BIAS_F32, 0, 1>, 0, 0},
            {"x2 f32 spread2", launch_x2<EPI_BIAS_F32, 0, 2>, 0, 0}, {"x2 f32 spread3", launch_x2<EPI_BIAS_F32, 0, 3>, 0, 0},
            {"x2 f32 spread4", launch_x2<EPI_BIAS_F32, 0, 4>, 0, 0}, {"x2 f32 spread2 rr", launch_x2<EPI_BIAS_F32, 0, 2, 2, 1>, 0, 0},
            {"x2 f32 spread2 rot0", launch_x2<EPI_BIAS_F32, 0, 2>, 0, 0, 1}, {"x2 f32 spread2 rot2", launch_x2<EPI_BIAS_F32, 0, 2>, 0, 0, 3},
            {"x2 f32 ABL2 same tile", launch_x2<EPI_BIAS_F32, 2, 2>, 0, 0}, {"x2 f32 ABL3 no-store", launch_x2<EPI_BIAS_F32, 3, 2>, 0, 0},
            {"x2 f32 ABL1 no-loads", launch_x2<EPI_BIAS_F32, 1, 2>, 0, 0}, {"x2 f32 ABL6 mfma only", launch_x2<EPI_BIAS_F32, 6, 2>, 0, 0}};
    }
    if (argc > 6) {  // (PMC passes: few dispatches - only the variants asked for)
      std::vector<V> keep;
      for (const V& v : vs)
        if (strstr(v.name, argv[6])) keep.push_back(v);
      vs = keep;
    }
    auto args_for = [&](const V& v) {
      GemmArgs b = a;
      b.nsplit = v.nsplit;
      b.nblock = v.nblock;
      if (v.x3) { b.A = A3; b.W = W3; b.lda = (int)lda3; b.ldw = (int)lda3; b.ldc = (int)ldc3; }
      else { b.A = A2; b.W = W2; b.lda = (int)lda2; b.ldw = (int)lda2; b.ldc = (int)ldc2; b.wscale = scale2; b.sat_flag = flag; }
      return b;
    };
    // correctness first (the residual epilogues add to a zeroed C), then interleaved timing rounds
    std::vector<double> errs, rmss;
    for (const V& v : vs) {
      const GemmArgs b = args_for(v);
      HIP_OK(hipMemsetAsync(C, 0, cbytes, st));
      v.fn(b, st);
      HIP_OK(hipGetLastError());
      check<<<32, 256, 0, st>>>(A, W, bias, C, sh.epi == 1 ? (v.x3 ? 3 : 2) : 0, b.ldc, M, sh.N, sh.K, chk);
      std::vector<double> h(8192 * 2);
      HIP_OK(hipMemcpyAsync(h.data(), chk, h.size() * 8, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      double worst = 0, big = 0, sq = 0;
      for (int s = 0; s < 8192; ++s) {
        const double d = fabs(h[2 * s] - h[2 * s + 1]);
        worst = fmax(worst, d); big = fmax(big, fabs(h[2 * s])); sq += d * d;
      }
      errs.push_back(worst / big);
      rmss.push_back(sqrt(sq / 8192) / big);
      if (worst / big > 1e-5 && !strstr(v.name, "ABL")) {   // a wrong product kernel: where?
        int shown = 0, bad = 0;
        for (int s = 0; s < 8192; ++s) {
          if (fabs(h[2 * s] - h[2 * s + 1]) <= 1e-4 * big) continue;
          ++bad;
          if (shown++ >= 24) continue;
          unsigned x = s * 0x9E3779B1u + 12345u;
          x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13;
          const int m = (s < 64) ? (M - 1 - s % std::min(M, 64)) : (int)(x % (unsigned)M);
          const int n = (int)((x >> 7) % (unsigned)sh.N);
          printf("  %s: sample %d m=%d (m%%256=%d) n=%d (n%%256=%d) ref %.6f got %.6f\n", v.name, s, m, m % 256, n, n % 256, h[2 * s], h[2 * s + 1]);
        }
        printf("  %s: %d of 8192 samples off\n", v.name, bad);
      }
    }
    HIP_OK(hipMemcpyAsync(&hflag, flag, 4, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<std::vector<float>> times(vs.size());
    for (int round = 0; round < rounds; ++round) {
      for (size_t vi = 0; vi < vs.size(); ++vi) {
        const GemmArgs b = args_for(vs[vi]);
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
      const int prods = vs[vi].x3 ? 6 : 3;
      printf("%-9s M=%d N=%d K=%d  %-30s min %7.3f med %7.3f ms  %6.1f TF/s fp32-eq (%6.1f on the pipe = %.3f of peak)  err max %.2e rms %.2e %s\n",
             sh.name, M, sh.N, sh.K, vs[vi].name, best, med, tf, prods * tf, prods * tf / 2500.0, errs[vi], rmss[vi],
             errs[vi] < 1e-5 ? "ok" : "WRONG(abl)");
    }
    printf("%s: saturation flag after the runs %d\n", sh.name, hflag);
    fflush(stdout);
    HIP_OK(hipFree(A)); HIP_OK(hipFree(W)); HIP_OK(hipFree(A2)); HIP_OK(hipFree(W2)); HIP_OK(hipFree(A3)); HIP_OK(hipFree(W3));
    HIP_OK(hipFree(C)); HIP_OK(hipFree(bias)); HIP_OK(hipFree(scale2)); HIP_OK(hipFree(flag)); HIP_OK(hipFree(chk));
  }
  return 0;
}

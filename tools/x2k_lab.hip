// Lab for the MFMA shape of the two-plane fp16 split-fp32 GEMM: the 32x32x16 kernel (tools/gemm_split2_m32.h, round 5's shipped
// kernel) next to the 16x16x32 kernel (fitclip_amd/csrc/gemm_split2.h) on the same x2 operands - accuracy against a float64 dot
// product of the fp32 operands on sampled outputs, interleaved timing rounds on the four block shapes, ablations of both.
// Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fitclip_amd/csrc -I include -I tools tools/x2k_lab.hip -o tools/bin/x2k_lab
//   tools/bin/x2k_lab [M] [reps] [rounds] [activation scale] [weight scale] [only variants whose name contains this]
#include "gemm_split2.hip"
#include "gemm_split2_m32.h"

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
  if (kind == 2) {
    acc = acc / (1.0 + exp(-1.702 * acc));
    const _Float16* line = reinterpret_cast<const _Float16*>(C) + (size_t)m * ldc + (size_t)(n / 32) * 64 + (n % 32);
    got = (double)(float)line[0] + (double)(float)line[32] / 2048.0;
  } else {
    got = reinterpret_cast<const float*>(C)[(size_t)m * ldc + n];
  }
  out[2 * s] = acc;
  out[2 * s + 1] = got;
}

// what v_mfma_f32_16x16x32_f16 does with subnormal fp16 inputs: D = A(all = a) x B(all = b), k = 32 products per element
__global__ void denorm_probe(float a, float b, float* out) {
  f16x8 va, vb;
  for (int e = 0; e < 8; ++e) { va[e] = static_cast<_Float16>(a); vb[e] = static_cast<_Float16>(b); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(va, vb, acc, 0, 0, 0);
  if (threadIdx.x == 0) {
    out[0] = acc[0];
    out[1] = (float)static_cast<_Float16>(a);
  }
}

template <int EPI, int ABL, int SPREAD = 0, int RW = 2, int RR = 0>
void launch_m32(const GemmArgs& a, hipStream_t st) {   // the 32x32x16 kernel
  auto kern = gemm_split2_m32_kernel<EPI, ABL, SPREAD, RW, RR>;
  static bool configured = false;
  if (!configured) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kSplit2Lds));
    configured = true;
  }
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles < 256 ? tiles : 256), dim3(512), kSplit2Lds, st, a);
}
template <int EPI, int ABL, int SPREAD = 0, int RW = 4, int PF = 1, int PRIO = 0, int GW = 2>
void launch_k32(const GemmArgs& a, hipStream_t st) {   // the 16x16x32 kernel
  auto kern = gemm_split2_kernel<EPI, ABL, SPREAD, RW, 0, 256, PF, PRIO, GW>;
  static bool configured = false;
  if (!configured) {
    HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kSplit2Lds));
    configured = true;
  }
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3(tiles < 256 ? tiles : 256), dim3(512), kSplit2Lds, st, a);
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
      float h[2];
      HIP_OK(hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      printf("denorm probe (16x16x32): a=%g (fp16 %g) b=%g: mfma sum of 32 products = %.9g, expected %.9g\n", c[0], h[1], c[1], h[0],
             32.0 * (double)h[1] * (double)(float)(_Float16)c[1]);
    }
    HIP_OK(hipFree(d));
  }
  struct Shape { const char* name; int N, K, epi; };   // epi: 0 = fp32 rows out, 1 = QuickGELU + plane rows out, 2 = residual update
  const Shape shapes[] = {{"qkv", 2304, 768, 0}, {"out_proj", 768, 768, 2}, {"c_fc", 3072, 768, 1}, {"c_proj", 768, 3072, 2}};
  for (const Shape& sh : shapes) {
    float *A, *W, *bias, *scale2;
    void *A2, *W2, *C;
    double* chk;
    int* flag;
    const long lda2 = x2_row_elems(sh.K);
    const long ldc2 = sh.epi == 1 ? x2_row_elems(sh.N) : sh.N;
    const size_t cbytes = (size_t)M * ldc2 * (sh.epi == 1 ? 2 : 4);
    HIP_OK(hipMalloc(&A, (size_t)M * sh.K * 4));
    HIP_OK(hipMalloc(&W, (size_t)sh.N * sh.K * 4));
    HIP_OK(hipMalloc(&A2, (size_t)M * lda2 * 2));
    HIP_OK(hipMalloc(&W2, (size_t)sh.N * lda2 * 2));
    HIP_OK(hipMalloc(&C, cbytes));
    HIP_OK(hipMalloc(&bias, sh.N * 4));
    HIP_OK(hipMalloc(&scale2, 8));
    HIP_OK(hipMalloc(&flag, 4));
    HIP_OK(hipMalloc(&chk, 8192 * 2 * 8));
    unsigned long long* dbg;
    HIP_OK(hipMalloc(&dbg, 256 * 8 * 8 * 8));
    HIP_OK(hipMemsetAsync(dbg, 0, 256 * 8 * 8 * 8, st));
    HIP_OK(hipMemsetAsync(flag, 0, 4, st));
    fill_f32<<<2048, 256, 0, st>>>(A, (size_t)M * sh.K, 1u, a_scale);
    fill_f32<<<2048, 256, 0, st>>>(W, (size_t)sh.N * sh.K, 2u, w_mul * 2.0f / sqrtf((float)sh.K));
    fill_f32<<<64, 256, 0, st>>>(bias, sh.N, 3u, 0.5f);
    if (launch_split2_rows(A, sh.K, A2, lda2, M, sh.K, flag, st) || launch_split2_weight(W, sh.K, W2, lda2, sh.N, sh.K, scale2, nullptr, st)) return 3;
    GemmArgs a{};
    a.bias = bias; a.C = C; a.alpha = 1.f;
    a.M = M; a.N = sh.N; a.K = sh.K;
    a.aux = reinterpret_cast<const float*>(dbg);
    a.A = A2; a.W = W2; a.lda = (int)lda2; a.ldw = (int)lda2; a.ldc = (int)ldc2; a.wscale = scale2; a.sat_flag = flag;
    struct V { const char* name; void (*fn)(const GemmArgs&, hipStream_t); int nsplit; };
    std::vector<V> vs;
    const int ns = sh.epi == 1 ? 4 : 0;
    if (sh.epi == 1) {
      vs = {{"m32 shipped", launch_m32<EPI_GELU_X2, 0, 3>, ns}, {"k32 spread3", launch_k32<EPI_GELU_X2, 0, 3>, ns},
            {"k32 spread5", launch_k32<EPI_GELU_X2, 0, 5>, ns}, {"k32 spread5 prio1", launch_k32<EPI_GELU_X2, 0, 5, 4, 1, 1>, ns}, {"k32 spread5 prio2", launch_k32<EPI_GELU_X2, 0, 5, 4, 1, 2>, ns}, {"k32 spread5 prio3", launch_k32<EPI_GELU_X2, 0, 5, 4, 1, 3>, ns}, {"k32 stamps prio1", launch_k32<EPI_GELU_X2, 4, 5, 4, 1, 1>, ns}, {"k32 stamps prio2", launch_k32<EPI_GELU_X2, 4, 5, 4, 1, 2>, ns}, {"k32 spread5 pf2", launch_k32<EPI_GELU_X2, 0, 5, 4, 2>, ns}, {"k32 spread5 gelu-pairs", launch_k32<EPI_GELU_X2, 0, 5, 4, 1, 0, 0>, ns}, {"k32 spread5 gelu-wide plain-split", launch_k32<EPI_GELU_X2, 0, 5, 4, 1, 0, 1>, ns}, {"k32 nsplit2 spread5", launch_k32<EPI_GELU_X2, 0, 5>, 2}, {"k32 nsplit1 spread5", launch_k32<EPI_GELU_X2, 0, 5>, 1}, {"k32 ABL4 stamps pf2", launch_k32<EPI_GELU_X2, 4, 5, 4, 2>, ns},
            {"m32 ABL1 no-loads", launch_m32<EPI_GELU_X2, 1, 3>, ns}, {"k32 ABL1 no-loads", launch_k32<EPI_GELU_X2, 1, 3>, ns},
            {"m32 ABL3 no-epilogue", launch_m32<EPI_GELU_X2, 3, 3>, ns}, {"k32 ABL3 no-epilogue", launch_k32<EPI_GELU_X2, 3, 3>, ns},
            {"m32 ABL6 mfma only", launch_m32<EPI_GELU_X2, 6, 3>, ns}, {"k32 ABL6 mfma only", launch_k32<EPI_GELU_X2, 6, 3>, ns},
            {"k32 ABL7 conflict-free patch writes", launch_k32<EPI_GELU_X2, 7, 5>, ns},
            {"k32 ABL4 stamps", launch_k32<EPI_GELU_X2, 4, 5>, ns}};
    } else if (sh.epi == 2) {
      vs = {{"m32 shipped", launch_m32<EPI_RESID3_F32, 0, 3>, ns}, {"k32 spread3", launch_k32<EPI_RESID3_F32, 0, 3>, ns},
            {"k32 spread5", launch_k32<EPI_RESID3_F32, 0, 5>, ns}, {"k32 spread5 prio1", launch_k32<EPI_RESID3_F32, 0, 5, 4, 1, 1>, ns}, {"k32 spread5 prio2", launch_k32<EPI_RESID3_F32, 0, 5, 4, 1, 2>, ns}, {"k32 spread5 prio3", launch_k32<EPI_RESID3_F32, 0, 5, 4, 1, 3>, ns}, {"k32 stamps prio1", launch_k32<EPI_RESID3_F32, 4, 5, 4, 1, 1>, ns}, {"k32 stamps prio2", launch_k32<EPI_RESID3_F32, 4, 5, 4, 1, 2>, ns}, {"k32 spread5 pf2", launch_k32<EPI_RESID3_F32, 0, 5, 4, 2>, ns}, {"k32 ABL4 stamps pf2", launch_k32<EPI_RESID3_F32, 4, 5, 4, 2>, ns},
            {"k32 ABL8 two products per line", launch_k32<EPI_RESID3_F32, 8, 5>, ns},
            {"k32 spread5 rw2", launch_k32<EPI_RESID3_F32, 0, 5, 2>, ns}, {"k32 spread5 rw6", launch_k32<EPI_RESID3_F32, 0, 5, 6>, ns}, {"k32 spread5 rw8", launch_k32<EPI_RESID3_F32, 0, 5, 8>, ns}, {"k32 spread5 rw16", launch_k32<EPI_RESID3_F32, 0, 5, 16>, ns},
            {"m32 ABL1 no-loads", launch_m32<EPI_RESID3_F32, 1, 3>, ns}, {"k32 ABL1 no-loads", launch_k32<EPI_RESID3_F32, 1, 3>, ns},
            {"m32 ABL3 no-epilogue", launch_m32<EPI_RESID3_F32, 3, 3>, ns}, {"k32 ABL3 no-epilogue", launch_k32<EPI_RESID3_F32, 3, 3>, ns},
            {"m32 ABL6 mfma only", launch_m32<EPI_RESID3_F32, 6, 3>, ns}, {"k32 ABL6 mfma only", launch_k32<EPI_RESID3_F32, 6, 3>, ns},
            {"k32 ABL4 stamps", launch_k32<EPI_RESID3_F32, 4, 5>, ns}};
    } else {
      vs = {{"m32 shipped", launch_m32<EPI_BIAS_F32, 0, 3>, ns}, {"k32 spread3", launch_k32<EPI_BIAS_F32, 0, 3>, ns},
            {"k32 spread5", launch_k32<EPI_BIAS_F32, 0, 5>, ns}, {"k32 spread5 prio1", launch_k32<EPI_BIAS_F32, 0, 5, 4, 1, 1>, ns}, {"k32 spread5 prio2", launch_k32<EPI_BIAS_F32, 0, 5, 4, 1, 2>, ns}, {"k32 spread5 prio3", launch_k32<EPI_BIAS_F32, 0, 5, 4, 1, 3>, ns}, {"k32 stamps prio1", launch_k32<EPI_BIAS_F32, 4, 5, 4, 1, 1>, ns}, {"k32 stamps prio2", launch_k32<EPI_BIAS_F32, 4, 5, 4, 1, 2>, ns}, {"k32 spread5 pf2", launch_k32<EPI_BIAS_F32, 0, 5, 4, 2>, ns},
            {"k32 ABL4 stamps pf2", launch_k32<EPI_BIAS_F32, 4, 5, 4, 2>, ns}, {"k32 ABL8 two products per line", launch_k32<EPI_BIAS_F32, 8, 5>, ns},
            {"m32 ABL1 no-loads", launch_m32<EPI_BIAS_F32, 1, 3>, ns}, {"k32 ABL1 no-loads", launch_k32<EPI_BIAS_F32, 1, 3>, ns},
            {"m32 ABL3 no-epilogue", launch_m32<EPI_BIAS_F32, 3, 3>, ns}, {"k32 ABL3 no-epilogue", launch_k32<EPI_BIAS_F32, 3, 3>, ns},
            {"m32 ABL6 mfma only", launch_m32<EPI_BIAS_F32, 6, 3>, ns}, {"k32 ABL6 mfma only", launch_k32<EPI_BIAS_F32, 6, 3>, ns},
            {"k32 ABL4 stamps", launch_k32<EPI_BIAS_F32, 4, 5>, ns}};
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
      return b;
    };
    // correctness first (the residual epilogues add to a zeroed C), then interleaved timing rounds
    std::vector<double> errs, rmss;
    for (const V& v : vs) {
      const GemmArgs b = args_for(v);
      HIP_OK(hipMemsetAsync(C, 0, cbytes, st));
      v.fn(b, st);
      HIP_OK(hipGetLastError());
      check<<<32, 256, 0, st>>>(A, W, bias, C, sh.epi == 1 ? 2 : 0, b.ldc, M, sh.N, sh.K, chk);
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
    int hflag;
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
      printf("%-9s M=%d N=%d K=%d  %-26s min %7.3f med %7.3f ms  %6.1f TF/s fp32-eq (%6.1f on the pipe = %.3f of peak)  err max %.2e rms %.2e %s\n",
             sh.name, M, sh.N, sh.K, vs[vi].name, best, med, tf, 3 * tf, 3 * tf / 2500.0, errs[vi], rmss[vi],
             errs[vi] < 1e-5 ? "ok" : "WRONG(abl)");
    }
    for (const V& v : vs) {
      if (!strstr(v.name, "stamps")) continue;
      // the stamped build once more, alone: per-wave sums of shader cycles (s_memtime)
      HIP_OK(hipMemsetAsync(dbg, 0, 256 * 8 * 8 * 8, st));
      for (int i = 0; i < 20; ++i) v.fn(args_for(v), st);   // (warm clocks; every launch overwrites the same slots)
      std::vector<unsigned long long> h(256 * 8 * 8);
      HIP_OK(hipMemcpyAsync(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost, st));
      HIP_OK(hipStreamSynchronize(st));
      double sd = 0, sb = 0, sk = 0, se = 0, tiles = 0, nkk = 0; int n = 0;
      double wd[8] = {0}, wb[8] = {0}, wf[8] = {0}, wh[8] = {0};
      for (int w = 0; w < 256 * 8; ++w) {
        const unsigned long long* d = &h[(size_t)w * 8];
        if (!d[4]) continue;
        sd += (double)d[0]; sb += (double)d[1]; sk += (double)d[2]; se += (double)d[3]; tiles += (double)d[4]; nkk = (double)d[5]; ++n;
        wd[w & 7] += (double)d[0] / d[4]; wb[w & 7] += (double)d[1] / d[4];
        wf[w & 7] += (double)d[6] / d[4] / (d[5] - 1); wh[w & 7] += (double)d[7] / d[4] / (d[5] - 1);
      }
      printf("%s stamps (%d waves): per tile: K loop %.0f cycles, epilogue %.0f; inside the K loop per K-step: wait for the next K-step's data %.0f, at the hand-over barrier %.0f (of %.0f per K-step)\n",
             sh.name, n, sk / tiles, se / tiles, sd / tiles / (nkk - 1), sb / tiles / (nkk - 1), sk / tiles / nkk);
      printf("%s stamps by wave: data wait per tile", sh.name);
      for (int w = 0; w < 8; ++w) printf(" %.0f", wd[w] / (n / 8));
      printf("; barrier wait per tile");
      for (int w = 0; w < 8; ++w) printf(" %.0f", wb[w] / (n / 8));
      printf("\n%s stamps by wave, per K-step: groups 0-3 (48 MFMAs + the LDS-DMA pieces)", sh.name);
      for (int w = 0; w < 8; ++w) printf(" %.0f", wf[w] / (n / 8));
      printf("; groups 4-6 (36 MFMAs)");
      for (int w = 0; w < 8; ++w) printf(" %.0f", wh[w] / (n / 8));
      printf("\n");
    }
    printf("%s: saturation flag after the runs %d\n", sh.name, hflag);
    fflush(stdout);
    HIP_OK(hipFree(A)); HIP_OK(hipFree(W)); HIP_OK(hipFree(A2)); HIP_OK(hipFree(W2));
    HIP_OK(hipFree(dbg));
    HIP_OK(hipFree(C)); HIP_OK(hipFree(bias)); HIP_OK(hipFree(scale2)); HIP_OK(hipFree(flag)); HIP_OK(hipFree(chk));
  }
  return 0;
}

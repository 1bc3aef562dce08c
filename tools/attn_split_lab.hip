// Split-fp32 attention lab: attn_split_kernel (attention_split.hip) against attn_f32_blocks_kernel's x3 output
// (attention.hip) on one bench pass, with ablations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fitclip_amd/csrc -I include tools/attn_split_lab.hip -o tools/bin/attn_split_lab
//   tools/bin/attn_split_lab [n_seq=768] [reps=10] [S=197]
#include "../fitclip_amd/csrc/attention.hip"
#include "../fitclip_amd/csrc/attention_split.hip"

#include <cmath>
#include <cstdarg>
#include <vector>

namespace fc {
void set_error(const std::string&) {}
hipError_t raise_dynamic_lds(const void* f, int bytes) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
int device_cus() { return 256; }
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

template <typename F>
float time_it(F&& launch, int reps) {
  hipEvent_t a, b;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&b));
  float best = 1e9f;
  for (int round = 0; round < 4; ++round) {
    HIP_OK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    HIP_OK(hipEventRecord(b, 0));
    HIP_OK(hipEventSynchronize(b));
    HIP_OK(hipGetLastError());
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    if (round) best = std::min(best, ms / reps);
  }
  return best;
}

template <int ABL>
void run_split(const float* qkv, char* out, int n_seq, int S, int heads, int reps, const char* what) {
  auto kern = attn_split_kernel<ABL>;
  HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_SPLIT_LDS));
  const float ms = time_it([&] { hipLaunchKernelGGL(kern, dim3(std::min(n_seq * heads, 256)), dim3(512), ATTN_SPLIT_LDS, 0, qkv, out, S, heads, n_seq * heads, (long long*)nullptr, (int*)nullptr); }, reps);
  printf("split ABL %2d  %-44s %8.3f ms\n", ABL, what, ms);
}

static float bf(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int n_seq = argc > 1 ? atoi(argv[1]) : 768, reps = argc > 2 ? atoi(argv[2]) : 10, S = argc > 3 ? atoi(argv[3]) : 197, heads = 12;
  const int D = heads * 64;
  const size_t rows = (size_t)n_seq * S, nq = rows * 3 * D, ob = rows * D * 8;
  float* qkv;
  char *o_old, *o_new;
  HIP_OK(hipMalloc(&qkv, nq * 4));
  HIP_OK(hipMalloc(&o_old, ob));
  HIP_OK(hipMalloc(&o_new, ob));
  fill_f32<<<1024, 256>>>(qkv, nq, 1u, 3.f);
  HIP_OK(hipMemset(o_new, 0x7f, ob));
  HIP_OK(hipDeviceSynchronize());
  if (launch_attention_x3(qkv, o_old, n_seq, S, heads, 0) != FC_OK) return 3;
  if (launch_attention_split(qkv, o_new, n_seq, S, heads, 0) != FC_OK) return 3;
  HIP_OK(hipDeviceSynchronize());
  {  // compare the decoded values of the first and last sequences + a float64 reference of a few rows
    std::vector<unsigned short> a(ob / 2), b(ob / 2);
    std::vector<float> hq(nq);
    HIP_OK(hipMemcpy(a.data(), o_old, ob, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(b.data(), o_new, ob, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hq.data(), qkv, nq * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0;
    size_t pad_bad = 0, noncanon = 0;
    for (size_t row = 0; row < rows; ++row)
      for (int grp = 0; grp < D / 16; ++grp) {
        const unsigned short* la = a.data() + (row * (D / 16) + grp) * 64;
        const unsigned short* lb = b.data() + (row * (D / 16) + grp) * 64;
        for (int c = 0; c < 16; ++c) {
          const float va = bf(la[c]) + bf(la[16 + c]) + bf(la[32 + c]);
          const float vb = bf(lb[c]) + bf(lb[16 + c]) + bf(lb[32 + c]);
          worst = std::max(worst, (double)fabsf(va - vb));
          scale = std::max(scale, (double)fabsf(va));
          if (lb[48 + c]) ++pad_bad;
        }
      }
    printf("old vs new: max abs diff %.3e (max |value| %.3e), nonzero pad positions %zu\n", worst, scale, pad_bad);
    // float64 reference for 3 (sequence, head) pairs
    double e_old = 0, e_new = 0;
    for (int pick = 0; pick < 3; ++pick) {
      const int seq = pick == 0 ? 0 : pick == 1 ? n_seq / 2 : n_seq - 1, h = (5 * pick + 1) % heads;
      for (int q = 0; q < S; q += 7) {
        std::vector<double> sc(S);
        double mx = -1e300;
        const float* qr = hq.data() + ((size_t)seq * S + q) * 3 * D + h * 64;
        for (int k = 0; k < S; ++k) {
          const float* kr = hq.data() + ((size_t)seq * S + k) * 3 * D + D + h * 64;
          double s = 0;
          for (int d = 0; d < 64; ++d) s += (double)qr[d] * kr[d];
          sc[k] = s / 8;
          mx = std::max(mx, sc[k]);
        }
        double sum = 0;
        for (int k = 0; k < S; ++k) { sc[k] = exp(sc[k] - mx); sum += sc[k]; }
        for (int d = 0; d < 64; ++d) {
          double o = 0;
          for (int k = 0; k < S; ++k) o += sc[k] * hq[((size_t)seq * S + k) * 3 * D + 2 * D + h * 64 + d];
          o /= sum;
          const size_t line = (((size_t)seq * S + q) * (D / 16) + h * 4 + d / 16) * 64;
          const int c = d % 16;
          e_old = std::max(e_old, fabs(o - (double)(bf(a[line + c]) + bf(a[line + 16 + c]) + bf(a[line + 32 + c]))));
          e_new = std::max(e_new, fabs(o - (double)(bf(b[line + c]) + bf(b[line + 16 + c]) + bf(b[line + 32 + c]))));
        }
      }
    }
    printf("against float64: fp32-MFMA kernel %.3e, split kernel %.3e\n", e_old, e_new);
    (void)noncanon;
  }
  const double mfma_ms = (double)n_seq * heads * (13.0 * 13 * 2 + 13.0 * 7 * 4) * 6 * 16 / 4 / 256 / 2.4e6;
  printf("%d x %d tokens x %d heads; pure bf16 MFMA issue time %.3f ms at 2.4 GHz\n", n_seq, S, heads, mfma_ms);
  {
    const float ms = time_it([&] { launch_attention_x3(qkv, o_old, n_seq, S, heads, 0); }, reps);
    printf("fp32-MFMA blocks kernel (x3 out)                        %8.3f ms\n", ms);
  }
  run_split<0>(qkv, o_new, n_seq, S, heads, reps, "product kernel");
  run_split<30>(qkv, o_new, n_seq, S, heads, reps, "waves 0, 4 stage nothing (instead of 0, 1)");
  run_split<31>(qkv, o_new, n_seq, S, heads, reps, "waves 0, 2 stage nothing");
  run_split<1>(qkv, o_new, n_seq, S, heads, reps, "no S MFMAs");
  run_split<2>(qkv, o_new, n_seq, S, heads, reps, "no P.V MFMAs");
  run_split<3>(qkv, o_new, n_seq, S, heads, reps, "no exponentials");
  run_split<4>(qkv, o_new, n_seq, S, heads, reps, "no staging");
  run_split<10>(qkv, o_new, n_seq, S, heads, reps, "no MFMAs, no exponentials");
  run_split<11>(qkv, o_new, n_seq, S, heads, reps, "no output stores");
  run_split<0>(qkv, o_new, n_seq, S, heads, reps, "product kernel (again)");
  return 0;
}

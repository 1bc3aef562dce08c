// fp32 attention lab: ablations of attn_f32_blocks_kernel at one bench pass (1024 images x 197 tokens x 12 heads).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fitclip_amd/csrc -I include tools/attn_lab_f32.hip -o tools/bin/attn_lab_f32
//   tools/bin/attn_lab_f32 [n_seq=1024] [reps=10]
#include "../fitclip_amd/csrc/attention.hip"

#include <cstdarg>
#include <algorithm>
#include <cstdlib>

namespace fc {
void set_error(const std::string&) {}
hipError_t raise_dynamic_lds(const void* f, int bytes) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); }
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

template <int ABL, int NW = 8>
float run(const float* qkv, float* out, int n_seq, int S, int heads, int reps, const char* what) {
  constexpr int lds = 2 * (64 * 256 + 16 * (1024 + 64));
  auto kern = attn_f32_blocks_kernel<NW, ABL>;
  HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t a, b;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&b));
  float best = 1e9f;
  for (int round = 0; round < 4; ++round) {  // the first round warms the clocks up; best of the rest
    HIP_OK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(n_seq * heads), dim3(NW * 64), lds, 0, qkv, out, S, heads);
    HIP_OK(hipEventRecord(b, 0));
    HIP_OK(hipEventSynchronize(b));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    if (round) best = std::min(best, ms / reps);
  }
  printf("ABL %d  %-42s %8.3f ms\n", ABL, what, best);
  return best;
}

__global__ void count_diff(const float* a, const float* b, size_t n, unsigned long long* cnt) {
  unsigned long long c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    c += __float_as_uint(a[i]) != __float_as_uint(b[i]);
  if (c) atomicAdd(cnt, c);
}

void compare(const float* a, const float* b, size_t n, const char* what) {
  unsigned long long *cnt, host = 0;
  HIP_OK(hipMalloc(&cnt, 8));
  HIP_OK(hipMemset(cnt, 0, 8));
  count_diff<<<1024, 256>>>(a, b, n, cnt);
  HIP_OK(hipMemcpy(&host, cnt, 8, hipMemcpyDeviceToHost));
  printf("       %-42s %llu of %zu values differ from the blocks kernel\n", what, host, n);
  HIP_OK(hipFree(cnt));
}

int main(int argc, char** argv) {
  const int n_seq = argc > 1 ? atoi(argv[1]) : 1024, reps = argc > 2 ? atoi(argv[2]) : 10, S = 197, heads = 12;
  const size_t nq = (size_t)n_seq * S * 3 * heads * 64, no = nq / 3;
  float *qkv, *out;
  HIP_OK(hipMalloc(&qkv, nq * 4));
  HIP_OK(hipMalloc(&out, no * 4));
  fill_f32<<<1024, 256>>>(qkv, nq, 1u, 1.f);
  HIP_OK(hipDeviceSynchronize());
  const double mfma_ms = (double)n_seq * heads * 13 * 13 * 32 * 32 / (256 * 4) / 2.4e6;
  printf("%d x %d tokens x %d heads; pure MFMA issue time %.3f ms at 2.4 GHz\n", n_seq, S, heads, mfma_ms);
  run<0>(qkv, out, n_seq, S, heads, reps, "product kernel");
  run<1>(qkv, out, n_seq, S, heads, reps, "no exponentials");
  run<2>(qkv, out, n_seq, S, heads, reps, "V operand from registers");
  run<3>(qkv, out, n_seq, S, heads, reps, "K operand from registers");
  run<4>(qkv, out, n_seq, S, heads, reps, "no staging, no barriers");
  run<5>(qkv, out, n_seq, S, heads, reps, "no P.V MFMAs (VALU stand-in)");
  run<6>(qkv, out, n_seq, S, heads, reps, "no S MFMAs (VALU stand-in)");
  run<0>(qkv, out, n_seq, S, heads, reps, "product kernel (again)");
  // (the persistent-workgroup form - bitwise equal, 6 % slower - was removed in round 5: profiles/r04_attn_persist_lab.log; so was
  // the two-heads-per-workgroup form with software-pipelined tiles: profiles/r05_attn_pair_lab.log)
  run<0>(qkv, out, n_seq, S, heads, reps, "product kernel (again)");
  return 0;
}

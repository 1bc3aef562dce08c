// Lab of the three-product attention (attention_split2.hip: attn_split2_kernel): the kernel and its ablations on one bench pass - where
// the 15 us per (sequence, head) go.  ABL 1 = no S products, 2 = no P.V products, 3 = no exponentials, 10 = none of the three (staging,
// splits and stores only), 4 = no staging of K / V (planes as they lie), 11 = no stores.  Results of the ablations are meaningless.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fitclip_amd/csrc -I include tools/attn2_lab.hip -o tools/bin/attn2_lab
//   tools/bin/attn2_lab [n_seq=2048] [reps=10] [S=197]
#include "../fitclip_amd/csrc/attention_split2.hip"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

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

template <int ABL>
void run(const float* qkv, char* out, int n_seq, int S, int heads, int reps, const char* what) {
  auto kern = attn_split2_kernel<ABL>;
  HIP_OK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, ATTN_SPLIT2_LDS));
  hipEvent_t a, b;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&b));
  float best = 1e9f;
  for (int round = 0; round < 5; ++round) {
    HIP_OK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i)
      hipLaunchKernelGGL(kern, dim3(std::min(n_seq * heads, 256)), dim3(NT), ATTN_SPLIT2_LDS, 0, qkv, out, S, heads, n_seq * heads, (int*)nullptr);
    HIP_OK(hipEventRecord(b, 0));
    HIP_OK(hipEventSynchronize(b));
    HIP_OK(hipGetLastError());
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    if (round) best = std::min(best, ms / reps);
  }
  printf("attn_split2 ABL %2d  %-60s %8.3f ms  (%.2f us per (sequence, head) and CU)\n", ABL, what, best, best * 1e3 * 256 / (n_seq * heads));
}

int main(int argc, char** argv) {
  const int n_seq = argc > 1 ? atoi(argv[1]) : 2048, reps = argc > 2 ? atoi(argv[2]) : 10, S = argc > 3 ? atoi(argv[3]) : 197, heads = 12;
  const int D = heads * 64;
  const size_t rows = (size_t)n_seq * S, nq = rows * 3 * D, ob = rows * D * 4;
  float* qkv;
  char* out;
  HIP_OK(hipMalloc(&qkv, nq * 4));
  HIP_OK(hipMalloc(&out, ob));
  fill_f32<<<1024, 256>>>(qkv, nq, 1u, 3.f);
  HIP_OK(hipDeviceSynchronize());
  printf("%d sequences x %d tokens x %d heads: q|k|v in %.2f GB, x2 rows out %.2f GB\n", n_seq, S, heads, nq * 4 / 1e9, ob / 1e9);
  run<0>(qkv, out, n_seq, S, heads, reps, "the kernel");
  run<1>(qkv, out, n_seq, S, heads, reps, "no S = q k^T products");
  run<2>(qkv, out, n_seq, S, heads, reps, "no P v products");
  run<3>(qkv, out, n_seq, S, heads, reps, "no exponentials");
  run<10>(qkv, out, n_seq, S, heads, reps, "no products, no exponentials (loads, splits, stores)");
  run<4>(qkv, out, n_seq, S, heads, reps, "no staging of K / V (q loads, products, softmax, stores)");
  run<11>(qkv, out, n_seq, S, heads, reps, "no stores");
  run<0>(qkv, out, n_seq, S, heads, reps, "the kernel (again)");
  return 0;
}

// Does the attention kernel's read pattern (128-byte row segments at a 4608-byte stride: one head's q/k/v slice of every
// token row of the packed [token][3*768] QKV matrix) cost HBM bandwidth compared with contiguous 25 KB blocks?
// Both kernels move the same bytes into LDS by LDS-DMA with two workgroups per CU and nothing else.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stride_lab.hip -o tools/bin/stride_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

// item = (seq, head); rows = 197 tokens x 3 parts (q, k, v), 128 B each.
template <bool HEAD_MAJOR>
__global__ void __launch_bounds__(448) read_kernel(const char* __restrict__ qkv, float* sink, int S, int heads) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rin = lane >> 3, pc = lane & 7;
  const int item = blockIdx.x, seq = item / heads, h = item % heads;
  const long ld = 3L * heads * 128;  // bytes per token row of the packed layout
  const int pieces = (3 * S + 7) / 8;
  for (int p = wave; p < pieces; p += 7) {
    const int row = min(p * 8 + rin, 3 * S - 1);  // 0 .. 3S-1: part = row / S, token = row % S
    const int part = row / S, tok = row - part * S;
    const char* src = HEAD_MAJOR ? qkv + (((long)seq * heads + h) * 3 * S + row) * 128 + pc * 16
                                 : qkv + ((long)seq * S + tok) * ld + part * (heads * 128) + h * 128 + pc * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + (p % 72) * 1024), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && smem[0] == 123 && smem[1] == 45 && smem[77] == 99) sink[0] = 1.f;
}

int main() {
  const int S = 197, heads = 12, n_seq = 512;
  const size_t bytes = (size_t)n_seq * S * 3 * heads * 128;
  char* buf; float* sink;
  HIP_OK(hipMalloc(&buf, bytes)); HIP_OK(hipMalloc(&sink, 4));
  HIP_OK(hipMemset(buf, 1, bytes));
  char* flush; HIP_OK(hipMalloc(&flush, (size_t)1 << 30));
  hipStream_t st; HIP_OK(hipStreamCreate(&st));
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const int lds = 72 * 1024;  // two workgroups per CU, as the attention kernel
  HIP_OK(hipFuncSetAttribute((const void*)read_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  HIP_OK(hipFuncSetAttribute((const void*)read_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int rep = 0; rep < 3; ++rep)
    for (int hm = 0; hm < 2; ++hm) {
      float total = 0.f;
      const int iters = 10;
      for (int i = 0; i < iters; ++i) {
        HIP_OK(hipMemsetAsync(flush, i, (size_t)1 << 30, st));  // evict L2 / Infinity Cache
        HIP_OK(hipEventRecord(e0, st));
        if (hm) hipLaunchKernelGGL(read_kernel<true>, dim3(n_seq * heads), dim3(448), lds, st, buf, sink, S, heads);
        else hipLaunchKernelGGL(read_kernel<false>, dim3(n_seq * heads), dim3(448), lds, st, buf, sink, S, heads);
        HIP_OK(hipEventRecord(e1, st));
        HIP_OK(hipStreamSynchronize(st));
        float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
      }
      printf("%-34s %.1f us  %.2f TB/s\n", hm ? "head-major (contiguous 75 KB items)" : "packed rows (128 B @ 4608 B stride)",
             total / iters * 1e3, bytes / (total / iters * 1e-3) / 1e12);
    }
  return 0;
}

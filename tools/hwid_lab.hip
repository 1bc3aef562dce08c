// Where do the waves of a workgroup land?  512-thread workgroups with 66 KiB of LDS (two per CU, as attn_f32_blocks_kernel): every
// wave records HW_ID (SIMD, wave slot, CU, SE) and XCC_ID; the host prints wave -> (SIMD, slot) for a few workgroups of the first
// wave of dispatches and of the steady state, and how often wave i sits on SIMD i % 4.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hwid_lab.hip -o tools/bin/hwid_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void __launch_bounds__(512) probe(unsigned* rec, int spin) {
  extern __shared__ char smem[];
  const int wave = threadIdx.x >> 6;
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float v = threadIdx.x;
  for (int i = 0; i < spin * (1 + (blockIdx.x % 3)); ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);  // uneven lifetimes: slots get recycled
  if ((threadIdx.x & 63) == 0) {
    rec[(blockIdx.x * 8 + wave) * 2] = hw;
    rec[(blockIdx.x * 8 + wave) * 2 + 1] = xcc;
  }
  if (v == 12345.f) smem[threadIdx.x] = 1;
}

int main() {
  const int nb = 4096;
  unsigned* rec;
  HIP_OK(hipMalloc(&rec, nb * 8 * 2 * 4));
  HIP_OK(hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 67584));
  hipLaunchKernelGGL(probe, dim3(nb), dim3(512), 67584, 0, rec, 20000);
  HIP_OK(hipDeviceSynchronize());
  std::vector<unsigned> h(nb * 16);
  HIP_OK(hipMemcpy(h.data(), rec, h.size() * 4, hipMemcpyDeviceToHost));
  long on_mod = 0, pair_same_simd[8] = {0};
  long slot_hist[16] = {0};
  for (int b = 0; b < nb; ++b) {
    for (int w = 0; w < 8; ++w) {
      const unsigned hw = h[(b * 8 + w) * 2];
      const int simd = (hw >> 4) & 3, slot = hw & 15;
      on_mod += simd == (w & 3);
      slot_hist[slot]++;
    }
  }
  printf("wave i on SIMD i %% 4: %ld of %d\n", on_mod, nb * 8);
  printf("wave slot histogram:");
  for (int s = 0; s < 16; ++s) printf(" %ld", slot_hist[s]);
  printf("\n");
  for (int b : {0, 1, 2, 3, 8, 9, 255, 256, 257, 600, 601, 2000, 2001, 3000, 4095}) {
    printf("wg %4d xcc %u se %u cu %2u :", b, h[b * 16 + 1] & 15, (h[b * 16] >> 13) & 7, (h[b * 16] >> 8) & 15);
    for (int w = 0; w < 8; ++w) printf("  w%d s%u/%u", w, (h[(b * 8 + w) * 2] >> 4) & 3, h[(b * 8 + w) * 2] & 15);
    printf("\n");
  }
  // how many distinct slot pairs per SIMD does a workgroup use, and are they {0,1} or {2,3}?
  long low = 0, high = 0, mixed = 0;
  for (int b = 0; b < nb; ++b) {
    int lo = 0, hi = 0;
    for (int w = 0; w < 8; ++w) ((h[(b * 8 + w) * 2] & 15) < 2 ? lo : hi)++;
    if (lo == 8) low++; else if (hi == 8) high++; else mixed++;
  }
  printf("workgroups with all waves in slots {0,1}: %ld, all in slots >= 2: %ld, mixed: %ld\n", low, high, mixed);
  return 0;
}

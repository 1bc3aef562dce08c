// What v_mfma_f32_16x16x4_f32 sustains as a function of the dependence pattern: CHAIN accumulators used in rotation (1 = every
// MFMA takes the previous one's result as its C operand, as the S^T chains of the fp32 attention do), 1 / 2 / 4 waves per SIMD,
// with and without VALU instructions between the MFMAs.  Reports cycles per MFMA and SIMD at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_f32_chain_lab.hip -o tools/bin/mfma_f32_chain_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int CHAIN, int VALU, int KIND = 0>
__global__ void __launch_bounds__(1024) chain_kernel(float* out, int iters, long long* cycles) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[CHAIN];
#pragma unroll
  for (int i = 0; i < CHAIN; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __shared__ float lds[1024];
  lds[threadIdx.x] = lane;
  __syncthreads();
  float a = lane * 1e-3f, b = 1.f - lane * 1e-3f, v = lane;
  float w[4] = {1.f + lane, 2.f + lane, 3.f + lane, 4.f + lane};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk = {1.f + lane, 2.f};
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      acc[k % CHAIN] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k % CHAIN], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < VALU; ++j) {
        if (KIND == 0) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        if (KIND == 1) w[j & 3] = __builtin_fmaf(w[j & 3], 1.0001f, 0.5f);
        if (KIND == 2) w[j & 3] = __builtin_amdgcn_exp2f(w[j & 3]);
        if (KIND == 3) pk = pk * f32x2{1.0001f, 0.9999f};
        if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(w[j & 3]) : "v"(a), "v"(b));
        if (KIND == 5) w[j & 3] += lds[(threadIdx.x + (k & 63) + (int)w[j & 3] * 0) & 1023];
      }
      if (VALU) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(KIND == 2 ? 0x400 : 0x002, VALU, 0);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < CHAIN; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v + w[0] + w[1] + w[2] + w[3] + pk[0] + pk[1];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int CHAIN, int VALU, int KIND = 0>
void run(float* out, long long* cyc, int threads, const char* what) {
  const int iters = 2000;
  hipEvent_t a, b;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&b));
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    HIP_OK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((chain_kernel<CHAIN, VALU, KIND>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    HIP_OK(hipEventRecord(b, 0));
    HIP_OK(hipEventSynchronize(b));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  long long c;
  HIP_OK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  const double mfma_per_simd = (double)iters * 64 * (threads / 256);
  // s_memtime / readcyclecounter ticks at 100 MHz on gfx9: use the wall time and report ns per MFMA and SIMD; at 2.4 GHz 32 cycles = 13.3 ns
  printf("chain %2d  valu %d  %d waves/SIMD  %-28s %7.2f ns per MFMA and SIMD  (%.1f cycles at 2.4 GHz)  [%lld ticks]\n", CHAIN, VALU,
         threads / 256, what, best * 1e6 / mfma_per_simd, best * 1e6 / mfma_per_simd * 2.4, c);
}

int main() {
  float* out;
  long long* cyc;
  HIP_OK(hipMalloc(&out, 256 * 1024 * 4));
  HIP_OK(hipMalloc(&cyc, 8));
  for (int threads : {256, 512, 1024}) {
    run<1, 0>(out, cyc, threads, "all dependent");
    run<4, 0>(out, cyc, threads, "four chains");
    run<4, 3, 0>(out, cyc, threads, "+ 3 chained v_fma");
    run<4, 4, 1>(out, cyc, threads, "+ 4 independent v_fma");
    run<4, 8, 1>(out, cyc, threads, "+ 8 independent v_fma");
    run<4, 1, 2>(out, cyc, threads, "+ 1 v_exp_f32");
    run<4, 2, 2>(out, cyc, threads, "+ 2 v_exp_f32");
    run<4, 4, 3>(out, cyc, threads, "+ 4 v_pk_mul_f32 (chained)");
    run<4, 4, 4>(out, cyc, threads, "+ 4 v_max3_f32");
    run<4, 4, 5>(out, cyc, threads, "+ 4 ds_read_b32 + v_add");
  }
  return 0;
}

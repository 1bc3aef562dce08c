// Bare-MFMA ceiling under the package power cap: what does v_mfma_f32_16x16x32_bf16 sustain on random operands when
// nothing else happens (variant 0), and with the LDS fragment traffic of the GEMM's wave tile (variant 1: 24
// ds_read_b128 per 64 MFMAs, as gemm_pipelined_kernel<256,256,2,4>)?  8 waves per CU (2 per SIMD), 256 accumulator
// registers per lane, one workgroup per CU.  Run under tools/clock_probe.sh to see clock and power.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_lab.hip -o tools/bin/mfma_lab ;  tools/bin/mfma_lab [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16;
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16; return x; }
__device__ inline bf16x8 rnd8(unsigned seed) {
  bf16x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned h = hash(seed * 8 + i);
    v[i] = static_cast<bf16>(((h & 0xFFFF) + (h >> 16)) * (1.f / 65536.f) - 1.f);
  }
  return v;
}

template <int VARIANT>
__global__ void __launch_bounds__(512) mfma_kernel(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 128 KiB of random operand data in LDS (also pins one workgroup per CU)
  for (int i = tid; i < 131072 / 16; i += 512) reinterpret_cast<bf16x8*>(smem)[i] = rnd8(blockIdx.x * 8192 + i);
  __syncthreads();
  f32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[a][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[2][8], wf[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int a = 0; a < 8; ++a) af[s][a] = rnd8(tid * 64 + s * 8 + a);
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[s][j] = rnd8(tid * 64 + 32 + s * 4 + j);
  }
  const char* base = smem + ((wave * 64 + lane) * 16) % 65536;
  for (int it = 0; it < iters; ++it) {
    if (VARIANT == 1) {  // the fragment traffic of one K-step: 16 + 8 ds_read_b128 (conflict-free: lane-linear)
      const char* p = base + (it & 3) * 16384;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int a = 0; a < 8; ++a) af[s][a] = *reinterpret_cast<const bf16x8*>(p + ((s * 8 + a) * 8192) % 65536 + (s * 8 + a) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[s][j] = *reinterpret_cast<const bf16x8*>(p + 65536 + ((s * 4 + j) * 8192) % 65536);
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[a][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], af[s][a], acc[a][j], 0, 0, 0);
  }
  float keep = 0.f;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int j = 0; j < 4; ++j) keep += acc[a][j][0] + acc[a][j][1] + acc[a][j][2] + acc[a][j][3];
  if (keep == 123.456f) out[0] = keep;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
  float* out; HIP_OK(hipMalloc(&out, 4));
  hipStream_t st; HIP_OK(hipStreamCreate(&st));
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  const int lds = 131072, iters = 4096;
  HIP_OK(hipFuncSetAttribute((const void*)mfma_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  HIP_OK(hipFuncSetAttribute((const void*)mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int variant = 0; variant < 2; ++variant) {
    const double flops = 256.0 * 8 * iters * 64 * (2.0 * 16 * 16 * 32);
    double total_ms = 0; int launches = 0;
    while (total_ms < seconds * 1e3) {
      HIP_OK(hipEventRecord(e0, st));
      for (int r = 0; r < 20; ++r) {
        if (variant == 0) hipLaunchKernelGGL(mfma_kernel<0>, dim3(256), dim3(512), lds, st, out, iters);
        else hipLaunchKernelGGL(mfma_kernel<1>, dim3(256), dim3(512), lds, st, out, iters);
      }
      HIP_OK(hipEventRecord(e1, st));
      HIP_OK(hipStreamSynchronize(st));
      float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
      total_ms += ms; launches += 20;
      printf("variant %d (%s): %.3f ms/launch  %.1f TF/s\n", variant, variant ? "MFMA + LDS fragment reads" : "bare MFMA",
             ms / 20, flops / (ms / 20 * 1e-3) / 1e12);
      fflush(stdout);
    }
  }
  return 0;
}

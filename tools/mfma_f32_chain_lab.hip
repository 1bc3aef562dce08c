// Do vector instructions hide under MFMAs when they are interleaved one by one in the instruction stream?
//   * v_mfma_f32_16x16x4_f32 (fp32 operands): NO.  The MFMA holds the SIMD's vector issue for all of its 32 cycles: every v_* between
//     two of them adds its full issue cost (v_fma_f32 ~3 cycles at four waves per SIMD, ~6 at one; v_exp_f32 8 - 16; v_max3 / packed
//     fp32 ~6), with 1, 2 or 4 waves per SIMD alike.  LDS reads (and scalar instructions) are free.  For an fp32-MFMA kernel the
//     vector instructions are therefore a cost to be COUNTED, not scheduled: the softmax of the fp32 attention was rewritten on
//     that basis (attention.hip), and no arrangement of it (software pipelining inside a wave, more waves) can hide it.
//   * v_mfma_f32_16x16x32_f16 (an XDL operation, 16 cycles): the MFMA holds the vector issue for 8 of its 16 cycles (the rule of
//     MI355X_MICROARCH.md): two v_fma_f32 per MFMA fit, four or eight add up (8 + 4 n cycles per MFMA).
//   * no MFMA at all (two or four waves per SIMD): v_fma_f32 2.7 cycles, v_pk_mul_f32 / v_max3_f32 / v_cvt_pkrtz 4.3 (per element the
//     packed forms win by a fifth - in a vector-only phase such as an epilogue; beside MFMAs they do not), v_exp_f32 8.2.
// Both sides are volatile asm: left to the scheduler, hipcc moves all vector instructions behind the 64 MFMAs of the loop body
// (sched_group_barrier or not) and the waves of a SIMD then run their two phases in lock step, which measures something else.
// Also: the dependence pattern (CHAIN accumulators in rotation; 1 = every MFMA takes the previous one's result) does not matter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_f32_chain_lab.hip -o tools/bin/mfma_f32_chain_lab     (profiles/r05_mfma_f32_chain_lab.log)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

template <int CHAIN, int VALU, int KIND = 0, int MF = 0>
__global__ void __launch_bounds__(1024) chain_kernel(float* out, int iters, long long* cycles) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[CHAIN];
#pragma unroll
  for (int i = 0; i < CHAIN; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __shared__ float lds[1024];
  lds[threadIdx.x] = lane;
  __syncthreads();
  float a = lane * 1e-3f, b = 1.f - lane * 1e-3f, v = lane;
  typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(lane * 1e-3f + i); hb[i] = (_Float16)(1.f - i * 0.1f); }
  float w[4] = {1.f + lane, 2.f + lane, 3.f + lane, 4.f + lane};
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pk = {1.f + lane, 2.f}, pk2 = {1.0001f, 0.9999f};
  f32x2 pkw[4] = {pk, pk2, pk, pk2};
  const unsigned ldsaddr = (unsigned)(threadIdx.x & 1023) * 4;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      // volatile asm on both sides: the instruction stream IS this order (left to the scheduler, hipcc moves all the VALU
      // instructions behind the 64 MFMAs, sched_group_barrier or not, and the waves of a SIMD then run their phases in lock step)
      if (MF == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[k % CHAIN]) : "v"(a), "v"(b));
      else if (MF == 2) {}  // no MFMA: the vector instructions alone
      else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[k % CHAIN]) : "v"(ha), "v"(hb));
#pragma unroll
      for (int j = 0; j < VALU; ++j) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(w[j & 3]) : "v"(a), "v"(b));
        if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(w[j & 3]));
        if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk) : "v"(pk2));
        if (KIND == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(w[j & 3]) : "v"(a), "v"(b));
        if (KIND == 5) asm volatile("ds_read_b32 %0, %1" : "=v"(w[j & 3]) : "v"(ldsaddr));
        if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pkw[j & 3]) : "v"(pk2));
        if (KIND == 7) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(w[j & 3]) : "v"(a), "v"(b));
      }
    }
    if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < CHAIN; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + v + w[0] + w[1] + w[2] + w[3] + pk[0] + pk[1] + pkw[0][0] + pkw[1][1] + pkw[2][0] + pkw[3][1];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int CHAIN, int VALU, int KIND = 0, int MF = 0>
void run(float* out, long long* cyc, int threads, const char* what) {
  const int iters = 2000;
  hipEvent_t a, b;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&b));
  float best = 1e9f;
  for (int r = 0; r < 3; ++r) {
    HIP_OK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((chain_kernel<CHAIN, VALU, KIND, MF>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
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
  printf("%s chain %2d  valu %d  %d waves/SIMD  %-28s %7.2f ns per MFMA and SIMD  (%.1f cycles at 2.4 GHz)  [%lld ticks]\n", MF == 2 ? "no MFMA     " : MF ? "f16 16x16x32" : "f32 16x16x4 ", CHAIN, VALU,
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
    run<4, 4, 5>(out, cyc, threads, "+ 4 ds_read_b32");
    run<4, 0, 0, 1>(out, cyc, threads, "four chains");
    run<4, 4, 1, 1>(out, cyc, threads, "+ 4 independent v_fma");
    run<4, 8, 1, 1>(out, cyc, threads, "+ 8 independent v_fma");
    run<4, 2, 2, 1>(out, cyc, threads, "+ 2 v_exp_f32");
    run<4, 2, 3, 1>(out, cyc, threads, "+ 2 v_pk_mul_f32 (chained)");
    run<4, 4, 3, 1>(out, cyc, threads, "+ 4 v_pk_mul_f32 (chained)");
    run<4, 4, 6, 1>(out, cyc, threads, "+ 4 v_pk_mul_f32 (independent)");
    run<4, 4, 6, 0>(out, cyc, threads, "+ 4 v_pk_mul_f32 (independent)");
    run<4, 4, 7, 1>(out, cyc, threads, "+ 4 v_cvt_pk_f16_f32 (v_cvt_pkrtz)");
    run<4, 8, 1, 2>(out, cyc, threads, "8 independent v_fma");
    run<4, 8, 6, 2>(out, cyc, threads, "8 v_pk_mul_f32 (independent)");
    run<4, 8, 3, 2>(out, cyc, threads, "8 v_pk_mul_f32 (chained)");
    run<4, 8, 0, 2>(out, cyc, threads, "8 chained v_fma");
    run<4, 8, 2, 2>(out, cyc, threads, "8 v_exp_f32");
    run<4, 8, 4, 2>(out, cyc, threads, "8 v_max3_f32");
    run<4, 8, 7, 2>(out, cyc, threads, "8 v_cvt_pkrtz_f16_f32");
  }
  return 0;
}

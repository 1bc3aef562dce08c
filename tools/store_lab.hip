// Store-pattern microbenchmark: how fast can one CU stream an output tile with different row/segment shapes?
//   hipcc --offload-arch=gfx950 -O3 tools/store_lab.hip -o tools/bin/store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2);} } while (0)

// Each block (512 threads) writes `tiles` tiles of 256 rows x 512 bytes (a 256x256 bf16 output tile), row stride `ld`
// bytes.  SEG = contiguous bytes per row written by one wave-instruction; BPL = bytes per lane (8 or 16).
template <int SEG, int BPL>
__global__ void __launch_bounds__(512) store_kernel(char* out, long ld, int tilesN, int ntiles, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int LPS = SEG / BPL;        // lanes per row segment
  constexpr int RPI = 64 / LPS;         // rows per instruction
  f32x4 v = {1.f * lane, 2.f, 3.f, 4.f};
  for (int it = 0; it < iters; ++it) {
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
      const long m0 = (long)(t / tilesN) * 256, n0 = (long)(t % tilesN) * 512;
      // tile = 256 rows x 512 B = 128 KB = 8 waves x 16 KB; wave w owns rows [32w, 32w+32) (all 512 B)... or column
      // slices, it does not matter for the memory system: what matters is the per-instruction footprint.
      constexpr int INSTR = 16384 / (64 * BPL);
      char* base = out + (m0 + wave * 32) * ld + n0;
#pragma unroll 4
      for (int k = 0; k < INSTR; ++k) {
        const int seg = k * RPI * LPS + lane;          // linear index of this lane's BPL-byte piece inside the wave's 32x512B
        const int piece_row = seg / (512 / BPL), piece_col = seg % (512 / BPL);
        // footprint per instruction: RPI rows x SEG bytes when SEG <= 512
        const int row = (k * RPI + lane / LPS) % 32 ;
        const int colb = ((k * RPI + lane / LPS) / 32) * SEG + (lane % LPS) * BPL;
        char* p = base + (long)row * ld + colb;
        (void)piece_row; (void)piece_col;
        if (BPL == 16) *reinterpret_cast<f32x4*>(p) = v; else *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
      }
    }
  }
}

static int g_grid = 256;
template <int SEG, int BPL>
void run(const char* name, char* out, long ld, int M, int Nbytes, hipStream_t st) {
  const int tilesN = Nbytes / 512, ntiles = (M / 256) * tilesN;
  hipEvent_t e0, e1; HIP_OK(hipEventCreate(&e0)); HIP_OK(hipEventCreate(&e1));
  store_kernel<SEG, BPL><<<g_grid, 512, 0, st>>>(out, ld, tilesN, ntiles, 1);
  HIP_OK(hipEventRecord(e0, st));
  const int iters = 10;
  store_kernel<SEG, BPL><<<g_grid, 512, 0, st>>>(out, ld, tilesN, ntiles, iters);
  HIP_OK(hipEventRecord(e1, st)); HIP_OK(hipStreamSynchronize(st));
  float ms; HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)ntiles * 131072 * iters;
  printf("%-30s grid=%3d M=%d: %.3f ms  %.2f TB/s  %.1f B/clk/CU(@2.1GHz)\n", name, g_grid, M, ms, bytes / ms / 1e9,
         bytes / (ms * 1e-3) / g_grid / 2.1e9);
}

int main(int argc, char** argv) {
  hipStream_t st; HIP_OK(hipStreamCreate(&st));
  for (int grid : {256, 128, 64, 32, 8}) {
    g_grid = grid;
    const int M = 12544, Nbytes = 2304 * 2;
    char* out; HIP_OK(hipMalloc(&out, (size_t)M * Nbytes));
    run<32, 8>("16 rows x 32 B (8 B/lane)", out, Nbytes, M, Nbytes, st);
    run<128, 16>("8 rows x 128 B (16 B/lane)", out, Nbytes, M, Nbytes, st);
    HIP_OK(hipFree(out));
  }
  return 0;
}

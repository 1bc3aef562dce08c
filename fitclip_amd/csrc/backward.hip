// HBM-bound kernels of the training step (KD fine-tuning of the student, aligner/teacher_student.py:99-183): backward of
// LayerNorm / pooling / L2-normalisation / the two similarity losses / the embeddings, QuickGELU forward on a saved
// pre-activation, AdamW.  Every reduction has a fixed order (deterministic; no float atomics anywhere): two-stage
// partial sums, and a row-ordered segmented sum for the token-embedding gradient.
#include "common.h"

#include <algorithm>
#include <climits>
#include <cmath>

namespace fc {

namespace {

constexpr int kLnBlocks = 512;  // partial rows of the dgamma / dbeta reduction

template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16>(const bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return f32x4{static_cast<float>(v[0]), static_cast<float>(v[1]), static_cast<float>(v[2]), static_cast<float>(v[3])};
}
template <typename T> __device__ __forceinline__ void st4(T* p, const f32x4& v);
template <> __device__ __forceinline__ void st4<float>(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16>(bf16* p, const f32x4& v) {
  bf16x4 o;
  o[0] = static_cast<bf16>(v[0]); o[1] = static_cast<bf16>(v[1]);
  o[2] = static_cast<bf16>(v[2]); o[3] = static_cast<bf16>(v[3]);
  *reinterpret_cast<bf16x4*>(p) = o;
}

// ------------------------------------------------------------------------------------------ LayerNorm backward
// y = (x - mean) * rstd * gamma + beta  (slip.py:350-356, eps 1e-5).  One wave per row, the row in registers, statistics
// recomputed from the saved input exactly as the forward computes them:
//   xhat = (x - mean) rstd;  dxh = dy gamma;  dx = rstd (dxh - mean(dxh) - xhat mean(dxh xhat))
//   dgamma += dy xhat;  dbeta += dy       (per-lane partial sums over the wave's rows -> block -> P[block][2][D])
// out[src] = dx or out[src] += dx (residual-stream gradient).  Row i reads x[src], src = gather ? gather[i] : i; dy row
// i (compact) or src.
template <int D, typename TDY>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, long x_stride,
                                                            const int* __restrict__ gather, const TDY* __restrict__ dy,
                                                            long dy_stride, int dy_compact,
                                                            const float* __restrict__ gamma, float* out,
                                                            long out_stride, int accumulate, int rows,
                                                            float* __restrict__ P) {
  static_assert(D % 256 == 0 || D == 128, "width");
  constexpr int V4 = D / 256, REM = (D % 256) / 64;
  constexpr int NV = V4 > 0 ? V4 : 1, NR = REM > 0 ? REM : 1;
  __shared__ float red[3][2 * D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wpb = blockDim.x >> 6;
  f32x4 dg[NV], db[NV];
  float dgs[NR], dbs[NR];
#pragma unroll
  for (int i = 0; i < NV; ++i) dg[i] = db[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NR; ++i) dgs[i] = dbs[i] = 0.f;
  for (int row = blockIdx.x * wpb + wave; row < rows; row += gridDim.x * wpb) {
    const long src = gather ? gather[row] : row;
    const float* xr = x + src * x_stride;
    const TDY* dr = dy + (dy_compact ? (long)row : src) * dy_stride;
    f32x4 v[NV], d[NV];
    float s[NR], ds[NR];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      v[i] = *reinterpret_cast<const f32x4*>(xr + i * 256 + lane * 4);
      d[i] = ld4<TDY>(dr + i * 256 + lane * 4);
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      s[i] = xr[V4 * 256 + i * 64 + lane];
      ds[i] = static_cast<float>(dr[V4 * 256 + i * 64 + lane]);
      sum += s[i];
    }
    const float mean = wave_sum(sum) * (1.f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] -= mean;
        sq += v[i][e] * v[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      s[i] -= mean;
      sq += s[i] * s[i];
    }
    const float rstd = 1.f / sqrtf(wave_sum(sq) * (1.f / D) + 1e-5f);
    float m1 = 0.f, m2 = 0.f;  // sum(dxh), sum(dxh * xhat)
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gamma + i * 256 + lane * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] *= rstd;                       // xhat
        dg[i][e] += d[i][e] * v[i][e];
        db[i][e] += d[i][e];
        d[i][e] *= g4[e];                      // dxh
        m1 += d[i][e];
        m2 += d[i][e] * v[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      s[i] *= rstd;
      dgs[i] += ds[i] * s[i];
      dbs[i] += ds[i];
      ds[i] *= gamma[V4 * 256 + i * 64 + lane];
      m1 += ds[i];
      m2 += ds[i] * s[i];
    }
    m1 = wave_sum(m1) * (1.f / D);
    m2 = wave_sum(m2) * (1.f / D);
    float* orow = out + src * out_stride;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rstd * (d[i][e] - m1 - v[i][e] * m2);
      f32x4* dst = reinterpret_cast<f32x4*>(orow + i * 256 + lane * 4);
      if (accumulate) o += *dst;
      *dst = o;
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      const int c = V4 * 256 + i * 64 + lane;
      const float o = rstd * (ds[i] - m1 - s[i] * m2);
      orow[c] = accumulate ? orow[c] + o : o;
    }
  }
  // block reduction of the affine gradients: waves 1..3 park theirs in LDS, wave 0 adds them in a fixed order
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < V4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[wave - 1][i * 256 + lane * 4 + e] = dg[i][e];
        red[wave - 1][D + i * 256 + lane * 4 + e] = db[i][e];
      }
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      red[wave - 1][V4 * 256 + i * 64 + lane] = dgs[i];
      red[wave - 1][D + V4 * 256 + i * 64 + lane] = dbs[i];
    }
  }
  __syncthreads();
  if (wave == 0) {
    float* p = P + (size_t)blockIdx.x * 2 * D;
#pragma unroll
    for (int i = 0; i < V4; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int c = i * 256 + lane * 4 + e;
        float a = dg[i][e], b = db[i][e];
        for (int w = 0; w + 1 < wpb; ++w) {
          a += red[w][c];
          b += red[w][D + c];
        }
        p[c] = a;
        p[D + c] = b;
      }
    }
#pragma unroll
    for (int i = 0; i < REM; ++i) {
      const int c = V4 * 256 + i * 64 + lane;
      float a = dgs[i], b = dbs[i];
      for (int w = 0; w + 1 < wpb; ++w) {
        a += red[w][c];
        b += red[w][D + c];
      }
      p[c] = a;
      p[D + c] = b;
    }
  }
}

template <typename TDY>
int layernorm_bwd_dispatch(const float* x, long xs, const int* gather, const TDY* dy, long dys, int dy_compact,
                           const float* gamma, float* out, long os, int accumulate, int rows, int D, float* P,
                           int blocks, hipStream_t st) {
#define FC_LN_BWD(W) hipLaunchKernelGGL((layernorm_bwd_kernel<W, TDY>), dim3(blocks), dim3(256), 0, st, x, xs, gather, dy, dys, dy_compact, gamma, out, os, accumulate, rows, P)
  switch (D) {
    case 128: FC_LN_BWD(128); break;
    case 256: FC_LN_BWD(256); break;
    case 512: FC_LN_BWD(512); break;
    case 768: FC_LN_BWD(768); break;
    case 1024: FC_LN_BWD(1024); break;
    default: return fail(FC_EINVAL, "layernorm backward: unsupported width %d", D);
  }
#undef FC_LN_BWD
  FC_CHECK_LAUNCH("layernorm backward");
  return FC_OK;
}

// ------------------------------------------------------------------------------------------------ elementwise
template <typename T>
__global__ void __launch_bounds__(256) quickgelu_kernel(const T* __restrict__ in, T* __restrict__ out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = ld4<T>(in + i * 4);
    v = quick_gelu_f32x4(v);  // slip.py:359-361; the same function as the GEMM epilogue
    st4<T>(out + i * 4, v);
  }
}

// torch.optim.AdamW (defaults: amsgrad off, maximize off), one fused pass with torch's own operation order:
//   p *= 1 - lr wd;  m = lerp(m, g, 1 - b1);  v = b2 v + (1 - b2) g g;  p -= (lr / bc1) * (m / (sqrt(v) / sqrt(bc2) + eps))
// The scalar factors are computed in double on the host (as Python does) and rounded once.
struct AdamScalars {
  float decay, w1, b2, w2, sqrt_bc2, eps, step_size;
};
__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamScalars& k) {
#pragma clang fp contract(off)  // torch evaluates mul_, lerp_, addcmul_, sqrt, div, add_, addcdiv_ as separate steps
  p = p * k.decay;
  m = m + k.w1 * (g - m);
  v = v * k.b2 + k.w2 * (g * g);
  const float denom = sqrtf(v) / k.sqrt_bc2 + k.eps;
  p = p - k.step_size * (m / denom);
}
__global__ void __launch_bounds__(256) adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n,
                                                    const AdamScalars k) {
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 pp = *reinterpret_cast<f32x4*>(p + i * 4), mm = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(g + i * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = pp[e], b = mm[e], c = vv[e];
      adamw_one(a, gg[e], b, c, k);
      pp[e] = a; mm[e] = b; vv[e] = c;
    }
    *reinterpret_cast<f32x4*>(p + i * 4) = pp;
    *reinterpret_cast<f32x4*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const size_t i = n4 * 4 + threadIdx.x;
    adamw_one(p[i], g[i], m[i], v[i], k);
  }
}

// ------------------------------------------------------------------------------ pooling / normalisation backward
// out[b] = (1/F) sum_f z[b,f] / ||z[b,f]||   (clip_video_text_encoder.py:85-89)  =>
// dz[b,f] = (dout[b] - (dout[b] . u) u) / (F ||z[b,f]||),  u = z / ||z||.     One block per frame row; F = 1: L2 normalise.
__global__ void __launch_bounds__(256) pool_normalize_bwd_kernel(const float* __restrict__ z,
                                                                 const float* __restrict__ dout,
                                                                 float* __restrict__ dz, int frames, int dim) {
  __shared__ float red[2][4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* zr = z + (long)row * dim;
  const float* dr = dout + (long)(row / frames) * dim;
  float zv[4], dv[4], sq = 0.f, dot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256;
    zv[i] = c < dim ? zr[c] : 0.f;
    dv[i] = c < dim ? dr[c] : 0.f;
    sq += zv[i] * zv[i];
    dot += zv[i] * dv[i];
  }
  sq = wave_sum(sq);
  dot = wave_sum(dot);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = sq;
    red[1][tid >> 6] = dot;
  }
  __syncthreads();
  const float n2 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  const float zd = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const float nrm = sqrtf(n2);
  const float scale = 1.f / ((float)frames * nrm);
  const float proj = zd / n2;  // (dout . u) / ||z||
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + i * 256;
    if (c < dim) dz[(long)row * dim + c] = (dv[i] - proj * zv[i]) * scale;
  }
}

// ---------------------------------------------------------------------------------------------- loss backward
__device__ __forceinline__ float block_sum_b(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_b(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// lse[line] = logsumexp(line) of s [R, C]: blocks [0, R) rows, [R, R + C) columns
__global__ void __launch_bounds__(256) lse_lines_kernel(const float* __restrict__ s, int R, int C,
                                                        float* __restrict__ lse) {
  __shared__ float red[4];
  const bool col = blockIdx.x >= R;
  const int i = col ? blockIdx.x - R : blockIdx.x;
  const int n = col ? R : C;
  const long sl = col ? 1 : C, se = col ? C : 1;
  const float* line = s + i * sl;
  float mx = -__builtin_inff();
  for (int j = threadIdx.x; j < n; j += 256) mx = fmaxf(mx, line[j * se]);
  mx = block_max_b(mx, red);
  float sum = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) sum += expf(line[j * se] - mx);
  sum = block_sum_b(sum, red);
  if (threadIdx.x == 0) lse[blockIdx.x] = mx + logf(sum);
}

// dS[i,j] = c ( (softmax_row(S)[i,j] - tr) / R + (softmax_col(S)[i,j] - tc) / C ) for S [R, C];
//   NCE (square, loss.py:13-26): tr = tc = delta_ij;   KD (loss.py:29-39): tr / tc = softmax_row / softmax_col of T
__global__ void __launch_bounds__(256) loss_bwd_kernel(const float* __restrict__ s, const float* __restrict__ lse,
                                                       const float* __restrict__ t, const float* __restrict__ lse_t,
                                                       int R, int C, float coef, float* __restrict__ ds) {
  const size_t total = (size_t)R * C;
  const float cr = coef / (float)R, cc = coef / (float)C;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / C), j = (int)(idx - (size_t)i * C);
    const float v = s[idx];
    float gr = expf(v - lse[i]), gc = expf(v - lse[R + j]);
    if (t) {
      const float tv = t[idx];
      gr -= expf(tv - lse_t[i]);
      gc -= expf(tv - lse_t[R + j]);
    } else if (i == j) {
      gr -= 1.f;
      gc -= 1.f;
    }
    ds[idx] = gr * cr + gc * cc;
  }
}

// One block per line (blocks [0, n): rows, [n, 2n): columns) of the KD loss (loss.py:29-39):
//   out[line] = sum_k dKL_line/dt_k * t_k,   dKL/dt_k = pt_k ((log pt_k - log ps_k) - KL_line)
// (what the teacher-student temperature, which multiplies every teacher score, receives through the teacher scores).
__global__ void __launch_bounds__(256) kd_teacher_lines_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ lse_t, int R, int C,
                                                               float* __restrict__ out) {
  __shared__ float red[4];
  const bool col = blockIdx.x >= R;
  const int i = col ? blockIdx.x - R : blockIdx.x;
  const int n = col ? R : C;
  const long sl = col ? 1 : C, se = col ? C : 1;
  const float* ls = s + i * sl;
  const float* lt = t + i * sl;
  const float a = lse[blockIdx.x], b = lse_t[blockIdx.x];
  float kl = 0.f, w1 = 0.f, w2 = 0.f;  // sum pt a, sum pt a t, sum pt t
  for (int j = threadIdx.x; j < n; j += 256) {
    const float tv = lt[j * se];
    const float lpt = tv - b, lps = ls[j * se] - a;
    const float pt = expf(lpt);
    const float d = pt > 0.f ? pt * (lpt - lps) : 0.f;
    kl += d;
    w1 += d * tv;
    w2 += pt * tv;
  }
  kl = block_sum_b(kl, red);
  w1 = block_sum_b(w1, red);
  w2 = block_sum_b(w2, red);
  if (threadIdx.x == 0) out[blockIdx.x] = w1 - kl * w2;
}
// out[0] = sum(a[0..R)) / R + sum(a[R..R+C)) / C
__global__ void __launch_bounds__(256) sum_lines_kernel(const float* __restrict__ a, int R, int C,
                                                        float* __restrict__ out) {
  __shared__ float red[4];
  float sr = 0.f, sc = 0.f;
  for (int j = threadIdx.x; j < R; j += 256) sr += a[j];
  for (int j = threadIdx.x; j < C; j += 256) sc += a[R + j];
  sr = block_sum_b(sr, red);
  sc = block_sum_b(sc, red);
  if (threadIdx.x == 0) out[0] = sr / (float)R + sc / (float)C;
}

// out[0] = beta * out[0] + alpha * sum_i a[i] b[i]   (single block, fixed order)
__global__ void __launch_bounds__(1024) dot_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t n,
                                                   float alpha, float beta, float* __restrict__ out) {
  __shared__ float red[16];
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 1024) acc += a[i] * b[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < 16; ++w) s += red[w];
    out[0] = (beta != 0.f ? beta * out[0] : 0.f) + alpha * s;
  }
}

// ---------------------------------------------------------------------------------------------- embeddings
// P[t, c] = sum_i g[(i * S + t), c]     (gradient of a positional embedding: sum over the sequences)
__global__ void __launch_bounds__(256) seq_sum_kernel(const float* __restrict__ g, int n_seq, int S, int D,
                                                      float* __restrict__ P) {
  __shared__ f32x4 red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strips = (D + 255) / 256;
  const int t = blockIdx.x / strips, strip = blockIdx.x - t * strips;
  const int c = strip * 256 + lane * 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (c < D)
    for (int i = wave; i < n_seq; i += 4) s += *reinterpret_cast<const f32x4*>(g + ((size_t)i * S + t) * D + c);
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && c < D)
    *reinterpret_cast<f32x4*>(P + (size_t)t * D + c) = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// ---- token_embedding gradient: dtok[id] = sum of g[row] over the rows that hold `id`, IN ROW ORDER.  No float atomics:
// SOT / EOT / pad rows collide in every caption, and an order-dependent sum there would make every later step of a run
// differ in the last bits from its own replay (a resumed run, a second rank group).  Two kernels:
//
// (1) `token_order_kernel`, ONE workgroup: a stable counting sort of the row indices by id.  Rows are walked in chunks of
//     1024; a row's position inside its id's segment = (rows of that id in earlier chunks, `cnt[id]` so far) + (earlier
//     rows of the chunk with the same id, counted against the chunk's ids in LDS, four per ds_read_b128).  The first row
//     of an id in the chunk then advances `cnt[id]` by the chunk's count - one writer per id, plain stores, ordered
//     by the workgroup barrier.  After the walk `cnt` is the histogram; an exclusive scan over the vocabulary gives every
//     segment's start, and `perm[start[id] + rank[row]] = row` lists each segment in ascending row order.
// (2) `token_segment_sum_kernel`: one wave per (row, 64-column slice); only the FIRST row of an id works: it walks its
//     id's segment of `perm` and adds the rows in that order (eight loads in flight, adds strictly sequential).
//     BOUND: the longest segment is a serial chain.  The pad id holds (77 - mean caption length) rows of every caption - about
//     60 % of all rows - so one wave per 64-column slice walks ~24 k of the 39 k rows of a 512-caption batch: ~3 ms (the step is
//     ~4 s there, ~50 us at 64 captions); it grows linearly with the captions of ONE rank, not with the global batch.  A fixed
//     tree (sub-blocks summed by separate waves, partials combined in index order) would keep the result reproducible but
//     change the bits away from numpy's sequential `np.add.at`, which the tests pin; not done while the cost is < 0.1 %.
__global__ void __launch_bounds__(1024) token_order_kernel(const int64_t* __restrict__ ids, int rows, int vocab,
                                                           int* __restrict__ cnt, int* __restrict__ start,
                                                           int* __restrict__ rank, int* __restrict__ perm) {
  __shared__ __attribute__((aligned(16))) int sid[1024];
  __shared__ int wtot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int v = tid; v < vocab; v += 1024) cnt[v] = 0;
  for (int base = 0; base < rows; base += 1024) {
    const int r = base + tid;
    int id = -1;
    if (r < rows) {
      const long raw = ids[r];
      id = (int)(raw < 0 ? 0 : (raw >= vocab ? vocab - 1 : raw));  // same clamp as the forward gather
    }
    sid[tid] = id;
    __syncthreads();  // chunk ids visible; the previous chunk's cnt updates too
    int before = 0, total = 0, prev = 0;
    if (id >= 0) {
      for (int j = 0; j < 1024; j += 4) {
        const int4 v = *reinterpret_cast<const int4*>(&sid[j]);
        const int e0 = v.x == id, e1 = v.y == id, e2 = v.z == id, e3 = v.w == id;
        total += e0 + e1 + e2 + e3;
        before += (e0 & (j < tid)) + (e1 & (j + 1 < tid)) + (e2 & (j + 2 < tid)) + (e3 & (j + 3 < tid));
      }
      prev = cnt[id];
      rank[r] = prev + before;
    }
    __syncthreads();  // every reader of cnt[id] and of sid is done
    if (id >= 0 && before == 0) cnt[id] = prev + total;
  }
  __syncthreads();
  // exclusive scan of cnt over the vocabulary: a contiguous slice per thread, slices combined through wave + LDS scans
  const int per = (vocab + 1023) / 1024;
  const int v0 = min(tid * per, vocab), v1 = min(v0 + per, vocab);
  int mine = 0;
  for (int v = v0; v < v1; ++v) mine += cnt[v];
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o, 64);
    if (lane >= o) incl += up;
  }
  if (lane == 63) wtot[wave] = incl;
  __syncthreads();
  int offset = incl - mine;
  for (int w = 0; w < wave; ++w) offset += wtot[w];
  for (int v = v0; v < v1; ++v) {
    start[v] = offset;
    offset += cnt[v];
  }
  __syncthreads();
  for (int r = tid; r < rows; r += 1024) {
    const long raw = ids[r];
    const int id = (int)(raw < 0 ? 0 : (raw >= vocab ? vocab - 1 : raw));
    perm[start[id] + rank[r]] = r;
  }
}

__global__ void __launch_bounds__(256) token_segment_sum_kernel(const int64_t* __restrict__ ids,
                                                                const float* __restrict__ g, float* __restrict__ dtok,
                                                                int rows, int D, int vocab, const int* __restrict__ cnt,
                                                                const int* __restrict__ start,
                                                                const int* __restrict__ rank,
                                                                const int* __restrict__ perm, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int slices = (D + 63) >> 6;
  const long widx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int row = (int)(widx / slices), c = (int)(widx - (long)row * slices) * 64 + lane;
  if (row >= rows || rank[row] != 0) return;  // wave-uniform: one wave per distinct id and column slice goes on
  const long raw = ids[row];
  const int id = (int)(raw < 0 ? 0 : (raw >= vocab ? vocab - 1 : raw));
  const int* seg = perm + start[id];
  const int n = cnt[id];
  if (c >= D) return;
  float acc = 0.f;
  int k = 0;
  for (; k + 8 <= n; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = g[(size_t)seg[k + u] * D + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; k < n; ++k) acc += g[(size_t)seg[k] * D + c];
  float* out = dtok + (size_t)id * D + c;
  *out = accumulate ? *out + acc : acc;
}

// x[i * S, :] = cls + pos0    (the CLS rows of the visual token stream; the fused inference entry never stores them)
__global__ void __launch_bounds__(256) fill_cls_kernel(float* __restrict__ x, const float* __restrict__ cls,
                                                       const float* __restrict__ pos0, int n_seq, int S, int D) {
  const size_t total = (size_t)n_seq * D;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t img = i / D;
    const int c = (int)(i - img * D);
    x[img * S * D + c] = cls[c] + pos0[c];
  }
}

inline int flat_blocks_b(size_t items) {
  const size_t b = (items + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

size_t layernorm_bwd_scratch_bytes(int D) { return (size_t)kLnBlocks * 2 * D * sizeof(float); }

// dgamma / dbeta: [D] each, out = beta_acc * out + sum (beta_acc 0 or 1)
int launch_layernorm_backward(const float* x, long x_stride, const int* gather, const void* dy, int dy_kind,
                              long dy_stride, int dy_compact, const float* gamma, float* out, long out_stride,
                              int accumulate, int rows, int D, float* dgamma, float* dbeta, float beta_acc,
                              float* scratch, size_t scratch_bytes, hipStream_t st) {
  if (rows <= 0) return FC_OK;
  if (scratch_bytes < layernorm_bwd_scratch_bytes(D)) return fail(FC_ENOMEM, "layernorm backward: scratch too small");
  if ((x_stride % 4) || (out_stride % 4) || (dy_stride % 4) ||
      (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)gamma | (uintptr_t)out | (uintptr_t)scratch | (uintptr_t)dgamma |
        (uintptr_t)dbeta) & 15))
    return fail(FC_EINVAL, "layernorm backward: operands must be 16-byte aligned");
  const int blocks = std::min((rows + 3) / 4, kLnBlocks);
  int rc = dy_kind == PREC_BF16
               ? layernorm_bwd_dispatch<bf16>(x, x_stride, gather, (const bf16*)dy, dy_stride, dy_compact, gamma, out,
                                              out_stride, accumulate, rows, D, scratch, blocks, st)
               : layernorm_bwd_dispatch<float>(x, x_stride, gather, (const float*)dy, dy_stride, dy_compact, gamma, out,
                                               out_stride, accumulate, rows, D, scratch, blocks, st);
  if (rc != FC_OK) return rc;
  // P is [blocks][2 D]: dgamma = columns [0, D), dbeta = columns [D, 2 D)
  // (two strided reductions over the same partial rows: treat P as `blocks` planes of 2 D and reduce each half)
  rc = launch_reduce_partials(scratch, blocks, 1, 2 * D, scratch, 2 * D, 1.f, 0.f, st);  // in place into plane 0
  if (rc != FC_OK) return rc;
  rc = launch_reduce_partials(scratch, 1, 1, D, dgamma, D, 1.f, beta_acc, st);
  if (rc != FC_OK) return rc;
  return launch_reduce_partials(scratch + D, 1, 1, D, dbeta, D, 1.f, beta_acc, st);
}

int launch_quickgelu(const void* in, void* out, int kind, size_t n, hipStream_t st) {
  if (n == 0) return FC_OK;
  if (n % 4 || (((uintptr_t)in | (uintptr_t)out) & 15)) return fail(FC_EINVAL, "quickgelu: alignment");
  if (kind == PREC_BF16)
    hipLaunchKernelGGL(quickgelu_kernel<bf16>, dim3(flat_blocks_b(n / 4)), dim3(256), 0, st, (const bf16*)in, (bf16*)out, n / 4);
  else
    hipLaunchKernelGGL(quickgelu_kernel<float>, dim3(flat_blocks_b(n / 4)), dim3(256), 0, st, (const float*)in, (float*)out, n / 4);
  FC_CHECK_LAUNCH("quickgelu");
  return FC_OK;
}

int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps,
                 double wd, int step, hipStream_t st) {
  if (n == 0) return FC_OK;
  if (step < 1) return fail(FC_EINVAL, "adamw: step counts from 1");
  const double bc1 = 1.0 - std::pow(b1, step), bc2 = 1.0 - std::pow(b2, step);
  AdamScalars k;
  k.decay = (float)(1.0 - lr * wd); k.w1 = (float)(1.0 - b1); k.b2 = (float)b2; k.w2 = (float)(1.0 - b2);
  k.sqrt_bc2 = (float)std::sqrt(bc2); k.eps = (float)eps; k.step_size = (float)(lr / bc1);
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(FC_EINVAL, "adamw: alignment");
  hipLaunchKernelGGL(adamw_kernel, dim3(flat_blocks_b(std::max<size_t>(1, n / 4))), dim3(256), 0, st, p, g, m, v, n, k);
  FC_CHECK_LAUNCH("adamw");
  return FC_OK;
}

int launch_pool_normalize_backward(const float* z, const float* dout, float* dz, int n_clips, int frames, int dim,
                                   hipStream_t st) {
  if (n_clips <= 0) return FC_OK;
  if (frames <= 0 || dim <= 0 || dim > 1024) return fail(FC_EINVAL, "pool_normalize backward: frames=%d dim=%d", frames, dim);
  hipLaunchKernelGGL(pool_normalize_bwd_kernel, dim3(n_clips * frames), dim3(256), 0, st, z, dout, dz, frames, dim);
  FC_CHECK_LAUNCH("pool_normalize backward");
  return FC_OK;
}

// scores [R, C]; ws: (R + C) floats (NCE) or 2 (R + C) floats (KD)
int launch_loss_backward(const float* scores, const float* teacher, int R, int C, float coef, float* dscores, float* ws,
                         hipStream_t st) {
  if (R <= 0 || C <= 0) return fail(FC_EINVAL, "loss backward: %d x %d", R, C);
  if (!teacher && R != C) return fail(FC_EINVAL, "loss backward: the NCE loss needs a square score matrix");
  hipLaunchKernelGGL(lse_lines_kernel, dim3(R + C), dim3(256), 0, st, scores, R, C, ws);
  if (teacher) hipLaunchKernelGGL(lse_lines_kernel, dim3(R + C), dim3(256), 0, st, teacher, R, C, ws + R + C);
  hipLaunchKernelGGL(loss_bwd_kernel, dim3(flat_blocks_b((size_t)R * C)), dim3(256), 0, st, scores, ws, teacher,
                     ws + R + C, R, C, coef, dscores);
  FC_CHECK_LAUNCH("loss backward");
  return FC_OK;
}

// out[0] = sum_ij dKD/dteacher_ij * teacher_ij  ("batchmean", rows + columns); ws: 3 (R + C) floats
int launch_kd_teacher_scale_grad(const float* scores, const float* teacher, int R, int C, float* out, float* ws,
                                 hipStream_t st) {
  if (R <= 0 || C <= 0) return fail(FC_EINVAL, "kd teacher grad: %d x %d", R, C);
  const int L = R + C;
  hipLaunchKernelGGL(lse_lines_kernel, dim3(L), dim3(256), 0, st, scores, R, C, ws);
  hipLaunchKernelGGL(lse_lines_kernel, dim3(L), dim3(256), 0, st, teacher, R, C, ws + L);
  hipLaunchKernelGGL(kd_teacher_lines_kernel, dim3(L), dim3(256), 0, st, scores, teacher, ws, ws + L, R, C, ws + 2 * L);
  hipLaunchKernelGGL(sum_lines_kernel, dim3(1), dim3(256), 0, st, ws + 2 * L, R, C, out);
  FC_CHECK_LAUNCH("kd teacher grad");
  return FC_OK;
}

int launch_dot(const float* a, const float* b, size_t n, float alpha, float beta, float* out, hipStream_t st) {
  hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, st, a, b, n, alpha, beta, out);
  FC_CHECK_LAUNCH("dot");
  return FC_OK;
}

int launch_seq_sum(const float* g, int n_seq, int S, int D, float* out_pos, float* out_row0, float beta, float* scratch,
                   size_t scratch_bytes, hipStream_t st) {
  if (n_seq <= 0 || S <= 0) return FC_OK;
  if (D % 4 || scratch_bytes < (size_t)S * D * sizeof(float)) return fail(FC_ENOMEM, "seq_sum: scratch / width");
  hipLaunchKernelGGL(seq_sum_kernel, dim3(S * ((D + 255) / 256)), dim3(256), 0, st, g, n_seq, S, D, scratch);
  FC_CHECK_LAUNCH("seq_sum");
  int rc = launch_reduce_partials(scratch, 1, S, D, out_pos, D, 1.f, beta, st);
  if (rc == FC_OK && out_row0) rc = launch_reduce_partials(scratch, 1, 1, D, out_row0, D, 1.f, beta, st);
  return rc;
}

size_t token_grad_scratch_bytes(int rows, int vocab) { return ((size_t)2 * vocab + (size_t)2 * rows) * sizeof(int); }

// dtok[id, :] (+)= sum of g[row, :] over the rows with ids[row] == id, summed in row order; rows of dtok whose id does not
// occur are left alone (the caller zeroes the table when it does not accumulate)
int launch_token_grad(const int64_t* ids, const float* g, float* dtok, int rows, int D, int vocab, int accumulate,
                      void* scratch, size_t scratch_bytes, hipStream_t st) {
  if (rows <= 0) return FC_OK;
  if (vocab <= 0 || D <= 0) return fail(FC_EINVAL, "token gradient: empty table");
  if (!scratch || scratch_bytes < token_grad_scratch_bytes(rows, vocab) || ((uintptr_t)scratch & 15))
    return fail(FC_ENOMEM, "token gradient: scratch needs %zu bytes", token_grad_scratch_bytes(rows, vocab));
  int* cnt = static_cast<int*>(scratch);
  int* start = cnt + vocab;
  int* rank = start + vocab;
  int* perm = rank + rows;
  hipLaunchKernelGGL(token_order_kernel, dim3(1), dim3(1024), 0, st, ids, rows, vocab, cnt, start, rank, perm);
  FC_CHECK_LAUNCH("token order");
  const long waves = (long)rows * ((D + 63) / 64);
  hipLaunchKernelGGL(token_segment_sum_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, ids, g, dtok, rows, D,
                     vocab, cnt, start, rank, perm, accumulate);
  FC_CHECK_LAUNCH("token segment sum");
  return FC_OK;
}

int launch_fill_cls(float* x, const float* cls, const float* pos0, int n_seq, int S, int D, hipStream_t st) {
  if (n_seq <= 0) return FC_OK;
  hipLaunchKernelGGL(fill_cls_kernel, dim3(flat_blocks_b((size_t)n_seq * D)), dim3(256), 0, st, x, cls, pos0, n_seq, S, D);
  FC_CHECK_LAUNCH("fill_cls");
  return FC_OK;
}

}  // namespace fc

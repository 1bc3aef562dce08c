// Scoring tail of `command=evaluate`: rank of the matching video inside each text row, and the NCE / teacher-student
// KD similarity losses.  All fp32, HBM/L2-bound, tiny next to the encoders.
//   ranks ........ aligner/metrics.py:16-20 (index of the target in the descending argsort of the row)
//   nce_loss ..... aligner/loss.py:13-26   (mean(-log_softmax(S).diag) + the same on S^T)
//   kd loss ...... aligner/loss.py:29-39   (KL(softmax(teacher) || softmax(student)), "batchmean", rows + columns)
#include "common.h"

namespace fc {

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// rank[i] = #{j : s[i,j] > s[i,t]} + #{j < t : s[i,j] == s[i,t]},  t = i + target_offset  (stable descending order)
__global__ void __launch_bounds__(256) ranks_kernel(const float* __restrict__ s, int ld, int n_cols, int target_offset,
                                                    const int32_t* __restrict__ targets, int32_t* __restrict__ ranks) {
  __shared__ int red[4];
  const int i = blockIdx.x;
  int t = targets ? targets[i] : i + target_offset;
  t = t < 0 ? 0 : (t >= n_cols ? n_cols - 1 : t);  // never read outside the row
  const float* row = s + (long)i * ld;
  const float ref = row[t];
  int cnt = 0;
  for (int j = threadIdx.x; j < n_cols; j += 256) {
    const float v = row[j];
    cnt += (v > ref) || (v == ref && j < t);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) ranks[i] = red[0] + red[1] + red[2] + red[3];
}

// One block per line (row: stride_e = 1, stride_l = n; column: stride_e = n, stride_l = 1).
// nce[line] = logsumexp(line) - line[line]
__global__ void __launch_bounds__(256) nce_lines_kernel(const float* __restrict__ s, int n, long stride_l, long stride_e,
                                                        float* __restrict__ out) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  const float* line = s + i * stride_l;
  float mx = -__builtin_inff();
  for (int j = threadIdx.x; j < n; j += 256) mx = fmaxf(mx, line[j * stride_e]);
  mx = block_max(mx, red);
  float sum = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) sum += expf(line[j * stride_e] - mx);
  sum = block_sum(sum, red);
  if (threadIdx.x == 0) out[i] = (mx + logf(sum)) - line[i * stride_e];
}

// kd[line] = sum_j pt_j * (log pt_j - log ps_j)   with pt = softmax(teacher line), ps = softmax(student line)
__global__ void __launch_bounds__(256) kd_lines_kernel(const float* __restrict__ s, const float* __restrict__ tch,
                                                       int n, long stride_l, long stride_e, float* __restrict__ out) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  const float* ls = s + i * stride_l;
  const float* lt = tch + i * stride_l;
  float ms = -__builtin_inff(), mt = -__builtin_inff();
  for (int j = threadIdx.x; j < n; j += 256) {
    ms = fmaxf(ms, ls[j * stride_e]);
    mt = fmaxf(mt, lt[j * stride_e]);
  }
  ms = block_max(ms, red);
  mt = block_max(mt, red);
  float ss = 0.f, st = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    ss += expf(ls[j * stride_e] - ms);
    st += expf(lt[j * stride_e] - mt);
  }
  ss = block_sum(ss, red);
  st = block_sum(st, red);
  const float lse_s = ms + logf(ss), lse_t = mt + logf(st);
  float kl = 0.f;
  for (int j = threadIdx.x; j < n; j += 256) {
    const float lpt = lt[j * stride_e] - lse_t;
    const float lps = ls[j * stride_e] - lse_s;
    const float pt = expf(lpt);
    kl += pt > 0.f ? pt * (lpt - lps) : 0.f;  // F.kl_div: zero where the target is zero
  }
  kl = block_sum(kl, red);
  if (threadIdx.x == 0) out[i] = kl;
}

// out[0] = sum(a[0..na)) / na + sum(b[0..nb)) / nb   (mean over rows + mean over columns; "batchmean" of the KD loss
// divides each direction by ITS number of lines: F.kl_div(input, ...) / input.size(0), loss.py:29-39)
__global__ void __launch_bounds__(256) finish_mean_kernel(const float* __restrict__ a, int na,
                                                          const float* __restrict__ b, int nb,
                                                          float* __restrict__ out) {
  __shared__ float red[4];
  float sa = 0.f, sb = 0.f;
  for (int j = threadIdx.x; j < na; j += 256) sa += a[j];
  for (int j = threadIdx.x; j < nb; j += 256) sb += b[j];
  sa = block_sum(sa, red);
  sb = block_sum(sb, red);
  if (threadIdx.x == 0) out[0] = sa / (float)na + sb / (float)nb;
}

}  // namespace

int launch_ranks(const float* scores, int ld, int n_rows, int n_cols, int target_offset, const int32_t* targets,
                 int32_t* ranks, hipStream_t stream) {
  if (n_rows <= 0) return FC_OK;
  if (n_cols <= 0 || ld < n_cols || (!targets && (target_offset < 0 || target_offset + n_rows > n_cols)))
    return fail(FC_EINVAL, "ranks: rows=%d cols=%d offset=%d ld=%d", n_rows, n_cols, target_offset, ld);
  hipLaunchKernelGGL(ranks_kernel, dim3(n_rows), dim3(256), 0, stream, scores, ld, n_cols, target_offset, targets,
                     ranks);
  FC_CHECK_LAUNCH("ranks");
  return FC_OK;
}

int launch_nce_loss(const float* scores, int n, float* out, float* ws, hipStream_t stream) {
  if (n <= 0) return fail(FC_EINVAL, "nce_loss: n=%d", n);
  hipLaunchKernelGGL(nce_lines_kernel, dim3(n), dim3(256), 0, stream, scores, n, (long)n, 1L, ws);
  hipLaunchKernelGGL(nce_lines_kernel, dim3(n), dim3(256), 0, stream, scores, n, 1L, (long)n, ws + n);
  hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(256), 0, stream, ws, n, ws + n, n, out);
  FC_CHECK_LAUNCH("nce_loss");
  return FC_OK;
}

// scores / teacher [rows, cols] contiguous (rows = videos, cols = texts or prompts); ws: rows + cols floats
int launch_kd_loss(const float* scores, const float* teacher, int rows, int cols, float* out, float* ws,
                   hipStream_t stream) {
  if (rows <= 0 || cols <= 0) return fail(FC_EINVAL, "kd_loss: %d x %d", rows, cols);
  hipLaunchKernelGGL(kd_lines_kernel, dim3(rows), dim3(256), 0, stream, scores, teacher, cols, (long)cols, 1L, ws);
  hipLaunchKernelGGL(kd_lines_kernel, dim3(cols), dim3(256), 0, stream, scores, teacher, rows, 1L, (long)cols, ws + rows);
  hipLaunchKernelGGL(finish_mean_kernel, dim3(1), dim3(256), 0, stream, ws, rows, ws + rows, cols, out);
  FC_CHECK_LAUNCH("kd_loss");
  return FC_OK;
}

}  // namespace fc

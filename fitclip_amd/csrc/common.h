// Shared declarations for the fitclip HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/fitclip_hip.h"
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace fc {

using bf16 = __bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum Precision : int { PREC_F32 = 0, PREC_BF16 = 1 };

// error codes: fc_status of the public header

void set_error(const std::string& msg);
int fail(int code, const char* fmt, ...);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel): the attribute belongs to the function on the
// CURRENT device, and a process may hold handles on several devices.  Thread-safe.  Returns hipSuccess or the error.
hipError_t raise_dynamic_lds(const void* kernel, int bytes);
// compute units of the CURRENT device (memoised per device; 256 if the query fails): grid size of the persistent kernels
int device_cus();

#define FC_CHECK_LAUNCH(what)                                                          \
  do {                                                                                 \
    hipError_t _e = hipGetLastError();                                                 \
    if (_e != hipSuccess) return ::fc::fail(FC_ELAUNCH, "%s: %s", what, hipGetErrorString(_e)); \
  } while (0)

// ------------------------------------------------------------------------------------------------ GEMM
// C[M,N] = epilogue(A[M,K] . W[N,K]^T).  A and W are both K-contiguous (W is a torch Linear weight as stored).
enum Epilogue : int {
  EPI_BIAS_T = 0,     // C(T)   = acc + bias
  EPI_GELU_T = 1,     // C(T)   = quickgelu(acc + bias)
  EPI_RESID_F32 = 2,  // C(f32) += acc + bias          (residual stream update, in place)
  EPI_PATCH_F32 = 3,  // C(f32)[m + m/P + 1] = acc + pos[(m % P) + 1]   (patch embedding into the token stream)
  EPI_STORE_F32 = 4,  // C(f32) = alpha * acc (+ bias if given)
  EPI_DGELU_T = 5,    // C(T)   = acc * quickgelu'(aux(T)[m, n])   (backward of c_fc's activation; aux laid out as C)
  EPI_BIAS_F32 = 6,   // gemm_split3 only: C(f32) = acc + bias                  (three-plane operands in, fp32 out)
  EPI_GELU_X3 = 7,    // gemm_split3 only: C(x3 rows [M, 4 N bf16]) = three_planes(quickgelu(acc + bias))   (the next GEMM's A operand)
  EPI_RESID3_F32 = 8, // gemm_split3 only: C(f32) += acc + bias           (three-plane operands in; the residual stream updated in place)
  EPI_RANKS_I32 = 9,  // f32 only (launch_similarity_ranks): C(int32[M]) += #{n : s[m,n] > s[m,t_m] or (== and n < t_m)}, s = alpha acc
  EPI_GELU_X2 = 10,   // gemm_split2 only: C(x2 rows [M, 2 N fp16]) = two_planes(quickgelu(acc + bias))   (the next GEMM's A operand)
};

// ---- split-fp32 operands.  An fp32 number is exactly the sum of three bf16 numbers, x = p1 + p2 + p3 (p1 = bf16(x),
// p2 = bf16(x - p1), p3 = bf16(x - p1 - p2), round to nearest), and a product x * y is recovered to 2^-26 from six bf16
// products p1q1 + p1q2 + p2q1 + p2q2 + p1q3 + p3q1, each exact in the bf16 MFMA's fp32 accumulator.
// Three-plane ("x3") rows, the operand format of gemm_split3.h: every plane is stored ONCE.  Every 16 columns of a row are
// one 128-byte line [p1 x16 | p2 x16 | p3 x16 | 32 bytes never read or written]; the six products are formed from
// registers by the kernel.  A row of K columns is K / 16 lines = 4 K bf16 positions (6 K bytes carry data).  (Round 2 laid
// the six products out along K - [p1 p1 p2 p2 p1 p3] x [q1 q2 q1 q2 q3 q1], 12 bytes per value - for the unmodified bf16
// dot-product kernel.)
constexpr int KIND_X3 = 3;             // element-kind argument of the row kernels: output = x3 rows
constexpr int X3_GROUP = 16;           // columns per line
constexpr int X3_GROUP_BYTES = 128;    // bytes per line
constexpr long x3_row_elems(long K) { return K / X3_GROUP * (X3_GROUP_BYTES / 2); }  // bf16 positions per row = 4 K
#ifdef __HIPCC__
// x = p1 + p2 + p3 exactly.  Contraction is switched off: hipcc would otherwise fuse `x - p1` with the multiply that
// produced x (fma on the UNROUNDED product), and the planes would describe a number that is not the fp32 value x.
__device__ __forceinline__ void split3(const f32x4& x, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    p1[e] = static_cast<bf16>(x[e]);
    const float r1 = x[e] - static_cast<float>(p1[e]);
    p2[e] = static_cast<bf16>(r1);
    p3[e] = static_cast<bf16>(r1 - static_cast<float>(p2[e]));
  }
}
#endif

// ---- two-plane fp16 operands ("x2" rows, gemm_split2.h): THREE fp16 products per fp32 product.
// fp16 carries 11 significand bits, so an fp32 value is h1 + 2^-11 h2 to 22 bits (h1 = fp16(x), h2 = fp16((x - h1) 2^11): the
// residual is stored SCALED, so it stays in fp16's normal range wherever h1 does - for every |x| in [2^-14, 65504] the pair is
// exact to 2^-23 |x|, below that to 2^-36 absolute) and a weight is g1 + g2 with g1 = fp16(s w), g2 = fp16(s w - g1), s a power
// of two per TENSOR chosen at pack time so that max |s w| lies in [2^14, 2^15) (the residual of every weight above 2^-18 of the
// largest one is a normal fp16 number; smaller ones keep 2^-40 of the largest, absolute).  The product is recovered from
//     h1 g1 + h1 g2 + h2 (2^-11 g1)            (the dropped h2 g2 term is 2^-22 of the product)
// - every term exact in the fp16 MFMA's fp32 accumulator; the third weight operand 2^-11 g1 is formed from the g1 FRAGMENT in
// registers (v_pk_mul_f16, exact for g1 >= 2^-3) - and the accumulator is multiplied by 1 / s in the epilogue (exact).
// Every 32 columns of a row are one 128-byte line [h1 x32 | h2 x32]: 4 bytes per value, no padding - the bytes of an fp32 row.
// Values beyond fp16's range (|x| > 65504) cannot be written: the producers set a device-side flag (fc_handle::sat_flag) and
// the planes hold infinities (the error is loud: NaNs downstream, FC_ERANGE from the next tower call) - never silently wrong.
constexpr int KIND_X2 = 4;             // element-kind argument of the row kernels: output = x2 rows
constexpr int X2_GROUP = 32;           // columns per line
constexpr int X2_GROUP_BYTES = 128;    // bytes per line
constexpr long x2_row_elems(long K) { return 2 * K; }  // fp16 positions per row
constexpr float X2_RESID_SCALE = 2048.f;               // 2^11
#ifdef __HIPCC__
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// activation planes of four values (contraction off: `x - h1` must be the exact fp32 difference of the ROUNDED x)
__device__ __forceinline__ void split2(const f32x4& x, f16x4& h1, f16x4& h2) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h1[e] = static_cast<_Float16>(x[e]);
    h2[e] = static_cast<_Float16>((x[e] - static_cast<float>(h1[e])) * X2_RESID_SCALE);
  }
}
// The same planes in four instructions per PAIR instead of six: h1 as one packed conversion, h2 = fp16(fma(h1, -2^11, 2^11 x)) on the
// mixed-precision fma (v_fma_mixlo / mixhi_f16 read h1 straight from the packed register and write either half of the packed h2) -
// 2^11 x is exact, the fma's exact value 2^11 (x - h1) is representable, so its one rounding is the rounding of split2's conversion:
// bit-identical planes.  hipcc has no builtin for the mix instructions and picks them for one pair in two at best (the other gets
// separate conversions and a v_perm), hence inline asm; plain VALU instructions, no memory, no hazard of their own.
__device__ __forceinline__ void split2_mix(const f32x4& x, f16x4& h1, f16x4& h2) {
  const f32x4 xs = x * X2_RESID_SCALE;
  unsigned a1, a2, b1, b2;
  const float neg = -X2_RESID_SCALE;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a1) : "v"(x[0]), "v"(x[1]));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(b1) : "v"(x[2]), "v"(x[3]));
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(a2) : "v"(a1), "v"(neg), "v"(xs[0]));
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(b2) : "v"(b1), "v"(neg), "v"(xs[2]));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a2) : "v"(a1), "v"(neg), "v"(xs[1]));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(b2) : "v"(b1), "v"(neg), "v"(xs[3]));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  h1 = __builtin_bit_cast(f16x4, u32x2{a1, b1});
  h2 = __builtin_bit_cast(f16x4, u32x2{a2, b2});
}
// weight planes of four values, xs = s * w: UNSCALED residual
__device__ __forceinline__ void split2w(const f32x4& xs, f16x4& g1, f16x4& g2) {
#pragma clang fp contract(off)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    g1[e] = static_cast<_Float16>(xs[e]);
    g2[e] = static_cast<_Float16>(xs[e] - static_cast<float>(g1[e]));
  }
}
#endif

struct GemmArgs {
  const void* A;      // T [M, lda]
  const void* W;      // T [N, ldw]
  const float* bias;  // [N] or nullptr
  void* C;            // [M(+), ldc]
  const float* aux;   // EPI_PATCH_F32: positional embedding [P + 1, N]; EPI_DGELU_T: pre-activation, T [M, ldc]
  float alpha;        // EPI_STORE_F32
  int M, N, K;
  int lda, ldw, ldc;  // in elements
  int P;              // EPI_PATCH_F32: patches per image
  int nblock;         // pipelined kernel: N-tiles per L2-resident weight block (0 = all)
  int nsplit;         // pipelined kernel: XCD groups that split the N range (1, 2 or 4; 0 = 1)
  // EPI_PATCH_F32 with T = float: gR > 0 makes A the FRAMES f32 [n, 3, gR, gR]; row m = (image, patch) and column
  // k = (channel, py, px) of the im2col matrix are gathered by the LDS-DMA loader itself (no im2col pass in HBM).
  int gR, gP;
  int hp;             // pipelined kernel with a tail of lower tiles (gemm_kernel.h, HT > 0): 256-row panels of the head
  int order;          // fp32 pipelined kernel: how the head tiles are dealt - 0 / 2 = round robin, 1 = XCD panel ranges
  // EPI_RANKS_I32: target column of row m = targets ? targets[m] : m + tgt_off (clamped to [0, N)); every workgroup walks `ctw`
  // column tiles of its row block (grid = row blocks x ceil(column tiles / ctw))
  const int32_t* targets;
  int tgt_off, ctw;
  // gemm_split2 only: wscale = device pointer to {s, 1 / s} of the weight tensor's x2 image (written by launch_split2_weight);
  // sat_flag = device int the EPI_GELU_X2 epilogue ORs 1 into when a value exceeds fp16's range (may be null)
  const float* wscale;
  int* sat_flag;
};

// tile: 0 = auto, 1 = 128x128 (4 waves), 2 = 256x256 (8 waves), 3 = persistent pipelined, 4..7 = pipelined with a forced row cut, 8 = 64x64 on a four-stage ring (gemm.hip)
int launch_gemm(int precision, int epilogue, const GemmArgs& a, int tile, hipStream_t stream);
// split-fp32 GEMM over three-plane operands (gemm_split3.h).  a.K counts fp32 columns; a.lda / a.ldw count bf16 positions of
// the x3 rows (>= 4 K); a.ldc counts floats (EPI_BIAS_F32) or bf16 positions of the x3 output rows (EPI_GELU_X3, >= 4 N)
int launch_gemm_split3(int epilogue, const GemmArgs& a, hipStream_t stream);
bool gemm_split3_ok(const GemmArgs& a);
// x3 image [rows, 4 K bf16] of fp32 rows [rows, K]
int launch_split3_rows(const float* in, long ld_in, void* out, long ld_out, long rows, int K, hipStream_t stream);
// split-fp32 GEMM over two-plane fp16 operands (gemm_split2.h).  a.K counts fp32 columns; a.lda / a.ldw count fp16 positions of
// the x2 rows (>= 2 K); a.ldc counts floats (EPI_BIAS_F32, EPI_RESID3_F32) or fp16 positions of the x2 output rows
// (EPI_GELU_X2, >= 2 N); a.wscale is required
// force_cut: 0 = tile height by the rounds of the busiest XCD (gemm_split2.hip: x2_tile_height), 1 = 256-row, 2 = 128-row, 3 = 192-row tiles (tests)
int launch_gemm_split2(int epilogue, const GemmArgs& a, hipStream_t stream, int force_cut = 0);
void gemm_split2_plan(int M, int N, int cus, int* tile_rows, int* workgroups);   // (host arithmetic only: the launch of an [M, N] problem)
bool gemm_split2_ok(const GemmArgs& a);
// x2 image [rows, 2 K fp16] of fp32 activation rows [rows, K] (scaled residual plane; sat_flag as in GemmArgs)
int launch_split2_rows(const float* in, long ld_in, void* out, long ld_out, long rows, int K, int* sat_flag, hipStream_t stream);
// x2 image of a weight tensor [rows, K] with its per-tensor power-of-two scale: scale2 <- {s, 1 / s} (two device floats), then
// the planes of s * w (unscaled residual plane).  No host synchronisation.  sat_flag (may be null): raised when the tensor holds an
// infinite or NaN weight (it cannot be split; scale 1, the planes carry the infinities / NaNs)
int launch_split2_weight(const float* w, long ld_in, void* out, long ld_out, long rows, int K, float* scale2, int* sat_flag,
                         hipStream_t stream);
// x2 rows [n g^2, 2 * 3 p^2 fp16 positions] of the im2col matrix of frames f32 [n, 3, res, res] (the patch embedding's operand in the
// three-product mode; patch % 8 == 0); launch_gemm_split2(EPI_PATCH_F32) adds the positional embedding and skips the class rows
int launch_im2col_x2(const float* frames, void* out, long ld_out, int n, int res, int patch, int* sat_flag, hipStream_t stream);
// raises *sat_flag when LayerNorm outputs under (gamma, beta) could leave fp16's range: sqrt(D) max|gamma| + max|beta| > 65504
int launch_x2_ln_bound(const float* gamma, const float* beta, int D, int* sat_flag, hipStream_t stream);
// the row cut of the fp32 pipelined kernel for an [M, N] output over K columns on the current device: 256-row panels of the head (whole tile
// rounds) and the height of the tail tiles in 64-row units (0 = no tail)
void gemm_tail_plan(int M, int N, int K, int* head_panels, int* tail_units);
// which kernel `tile` = 0 resolves to: 1 / 2 = one-tile-per-workgroup 128x128 / 256x256, 3 = persistent pipelined
int gemm_resolved_tile(int precision, int epilogue, const GemmArgs& a, int tile);

// ------------------------------------------------------------------------------------------- attention
// qkv: T [n_seq * S, 3 * D] (q | k | v, heads of 64 inside each), out: T [n_seq * S, D]
int launch_attention(int precision, const void* qkv, void* out, int n_seq, int S, int heads, int causal,
                     hipStream_t stream);
// fp32 attention whose output is written as x3 rows [n_seq * S, 4 * heads * 64 bf16] (out_proj's operand in split-fp32 mode)
bool attention_x3_supported(int S, int causal);
int launch_attention_x3(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t stream);
// split-fp32 attention (attention_split.hip): fp32 qkv in, x3 rows out, both products as six bf16 products on the bf16
// matrix cores; non-causal, 193..208 tokens (the ViT's 197)
constexpr int ATTN_SPLIT = 4;  // `precision` argument of fc_attention
constexpr int ATTN_SPLIT_X2 = 5;  // ... the same kernel with x2 rows out
constexpr int ATTN_SPLIT2 = 6;    // attention_split2.hip: both products as THREE fp16 products per fp32 product, x2 rows out
bool attention_split_supported(int S, int causal);
// out_kind: KIND_X3 (x3 rows) or KIND_X2 (x2 rows [n_seq * S, 2 * heads * 64 fp16 positions]; sat_flag as in GemmArgs)
int launch_attention_split(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t stream, int out_kind = 3,
                           int* sat_flag = nullptr);

// split-fp32 attention on two fp16 planes (attention_split2.hip): fp32 qkv in, x2 rows out, three fp16 products per fp32 product
// with the cross terms in a second accumulator; non-causal, 193..208 tokens; sat_flag as in GemmArgs
bool attention_split2_supported(int S, int causal);
int launch_attention_split2(const void* qkv, void* out, int n_seq, int S, int heads, hipStream_t stream, int* sat_flag = nullptr);

// --------------------------------------------------------------------------------------------- row ops
// y[i] = LN(x[row(i)]) * gamma + beta.  row(i) = gather ? gather[i] : i; x row r at x + r * x_stride.
// out_kind: 0 = f32, 1 = bf16.
int launch_layernorm(const float* x, long x_stride, const int* gather, const float* gamma, const float* beta,
                     void* y, long y_stride, int out_kind, int rows, int D, hipStream_t stream);
// v = x[r] + delta[d] (delta of element kind `kind`, as y); x[r] = v if write_x; y[i] = LN(v).  r = gather ? gather[i] : i;
// d = delta_compact ? i : r.
int launch_layernorm_pair(float* x, const float* cls, const float* pos0, int tokens, const float* g0, const float* b0,
                          const float* g1, const float* b1, void* y, int out_kind, int rows, int D,
                          hipStream_t stream);
int launch_add_layernorm(float* x, long x_stride, const void* delta, long d_stride, const int* gather,
                         const float* gamma, const float* beta, void* y, long y_stride, int kind, int rows, int D,
                         int write_x, int delta_compact, hipStream_t stream, float* x_out = nullptr);
// dst[i] = src row (idx ? idx[i] : i * step), rows of row_bytes bytes (multiple of 16)
int launch_gather_rows(const void* src, const int* idx, long step, void* dst, int n, int row_bytes, hipStream_t stream);
int launch_im2col(const float* frames, void* patches, int out_kind, int n, int res, int patch, int Kp,
                  hipStream_t stream);
int launch_convert_rows(const float* in, void* out, int out_kind, long rows, int K, int Kp, hipStream_t stream);
// uint8 [n,H,W,3] -> f32 NCHW [n,3,R,R]: /255, bicubic resize (shorter side R), centre crop, mean/std (host arrays)
int launch_preprocess_u8(const unsigned char* frames, float* out, int n, int H, int W, int R, const float* mean3,
                         const float* std3, hipStream_t stream);
int launch_text_embed(const int64_t* ids, const float* tok, const float* pos, float* x, int* eot, int n, int L,
                      int D, int vocab, hipStream_t stream);
int launch_pool_normalize(const float* frame_emb, float* out, int n_clips, int frames, int dim, hipStream_t stream);
int launch_l2_normalize(const float* in, float* out, int n, int dim, hipStream_t stream);
int launch_group_mean(const float* in, float* out, int n_groups, int group, int dim, hipStream_t stream);
int launch_convert(const float* in, void* out, int out_kind, size_t n, hipStream_t stream);
int launch_transpose_convert(const float* in, void* out, int out_kind, int rows, int cols, hipStream_t stream);
int launch_wise(const float* a, const float* b, double w, float* out, size_t n, hipStream_t stream);

// ----------------------------------------------------------------------------------------------- score
// ranks[i] = rank of column t_i in the stable descending order of row i of alpha * T @ V^T, WITHOUT the [nt, nv] matrix in
// memory: the comparison runs in the epilogue of the scoring GEMM (gemm_kernel.h, EPI_RANKS_I32); same bits as
// launch_gemm(EPI_STORE_F32) + launch_ranks
int launch_similarity_ranks(const float* T, const float* V, int nt, int nv, int dim, float alpha, int target_offset,
                            const int32_t* targets, int32_t* ranks, hipStream_t stream);
// rank of column targets[i] (or i + target_offset when targets == nullptr) in the stable descending order of row i
int launch_ranks(const float* scores, int ld, int n_rows, int n_cols, int target_offset, const int32_t* targets,
                 int32_t* ranks, hipStream_t stream);
int launch_nce_loss(const float* scores, int n, float* out, float* ws, hipStream_t stream);
int launch_kd_loss(const float* scores, const float* teacher, int rows, int cols, float* out, float* ws,
                   hipStream_t stream);

// ------------------------------------------------------------------------------------------- training step
// attention backward (attention_bwd.hip); qkv / o are the forward's input / output, d_o the gradient of o
int launch_attention_backward(int precision, const void* qkv, const void* o, const void* d_o, void* dqkv, int n_seq,
                              int S, int heads, int causal, hipStream_t stream);
// C[N1, N2] = beta C + alpha sum_m A[m, N1] B[m, N2]  (wgrad.hip; fp32).  a_skip > 0: A row m lives at m + m / a_skip + 1;
// colsum: null, or [N1]: beta colsum + sum_m A[m, N1] from the same pass (the bias gradient next to the weight gradient)
size_t gemm_tn_scratch_bytes(int M, int N1, int N2);
int launch_gemm_tn(const float* A, const float* B, int M, int N1, int N2, int lda, int ldb, int a_skip, float alpha,
                   float beta, float* C, int ldc, float* scratch, size_t scratch_bytes, const float* zeros,
                   hipStream_t st, float* colsum = nullptr);
int launch_reduce_partials(const float* P, int splits, int rows, int cols, float* C, int ldc, float alpha, float beta,
                           hipStream_t st);
// backward.hip
size_t layernorm_bwd_scratch_bytes(int D);
int launch_layernorm_backward(const float* x, long x_stride, const int* gather, const void* dy, int dy_kind,
                              long dy_stride, int dy_compact, const float* gamma, float* out, long out_stride,
                              int accumulate, int rows, int D, float* dgamma, float* dbeta, float beta_acc,
                              float* scratch, size_t scratch_bytes, hipStream_t st);
int launch_quickgelu(const void* in, void* out, int kind, size_t n, hipStream_t st);
int launch_adamw(float* p, const float* g, float* m, float* v, size_t n, double lr, double b1, double b2, double eps,
                 double wd, int step, hipStream_t st);
int launch_pool_normalize_backward(const float* z, const float* dout, float* dz, int n_clips, int frames, int dim,
                                   hipStream_t st);
int launch_loss_backward(const float* scores, const float* teacher, int R, int C, float coef, float* dscores, float* ws,
                         hipStream_t st);
int launch_dot(const float* a, const float* b, size_t n, float alpha, float beta, float* out, hipStream_t st);
int launch_kd_teacher_scale_grad(const float* scores, const float* teacher, int R, int C, float* out, float* ws,
                                 hipStream_t st);
// sums[t, :] = sum_i g[i * S + t, :] into scratch (S * D floats), then out_pos = beta out_pos + sums and, if given,
// out_row0 = beta out_row0 + sums[0]
int launch_seq_sum(const float* g, int n_seq, int S, int D, float* out_pos, float* out_row0, float beta, float* scratch,
                   size_t scratch_bytes, hipStream_t st);
// token_embedding gradient as a fixed-order segmented sum (no float atomics): scratch holds 2 vocab + 2 rows ints
size_t token_grad_scratch_bytes(int rows, int vocab);
int launch_token_grad(const int64_t* ids, const float* g, float* dtok, int rows, int D, int vocab, int accumulate,
                      void* scratch, size_t scratch_bytes, hipStream_t st);
int launch_fill_cls(float* x, const float* cls, const float* pos0, int n_seq, int S, int D, hipStream_t st);

// exp(x) for x <= 0 (softmax terms) in fp32: 2^(x log2 e) on v_exp_f32 (1 ulp) with the product's rounding error and the low
// half of log2 e carried along: 2^(t + r) = 2^t (1 + r ln 2).  1.2e-7 max relative error against float64 over [-80, 0]
// (a plain exp2f(x * log2e) has 3.7e-6); arguments below -80 (incl. the -inf of masked keys) give exactly 0.
__device__ __forceinline__ float exp_neg_f32(float x) {
  constexpr float kHi = 1.4426950216293335f, kLo = 1.925963033500011e-08f;  // hi + lo = log2(e)
  const float xs = fmaxf(x, -80.f);
  const float t = xs * kHi;
  const float r = __builtin_fmaf(xs, kHi, -t) + xs * kLo;
  float e = __builtin_amdgcn_exp2f(t);
  e = __builtin_fmaf(e, r * 0.6931471805599453f, e);
  return x > -80.f ? e : 0.f;
}

// exp(x) for FINITE x <= 0 (no masked keys in the tile): exp_neg_f32 (common.h) without its clamp and select
__device__ __forceinline__ float exp_neg_finite_f32(float x) {
  constexpr float kHi = 1.4426950216293335f, kLo = 1.925963033500011e-08f;
  const float t = x * kHi;
  const float r = __builtin_fmaf(x, kLo, __builtin_fmaf(x, kHi, -t));
  const float e = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(e, r * 0.6931471805599453f, e);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// exp_neg_finite_f32 (common.h) on a pair: 2^(x log2 e) with the product's rounding error and the low half of log2 e
// carried along
__device__ __forceinline__ f32x2 exp_neg_finite_pair(const f32x2 x) {
  constexpr float kHi = 1.4426950216293335f, kLo = 1.925963033500011e-08f, kLn2 = 0.6931471805599453f;
  const f32x2 tt = x * kHi;
  f32x2 rr = __builtin_elementwise_fma(x, f32x2{kHi, kHi}, -tt);
  rr = __builtin_elementwise_fma(x, f32x2{kLo, kLo}, rr);
  const f32x2 ee = {__builtin_amdgcn_exp2f(tt[0]), __builtin_amdgcn_exp2f(tt[1])};
  return __builtin_elementwise_fma(ee, rr * kLn2, ee);
}

// max / sum over the four lanes {l, l ^ 16, l ^ 32, l ^ 48} on v_permlane32_swap / v_permlane16_swap (no LDS round trip)
__device__ __forceinline__ float max_over_lane_groups(float x) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum_over_lane_groups(float x) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}


// QuickGELU in fp32 (slip.py:359-361): x * sigmoid(1.702 x) = x / (1 + 2^t), t = -(1.702 log2 e) x.
// v_exp_f32 / v_rcp_f32 are 1-ulp instructions; what would cost accuracy is the rounding of the product t (|t| up to ~40),
// so t carries its exact fma residual and the low half of the constant: 2^(t + r) = 2^t (1 + r ln 2).  1.9e-7 max relative
// error against float64 (the plain float32 expression x / (1 + expf(-1.702f * x)) has 2.7e-6), in 8 VALU instructions
// instead of the 27 of expf + IEEE division.  The ONE definition used by the GEMM epilogues and the training kernels.
__device__ __forceinline__ float sigmoid_1702(float x) {
  constexpr float kHi = -2.4554669857025146f, kLo = 2.6109498563187117e-08f;  // hi + lo = -1.702 * log2(e)
  const float p = x * kHi;
  const float t = fminf(p, 126.f);                                           // 2^126: still finite, sigmoid ~ 1e-38
  const float r = __builtin_fmaf(x, kLo, __builtin_fmaf(x, kHi, -p));        // (every fma spelled out: the pair form below
  float e = __builtin_amdgcn_exp2f(t);                                       //  must perform the SAME operations)
  e = __builtin_fmaf(e, r * 0.6931471805599453f, e);
  return __builtin_amdgcn_rcpf(1.f + e);
}
__device__ __forceinline__ float quick_gelu_f32(float x) { return x * sigmoid_1702(x); }
// The same function on a pair / on four values with the packed-fp32 instructions (v_pk_mul_f32, v_pk_fma_f32, v_pk_add_f32: two
// elements per issue; only min, exp2 and rcp stay per element): 6.5 instead of 9.5 VALU instructions per element in the
// epilogue of the c_fc GEMM, the same IEEE operations in the same order - bit-identical to quick_gelu_f32.
__device__ __forceinline__ f32x2 quick_gelu_pair(const f32x2 x) {
  constexpr float kHi = -2.4554669857025146f, kLo = 2.6109498563187117e-08f;
  const f32x2 p = x * kHi;
  const f32x2 t = {fminf(p[0], 126.f), fminf(p[1], 126.f)};
  f32x2 r = __builtin_elementwise_fma(x, f32x2{kHi, kHi}, -p);
  r = __builtin_elementwise_fma(x, f32x2{kLo, kLo}, r);
  f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
  e = __builtin_elementwise_fma(e, r * 0.6931471805599453f, e);
  const f32x2 d = e + 1.f;
  const f32x2 s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  return x * s;
}
__device__ __forceinline__ f32x4 quick_gelu_f32x4(const f32x4 x) {
  const f32x2 lo = quick_gelu_pair(f32x2{x[0], x[1]}), hi = quick_gelu_pair(f32x2{x[2], x[3]});
  return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
// The same function with every step spelled on all FOUR values (the two packed halves of a step are adjacent in the source): the
// two chains of quick_gelu_f32x4 above come out of hipcc one after the other, each packed instruction waiting for its predecessor
// (an `s_nop` after almost every one in a VALU-bound epilogue); interleaved they cover each other.  The same IEEE operations in the
// same order per element: bit-identical to quick_gelu_f32.
__device__ __forceinline__ f32x4 quick_gelu_f32x4_wide(const f32x4 x) {
  constexpr float kHi = -2.4554669857025146f, kLo = 2.6109498563187117e-08f;
  const f32x4 hi4 = {kHi, kHi, kHi, kHi}, lo4 = {kLo, kLo, kLo, kLo};
  const f32x4 p = x * kHi;
  const f32x4 t = {fminf(p[0], 126.f), fminf(p[1], 126.f), fminf(p[2], 126.f), fminf(p[3], 126.f)};
  f32x4 r = __builtin_elementwise_fma(x, hi4, -p);
  r = __builtin_elementwise_fma(x, lo4, r);
  f32x4 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1]), __builtin_amdgcn_exp2f(t[2]), __builtin_amdgcn_exp2f(t[3])};
  e = __builtin_elementwise_fma(e, r * 0.6931471805599453f, e);
  const f32x4 d = e + 1.f;
  const f32x4 s = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1]), __builtin_amdgcn_rcpf(d[2]), __builtin_amdgcn_rcpf(d[3])};
  return x * s;
}
// d/dx [x * sigmoid(1.702 x)] = s * (1 + 1.702 x (1 - s))
__device__ __forceinline__ float quick_gelu_grad_f32(float x) {
  const float s = sigmoid_1702(x);
  return s * (1.f + 1.702f * x * (1.f - s));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace fc

/* fitclip_hip.h - C ABI of libfitclip_hip.so: the MI355X (gfx950) encode-and-score path of FitCLIP.
 *
 * The reference (bryant1410/fitclip) has no FFI: its hot path is Python calling `clip.model.CLIP.encode_image /
 * encode_text` (third-party openai/CLIP) from `aligner/encoder/clip_video_text_encoder.py`.  This header is the C
 * boundary a binding for that path would use; each entry point names the reference code it replaces (paths relative
 * to the reference repository root).  INTEGRATION.md shows the ctypes / cffi stub for the reference side.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer marked "dev" is DEVICE memory owned by the caller (PyTorch-ROCm
 *     tensors: pass `tensor.data_ptr()`); the library allocates no device memory and never synchronises.
 *   - every launch goes to the `hipStream_t` passed in (`torch.cuda.current_stream().cuda_stream`); all functions
 *     may be captured into a hipGraph (tests/test_gpu_graph.py: the whole encode-and-score call, replayed bitwise).
 *   - return 0 on success, a negative fc_status otherwise; `fc_last_error()` returns a thread-local message.
 *   - one handle per device / model; handles are independent (no global state besides the thread-local error string and
 *     per-device caches of two device attributes: the compute-unit count and which kernels had their dynamic-LDS limit
 *     raised); the library reads no environment variable.
 *   - layouts: row-major, fp32 unless stated; video frames NCHW fp32 (the layout the reference's eval transform
 *     produces, clip_video_text_encoder.py:125-133); token ids int64 [n, context_length] (clip.tokenize output).
 */
#ifndef FITCLIP_HIP_H
#define FITCLIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(FITCLIP_BUILD)
#define FC_API __attribute__((visibility("default")))
#else
#define FC_API
#endif

typedef struct ihipStream_t* fc_stream; /* == hipStream_t */

typedef enum {
  FC_OK = 0,
  FC_EINVAL = -1,  /* bad argument (shape, alignment, unknown name) */
  FC_ELAUNCH = -2, /* HIP launch / runtime error */
  FC_ENOMEM = -3,  /* workspace or arena too small */
  FC_ESTATE = -4,  /* call order (weights missing / not packed) */
  FC_ERANGE = -5   /* split_gemm = 2: an activation (or a weight) beyond fp16's range, or not finite, was met - results not valid */
} fc_status;

typedef enum {
  FC_PREC_F32 = 0, /* exact fp32: v_mfma_f32_16x16x4_f32 GEMMs, fp32 activations (parity path)                */
  FC_PREC_BF16 = 1 /* bf16 MFMA operands, fp32 accumulate, fp32 residual stream / LayerNorm / softmax          */
} fc_precision;

/* ABI revision of this header: bumped whenever a struct or a signature changes; the tail of fc_version() names it. */
#define FC_ABI_VERSION 5

/* Architecture: the keys of config/encoder/clip_from_scratch_vit_b_16.yaml:5-16 (heads = width / 64 as in
 * clip.model.CLIP).  Head dimension must be 64.
 * `struct_size` is the FIRST field and must hold sizeof(fc_config) AS THE CALLER DECLARED IT: fc_create reads it before
 * anything else and rejects a binding written against another revision of this struct (FC_EINVAL, with both sizes in
 * fc_last_error()) instead of reading past the caller's memory.  Use FC_CONFIG_INIT in C. */
typedef struct {
  int32_t struct_size;        /* sizeof(fc_config) */
  int32_t embed_dim;          /* 512  */
  int32_t image_resolution;   /* 224  */
  int32_t vision_layers;      /* 12   */
  int32_t vision_width;       /* 768  */
  int32_t vision_patch_size;  /* 16   */
  int32_t context_length;     /* 77   */
  int32_t vocab_size;         /* 49408 */
  int32_t transformer_width;  /* 512  */
  int32_t transformer_heads;  /* 8    */
  int32_t transformer_layers; /* 12   */
  int32_t precision;          /* fc_precision */
  int32_t chunk_frames;       /* frames per pass of the visual tower; 0 = default */
  int32_t chunk_texts;        /* texts per pass of the text tower;    0 = default */
  int32_t gemm_tile;          /* 0 = auto, 1 = 128x128, 2 = 256x256 (tuning / tests) */
  int32_t prune_last_block;   /* 1: after the attention of the LAST block only the pooled rows (CLS / EOT) go through
                                 out_proj, LayerNorm 2 and the MLP - nothing else is read afterwards; identical
                                 embeddings, 6 % fewer FLOPs.  0 (default): every row, as the reference computes it */
  int32_t split_gemm;         /* fp32 precision only.  1: the four block GEMMs of the VISUAL tower run on the bf16 matrix
                                 cores over split-fp32 operands - every fp32 value as three bf16 numbers ("x3" rows,
                                 fc_split3), every product as six bf16 products accumulated in fp32 (fc_gemm_split3) -
                                 at fp32 accuracy and ~1.6x the fp32-MFMA rate; LayerNorm, attention, residual stream,
                                 patch embedding, the text tower and passes too small for that GEMM stay on the plain
                                 fp32 path.  2: the same GEMMs on the fp16 matrix cores over TWO-plane operands ("x2" rows,
                                 fc_split2: x = h1 + 2^-11 h2 in fp16) with THREE products per fp32 product
                                 (fc_gemm_split2) - fp32 accuracy at ~2.7x the fp32-MFMA rate; so does the patch embedding
                                 (patch sizes that are multiples of 8) and, in an fc_encode_text CALL of 2048 token rows or
                                 more (27 captions of 77 tokens; text widths that are multiples of 256), the four block
                                 GEMMs of the text tower - a smaller call is one tile's latency with that kernel and
                                 keeps the fp32 kernels: one arithmetic per call, rows independent of the rest of the call
                                 within either kind.  fp16 planes hold
                                 |x| <= 65504: a value beyond that (or an infinite / NaN value or weight) raises a device-side
                                 flag and the NEXT fc_encode_image / large fc_encode_text (and fc_range_status) returns
                                 FC_ERANGE.  A caller MUST ask fc_range_status(wait = 1) after its last batch before it uses
                                 or saves the embeddings - or switch fc_range_strict on, and every such call answers for itself (one host
                                 synchronisation per call) - never silently wrong.
                                 0 (default): fp32-input MFMA everywhere */
} fc_config;
#define FC_CONFIG_INIT {(int32_t)sizeof(fc_config)}

typedef struct fc_handle fc_handle;

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
FC_API int fc_create(const fc_config* cfg, fc_handle** out);
FC_API void fc_destroy(fc_handle* h);
FC_API const char* fc_last_error(void);
FC_API const char* fc_version(void);

/* ---- weights: replaces `clip.load(...)` + `load_state_dict` (clip_video_text_encoder.py:22-61) ----------------
 * `name` is the OpenAI-CLIP state-dict key ("visual.conv1.weight", "transformer.resblocks.3.attn.in_proj_weight",
 * "token_embedding.weight", ...; `logit_scale`, `input_resolution`, `context_length`, `vocab_size` are accepted and
 * ignored).  `dev_f32` is BORROWED: it must stay valid (and may be updated in place, e.g. by WiSE) until the handle
 * is destroyed; call fc_pack_weights again after changing values. */
FC_API int fc_set_weight(fc_handle* h, const char* name, const float* dev_f32, const int64_t* shape, int32_t ndim);
FC_API int fc_num_weights(const fc_handle* h);                       /* how many names fc_set_weight expects */
FC_API const char* fc_weight_name(const fc_handle* h, int32_t index); /* the index-th expected name */
/* Kernel-layout copies of the GEMM weights (bf16 conversion, [K,N] -> [N,K] transposes of the two projections). */
FC_API size_t fc_packed_bytes(const fc_handle* h);
FC_API int fc_pack_weights(fc_handle* h, void* dev_arena, size_t arena_bytes, fc_stream stream);

/* ---- encoders -------------------------------------------------------------------------------------------------
 * fc_encode_image: `CLIP.encode_image(images)` as called at clip_video_text_encoder.py:84.
 *   frames dev f32 [n_frames, 3, R, R] -> out dev f32 [n_frames, embed_dim] (NOT normalised).
 * fc_encode_text: `CLIP.encode_text(ids)` as called at clip_video_text_encoder.py:93 (in-tree twin
 *   aligner/encoder/slip.py:468-480): ids dev int64 [n_texts, context_length] -> out dev f32 [n_texts, embed_dim].
 * `workspace` is caller-owned scratch of at least fc_workspace_bytes(h, tower, n) bytes (tower 0 = visual, 1 = text);
 * a smaller workspace is accepted as long as one chunk of at least one item fits (more passes). */
FC_API size_t fc_workspace_bytes(const fc_handle* h, int32_t tower, int32_t n);
FC_API int fc_encode_image(fc_handle* h, const float* frames, int32_t n_frames, float* out, void* workspace,
                    size_t workspace_bytes, fc_stream stream);
FC_API int fc_encode_text(fc_handle* h, const int64_t* ids, int32_t n_texts, float* out, void* workspace,
                   size_t workspace_bytes, fc_stream stream);
/* split_gemm = 2 only (FC_OK otherwise): FC_ERANGE when a writer of fp16 planes met a value beyond 65504 since
 * fc_pack_weights (activations of the towers' blocks; or LayerNorm weights whose outputs could exceed it).  wait != 0:
 * first waits for the work queued on `stream` (one host synchronisation - call it after the last batch of an evaluation);
 * wait == 0: what the flag copies of the calls completed so far have shown (fc_encode_image itself checks this on entry). */
FC_API int fc_range_status(fc_handle* h, fc_stream stream, int32_t wait);
/* on != 0: every fc_encode_image (and every fc_encode_text that uses fp16 planes) of a split_gemm = 2 handle waits for its own
 * range flag before it returns (one host synchronisation per call; not capturable into a hipGraph) and returns FC_ERANGE ITSELF
 * when its values left fp16's range.  Off by default: the flag is then reported by the NEXT such call and by fc_range_status. */
FC_API int fc_range_strict(fc_handle* h, int32_t on);

/* Eval transform on the device (clip_video_text_encoder.py:125-133): frames dev uint8 [n, H, W, 3]
 * -> out dev f32 [n, 3, R, R] = normalise(center_crop(bicubic_resize(frames / 255, shorter side R), R)).  mean3 / std3
 * are HOST arrays of 3 floats. */
FC_API int fc_preprocess_u8(const uint8_t* frames, float* out, int32_t n, int32_t H, int32_t W, int32_t R,
                     const float* mean3, const float* std3, fc_stream stream);

/* clip_video_text_encoder.py:85-89: out[b] = mean_f(e[b,f] / ||e[b,f]||), NOT re-normalised.  e [n_clips*frames, dim] */
FC_API int fc_pool_normalize(const float* frame_emb, float* out, int32_t n_clips, int32_t frames, int32_t dim,
                      fc_stream stream);
/* clip_video_text_encoder.py:94: out[i] = in[i] / ||in[i]|| */
FC_API int fc_l2_normalize(const float* in, float* out, int32_t n, int32_t dim, fc_stream stream);

/* ---- scoring ---------------------------------------------------------------------------------------------------
 * fc_similarity: out[na, nb] = alpha * A[na, dim] . B[nb, dim]^T in exact fp32 (text_video_retrieval.py:50,74;
 *   video_text_module.py:63).  dim % 32 == 0, nb % 4 == 0, ldo >= nb.
 * fc_ranks: ranks[i] = position of column (i + target_offset) in the stable descending order of row i
 *   (aligner/metrics.py:16-20).  scores [n_rows, ld].
 * fc_nce_loss / fc_kd_loss: aligner/loss.py:13-26 / 29-39 (KD with reduction="batchmean", teacher_student.py:72-73)
 *   on square [n, n] matrices; `ws` dev f32 scratch of 2 * n floats; result in out[0]. */
FC_API int fc_similarity(const float* A, const float* B, int32_t na, int32_t nb, int32_t dim, float alpha, float* out,
                  int32_t ldo, fc_stream stream);
FC_API int fc_ranks(const float* scores, int32_t ld, int32_t n_rows, int32_t n_cols, int32_t target_offset, int32_t* ranks,
             fc_stream stream);
/* Scoring WITHOUT the score matrix ("compute ranks tile-wise"; aligner/text_video_retrieval.py:70-80 +
 * aligner/metrics.py:16-20): ranks[i] = position of column t_i (= targets ? targets[i], clamped to a valid column : i + target_offset, which must be one) in the
 * stable descending order of row i of alpha * T @ V^T, T [nt, dim], V [nv, dim], dim % 32 == 0.  The comparison runs in the
 * epilogue of the exact-fp32 scoring GEMM; every score has the bits fc_similarity would store, so the ranks equal
 * fc_similarity + fc_ranks / fc_ranks_of exactly (ties included) while nothing of size nt x nv touches memory
 * (8192 x 8192: 268 MB).  ranks: int32 [nt], overwritten. */
FC_API int fc_similarity_ranks(const float* T, const float* V, int32_t nt, int32_t nv, int32_t dim, float alpha,
                        int32_t target_offset, const int32_t* targets, int32_t* ranks, fc_stream stream);
/* Zero-shot classification (aligner/video_text_classification.py): fc_ranks_of = rank of the label
 * column targets[i] in row i (Accuracy@k = rank < k, MedianRank); fc_group_mean = mean over the `group` template
 * prompts of each label (:88-90), out [n_groups, dim]. */
FC_API int fc_ranks_of(const float* scores, int32_t ld, int32_t n_rows, int32_t n_cols, const int32_t* targets,
                int32_t* ranks, fc_stream stream);
FC_API int fc_group_mean(const float* in, float* out, int32_t n_groups, int32_t group, int32_t dim, fc_stream stream);
FC_API int fc_nce_loss(const float* scores, int32_t n, float* out, float* ws, fc_stream stream);
FC_API int fc_kd_loss(const float* scores, const float* teacher_scores, int32_t n, float* out, float* ws, fc_stream stream);
/* the same on [rows, cols] matrices (videos x prompts, teacher_student.py:111-138): each direction's "batchmean" divides
 * by its own number of lines; ws: rows + cols floats */
FC_API int fc_kd_loss_rect(const float* scores, const float* teacher_scores, int32_t rows, int32_t cols, float* out,
                    float* ws, fc_stream stream);

/* ---- WiSE (aligner/wise.py:16): out = (1 - weight_for_2) * a + weight_for_2 * b over n floats ------------------ */
FC_API int fc_wise(const float* a, const float* b, double weight_for_2, float* out, size_t n, fc_stream stream);

/* ---- single operators (unit parity tests, tuning) --------------------------------------------------------------
 * element kind: 0 = f32, 1 = bf16 (as fc_precision).  See csrc/common.h for the epilogue ids. */
FC_API int fc_gemm(int32_t precision, int32_t epilogue, const void* A, const void* W, const float* bias, void* C,
            const float* aux, float alpha, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc,
            int32_t P, int32_t tile, fc_stream stream);
/* How the fp32 persistent GEMM cuts the rows of an [M, N] output over K columns on the current device: `head_panels` 256-row panels whose
 * tiles fill whole rounds over the compute units, then a tail of tiles of `tail_units` x 64 rows (0: no tail) in the same
 * launch.  fc_gemm's `tile`: 0 auto, 1 / 2 one 128 x 128 / 256 x 256 tile per workgroup, 3 persistent with this cut,
 * 4 persistent with whole 256-row tiles only, 5..7 persistent with a tail of 1..3 units forced (tests), 8 one 64 x 64 tile
 * per workgroup on a four-stage LDS ring (what small fp32 problems resolve to).  Results do not depend on the cut or the
 * kernel. */
FC_API int fc_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t* head_panels, int32_t* tail_units);
FC_API int fc_layernorm(const float* x, int64_t x_stride, const int32_t* gather, const float* gamma, const float* beta,
                 void* y, int64_t y_stride, int32_t out_kind, int32_t rows, int32_t D, fc_stream stream);
/* v = x[r] + delta[r]; if write_x: x[r] = v; y[i] = LayerNorm(v) * gamma + beta, r = gather ? gather[i] : i.  The
 * residual update of a pre-LN block (slip.py:382-385) folded into the LayerNorm that follows it.  delta and y have
 * element kind `kind`; kind 3: delta fp32, y x3 rows (fc_split3 layout, y_stride >= 4 D bf16 positions, 128-byte aligned;
 * fc_layernorm too); kind 4: delta fp32, y x2 rows (fc_split2 layout, y_stride >= 2 D fp16 positions, 128-byte aligned;
 * fc_layernorm too). */
FC_API int fc_add_layernorm(float* x, int64_t x_stride, const void* delta, int64_t d_stride, const int32_t* gather,
                     const float* gamma, const float* beta, void* y, int64_t y_stride, int32_t kind, int32_t rows,
                     int32_t D, int32_t write_x, fc_stream stream);
/* Multi-head attention over packed rows: qkv [n_seq * S, 3 * heads * 64] (q | k | v, head dim 64) -> out
 * [n_seq * S, heads * 64], softmax(q k^T / 8 [+ causal mask]) v per (sequence, head), as nn.MultiheadAttention inside
 * slip.py:366-380.  Any S in fp32; bf16: causal up to 224 tokens, non-causal any S (K/V streamed beyond 224).
 * precision 3: fp32 qkv in, x3 rows out (fc_split3 layout; non-causal, 113..224 tokens), the fp32 kernel's values.
 * precision 4: the same operands and layout, both products formed as six bf16 products per fp32 product on the bf16 matrix
 * cores (fc_gemm_split3's arithmetic; fp32 accuracy, not the fp32 kernel's bits; non-causal, 193..208 tokens): the attention
 * of the split_gemm = 1 mode.  precision 5: as 4, x2 rows out (fc_split2 layout).  precision 6: fp32 qkv in, x2 rows out, both
 * products as THREE fp16 products per fp32 product (two fp16 planes per operand, the cross terms in a second accumulator;
 * fp32 accuracy; non-causal, 193..208 tokens): the attention of split_gemm = 2. */
FC_API int fc_attention(int32_t precision, const void* qkv, void* out, int32_t n_seq, int32_t S, int32_t heads,
                 int32_t causal, fc_stream stream);
FC_API int fc_convert(const float* in, void* out, int32_t out_kind, size_t n, fc_stream stream);
/* Split-fp32 operands ("x3" rows): fp32 rows [rows, K] (row stride ld_in floats) -> three bf16 planes per value, x = p1 + p2
 * + p3 exactly.  Every 16 columns become one 128-byte line [p1 x16 | p2 x16 | p3 x16 | 32 unused bytes]; a row is K / 16 lines
 * = 4 K bf16 positions (ld_out counts bf16 positions, a multiple of 64; `out` 128-byte aligned).  K % 16 == 0. */
FC_API int fc_split3(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int32_t K, fc_stream stream);
/* C = epilogue(A . W^T) over x3 operands A3 [M, K], W3 [N, K] (lda / ldw in bf16 positions, >= 4 K): the six bf16 products
 * p1q1 + p1q2 + p2q1 + p2q2 + p1q3 + p3q1 of every fp32 product, formed from registers on v_mfma_f32_32x32x16_bf16 and
 * accumulated in fp32 - the reference's fp32 `F.linear` (slip.py:366-390) to 2^-26 per product, at the bf16 matrix cores'
 * rate.  epilogue 6: C fp32 [M, N] = acc + bias (ldc floats); epilogue 7: C x3 rows [M, 4 N] = planes(QuickGELU(acc + bias))
 * (ldc bf16 positions), the next GEMM's A operand; epilogue 8: C fp32 [M, N] += acc + bias, in place (the residual update
 * x = x + proj(..) of slip.py:382-385 in the projection's epilogue).  K % 32 == 0, K >= 64, N % 32 == 0; operands below 4 GiB. */
FC_API int fc_gemm_split3(int32_t epilogue, const void* A3, const void* W3, const float* bias, void* C, int32_t M, int32_t N,
                   int32_t K, int32_t lda, int32_t ldw, int32_t ldc, fc_stream stream);

/* Two-plane fp16 operands ("x2" rows, split_gemm = 2): fp32 rows [rows, K] -> h1 = fp16(x), h2 = fp16((x - h1) 2^11), i.e.
 * x = h1 + 2^-11 h2 to 2^-23 |x| for |x| in [2^-14, 65504] (2^-36 absolute below).  Every 32 columns become one 128-byte line
 * [h1 x32 | h2 x32]; a row is 2 K fp16 positions = the bytes of the fp32 row (ld_out counts fp16 positions, a multiple of 64;
 * `out` 128-byte aligned).  K % 32 == 0.  sat_flag (may be null): device int, 1 is ORed in when |x| > 65504, an infinity or a NaN
 * was met. */
FC_API int fc_split2(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int32_t K, int32_t* sat_flag,
              fc_stream stream);
/* ... of a WEIGHT tensor [rows, K]: scale2 (two device floats) <- {s, 1 / s} with s the power of two that puts max |s w| into
 * [2^14, 2^15), then g1 = fp16(s w), g2 = fp16(s w - g1) (unscaled residual) in the same line layout.  No host synchronisation.
 * sat_flag (may be null): device int, 1 is ORed in when the tensor holds an infinite or NaN weight (it cannot be split: scale 1,
 * the planes carry the infinities / NaNs into every product). */
FC_API int fc_split2_weight(const float* w, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int32_t K, float* scale2,
                     int32_t* sat_flag, fc_stream stream);
/* C = epilogue((A . W^T) / s) over x2 operands A2 [M, K] (fc_split2, LayerNorm kind 4, attention precision 5, epilogue 10) and
 * W2 [N, K] (fc_split2_weight, with its scale2): the three fp16 products h1 g1 + h1 g2 + h2 (2^-11 g1) of every fp32 product
 * on v_mfma_f32_16x16x32_f16, accumulated in fp32 - the reference's fp32 `F.linear` (slip.py:366-390) to 2^-22 per product.
 * epilogue 6: C fp32 [M, N] = acc + bias; 8: C fp32 += acc + bias in place; 10: C x2 rows [M, 2 N fp16] =
 * planes(QuickGELU(acc + bias)) (sat_flag as in fc_split2).  K % 64 == 0, K >= 128, N % 32 == 0; operands below 4 GiB.
 * `cut`: the tile height of the persistent kernel - 0: 256, 192 or 128 rows, whichever costs the busiest XCD the fewest rounds x rows
 * (small and mid-size batches leave the big tile), 1: 256 rows, 2: 128 rows, 3: 192 rows (tests).  Results do not depend on it. */
FC_API int fc_gemm_split2(int32_t epilogue, const void* A2, const void* W2, const float* scale2, const float* bias, void* C,
                   int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc, int32_t* sat_flag, int32_t cut,
                   fc_stream stream);
/* What fc_gemm_split2 (cut 0) does with an [M, N] problem on `compute_units` CUs (<= 0: the current device): the tile height it
 * picks (256, 192 or 128 rows) and the workgroups it launches - 8 x the tiles of the busiest XCD under the kernel's schedule, at most
 * one per CU.  Host arithmetic only: callable without a GPU when compute_units is given. */
FC_API int fc_gemm_split2_plan(int32_t M, int32_t N, int32_t compute_units, int32_t* tile_rows, int32_t* workgroups);

/* ---- training: the KD fine-tuning step of the student -------------------------------------------
 * Replaces autograd + torch.optim.AdamW for `TeacherStudentLightningModule.training_step / training_step_end /
 * optimizer_step` (aligner/teacher_student.py:99-183, aligner/video_text_module.py:94-97, aligner/cli.py:129).
 * Only precision fp32 handles train (the reference trains in float32).  Call order per optimiser step:
 *   fc_train_prepare (transposed weight copies for the dgrad GEMMs; after fc_pack_weights, again after every update)
 *   fc_encode_image_train / fc_encode_text_train   forward of a tower, every activation kept in `arena`
 *   ... loss and its gradient w.r.t. the tower outputs (fc_similarity, fc_*_loss_backward, fc_gemm_tn,
 *       fc_pool_normalize_backward) ...
 *   fc_encode_image_backward / fc_encode_text_backward   parameter gradients into the buffers given by fc_set_grad
 *   fc_adamw over the (flat) parameter / gradient / moment buffers.
 * `arena` (>= fc_train_arena_bytes) and `scratch` (>= fc_train_scratch_bytes) are caller-owned, 256-byte aligned. */
FC_API int fc_set_grad(fc_handle* h, const char* name, float* dev_f32);        /* borrowed, same shape as the weight */
FC_API size_t fc_train_weights_bytes(const fc_handle* h);
FC_API int fc_train_prepare(fc_handle* h, void* dev_arena, size_t arena_bytes, fc_stream stream);
FC_API size_t fc_train_arena_bytes(const fc_handle* h, int32_t tower, int32_t n);
FC_API size_t fc_train_scratch_bytes(const fc_handle* h, int32_t tower, int32_t n);
/* out = CLIP.encode_image(frames) / CLIP.encode_text(ids) (NOT normalised), bit-identical to the inference entry points */
FC_API int fc_encode_image_train(fc_handle* h, const float* frames, int32_t n, float* out, void* arena, size_t arena_bytes,
                          fc_stream stream);
FC_API int fc_encode_text_train(fc_handle* h, const int64_t* ids, int32_t n, float* out, void* arena, size_t arena_bytes,
                         fc_stream stream);
/* d_out dev f32 [n, embed_dim] = dLoss / d(out of the *_train call that filled `arena`).  accumulate = 0: gradients
 * are overwritten; 1: added (several micro-batches per optimiser step). */
FC_API int fc_encode_image_backward(fc_handle* h, const float* d_out, int32_t n, void* arena, size_t arena_bytes,
                             void* scratch, size_t scratch_bytes, int32_t accumulate, fc_stream stream);
FC_API int fc_encode_text_backward(fc_handle* h, const int64_t* ids, const float* d_out, int32_t n, void* arena,
                            size_t arena_bytes, void* scratch, size_t scratch_bytes, int32_t accumulate,
                            fc_stream stream);
/* backward of fc_pool_normalize (frames = 1: of fc_l2_normalize): z [n_clips * frames, dim] the un-normalised tower
 * output, d_out [n_clips, dim] -> d_z [n_clips * frames, dim] */
FC_API int fc_pool_normalize_backward(const float* z, const float* d_out, float* d_z, int32_t n_clips, int32_t frames,
                               int32_t dim, fc_stream stream);
/* d_scores = coef * dLoss/dscores of fc_nce_loss ([n, n]) / fc_kd_loss(_rect) ([rows, cols]) (aligner/loss.py:13-39,
 * "mean" / "batchmean"); ws: 2 n (NCE) or 2 (rows + cols) (KD) floats */
FC_API int fc_nce_loss_backward(const float* scores, int32_t n, float coef, float* d_scores, float* ws, fc_stream stream);
FC_API int fc_kd_loss_backward(const float* scores, const float* teacher_scores, int32_t rows, int32_t cols, float coef,
                        float* d_scores, float* ws, fc_stream stream);
/* out[0] = sum_ij dKD/dteacher_scores[i,j] * teacher_scores[i,j]: what the teacher-student temperature (a factor of
 * every teacher score, teacher_student.py:68-69,157) receives through the teacher scores.  ws: 3 (rows + cols) floats */
FC_API int fc_kd_teacher_scale_grad(const float* scores, const float* teacher_scores, int32_t rows, int32_t cols,
                             float* out, float* ws, fc_stream stream);
/* C[N1, N2] = beta C + alpha sum_m A[m, N1] B[m, N2], exact fp32 (weight gradients dW = dY^T X; and dV = dS T,
 * dT = dS^T V of the similarity, video_text_module.py:63).  scratch >= fc_gemm_tn_scratch_bytes, 256-byte aligned. */
FC_API size_t fc_gemm_tn_scratch_bytes(int32_t M, int32_t N1, int32_t N2);
FC_API int fc_gemm_tn(const float* A, const float* B, int32_t M, int32_t N1, int32_t N2, int32_t lda, int32_t ldb,
               float alpha, float beta, float* C, int32_t ldc, void* scratch, size_t scratch_bytes, fc_stream stream);
FC_API int fc_attention_backward(int32_t precision, const void* qkv, const void* out, const void* d_out, void* d_qkv,
                          int32_t n_seq, int32_t S, int32_t heads, int32_t causal, fc_stream stream);
FC_API size_t fc_layernorm_backward_scratch_bytes(int32_t D);
FC_API int fc_layernorm_backward(const float* x, const float* d_y, const float* gamma, float* d_x, int32_t accumulate,
                          int32_t rows, int32_t D, float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes,
                          fc_stream stream);
/* Gradient of `self.token_embedding(text)` (aligner/encoder/slip.py:469; what autograd's embedding backward does in
 * teacher_student.py:99-140): d_table[id, :] (+)= the sum of d_rows[r, :] over the rows r with ids[r] == id, added IN ROW
 * ORDER - a fixed-order segmented sum, no float atomics, so a step is bit-reproducible (SOT / EOT / pad ids collide in
 * every caption).  ids dev int64 [rows] (clamped to the table like the forward gather), d_rows dev f32 [rows, D], d_table
 * dev f32 [vocab, D]; rows of d_table whose id does not occur are left untouched (zero the table first unless
 * accumulate = 1).  scratch: fc_token_embedding_backward_scratch_bytes(rows, vocab) bytes, 16-byte aligned. */
FC_API size_t fc_token_embedding_backward_scratch_bytes(int32_t rows, int32_t vocab);
FC_API int fc_token_embedding_backward(const int64_t* ids, const float* d_rows, float* d_table, int32_t rows, int32_t D,
                                int32_t vocab, int32_t accumulate, void* scratch, size_t scratch_bytes, fc_stream stream);
FC_API int fc_dot(const float* a, const float* b, size_t n, float alpha, float beta, float* out, fc_stream stream);
FC_API int fc_transpose(const float* in, float* out, int32_t rows, int32_t cols, fc_stream stream);
/* torch.optim.AdamW step `step` (counted from 1) over n floats (aligner/cli.py:129, config/trainer.yaml:21-23) */
FC_API int fc_adamw(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double lr, double beta1,
             double beta2, double eps, double weight_decay, int32_t step, fc_stream stream);

/* ---- CLIP BPE tokenizer (host C++) -----------------------------------------------------------------
 * `clip.tokenize(texts, truncate=True)` (clip_video_text_encoder.py:64-65) = `SimpleTokenizer` of
 * aligner/encoder/slip.py:75-164 over a LOCAL gzip merges file (`bpe_simple_vocab_16e6.txt.gz` layout: a header line,
 * then one "a b" merge per line, the first 48 894 kept).  Texts are UTF-8, already cleaned and lower-cased by the caller
 * (html.unescape x2, white-space collapse, str.lower: slip.py:60-72,138).  Not thread-safe per handle (merge cache). */
typedef struct fc_bpe fc_bpe;
FC_API int fc_bpe_create(const char* merges_gz_path, int32_t context_length, fc_bpe** out);
FC_API void fc_bpe_destroy(fc_bpe* t);
FC_API int32_t fc_bpe_vocab_size(const fc_bpe* t);          /* len(encoder): 49 408 for the published file */
FC_API int32_t fc_bpe_sot(const fc_bpe* t);                 /* id of <|startoftext|> (49 406) */
FC_API int32_t fc_bpe_eot(const fc_bpe* t);                 /* id of <|endoftext|>   (49 407) */
/* ids of one text WITHOUT framing (SimpleTokenizer.encode); returns the count (may exceed `capacity`: call again) */
FC_API int32_t fc_bpe_encode(fc_bpe* t, const char* text_utf8, int64_t* out_ids, int32_t capacity);
/* out_ids host int64 [n, context_length]: SOT, ids, EOT, zero padding; too long: cut with EOT in the last slot when
 * `truncate`, FC_EINVAL otherwise */
FC_API int fc_bpe_tokenize(fc_bpe* t, const char* const* texts_utf8, int32_t n, int32_t truncate, int64_t* out_ids);
/* UTF-8 bytes of the decoded text ("</w>" -> " "); returns the byte count (NUL-terminated if it fits `capacity`) */
FC_API int32_t fc_bpe_decode(const fc_bpe* t, const int64_t* ids, int32_t n, char* out, int32_t capacity);

/* ---- kernel timing (bench.py roofline leg): hipEvent pairs around the GEMM, attention and add+LayerNorm launches
 * of the transformer blocks, on the caller's stream ----------------------------------------------------------------- */
typedef struct {
  int32_t kind;      /* 0 = gemm (M,N,K); 1 = attention (M = sequences, N = heads, K = tokens); 2 = add+LayerNorm (M = rows, N = width) */
  int32_t precision; /* fc_precision */
  int32_t epilogue;  /* gemm: the epilogue id; attention: the arithmetic of the kernel that ran, as fc_attention's precision code (0 fp32
                        MFMA, 1 bf16, 3 fp32 with x3 rows out, 4 / 5 six bf16 products, 6 three fp16 products) */
  int32_t tile;      /* gemm: the tile / row cut chosen (the three-product GEMM: its tile height, 256 / 192 / 128 rows); attention: 1 when
                        a split pass over the fp32 output followed */
  int32_t M, N, K;
  float ms;          /* elapsed between the two events; valid after the stream has been synchronised */
} fc_prof_record;
FC_API int fc_profile_enable(fc_handle* h, int32_t max_records); /* 0 disables and frees the events */
/* Only launches with (kind_mask >> kind) & 1 and, for GEMMs, (epilogue_mask >> epilogue) & 1 are recorded (default: all).
 * An event pair costs a few microseconds of dispatch serialisation per launch: a timed run keeps only the kernel it
 * reports on. */
FC_API int fc_profile_select(fc_handle* h, uint32_t kind_mask, uint32_t epilogue_mask);
FC_API int fc_profile_reset(fc_handle* h);
FC_API int fc_profile_read(fc_handle* h, fc_prof_record* out, int32_t max_records); /* returns the record count */

#ifdef __cplusplus
}
#endif
#endif /* FITCLIP_HIP_H */

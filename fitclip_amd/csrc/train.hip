// C ABI of the KD training step (include/fitclip_hip.h, "training"): forward of a tower that KEEPS every activation the
// backward needs, and the backward that turns d(tower output) into parameter gradients.  Replaces, for the student
// encoder, what autograd does for `TeacherStudentLightningModule.training_step` (aligner/teacher_student.py:99-140) on
// `ClipVideoTextEncoder.encode_video / encode_text` (aligner/encoder/clip_video_text_encoder.py:80-94).
//
// Memory plan (sized for 288 GB of HBM): nothing is recomputed.  Per transformer block and token row the forward saves
//   x_in f32[w] | LN1(x) T[w] | qkv T[3w] | attention out T[w] | x_mid f32[w] | LN2(x_mid) T[w] | c_fc pre-act T[4w] |
//   QuickGELU out T[4w]                                                   = 64 w bytes in fp32 (49 KB at w = 768)
// i.e. 116 MB per 224^2 frame for ViT-B/16: 512 frames (one rank's share of BASELINE configs[4]) = 60 GB.
// All GEMMs are the inference kernels (gemm_kernel.h) - the dgrad products read transposed weight copies refreshed once
// per optimiser step (fc_train_prepare) - plus the split-M "TN" kernel of wgrad.hip for the weight gradients.
// Only the exact-fp32 mode trains (the reference trains in fp32: Trainer precision 32).
#include "../../include/fitclip_hip.h"
#include "common.h"
#include "handle.h"

#include <algorithm>
#include <string>

using namespace fc;

namespace {

#define FC_TRY(expr)              \
  do {                            \
    int _rc = (expr);             \
    if (_rc != FC_OK) return _rc; \
  } while (0)

constexpr size_t kZeroBytes = 16384;

struct Dims {
  int n, S, w, heads, L, causal;
  long M() const { return (long)n * S; }
};

Dims tower_dims(const fc_handle* h, int tower, int n) {
  const fc_config& c = h->cfg;
  if (tower == 0) return {n, h->vtokens(), c.vision_width, h->vheads(), c.vision_layers, 0};
  return {n, c.context_length, c.transformer_width, c.transformer_heads, c.transformer_layers, 1};
}

struct Layer {
  float *x_in, *xn1, *qkv, *ao, *x_mid, *xn2, *hpre, *hact;
};
struct Arena {
  std::vector<Layer> layers;
  float *x_fin, *hn, *delta, *x_raw, *patches;
  int* eot;
  size_t total;
};

// carve the activation arena (base may be null: sizes only)
Arena carve_arena(const fc_handle* h, int tower, int n, char* base) {
  const Dims d = tower_dims(h, tower, n);
  const size_t M = (size_t)d.M(), w = d.w;
  Arena a{};
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align_up(floats * sizeof(float));
    return p;
  };
  a.layers.resize(d.L);
  for (auto& l : a.layers) {
    l.x_in = take(M * w); l.xn1 = take(M * w); l.qkv = take(M * 3 * w); l.ao = take(M * w);
    l.x_mid = take(M * w); l.xn2 = take(M * w); l.hpre = take(M * 4 * w); l.hact = take(M * 4 * w);
  }
  a.x_fin = take(M * w);
  a.hn = take((size_t)n * w);
  a.delta = take(M * w);
  if (tower == 0) {
    a.x_raw = take(M * w);
    a.patches = take((size_t)n * h->patches() * h->patch_kp());
  } else {
    a.eot = reinterpret_cast<int*>(take((size_t)n));
  }
  a.total = off;
  return a;
}

struct Scratch {
  float *g, *dA, *dB, *dh, *tn, *small;
  void* tok;
  size_t tn_bytes, small_bytes, tok_bytes, total;
};

Scratch carve_scratch(const fc_handle* h, int tower, int n, char* base) {
  const Dims d = tower_dims(h, tower, n);
  const size_t M = (size_t)d.M(), w = d.w;
  const int E = h->cfg.embed_dim;
  Scratch s{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    float* p = base ? reinterpret_cast<float*>(base + off) : nullptr;
    off += align_up(bytes);
    return p;
  };
  s.g = take(M * w * 4);
  s.dA = take(M * 4 * w * 4);
  s.dB = take(M * w * 4);
  s.dh = take((size_t)n * w * 4);
  const int Mi = (int)M;
  size_t tn = std::max({gemm_tn_scratch_bytes(Mi, d.w, 4 * d.w), gemm_tn_scratch_bytes(Mi, 4 * d.w, d.w),
                        gemm_tn_scratch_bytes(Mi, d.w, d.w), gemm_tn_scratch_bytes(Mi, 3 * d.w, d.w),
                        gemm_tn_scratch_bytes(n, d.w, E)});
  if (tower == 0) tn = std::max(tn, gemm_tn_scratch_bytes(n * h->patches(), d.w, h->patch_k()));
  s.tn_bytes = tn;
  s.tn = take(tn);
  s.small_bytes = std::max(layernorm_bwd_scratch_bytes(d.w), (size_t)d.S * w * 4);
  s.small = take(s.small_bytes);
  if (tower == 1) {
    s.tok_bytes = token_grad_scratch_bytes(Mi, h->cfg.vocab_size);
    s.tok = take(s.tok_bytes);
  }
  s.total = off;
  return s;
}

int gemm_f32(int epi, const void* A, const void* W, const float* bias, void* C, const float* aux, int M, int N, int K,
             int ldc, int P, hipStream_t st) {
  GemmArgs a{};
  a.A = A; a.W = W; a.bias = bias; a.C = C; a.aux = aux; a.alpha = 1.f;
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = ldc; a.P = P;
  return launch_gemm(PREC_F32, epi, a, 0, st);
}

struct BlockNames {
  std::string in_w, in_b, out_w, out_b, ln1_w, ln1_b, fc_w, fc_b, proj_w, proj_b, ln2_w, ln2_b;
};
BlockNames block_names(const std::string& prefix, int i) {
  const std::string b = prefix + ".resblocks." + std::to_string(i);
  return {b + ".attn.in_proj_weight", b + ".attn.in_proj_bias", b + ".attn.out_proj.weight", b + ".attn.out_proj.bias",
          b + ".ln_1.weight", b + ".ln_1.bias", b + ".mlp.c_fc.weight", b + ".mlp.c_fc.bias",
          b + ".mlp.c_proj.weight", b + ".mlp.c_proj.bias", b + ".ln_2.weight", b + ".ln_2.bias"};
}

int check_train(const fc_handle* h, const char* what) {
  if (!h) return fail(FC_EINVAL, "%s: null handle", what);
  if (h->cfg.precision != FC_PREC_F32)
    return fail(FC_EINVAL, "%s: only precision fp32 handles can train (the reference trains in float32)", what);
  if (!h->packed) return fail(FC_ESTATE, "%s: call fc_pack_weights first", what);
  if (!h->train_ready) return fail(FC_ESTATE, "%s: call fc_train_prepare after every weight update", what);
  return FC_OK;
}

// The residual blocks with every intermediate kept (see the memory plan above); `a.layers[0].x_in` holds the block input.
int forward_blocks(fc_handle* h, const Tower& t, const Arena& a, const Dims& d, const float* fin_w, const float* fin_b,
                   const int* pool_idx, long pool_step, hipStream_t st) {
  const int M = (int)d.M(), w = d.w;
  const long xs_pool = pool_idx ? w : pool_step * w;
  FC_TRY(launch_layernorm(a.layers[0].x_in, w, nullptr, t.blocks[0].ln1_w, t.blocks[0].ln1_b, a.layers[0].xn1, w, 0, M, w, st));
  for (int l = 0; l < d.L; ++l) {
    const Block& b = t.blocks[l];
    const Layer& s = a.layers[l];
    FC_TRY(gemm_f32(EPI_BIAS_T, s.xn1, b.in_w, b.in_b, s.qkv, nullptr, M, 3 * w, w, 3 * w, 0, st));
    FC_TRY(launch_attention(PREC_F32, s.qkv, s.ao, d.n, d.S, d.heads, d.causal, st));
    FC_TRY(gemm_f32(EPI_BIAS_T, s.ao, b.out_w, b.out_b, a.delta, nullptr, M, w, w, w, 0, st));
    FC_TRY(launch_add_layernorm(s.x_in, w, a.delta, w, nullptr, b.ln2_w, b.ln2_b, s.xn2, w, 0, M, w, 1, 0, st, s.x_mid));
    FC_TRY(gemm_f32(EPI_BIAS_T, s.xn2, b.fc_w, b.fc_b, s.hpre, nullptr, M, 4 * w, w, 4 * w, 0, st));
    FC_TRY(launch_quickgelu(s.hpre, s.hact, PREC_F32, (size_t)M * 4 * w, st));
    FC_TRY(gemm_f32(EPI_BIAS_T, s.hact, b.proj_w, b.proj_b, a.delta, nullptr, M, w, 4 * w, w, 0, st));
    if (l + 1 < d.L) {
      const Block& nb = t.blocks[l + 1];
      FC_TRY(launch_add_layernorm(s.x_mid, w, a.delta, w, nullptr, nb.ln1_w, nb.ln1_b, a.layers[l + 1].xn1, w, 0, M, w, 1,
                                  0, st, a.layers[l + 1].x_in));
    } else {  // only the pooled rows (CLS / EOT) are read after the last block: their sum and final LayerNorm
      FC_TRY(launch_add_layernorm(s.x_mid, xs_pool, a.delta, xs_pool, pool_idx, fin_w, fin_b, a.hn, w, 0, d.n, w, 1, 0, st,
                                  a.x_fin));
    }
  }
  return FC_OK;
}

struct GradCtx {
  fc_handle* h;
  const Scratch& sc;
  float beta;
  hipStream_t st;
  float* grad(const std::string& n) const { return h->grad(n); }
  // dW = A^T B (A = dY), and with `bias` the bias gradient = column sums of dY from the same pass over it
  int tn(const float* A, const float* B, int M, int N1, int N2, int lda, int ldb, int a_skip, float* C, int ldc,
         float* bias = nullptr) const {
    return launch_gemm_tn(A, B, M, N1, N2, lda, ldb, a_skip, 1.f, beta, C, ldc, sc.tn, sc.tn_bytes, h->zeros, st, bias);
  }
  int ln_bwd(const float* x, long xs, const int* gather, const float* dy, long dys, int dy_compact, const float* gamma,
             float* out, long os, int acc, int rows, int D, float* dgamma, float* dbeta) const {
    return launch_layernorm_backward(x, xs, gather, dy, PREC_F32, dys, dy_compact, gamma, out, os, acc, rows, D, dgamma,
                                     dbeta, beta, sc.small, sc.small_bytes, st);
  }
};

// From sc.g = dL/d(block-stack output) (zero outside the pooled rows) to sc.g = dL/d(layers[0].x_in), accumulating
// every block parameter's gradient on the way.
int backward_blocks(const GradCtx& c, const Tower& t, const std::string& prefix, const Arena& a, const Dims& d) {
  const int M = (int)d.M(), w = d.w;
  const Scratch& sc = c.sc;
  const float* zeros = c.h->zeros;
  for (int l = d.L - 1; l >= 0; --l) {
    const Block& b = t.blocks[l];
    const Layer& s = a.layers[l];
    const BlockNames nm = block_names(prefix, l);
    // ---- MLP branch: x_out = x_mid + c_proj(quickgelu(c_fc(LN2(x_mid))))
    FC_TRY(c.tn(sc.g, s.hact, M, w, 4 * w, w, 4 * w, 0, c.grad(nm.proj_w), 4 * w, c.grad(nm.proj_b)));
    FC_TRY(gemm_f32(EPI_DGELU_T, sc.g, b.proj_wT, zeros, sc.dA, s.hpre, M, 4 * w, w, 4 * w, 0, c.st));  // d pre-activation
    FC_TRY(c.tn(sc.dA, s.xn2, M, 4 * w, w, 4 * w, w, 0, c.grad(nm.fc_w), w, c.grad(nm.fc_b)));
    FC_TRY(gemm_f32(EPI_BIAS_T, sc.dA, b.fc_wT, zeros, sc.dB, nullptr, M, w, 4 * w, w, 0, c.st));       // d LN2 output
    FC_TRY(c.ln_bwd(s.x_mid, w, nullptr, sc.dB, w, 0, b.ln2_w, sc.g, w, 1, M, w, c.grad(nm.ln2_w), c.grad(nm.ln2_b)));
    // ---- attention branch: x_mid = x_in + out_proj(attention(in_proj(LN1(x_in))))
    FC_TRY(c.tn(sc.g, s.ao, M, w, w, w, w, 0, c.grad(nm.out_w), w, c.grad(nm.out_b)));
    FC_TRY(gemm_f32(EPI_BIAS_T, sc.g, b.out_wT, zeros, sc.dB, nullptr, M, w, w, w, 0, c.st));           // d attention out
    FC_TRY(launch_attention_backward(PREC_F32, s.qkv, s.ao, sc.dB, sc.dA, d.n, d.S, d.heads, d.causal, c.st));
    FC_TRY(c.tn(sc.dA, s.xn1, M, 3 * w, w, 3 * w, w, 0, c.grad(nm.in_w), w, c.grad(nm.in_b)));
    FC_TRY(gemm_f32(EPI_BIAS_T, sc.dA, b.in_wT, zeros, sc.dB, nullptr, M, w, 3 * w, w, 0, c.st));       // d LN1 output
    FC_TRY(c.ln_bwd(s.x_in, w, nullptr, sc.dB, w, 0, b.ln1_w, sc.g, w, 1, M, w, c.grad(nm.ln1_w), c.grad(nm.ln1_b)));
  }
  return FC_OK;
}

// d(tower output z [n, E]) -> gradient of the projection, of the final LayerNorm, and sc.g = dL/d(block-stack output)
int backward_head(const GradCtx& c, const Arena& a, const Dims& d, const float* dz, const std::string& proj_name,
                  const std::string& ln_name, const int* pool_idx, long pool_step) {
  const fc_handle* h = c.h;
  const int E = h->cfg.embed_dim, w = d.w;
  const Scratch& sc = c.sc;
  const long xs_pool = pool_idx ? w : pool_step * w;
  FC_TRY(c.tn(a.hn, dz, d.n, w, E, w, E, 0, c.grad(proj_name), E));                                       // d proj [w, E]
  FC_TRY(gemm_f32(EPI_STORE_F32, dz, h->w(proj_name), nullptr, sc.dh, nullptr, d.n, w, E, w, 0, c.st));  // dz . proj^T
  if (hipMemsetAsync(sc.g, 0, (size_t)d.M() * w * sizeof(float), c.st) != hipSuccess)
    return fail(FC_ELAUNCH, "backward: hipMemsetAsync failed");
  return c.ln_bwd(a.x_fin, xs_pool, pool_idx, sc.dh, w, 1, h->w(ln_name + ".weight"), sc.g, xs_pool, 0, d.n, w,
                  c.grad(ln_name + ".weight"), c.grad(ln_name + ".bias"));
}

int check_grads(const fc_handle* h, const char* what, bool visual) {
  for (auto& n : h->names) {
    const bool is_visual = n.rfind("visual.", 0) == 0;
    if (is_visual != visual) continue;
    if (!h->slots.at(n).grad) return fail(FC_ESTATE, "%s: no gradient buffer for \"%s\" (fc_set_grad)", what, n.c_str());
  }
  return FC_OK;
}

}  // namespace

extern "C" {

int fc_set_grad(fc_handle* h, const char* name, float* dev) {
  if (!h || !name) return fail(FC_EINVAL, "fc_set_grad: null argument");
  auto it = h->slots.find(name);
  if (it == h->slots.end()) return fail(FC_EINVAL, "fc_set_grad: unexpected key \"%s\"", name);
  if (dev && ((uintptr_t)dev & 15)) return fail(FC_EINVAL, "fc_set_grad: \"%s\" must be 16-byte aligned", name);
  it->second.grad = dev;
  return FC_OK;
}

size_t fc_train_weights_bytes(const fc_handle* h) {
  if (!h) return 0;
  size_t total = kZeroBytes;
  auto add = [&](int width, int layers) { total += (size_t)layers * 4 * align_up((size_t)4 * width * width * 4); };
  add(h->cfg.vision_width, h->cfg.vision_layers);
  add(h->cfg.transformer_width, h->cfg.transformer_layers);
  return total;
}

int fc_train_prepare(fc_handle* h, void* arena, size_t bytes, fc_stream st) {
  if (!h) return fail(FC_EINVAL, "fc_train_prepare: null handle");
  if (h->cfg.precision != FC_PREC_F32) return fail(FC_EINVAL, "fc_train_prepare: only precision fp32 handles can train");
  if (!h->packed) return fail(FC_ESTATE, "fc_train_prepare: call fc_pack_weights first");
  if (!arena || ((uintptr_t)arena & 255) || bytes < fc_train_weights_bytes(h))
    return fail(FC_ENOMEM, "fc_train_prepare: arena needs %zu bytes, 256-byte aligned", fc_train_weights_bytes(h));
  char* base = static_cast<char*>(arena);
  if (hipMemsetAsync(base, 0, kZeroBytes, st) != hipSuccess) return fail(FC_ELAUNCH, "fc_train_prepare: memset failed");
  h->zeros = reinterpret_cast<const float*>(base);
  size_t off = kZeroBytes;
  auto fill = [&](Tower& t, int width) -> int {
    const size_t slot = align_up((size_t)4 * width * width * 4);
    for (auto& b : t.blocks) {
      struct { const void* src; const void** dst; int rows, cols; } items[4] = {
          {b.in_w, &b.in_wT, 3 * width, width}, {b.out_w, &b.out_wT, width, width},
          {b.fc_w, &b.fc_wT, 4 * width, width}, {b.proj_w, &b.proj_wT, width, 4 * width}};
      for (auto& it : items) {
        void* dst = base + off;
        FC_TRY(launch_transpose_convert(static_cast<const float*>(it.src), dst, PREC_F32, it.rows, it.cols, st));
        *it.dst = dst;
        off += slot;
      }
    }
    return FC_OK;
  };
  FC_TRY(fill(h->vis, h->cfg.vision_width));
  FC_TRY(fill(h->txt, h->cfg.transformer_width));
  h->train_ready = true;
  return FC_OK;
}

size_t fc_train_arena_bytes(const fc_handle* h, int32_t tower, int32_t n) {
  if (!h || n <= 0 || tower < 0 || tower > 1) return 0;
  return carve_arena(h, tower, n, nullptr).total;
}
size_t fc_train_scratch_bytes(const fc_handle* h, int32_t tower, int32_t n) {
  if (!h || n <= 0 || tower < 0 || tower > 1) return 0;
  return carve_scratch(h, tower, n, nullptr).total;
}

int fc_encode_image_train(fc_handle* h, const float* frames, int32_t n, float* out, void* arena, size_t arena_bytes,
                          fc_stream st) {
  FC_TRY(check_train(h, "fc_encode_image_train"));
  if (n <= 0 || !frames || !out || !arena) return fail(FC_EINVAL, "fc_encode_image_train: bad argument");
  if (((uintptr_t)frames | (uintptr_t)out) & 15 || ((uintptr_t)arena & 255))
    return fail(FC_EINVAL, "fc_encode_image_train: unaligned pointer");
  const Dims d = tower_dims(h, 0, n);
  const Arena a = carve_arena(h, 0, n, static_cast<char*>(arena));
  if (a.total > arena_bytes) return fail(FC_ENOMEM, "fc_encode_image_train: arena needs %zu bytes", a.total);
  const fc_config& c = h->cfg;
  const int P = h->patches(), Kp = h->patch_kp(), w = d.w;
  FC_TRY(launch_im2col(frames, a.patches, 0, n, c.image_resolution, c.vision_patch_size, Kp, st));
  FC_TRY(gemm_f32(EPI_PATCH_F32, a.patches, h->conv_w, nullptr, a.x_raw, h->w("visual.positional_embedding"), n * P, w,
                  Kp, w, P, st));
  FC_TRY(launch_fill_cls(a.x_raw, h->w("visual.class_embedding"), h->w("visual.positional_embedding"), n, d.S, w, st));
  FC_TRY(launch_layernorm(a.x_raw, w, nullptr, h->w("visual.ln_pre.weight"), h->w("visual.ln_pre.bias"),
                          a.layers[0].x_in, w, 0, (int)d.M(), w, st));
  FC_TRY(forward_blocks(h, h->vis, a, d, h->w("visual.ln_post.weight"), h->w("visual.ln_post.bias"), nullptr, d.S, st));
  return gemm_f32(EPI_STORE_F32, a.hn, h->vproj_t, nullptr, out, nullptr, n, c.embed_dim, w, c.embed_dim, 0, st);
}

int fc_encode_text_train(fc_handle* h, const int64_t* ids, int32_t n, float* out, void* arena, size_t arena_bytes,
                         fc_stream st) {
  FC_TRY(check_train(h, "fc_encode_text_train"));
  if (n <= 0 || !ids || !out || !arena) return fail(FC_EINVAL, "fc_encode_text_train: bad argument");
  if (((uintptr_t)out & 15) || ((uintptr_t)arena & 255)) return fail(FC_EINVAL, "fc_encode_text_train: unaligned pointer");
  const Dims d = tower_dims(h, 1, n);
  const Arena a = carve_arena(h, 1, n, static_cast<char*>(arena));
  if (a.total > arena_bytes) return fail(FC_ENOMEM, "fc_encode_text_train: arena needs %zu bytes", a.total);
  const fc_config& c = h->cfg;
  FC_TRY(launch_text_embed(ids, h->w("token_embedding.weight"), h->w("positional_embedding"), a.layers[0].x_in, a.eot, n,
                           d.S, d.w, c.vocab_size, st));
  FC_TRY(forward_blocks(h, h->txt, a, d, h->w("ln_final.weight"), h->w("ln_final.bias"), a.eot, 0, st));
  return gemm_f32(EPI_STORE_F32, a.hn, h->tproj_t, nullptr, out, nullptr, n, c.embed_dim, d.w, c.embed_dim, 0, st);
}

int fc_encode_image_backward(fc_handle* h, const float* dz, int32_t n, void* arena, size_t arena_bytes, void* scratch,
                             size_t scratch_bytes, int32_t accumulate, fc_stream st) {
  FC_TRY(check_train(h, "fc_encode_image_backward"));
  FC_TRY(check_grads(h, "fc_encode_image_backward", true));
  if (n <= 0 || !dz || !arena || !scratch || ((uintptr_t)dz & 15) || ((uintptr_t)arena & 255) || ((uintptr_t)scratch & 255))
    return fail(FC_EINVAL, "fc_encode_image_backward: bad argument");
  const Dims d = tower_dims(h, 0, n);
  const Arena a = carve_arena(h, 0, n, static_cast<char*>(arena));
  const Scratch sc = carve_scratch(h, 0, n, static_cast<char*>(scratch));
  if (a.total > arena_bytes || sc.total > scratch_bytes)
    return fail(FC_ENOMEM, "fc_encode_image_backward: arena needs %zu bytes, scratch %zu", a.total, sc.total);
  const GradCtx c{h, sc, accumulate ? 1.f : 0.f, st};
  const int w = d.w, P = h->patches();
  FC_TRY(backward_head(c, a, d, dz, "visual.proj", "visual.ln_post", nullptr, d.S));
  FC_TRY(backward_blocks(c, h->vis, "visual.transformer", a, d));
  // x_in[0] = ln_pre(x_raw);  x_raw = [cls + pos[0] | patches . conv^T + pos[1:]]
  FC_TRY(c.ln_bwd(a.x_raw, w, nullptr, sc.g, w, 0, h->w("visual.ln_pre.weight"), sc.g, w, 0, (int)d.M(), w,
                  c.grad("visual.ln_pre.weight"), c.grad("visual.ln_pre.bias")));
  FC_TRY(launch_seq_sum(sc.g, n, d.S, w, c.grad("visual.positional_embedding"), c.grad("visual.class_embedding"), c.beta,
                        sc.small, sc.small_bytes, st));
  return c.tn(sc.g, a.patches, n * P, w, h->patch_k(), w, h->patch_kp(), P, c.grad("visual.conv1.weight"), h->patch_k());
}

int fc_encode_text_backward(fc_handle* h, const int64_t* ids, const float* dz, int32_t n, void* arena, size_t arena_bytes,
                            void* scratch, size_t scratch_bytes, int32_t accumulate, fc_stream st) {
  FC_TRY(check_train(h, "fc_encode_text_backward"));
  FC_TRY(check_grads(h, "fc_encode_text_backward", false));
  if (n <= 0 || !ids || !dz || !arena || !scratch || ((uintptr_t)dz & 15) || ((uintptr_t)arena & 255) ||
      ((uintptr_t)scratch & 255))
    return fail(FC_EINVAL, "fc_encode_text_backward: bad argument");
  const Dims d = tower_dims(h, 1, n);
  const Arena a = carve_arena(h, 1, n, static_cast<char*>(arena));
  const Scratch sc = carve_scratch(h, 1, n, static_cast<char*>(scratch));
  if (a.total > arena_bytes || sc.total > scratch_bytes)
    return fail(FC_ENOMEM, "fc_encode_text_backward: arena needs %zu bytes, scratch %zu", a.total, sc.total);
  const GradCtx c{h, sc, accumulate ? 1.f : 0.f, st};
  FC_TRY(backward_head(c, a, d, dz, "text_projection", "ln_final", a.eot, 0));
  FC_TRY(backward_blocks(c, h->txt, "transformer", a, d));
  // x_in[0] = token_embedding[ids] + positional_embedding
  FC_TRY(launch_seq_sum(sc.g, n, d.S, d.w, c.grad("positional_embedding"), nullptr, c.beta, sc.small, sc.small_bytes, st));
  float* dtok = c.grad("token_embedding.weight");
  if (!accumulate &&
      hipMemsetAsync(dtok, 0, (size_t)h->cfg.vocab_size * d.w * sizeof(float), st) != hipSuccess)
    return fail(FC_ELAUNCH, "fc_encode_text_backward: hipMemsetAsync failed");
  return launch_token_grad(ids, sc.g, dtok, (int)d.M(), d.w, h->cfg.vocab_size, accumulate ? 1 : 0, sc.tok, sc.tok_bytes, st);
}

int fc_pool_normalize_backward(const float* z, const float* dout, float* dz, int32_t n_clips, int32_t frames, int32_t dim,
                               fc_stream st) {
  if (!z || !dout || !dz) return fail(FC_EINVAL, "fc_pool_normalize_backward: null argument");
  return launch_pool_normalize_backward(z, dout, dz, n_clips, frames, dim, st);
}
int fc_nce_loss_backward(const float* scores, int32_t n, float coef, float* dscores, float* ws, fc_stream st) {
  if (!scores || !dscores || !ws) return fail(FC_EINVAL, "fc_nce_loss_backward: null argument");
  return launch_loss_backward(scores, nullptr, n, n, coef, dscores, ws, st);
}
int fc_kd_loss_backward(const float* scores, const float* teacher, int32_t rows, int32_t cols, float coef,
                        float* dscores, float* ws, fc_stream st) {
  if (!scores || !teacher || !dscores || !ws) return fail(FC_EINVAL, "fc_kd_loss_backward: null argument");
  return launch_loss_backward(scores, teacher, rows, cols, coef, dscores, ws, st);
}

int fc_kd_teacher_scale_grad(const float* scores, const float* teacher, int32_t rows, int32_t cols, float* out, float* ws,
                             fc_stream st) {
  if (!scores || !teacher || !out || !ws) return fail(FC_EINVAL, "fc_kd_teacher_scale_grad: null argument");
  return launch_kd_teacher_scale_grad(scores, teacher, rows, cols, out, ws, st);
}

size_t fc_gemm_tn_scratch_bytes(int32_t M, int32_t N1, int32_t N2) { return 1024 + gemm_tn_scratch_bytes(M, N1, N2); }
int fc_gemm_tn(const float* A, const float* B, int32_t M, int32_t N1, int32_t N2, int32_t lda, int32_t ldb, float alpha,
               float beta, float* C, int32_t ldc, void* scratch, size_t scratch_bytes, fc_stream st) {
  if (!scratch || scratch_bytes < 1024 || ((uintptr_t)scratch & 255)) return fail(FC_ENOMEM, "fc_gemm_tn: scratch");
  if (hipMemsetAsync(scratch, 0, 1024, st) != hipSuccess) return fail(FC_ELAUNCH, "fc_gemm_tn: memset failed");
  return launch_gemm_tn(A, B, M, N1, N2, lda, ldb, 0, alpha, beta, C, ldc,
                        reinterpret_cast<float*>(static_cast<char*>(scratch) + 1024), scratch_bytes - 1024,
                        static_cast<const float*>(scratch), st);
}
int fc_attention_backward(int32_t precision, const void* qkv, const void* o, const void* d_o, void* dqkv, int32_t n_seq,
                          int32_t S, int32_t heads, int32_t causal, fc_stream st) {
  return launch_attention_backward(precision, qkv, o, d_o, dqkv, n_seq, S, heads, causal, st);
}
size_t fc_layernorm_backward_scratch_bytes(int32_t D) { return layernorm_bwd_scratch_bytes(D); }
int fc_layernorm_backward(const float* x, const float* dy, const float* gamma, float* dx, int32_t accumulate,
                          int32_t rows, int32_t D, float* dgamma, float* dbeta, void* scratch, size_t scratch_bytes,
                          fc_stream st) {
  return launch_layernorm_backward(x, D, nullptr, dy, PREC_F32, D, 0, gamma, dx, D, accumulate, rows, D, dgamma, dbeta, 0.f,
                                   static_cast<float*>(scratch), scratch_bytes, st);
}
size_t fc_token_embedding_backward_scratch_bytes(int32_t rows, int32_t vocab) {
  return rows < 0 || vocab < 0 ? 0 : token_grad_scratch_bytes(rows, vocab);
}
int fc_token_embedding_backward(const int64_t* ids, const float* d_rows, float* d_table, int32_t rows, int32_t D,
                                int32_t vocab, int32_t accumulate, void* scratch, size_t scratch_bytes, fc_stream st) {
  if (!ids || !d_rows || !d_table) return fail(FC_EINVAL, "fc_token_embedding_backward: null argument");
  return launch_token_grad(ids, d_rows, d_table, rows, D, vocab, accumulate ? 1 : 0, scratch, scratch_bytes, st);
}
int fc_dot(const float* a, const float* b, size_t n, float alpha, float beta, float* out, fc_stream st) {
  if (!a || !b || !out) return fail(FC_EINVAL, "fc_dot: null argument");
  return launch_dot(a, b, n, alpha, beta, out, st);
}
int fc_transpose(const float* in, float* out, int32_t rows, int32_t cols, fc_stream st) {
  if (!in || !out) return fail(FC_EINVAL, "fc_transpose: null argument");
  return launch_transpose_convert(in, out, PREC_F32, rows, cols, st);
}
int fc_adamw(float* p, const float* g, float* m, float* v, size_t n, double lr, double beta1, double beta2, double eps,
             double weight_decay, int32_t step, fc_stream st) {
  if (!p || !g || !m || !v) return fail(FC_EINVAL, "fc_adamw: null argument");
  return launch_adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, st);
}

}  // extern "C"

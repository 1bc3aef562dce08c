// C ABI of libfitclip_hip.so (see include/fitclip_hip.h): handle, weight table, and the two tower forwards that
// sequence the HIP kernels of this directory on the caller's stream.  No device allocation, no synchronisation.
#include "../../include/fitclip_hip.h"
#include "common.h"
#include "handle.h"
#include <cstdlib>

#include <cstdarg>
#include <map>
#include <mutex>
#include <set>
#include <memory>
#include <vector>

namespace fc {

static thread_local std::string g_error;

void set_error(const std::string& msg) { g_error = msg; }

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

hipError_t raise_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({dev, kernel})) return hipSuccess;
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.insert({dev, kernel});
  return e;
}

int device_cus() {
  static std::mutex mu;
  static std::map<int, int> known;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::lock_guard<std::mutex> lock(mu);
  auto it = known.find(dev);
  if (it != known.end()) return it->second;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  known[dev] = n;
  return n;
}

}  // namespace fc

namespace fc {
namespace {

void add_blocks(fc_handle* h, const std::string& prefix, int width, int layers) {
  auto add = [&](const std::string& n, std::vector<int64_t> shape) {
    h->names.push_back(n);
    h->slots[n].shape = std::move(shape);
  };
  for (int i = 0; i < layers; ++i) {
    const std::string b = prefix + ".resblocks." + std::to_string(i);
    add(b + ".attn.in_proj_weight", {3L * width, width});
    add(b + ".attn.in_proj_bias", {3L * width});
    add(b + ".attn.out_proj.weight", {width, width});
    add(b + ".attn.out_proj.bias", {width});
    add(b + ".ln_1.weight", {width});
    add(b + ".ln_1.bias", {width});
    add(b + ".mlp.c_fc.weight", {4L * width, width});
    add(b + ".mlp.c_fc.bias", {4L * width});
    add(b + ".mlp.c_proj.weight", {width, 4L * width});
    add(b + ".mlp.c_proj.bias", {width});
    add(b + ".ln_2.weight", {width});
    add(b + ".ln_2.bias", {width});
  }
}

void build_names(fc_handle* h) {
  const fc_config& c = h->cfg;
  auto add = [&](const std::string& n, std::vector<int64_t> shape) {
    h->names.push_back(n);
    h->slots[n].shape = std::move(shape);
  };
  const int64_t vw = c.vision_width, tw = c.transformer_width, p = c.vision_patch_size;
  add("positional_embedding", {c.context_length, tw});
  add("text_projection", {tw, c.embed_dim});
  add("visual.class_embedding", {vw});
  add("visual.positional_embedding", {h->vtokens(), vw});
  add("visual.proj", {vw, c.embed_dim});
  add("visual.conv1.weight", {vw, 3, p, p});
  add("visual.ln_pre.weight", {vw});
  add("visual.ln_pre.bias", {vw});
  add_blocks(h, "visual.transformer", (int)vw, c.vision_layers);
  add("visual.ln_post.weight", {vw});
  add("visual.ln_post.bias", {vw});
  add_blocks(h, "transformer", (int)tw, c.transformer_layers);
  add("token_embedding.weight", {c.vocab_size, tw});
  add("ln_final.weight", {tw});
  add("ln_final.bias", {tw});
}

size_t numel(const std::vector<int64_t>& s) {
  size_t n = 1;
  for (auto v : s) n *= (size_t)v;
  return n;
}

// GEMM weights that get a kernel-layout copy
enum PackMode { PACK_CONVERT, PACK_TRANSPOSE, PACK_PAD_ROWS, PACK_SPLIT3, PACK_SPLIT2 };
struct PackItem {
  std::string name;
  PackMode mode;
};
std::vector<PackItem> packed_list(const fc_handle* h) {
  std::vector<PackItem> l;
  const bool conv = h->cfg.precision == FC_PREC_BF16;  // f32 mode uses the caller's tensors directly
  auto blocks = [&](const std::string& prefix, int layers) {
    for (int i = 0; i < layers; ++i) {
      const std::string b = prefix + ".resblocks." + std::to_string(i);
      l.push_back({b + ".attn.in_proj_weight", PACK_CONVERT});
      l.push_back({b + ".attn.out_proj.weight", PACK_CONVERT});
      l.push_back({b + ".mlp.c_fc.weight", PACK_CONVERT});
      l.push_back({b + ".mlp.c_proj.weight", PACK_CONVERT});
    }
  };
  const bool padded = h->patch_kp() != h->patch_k();
  if (conv || padded) l.push_back({"visual.conv1.weight", padded ? PACK_PAD_ROWS : PACK_CONVERT});
  if (conv) {
    blocks("visual.transformer", h->cfg.vision_layers);
    blocks("transformer", h->cfg.transformer_layers);
  }
  if (h->split()) {  // x3 rows (three bf16 planes) of the visual tower's block weights (the fp32 tensors stay in use as well)
    for (int i = 0; i < h->cfg.vision_layers; ++i) {
      const std::string b = "visual.transformer.resblocks." + std::to_string(i);
      for (const char* n : {".attn.in_proj_weight", ".attn.out_proj.weight", ".mlp.c_fc.weight", ".mlp.c_proj.weight"})
        l.push_back({b + n, h->split2() ? PACK_SPLIT2 : PACK_SPLIT3});
    }
  }
  if (h->x2_text()) {  // split_gemm 2: the text tower's block weights too (large calls run them on the fp16 pipe: fc_encode_text)
    for (int i = 0; i < h->cfg.transformer_layers; ++i) {
      const std::string b = "transformer.resblocks." + std::to_string(i);
      for (const char* n : {".attn.in_proj_weight", ".attn.out_proj.weight", ".mlp.c_fc.weight", ".mlp.c_proj.weight"})
        l.push_back({b + n, PACK_SPLIT2});
    }
  }
  if (h->x2_patch()) l.push_back({"visual.conv1.weight", PACK_SPLIT2});   // [width, 3, p, p] = [width, 3 p^2] rows
  l.push_back({"visual.proj", PACK_TRANSPOSE});
  l.push_back({"text_projection", PACK_TRANSPOSE});
  return l;
}
// a weight as a matrix [rows, K]: the leading dimension against everything behind it (conv1.weight [width, 3, p, p])
inline long weight_cols(const std::vector<int64_t>& shape) { return shape.empty() || !shape[0] ? 0 : (long)(numel(shape) / (size_t)shape[0]); }
size_t packed_item_bytes(const fc_handle* h, const PackItem& e) {
  const auto& shape = h->slots.at(e.name).shape;
  if (e.mode == PACK_SPLIT3) return align_up((size_t)shape[0] * (size_t)x3_row_elems(shape[1]) * 2);
  if (e.mode == PACK_SPLIT2) return align_up((size_t)shape[0] * (size_t)x2_row_elems(weight_cols(shape)) * 2) + 256;  // + the scale pair
  const size_t n = e.mode == PACK_PAD_ROWS ? (size_t)shape[0] * h->patch_kp() : numel(shape);
  return align_up(n * h->esz);
}

struct ProfScope {
  fc_handle* h;
  hipStream_t st;
  int idx = -1;
  ProfScope(fc_handle* h_, hipStream_t st_, int prec, int epi, int tile, const GemmArgs& a) : h(h_), st(st_) {
    if (h && h->prof_cap && (h->prof_kinds & 1u) && (h->prof_epis >> epi & 1u) && (int)h->recs.size() < h->prof_cap) {
      idx = (int)h->recs.size();
      fc_prof_record r{};
      r.kind = 0; r.precision = prec; r.epilogue = epi; r.tile = tile; r.M = a.M; r.N = a.N; r.K = a.K; r.ms = -1.f;
      h->recs.push_back(r);
      (void)hipEventRecord(h->ev[2 * idx], st);
    }
  }
  // kind 1 = attention (M = sequences, N = heads, K = tokens), kind 2 = (add+)LayerNorm (M = rows, N = width, K = 0/1 = with add)
  ProfScope(fc_handle* h_, hipStream_t st_, int kind, int M, int N, int K) : h(h_), st(st_) {
    if (h && h->prof_cap && (h->prof_kinds >> kind & 1u) && (int)h->recs.size() < h->prof_cap) {
      idx = (int)h->recs.size();
      fc_prof_record r{};
      r.kind = kind; r.precision = h->cfg.precision; r.epilogue = -1; r.tile = 0; r.M = M; r.N = N; r.K = K; r.ms = -1.f;
      h->recs.push_back(r);
      (void)hipEventRecord(h->ev[2 * idx], st);
    }
  }
  ~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(h->ev[2 * idx + 1], st);
  }
  // attention records: which arithmetic the launch ran - `epilogue` = the fc_attention precision code of the kernel (0 fp32 MFMA,
  // 1 bf16, 3 fp32 with x3 rows out, 4 / 5 six bf16 products, 6 three fp16 products), `tile` = 1 when a split pass over its fp32
  // output followed (sequence lengths without a fused split attention)
  void form(int code, int split_pass = 0) {
    if (idx >= 0) { h->recs[idx].epilogue = code; h->recs[idx].tile = split_pass; }
  }
};

int gemm(fc_handle* h, int epi, const void* A, const void* W, const float* bias, void* C, const float* aux, int M,
         int N, int K, int ldc, int P, hipStream_t st) {
  GemmArgs a{};
  a.A = A; a.W = W; a.bias = bias; a.C = C; a.aux = aux; a.alpha = 1.f;
  a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = ldc; a.P = P;
  ProfScope ps(h, st, h->cfg.precision, epi, gemm_resolved_tile(h->cfg.precision, epi, a, h->cfg.gemm_tile), a);
  return launch_gemm(h->cfg.precision, epi, a, h->cfg.gemm_tile, st);
}

#define FC_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != FC_OK) return _rc; \
  } while (0)

#ifdef FITCLIP_DEBUG
// Debug builds only (`python -m fitclip_amd.build --debug` -> tools/bin/libfitclip_hip_debug.so): every tower
// call ends with a scan of its output for NaN / Inf and a HOST synchronisation - a tower that produced non-finite embeddings
// returns FC_ELAUNCH naming the call instead of handing them to the scoring.  The product library never defines FITCLIP_DEBUG (it
// allocates nothing and never synchronises).
__global__ void __launch_bounds__(256) nonfinite_kernel(const float* __restrict__ x, size_t n, int* __restrict__ count) {
  int c = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += !isfinite(x[i]);
  if (c) atomicAdd(count, c);
}
int debug_scan(const float* x, size_t n, hipStream_t st, const char* what) {
  static thread_local int* d_count = nullptr;
  if (!d_count && hipMalloc(reinterpret_cast<void**>(&d_count), sizeof(int)) != hipSuccess) return fail(FC_ENOMEM, "%s: debug scan", what);
  int h = 0;
  if (hipMemsetAsync(d_count, 0, sizeof(int), st) != hipSuccess) return fail(FC_ELAUNCH, "%s: debug scan", what);
  hipLaunchKernelGGL(nonfinite_kernel, dim3(256), dim3(256), 0, st, x, n, d_count);
  if (hipMemcpyAsync(&h, d_count, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    return fail(FC_ELAUNCH, "%s: debug scan", what);
  return h ? fail(FC_ELAUNCH, "%s: %d of %zu output values are NaN or Inf (FITCLIP_DEBUG scan)", what, h, n) : FC_OK;
}
#define FC_DEBUG_SCAN(x, n, st, what) FC_TRY(debug_scan(x, n, st, what))
#else
#define FC_DEBUG_SCAN(x, n, st, what) do { } while (0)
#endif

// split_gemm 2: the last line of defence of the range flag.  The writers of fp16 planes test |x| <= 65504 as they go (max on
// v_max3_f32 / fmaxf: fast, but a max drops NaN operands), and an infinity or a NaN anywhere in the blocks - an overflowed plane, a
// NaN frame, a non-finite weight of the fp32 part of the tower - reaches the frame's embedding through LayerNorm and the
// projections.  So every visual-tower call scans its [n, embed_dim] output (a few KB) and ORs the flag: no value leaves with rc 0.
__global__ void __launch_bounds__(256) nonfinite_flag_kernel(const float* __restrict__ x, size_t n, int* __restrict__ flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    bad |= (__float_as_uint(x[i]) & 0x7f800000u) == 0x7f800000u;
  if (bad) atomicOr(flag, 1);
}

// The end of a tower call that wrote fp16 planes: the output scan, the flag's copy to its pinned mirror (no synchronisation: the
// NEXT call and fc_range_status read it) and - fc_range_strict - the wait that makes THIS call answer for its own values.
int finish_range(fc_handle* h, const float* out, size_t total, hipStream_t st, const char* who) {
  if (h->sat_flag) {
    hipLaunchKernelGGL(nonfinite_flag_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 256)), dim3(256), 0, st, out, total, h->sat_flag);
    FC_CHECK_LAUNCH("output scan");
    if (hipMemcpyAsync(h->sat_host, h->sat_flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess)
      return fail(FC_ELAUNCH, "%s: range flag copy", who);
  }
  FC_DEBUG_SCAN(out, total, st, who);
  if (h->sat_flag && h->strict_range) {   // one host synchronisation
    if (hipStreamSynchronize(st) != hipSuccess) return fail(FC_ELAUNCH, "%s: synchronise for the range flag", who);
    if (*h->sat_host)
      return fail(FC_ERANGE, "%s: an activation (or a weight) beyond fp16's range (65504) or not finite was met: split_gemm = 2 "
                             "cannot represent this model's values; the embeddings of this call are not valid", who);
  }
  return FC_OK;
}

struct Scratch {
  float* x;
  char* xn;
  char* big;
  float* d;   // split mode only: fp32 [M, w] projection deltas (out_proj, c_proj)
  char* clsn;
  int* eot;
  size_t total;
};

// workspace carve for `c` items of `tokens` tokens and width `w` (base may be null: sizes only).  `split`: the layout
// of the split-fp32 visual tower - xn holds x3 rows (8 w bytes of address space), big the fp32 QKV rows or the x3 MLP
// hidden rows (32 w bytes), d the fp32 deltas; it contains the plain fp32 layout, which small passes fall back to.
// split = 2 (x2 rows: 4 bytes per value, the fp32 layout's sizes) only adds d.
Scratch carve(char* base, int c, int tokens, int w, int esz, int min_big_cols, int split = 0) {
  Scratch s{};
  const size_t M = (size_t)c * tokens;
  const size_t big_cols = (size_t)std::max(4 * w, min_big_cols);
  const size_t xn_row = split == 1 ? (size_t)x3_row_elems(w) * 2 : (size_t)w * esz;
  const size_t big_row = split == 1 ? std::max(big_cols * esz, (size_t)x3_row_elems(4 * w) * 2) : big_cols * esz;
  const size_t o_x = 0;
  const size_t o_xn = o_x + align_up(M * w * 4);
  const size_t o_big = o_xn + align_up(M * xn_row);
  const size_t o_d = o_big + align_up(M * big_row);
  const size_t o_cls = o_d + (split ? align_up(M * w * 4) : 0);
  const size_t o_eot = o_cls + align_up((size_t)c * w * esz);
  s.total = o_eot + align_up((size_t)c * 4);
  if (base) {
    s.x = reinterpret_cast<float*>(base + o_x);
    s.xn = base + o_xn;
    s.big = base + o_big;
    s.d = split ? reinterpret_cast<float*>(base + o_d) : nullptr;
    s.clsn = base + o_cls;
    s.eot = reinterpret_cast<int*>(base + o_eot);
  }
  return s;
}

// The 12 pre-LN residual blocks followed by the final LayerNorm of the pooled rows (CLS rows: pool_idx == nullptr,
// row i * pool_step; EOT rows: pool_idx).  out_proj and c_proj update the residual stream IN their epilogue
// (EPI_RESID_F32: x += acc + bias), so the LayerNorm behind them reads one fp32 row and writes none back: fp32 8 instead of
// 16 bytes per element in a pass that is HBM-bound, against 4 more bytes read by a GEMM that is MFMA-bound; the same fp32
// add a fused add+LayerNorm performs - same bits (+1.3 % on the bench step).  bf16: 6 instead of 12 bytes in the LayerNorm
// pass against 6 more in the GEMMs, which are closer to the memory there (+1.5 %); the stream receives the fp32 accumulator
// instead of its bf16 rounding.
//
// cfg.prune_last_block: the towers only read the pooled row of each sequence after the last block, so in the last block
// everything after the attention (out_proj, LayerNorm 2, c_fc, c_proj: 72 % of a block's FLOPs) is only computed for
// those rows.  Rows of a GEMM / LayerNorm are independent, so the pooled rows come out bit-identical.
// Visual tower entry folded into the first LayerNorm of the first block (layernorm_pair_kernel): class embedding and
// positional_embedding[0] for the CLS rows, ln_pre weights.
struct TowerEntry {
  const float *cls, *pos0, *pre_w, *pre_b;
};

#ifdef FITCLIP_LAB
// (tools/ only: libfitclip_hip_lab.so) FITCLIP_LAB_SKIP_LN=1: the LayerNorm launches of blocks 1.. are skipped - the GEMMs run on
// the first block's normalised rows, results are meaningless - to time the CEILING of fusing LayerNorm into the GEMMs
static bool lab_skip_ln() {
  static const bool v = [] { const char* e = getenv("FITCLIP_LAB_SKIP_LN"); return e && atoi(e) != 0; }();
  return v;
}
// FITCLIP_LAB_LN_FUSE=1: what the algebraic LayerNorm fusion would cost with a statistics-only pass in place of every block
// LayerNorm (mean, 1 / std per row; no normalised row) and the correction rstd (acc - mean g) + c in the QKV / c_fc epilogues
// (gemm_kernel.h, on stand-in vectors; the switch travels in GemmArgs::P, which these epilogues do not use): results
// meaningless, timing valid (tools/ln_ceiling.py)
static bool lab_ln_fuse() {
  static const bool v = [] { const char* e = getenv("FITCLIP_LAB_LN_FUSE"); return e && atoi(e) != 0; }();
  return v;
}
// FITCLIP_LAB_GELU=2: the c_fc epilogue without its QuickGELU (what the function costs in place); 3: the plain form
// x / (1 + 2^(-1.702 log2 e x)) without the compensated exponent (what dropping the compensation would buy)
static int lab_gelu() {
  static const int v = [] { const char* e = getenv("FITCLIP_LAB_GELU"); return e ? atoi(e) : 0; }();
  return v;
}
#else
static constexpr bool lab_skip_ln() { return false; }
static constexpr bool lab_ln_fuse() { return false; }
static constexpr int lab_gelu() { return 0; }
#endif

int run_blocks(fc_handle* h, const Tower& t, const Scratch& s, int n_seq, int S, int w, int heads, int causal,
               const float* fin_w, const float* fin_b, const int* pool_idx, long pool_step, hipStream_t st,
               const TowerEntry* entry = nullptr) {
  const int M = n_seq * S;
  const int kind = h->cfg.precision;
  const int esz = h->esz;
  const size_t nl = t.blocks.size();
  const long xs_pool = pool_idx ? w : pool_step * w;  // x row stride seen through the pooling index
  for (size_t l = 0; l < nl; ++l) {
    const Block& b = t.blocks[l];
    if (l == 0 && entry) {
      FC_TRY(launch_layernorm_pair(s.x, entry->cls, entry->pos0, S, entry->pre_w, entry->pre_b, b.ln1_w, b.ln1_b, s.xn,
                                   kind, M, w, st));
    } else if (l == 0) {
      FC_TRY(launch_layernorm(s.x, w, nullptr, b.ln1_w, b.ln1_b, s.xn, w, kind, M, w, st));
    } else if (!lab_skip_ln()) {
      ProfScope ps(h, st, 2, M, w, 1);
      FC_TRY(launch_layernorm(s.x, w, nullptr, b.ln1_w, b.ln1_b, s.xn, lab_ln_fuse() && kind == 0 ? -w : w, kind, M, w, st));
    }
    FC_TRY(gemm(h, EPI_BIAS_T, s.xn, b.in_w, b.in_b, s.big, nullptr, M, 3 * w, w, 3 * w, lab_ln_fuse() && kind == 0 && l > 0 ? 1 : 0, st));
    {
      ProfScope ps(h, st, 1, n_seq, heads, S);
      ps.form(kind);
      FC_TRY(launch_attention(kind, s.big, s.xn, n_seq, S, heads, causal, st));
    }
    if (l + 1 == nl && h->cfg.prune_last_block) {
      // the pooled rows of the attention output and of the residual stream, compact; then the block's tail as above
      const size_t blk = align_up((size_t)n_seq * w * 4);
      char* ao = s.big;                                   // [n, w] T   attention output
      float* xp = reinterpret_cast<float*>(s.big + blk);  // [n, w] f32 residual stream
      char* xc = s.big + 2 * blk;                         // [n, w] T   LayerNorm 2 output
      char* hc = s.big + 3 * blk;                         // [n, 4w] T  MLP hidden
      FC_TRY(launch_gather_rows(s.xn, pool_idx, pool_step, ao, n_seq, w * esz, st));
      FC_TRY(launch_gather_rows(reinterpret_cast<const char*>(s.x), pool_idx, pool_step, reinterpret_cast<char*>(xp), n_seq, w * 4, st));
      FC_TRY(gemm(h, EPI_RESID_F32, ao, b.out_w, b.out_b, xp, nullptr, n_seq, w, w, w, 0, st));
      FC_TRY(launch_layernorm(xp, w, nullptr, b.ln2_w, b.ln2_b, xc, w, kind, n_seq, w, st));
      FC_TRY(gemm(h, EPI_GELU_T, xc, b.fc_w, b.fc_b, hc, nullptr, n_seq, 4 * w, w, 4 * w, 0, st));
      FC_TRY(gemm(h, EPI_RESID_F32, hc, b.proj_w, b.proj_b, xp, nullptr, n_seq, w, 4 * w, w, 0, st));
      return launch_layernorm(xp, w, nullptr, fin_w, fin_b, s.clsn, w, kind, n_seq, w, st);
    }
    FC_TRY(gemm(h, EPI_RESID_F32, s.xn, b.out_w, b.out_b, s.x, nullptr, M, w, w, w, 0, st));
    if (!(lab_skip_ln() && l > 0)) {
      ProfScope ps(h, st, 2, M, w, 1);
      FC_TRY(launch_layernorm(s.x, w, nullptr, b.ln2_w, b.ln2_b, s.xn, lab_ln_fuse() && kind == 0 && l > 0 ? -w : w, kind, M, w, st));
    }
    FC_TRY(gemm(h, EPI_GELU_T, s.xn, b.fc_w, b.fc_b, s.big, nullptr, M, 4 * w, w, 4 * w, kind == 0 ? (lab_ln_fuse() && l > 0 ? 1 : lab_gelu()) : 0, st));
    FC_TRY(gemm(h, EPI_RESID_F32, s.big, b.proj_w, b.proj_b, s.x, nullptr, M, w, 4 * w, w, 0, st));
  }
  return launch_layernorm(s.x, xs_pool, pool_idx, fin_w, fin_b, s.clsn, w, kind, n_seq, w, st);
}

// ---- split-fp32 visual tower (cfg.split_gemm): the same block sequence with the four big GEMMs on the bf16 matrix
// cores over three-plane operands (gemm_split3.h, common.h).  Producers write x3 rows directly: LayerNorm (KIND_X3), the
// attention kernel (attention_split.hip: its two products in the same arithmetic), c_fc's QuickGELU epilogue (EPI_GELU_X3);
// QKV returns fp32 (EPI_BIAS_F32), out_proj / c_proj add to the fp32 residual stream in place (EPI_RESID3_F32).
// Everything else - residual stream, LayerNorm statistics, softmax arithmetic - is the fp32 path's code.
int gemm_x3(fc_handle* h, int epi, const void* A3, const void* W3, const float* bias, void* C, int M, int N, int K,
            int ldc, hipStream_t st) {
  GemmArgs a{};
  a.A = A3; a.W = W3; a.bias = bias; a.C = C; a.aux = nullptr; a.alpha = 1.f;
  a.M = M; a.N = N; a.K = K; a.lda = (int)x3_row_elems(K); a.ldw = a.lda; a.ldc = ldc; a.P = 0;
  GemmArgs rec = a;  // the profiling record counts the bf16 work: six products per fp32 product
  rec.K = 6 * K;
  ProfScope ps(h, st, PREC_BF16, epi, 3, rec);
  return launch_gemm_split3(epi, a, st);
}

// A pass qualifies for the three-plane GEMMs when K spans the K-steps the kernel's prologue assumes (the operands are
// addressed from 64-bit tile bases: no size limit on the pass).
bool x3_pass_ok(int M, int w) { return M > 0 && w >= 64 && w % 32 == 0; }

int run_blocks_x3(fc_handle* h, const Tower& t, const Scratch& s, int n_seq, int S, int w, int heads, const float* fin_w,
                  const float* fin_b, long pool_step, hipStream_t st, const TowerEntry& entry) {
  const int M = n_seq * S;
  const long ld3 = x3_row_elems(w);
  const size_t nl = t.blocks.size();
  // out_proj and c_proj update the residual stream in their epilogue (EPI_RESID3_F32, as the fp32 path's EPI_RESID_F32): the
  // LayerNorm behind them reads one fp32 row and writes the x3 rows - 12 instead of 20 bytes per element
  for (size_t l = 0; l < nl; ++l) {
    const Block& b = t.blocks[l];
    if (l == 0) {
      FC_TRY(launch_layernorm_pair(s.x, entry.cls, entry.pos0, S, entry.pre_w, entry.pre_b, b.ln1_w, b.ln1_b, s.xn,
                                   KIND_X3, M, w, st));
    } else {
      ProfScope ps(h, st, 2, M, w, 1);
      FC_TRY(launch_layernorm(s.x, w, nullptr, b.ln1_w, b.ln1_b, s.xn, ld3, KIND_X3, M, w, st));
    }
    FC_TRY(gemm_x3(h, EPI_BIAS_F32, s.xn, b.in_w3, b.in_b, s.big, M, 3 * w, w, 3 * w, st));
    {
      ProfScope ps(h, st, 1, n_seq, heads, S);
      if (attention_split_supported(S, 0)) {
        ps.form(ATTN_SPLIT);
        FC_TRY(launch_attention_split(s.big, s.xn, n_seq, S, heads, st));
      } else if (attention_x3_supported(S, 0)) {
        ps.form(KIND_X3);
        FC_TRY(launch_attention_x3(s.big, s.xn, n_seq, S, heads, st));
      } else {  // other sequence lengths: the fp32 kernel of that length, then the split as a pass of its own
        ps.form(PREC_F32, 1);
        FC_TRY(launch_attention(PREC_F32, s.big, s.d, n_seq, S, heads, 0, st));
        FC_TRY(launch_split3_rows(s.d, w, s.xn, ld3, M, w, st));
      }
    }
    FC_TRY(gemm_x3(h, EPI_RESID3_F32, s.xn, b.out_w3, b.out_b, s.x, M, w, w, w, st));
    {
      ProfScope ps(h, st, 2, M, w, 1);
      FC_TRY(launch_layernorm(s.x, w, nullptr, b.ln2_w, b.ln2_b, s.xn, ld3, KIND_X3, M, w, st));
    }
    FC_TRY(gemm_x3(h, EPI_GELU_X3, s.xn, b.fc_w3, b.fc_b, s.big, M, 4 * w, w, (int)x3_row_elems(4 * w), st));
    FC_TRY(gemm_x3(h, EPI_RESID3_F32, s.big, b.proj_w3, b.proj_b, s.x, M, w, 4 * w, w, st));
  }
  return launch_layernorm(s.x, pool_step * w, nullptr, fin_w, fin_b, s.clsn, w, PREC_F32, n_seq, w, st);
}

// ---- split-fp32 visual tower on TWO fp16 planes (cfg.split_gemm = 2): the sequence of run_blocks_x3 over x2 rows - three fp16
// products per fp32 product (gemm_split2.h).  Producers: LayerNorm (KIND_X2), the split attention (x2 rows out), c_fc's QuickGELU
// epilogue (EPI_GELU_X2).  Every writer of x2 rows raises h->sat_flag when a value exceeds fp16's range.
int gemm_x2(fc_handle* h, int epi, const void* A2, const void* W2, const float* scale2, const float* bias, void* C, int M, int N,
            int K, int ldc, hipStream_t st, const float* aux = nullptr, int P = 0) {
  GemmArgs a{};
  a.A = A2; a.W = W2; a.bias = bias; a.C = C; a.aux = aux; a.alpha = 1.f; a.wscale = scale2; a.sat_flag = h->sat_flag;
  a.M = M; a.N = N; a.K = K; a.lda = (int)x2_row_elems(K); a.ldw = a.lda; a.ldc = ldc; a.P = P;
  GemmArgs rec = a;  // the profiling record counts the fp16 work: three products per fp32 product (precision code 2 = fp16 pipe)
  rec.K = 3 * K;
  int rows = 0, wgs = 0;
  gemm_split2_plan(M, N, 0, &rows, &wgs);   // (the record's `tile`: the tile height of this launch)
  ProfScope ps(h, st, 2, epi, rows, rec);
  return launch_gemm_split2(epi, a, st);
}

bool x2_pass_ok(int M, int w) { return M > 0 && w >= 128 && w % 64 == 0; }

// The text tower (width 512 in every CLIP the reference loads: N = 512 gives the 256 x 256-tile kernel two tile columns) joins
// split_gemm 2 per CALL, from this many token rows on.  Measured (tools/text_gemm_x2_probe.py, the four GEMMs of a block, fp32
// kernels -> three-product kernel): 256 captions = 19 712 rows 1027 -> 341 us; 32 captions = 2 464 rows 192 -> 128 us; 24 captions
// 135 -> 125; 16 captions 117 -> 125 (+ the split pass behind the causal attention): below ~2000 rows a launch is ONE tile's latency
// (c_proj: 32 K-steps = 66 us whatever the row count), which the fp32 path's 64-row tiles undercut.
constexpr long kTextX2MinRows = 2048;
bool text_x2_call(const fc_handle* h, int n) {
  return h->x2_text() && x2_pass_ok(n, h->cfg.transformer_width) && (long)n * h->cfg.context_length >= kTextX2MinRows;
}

int run_blocks_x2(fc_handle* h, const Tower& t, const Scratch& s, int n_seq, int S, int w, int heads, int causal, const float* fin_w,
                  const float* fin_b, const int* pool_idx, long pool_step, hipStream_t st, const TowerEntry* entry) {
  const int M = n_seq * S;
  const long ld2 = x2_row_elems(w);
  const size_t nl = t.blocks.size();
  for (size_t l = 0; l < nl; ++l) {
    const Block& b = t.blocks[l];
    if (l == 0 && entry) {
      FC_TRY(launch_layernorm_pair(s.x, entry->cls, entry->pos0, S, entry->pre_w, entry->pre_b, b.ln1_w, b.ln1_b, s.xn,
                                   KIND_X2, M, w, st));
    } else if (l == 0) {
      FC_TRY(launch_layernorm(s.x, w, nullptr, b.ln1_w, b.ln1_b, s.xn, ld2, KIND_X2, M, w, st));
    } else {
      ProfScope ps(h, st, 2, M, w, 1);
      FC_TRY(launch_layernorm(s.x, w, nullptr, b.ln1_w, b.ln1_b, s.xn, ld2, KIND_X2, M, w, st));
    }
    FC_TRY(gemm_x2(h, EPI_BIAS_F32, s.xn, b.in_w2, b.in_s2, b.in_b, s.big, M, 3 * w, w, 3 * w, st));
    {
      ProfScope ps(h, st, 1, n_seq, heads, S);
      if (attention_split2_supported(S, causal)) {
        ps.form(ATTN_SPLIT2);
        FC_TRY(launch_attention_split2(s.big, s.xn, n_seq, S, heads, st, h->sat_flag));
      } else {  // other sequence lengths, causal attention: the fp32 kernel of that form, then the split as a pass of its own
        ps.form(PREC_F32, 1);
        FC_TRY(launch_attention(PREC_F32, s.big, s.d, n_seq, S, heads, causal, st));
        FC_TRY(launch_split2_rows(s.d, w, s.xn, ld2, M, w, h->sat_flag, st));
      }
    }
    FC_TRY(gemm_x2(h, EPI_RESID3_F32, s.xn, b.out_w2, b.out_s2, b.out_b, s.x, M, w, w, w, st));
    {
      ProfScope ps(h, st, 2, M, w, 1);
      FC_TRY(launch_layernorm(s.x, w, nullptr, b.ln2_w, b.ln2_b, s.xn, ld2, KIND_X2, M, w, st));
    }
    FC_TRY(gemm_x2(h, EPI_GELU_X2, s.xn, b.fc_w2, b.fc_s2, b.fc_b, s.big, M, 4 * w, w, (int)x2_row_elems(4 * w), st));
    FC_TRY(gemm_x2(h, EPI_RESID3_F32, s.big, b.proj_w2, b.proj_s2, b.proj_b, s.x, M, w, 4 * w, w, st));
  }
  return launch_layernorm(s.x, pool_idx ? w : pool_step * w, pool_idx, fin_w, fin_b, s.clsn, w, PREC_F32, n_seq, w, st);
}

// Items per pass of a tower over `n` items: what bounds a pass is only the workspace it needs (3.6 MB per ViT-B/16 frame in
// fp32; the GEMMs address their activation rows from 64-bit tile bases).  bf16 and fp32: up to 2048 frames - the measured
// size (the bench step in ONE pass: bf16 3003 / 3023 / 3066 pairs/s with passes of 512 / 1024 / 2048 frames; fp32 464 -> 474
// against 1663 + 385).  fp32 needs no "whole round" pass sizes any more (rounds 2-3 planned them): the GEMM launch cuts its
// own rows into a head of whole tile rounds and a tail of lower tiles (gemm.hip: plan_tail).
// Split-fp32 (bf16-pipe GEMMs): 768 frames - measured on the 2048-frame bench step (pairs/s): passes of 256 frames 594, 384:
// 613, 512: 624, 600: 626, 700: 632, 768: 633-635, 850: 626, 886: 620, 665 (whole rounds): 607; 768: 704, 1024: 691, 2048: 697.
int planned_chunk(const fc_handle* h, int tower, int n) {
  (void)n;
  const fc_config& c = h->cfg;
  if (tower == 1) return c.chunk_texts > 0 ? c.chunk_texts : 1024;
  if (c.chunk_frames > 0) return c.chunk_frames;
  // split2 (three fp16 products): round 5's 32x32x16 kernel ran 768 / 1024 / 2048 alike; on the 16x16x32 kernel of round 6 the bench
  // step takes 201.3 / 199.6 / 199.5 / 198.7 ms in passes of 512 / 768 / 1024 / 2048 frames (tools/x3_e2e.py, profiles/r06_x3_e2e.log)
  return h->split2() ? 2048 : h->split() ? 768 : 2048;
}

size_t per_item_bytes(const fc_handle* h, int tower) {
  const fc_config& c = h->cfg;
  if (tower == 0) return carve(nullptr, 1, h->vtokens(), c.vision_width, h->esz, h->patch_kp(), h->cfg.precision == FC_PREC_F32 ? h->cfg.split_gemm : 0).total;
  return carve(nullptr, 1, c.context_length, c.transformer_width, h->esz, 0, h->x2_text() ? 2 : 0).total;
}

}  // namespace
}  // namespace fc

using namespace fc;

extern "C" {

const char* fc_last_error(void) { return g_error.c_str(); }
#define FC_STR2(x) #x
#define FC_STR(x) FC_STR2(x)
const char* fc_version(void) { return "fitclip_hip 0.3 (gfx950) abi " FC_STR(FC_ABI_VERSION); }

int fc_create(const fc_config* cfg, fc_handle** out) {
  if (!cfg || !out) return fail(FC_EINVAL, "fc_create: null argument");
  // the first four bytes say how large the CALLER believes the struct is: nothing else is read before they match
  if (cfg->struct_size != (int32_t)sizeof(fc_config))
    return fail(FC_EINVAL,
                "fc_create: fc_config.struct_size is %d but this library's fc_config has %zu bytes (ABI %d): the binding "
                "was written against another revision of include/fitclip_hip.h",
                cfg->struct_size, sizeof(fc_config), FC_ABI_VERSION);
  const fc_config& c = *cfg;
  if (c.precision != FC_PREC_F32 && c.precision != FC_PREC_BF16) return fail(FC_EINVAL, "fc_create: precision");
  if (c.split_gemm && (c.precision != FC_PREC_F32 || c.vision_width % 256 || c.prune_last_block || c.split_gemm < 0 || c.split_gemm > 2))
    return fail(FC_EINVAL, "fc_create: split_gemm (1 or 2) needs the fp32 precision, a vision width that is a multiple of 256 and prune_last_block = 0");
  if (c.vision_width % 64 || c.transformer_width % 64 || c.vision_width <= 0 || c.transformer_width <= 0)
    return fail(FC_EINVAL, "fc_create: widths must be positive multiples of 64 (head dim 64)");
  if (c.transformer_heads * 64 != c.transformer_width)
    return fail(FC_EINVAL, "fc_create: transformer_heads * 64 must equal transformer_width");
  if (c.vision_patch_size <= 0 || c.image_resolution <= 0 || c.image_resolution % c.vision_patch_size)
    return fail(FC_EINVAL, "fc_create: resolution %d / patch %d", c.image_resolution, c.vision_patch_size);
  if (c.embed_dim % 4 || c.embed_dim <= 0 || c.vision_layers <= 0 || c.transformer_layers <= 0 ||
      c.context_length <= 0 || c.vocab_size <= 0)
    return fail(FC_EINVAL, "fc_create: bad dimension");
  auto h = std::make_unique<fc_handle>();
  h->cfg = c;
  h->esz = c.precision == FC_PREC_BF16 ? 2 : 4;
  build_names(h.get());
  if (c.precision == FC_PREC_BF16 && c.context_length > 224)
    return fail(FC_EINVAL, "fc_create: text contexts longer than 224 tokens are not supported in bf16 mode");
  *out = h.release();
  return FC_OK;
}

void fc_destroy(fc_handle* h) {
  if (!h) return;
  for (auto e : h->ev) (void)hipEventDestroy(e);
  if (h->sat_host) (void)hipHostFree(h->sat_host);
  delete h;
}

int fc_num_weights(const fc_handle* h) { return h ? (int)h->names.size() : 0; }
const char* fc_weight_name(const fc_handle* h, int32_t i) {
  return (h && i >= 0 && i < (int)h->names.size()) ? h->names[i].c_str() : nullptr;
}

int fc_set_weight(fc_handle* h, const char* name, const float* dev, const int64_t* shape, int32_t ndim) {
  if (!h || !name) return fail(FC_EINVAL, "fc_set_weight: null argument");
  const std::string n(name);
  if (n == "logit_scale" || n == "input_resolution" || n == "context_length" || n == "vocab_size") return FC_OK;
  auto it = h->slots.find(n);
  if (it == h->slots.end()) return fail(FC_EINVAL, "fc_set_weight: unexpected key \"%s\"", name);
  const auto& want = it->second.shape;
  bool ok = (int)want.size() == ndim;
  for (int i = 0; ok && i < ndim; ++i) ok = want[i] == shape[i];
  if (!ok) return fail(FC_EINVAL, "fc_set_weight: shape mismatch for \"%s\"", name);
  if (!dev || ((uintptr_t)dev & 15)) return fail(FC_EINVAL, "fc_set_weight: \"%s\" must be a 16-byte aligned device pointer", name);
  it->second.ptr = dev;
  h->packed = false;
  h->train_ready = false;
  return FC_OK;
}

size_t fc_packed_bytes(const fc_handle* h) {
  if (!h) return 0;
  size_t total = 0;
  for (auto& e : packed_list(h)) total += packed_item_bytes(h, e);
  if (h->split2()) total += 256;  // the range flag of the x2 writers
  if (h->x2_patch()) total += align_up((size_t)h->cfg.vision_width * 4);  // the zero bias of the patch-embedding GEMM (behind the flag)
  return total;
}

int fc_pack_weights(fc_handle* h, void* arena, size_t bytes, fc_stream stream) {
  if (!h) return fail(FC_EINVAL, "fc_pack_weights: null handle");
  for (auto& n : h->names)
    if (!h->slots[n].ptr) return fail(FC_ESTATE, "fc_pack_weights: missing key \"%s\"", n.c_str());
  if (bytes < fc_packed_bytes(h) || !arena || ((uintptr_t)arena & 255))
    return fail(FC_ENOMEM, "fc_pack_weights: arena needs %zu bytes, 256-byte aligned", fc_packed_bytes(h));
  const int kind = h->cfg.precision;
  std::map<std::string, const void*> packed;
  size_t off = 0;
  // split_gemm 2: the range flag (a device int behind the packed items, with a pinned host mirror) starts clear BEFORE the weights
  // are split - a tensor with an infinite or NaN weight raises it
  int* flag = nullptr;
  if (h->split2()) {
    const size_t tail = 256 + (h->x2_patch() ? align_up((size_t)h->cfg.vision_width * 4) : 0);   // [flag | zero bias]
    flag = reinterpret_cast<int*>(static_cast<char*>(arena) + (fc_packed_bytes(h) - tail));
    h->zero_bias = nullptr;
    if (h->x2_patch()) {
      float* zb = reinterpret_cast<float*>(reinterpret_cast<char*>(flag) + 256);
      if (hipMemsetAsync(zb, 0, (size_t)h->cfg.vision_width * 4, stream) != hipSuccess) return fail(FC_ELAUNCH, "fc_pack_weights: memset");
      h->zero_bias = zb;
    }
    if (!h->sat_host && hipHostMalloc(reinterpret_cast<void**>(&h->sat_host), 64, hipHostMallocDefault) != hipSuccess)
      return fail(FC_ENOMEM, "fc_pack_weights: cannot allocate the pinned mirror of the range flag");
    *h->sat_host = 0;
    if (hipMemsetAsync(flag, 0, 256, stream) != hipSuccess) return fail(FC_ELAUNCH, "fc_pack_weights: memset");
  }
  for (auto& e : packed_list(h)) {
    const auto& slot = h->slots.at(e.name);
    void* dst = static_cast<char*>(arena) + off;
    if (e.mode == PACK_TRANSPOSE) {
      FC_TRY(launch_transpose_convert(slot.ptr, dst, kind, (int)slot.shape[0], (int)slot.shape[1], stream));
    } else if (e.mode == PACK_PAD_ROWS) {
      FC_TRY(launch_convert_rows(slot.ptr, dst, kind, (long)slot.shape[0], h->patch_k(), h->patch_kp(), stream));
    } else if (e.mode == PACK_SPLIT3) {
      FC_TRY(launch_split3_rows(slot.ptr, (long)slot.shape[1], dst, x3_row_elems(slot.shape[1]), (long)slot.shape[0],
                                (int)slot.shape[1], stream));
      packed[e.name + "#x3"] = dst;
      off += packed_item_bytes(h, e);
      continue;
    } else if (e.mode == PACK_SPLIT2) {  // [the x2 image | 256 bytes: the {s, 1 / s} pair]
      const size_t img = packed_item_bytes(h, e) - 256;
      float* sc = reinterpret_cast<float*>(static_cast<char*>(dst) + img);
      const long wk = weight_cols(slot.shape);
      FC_TRY(launch_split2_weight(slot.ptr, wk, dst, x2_row_elems(wk), (long)slot.shape[0], (int)wk, sc, flag, stream));
      packed[e.name + "#x2"] = dst;
      packed[e.name + "#s2"] = sc;
      off += packed_item_bytes(h, e);
      continue;
    } else {
      FC_TRY(launch_convert(slot.ptr, dst, kind, numel(slot.shape), stream));
    }
    packed[e.name] = dst;
    off += packed_item_bytes(h, e);
  }
  auto gw = [&](const std::string& n) -> const void* {
    auto it = packed.find(n);
    return it != packed.end() ? it->second : static_cast<const void*>(h->w(n));
  };
  auto fill = [&](Tower& t, const std::string& prefix, int layers) {
    t.blocks.assign(layers, Block{});
    for (int i = 0; i < layers; ++i) {
      const std::string b = prefix + ".resblocks." + std::to_string(i);
      Block& k = t.blocks[i];
      k.ln1_w = h->w(b + ".ln_1.weight"); k.ln1_b = h->w(b + ".ln_1.bias");
      k.ln2_w = h->w(b + ".ln_2.weight"); k.ln2_b = h->w(b + ".ln_2.bias");
      k.in_b = h->w(b + ".attn.in_proj_bias"); k.out_b = h->w(b + ".attn.out_proj.bias");
      k.fc_b = h->w(b + ".mlp.c_fc.bias"); k.proj_b = h->w(b + ".mlp.c_proj.bias");
      k.in_w = gw(b + ".attn.in_proj_weight"); k.out_w = gw(b + ".attn.out_proj.weight");
      k.fc_w = gw(b + ".mlp.c_fc.weight"); k.proj_w = gw(b + ".mlp.c_proj.weight");
      auto x3 = [&](const char* n) -> const void* {
        auto it = packed.find(b + n + "#x3");
        return it != packed.end() ? it->second : nullptr;
      };
      k.in_w3 = x3(".attn.in_proj_weight"); k.out_w3 = x3(".attn.out_proj.weight");
      k.fc_w3 = x3(".mlp.c_fc.weight"); k.proj_w3 = x3(".mlp.c_proj.weight");
      auto x2 = [&](const char* n, const char* tag) -> const void* {
        auto it = packed.find(b + n + tag);
        return it != packed.end() ? it->second : nullptr;
      };
      k.in_w2 = x2(".attn.in_proj_weight", "#x2"); k.out_w2 = x2(".attn.out_proj.weight", "#x2");
      k.fc_w2 = x2(".mlp.c_fc.weight", "#x2"); k.proj_w2 = x2(".mlp.c_proj.weight", "#x2");
      k.in_s2 = static_cast<const float*>(x2(".attn.in_proj_weight", "#s2")); k.out_s2 = static_cast<const float*>(x2(".attn.out_proj.weight", "#s2"));
      k.fc_s2 = static_cast<const float*>(x2(".mlp.c_fc.weight", "#s2")); k.proj_s2 = static_cast<const float*>(x2(".mlp.c_proj.weight", "#s2"));
    }
  };
  fill(h->vis, "visual.transformer", h->cfg.vision_layers);
  fill(h->txt, "transformer", h->cfg.transformer_layers);
  if (h->split2()) {
    // LayerNorm outputs are bounded by sqrt(D) max|gamma| + max|beta|: a tower whose LayerNorm weights could leave fp16's range
    // (or are not finite) raises the flag here, once
    h->sat_flag = flag;
    for (const Block& k : h->vis.blocks) {
      FC_TRY(launch_x2_ln_bound(k.ln1_w, k.ln1_b, h->cfg.vision_width, h->sat_flag, stream));
      FC_TRY(launch_x2_ln_bound(k.ln2_w, k.ln2_b, h->cfg.vision_width, h->sat_flag, stream));
    }
    if (h->x2_text())
      for (const Block& k : h->txt.blocks) {
        FC_TRY(launch_x2_ln_bound(k.ln1_w, k.ln1_b, h->cfg.transformer_width, h->sat_flag, stream));
        FC_TRY(launch_x2_ln_bound(k.ln2_w, k.ln2_b, h->cfg.transformer_width, h->sat_flag, stream));
      }
  } else {
    h->sat_flag = nullptr;
  }
  h->conv_w = gw("visual.conv1.weight");
  h->conv_w2 = nullptr;
  h->conv_s2 = nullptr;
  if (h->x2_patch()) {
    h->conv_w2 = packed.at("visual.conv1.weight#x2");
    h->conv_s2 = static_cast<const float*>(packed.at("visual.conv1.weight#s2"));
    h->conv_w = h->w("visual.conv1.weight");   // (the fp32 tensor: passes too small for the plane GEMM take the fp32 path)
  }
  h->vproj_t = gw("visual.proj");
  h->tproj_t = gw("text_projection");
  h->packed = true;
  h->train_ready = false;  // the transposed copies of fc_train_prepare belong to the previous weights
  return FC_OK;
}

size_t fc_workspace_bytes(const fc_handle* h, int32_t tower, int32_t n) {
  if (!h || n <= 0 || tower < 0 || tower > 1) return 0;
  const int c = std::min(n, planned_chunk(h, tower, n));
  const fc_config& k = h->cfg;
  return tower == 0 ? carve(nullptr, c, h->vtokens(), k.vision_width, h->esz, h->patch_kp(), h->cfg.precision == FC_PREC_F32 ? h->cfg.split_gemm : 0).total
                    : carve(nullptr, c, k.context_length, k.transformer_width, h->esz, 0, h->x2_text() ? 2 : 0).total;
}

int fc_encode_image(fc_handle* h, const float* frames, int32_t n, float* out, void* ws, size_t ws_bytes,
                    fc_stream st) {
  if (!h) return fail(FC_EINVAL, "fc_encode_image: null handle");
  if (!h->packed) return fail(FC_ESTATE, "fc_encode_image: call fc_pack_weights first");
  if (n == 0) return FC_OK;
  if (h->sat_host && *h->sat_host)
    return fail(FC_ERANGE, "fc_encode_image: an earlier call met an activation (or LayerNorm weights) beyond fp16's range (65504): "
                           "split_gemm = 2 cannot represent this model's values; its results since then are not valid");
  if (n < 0 || !frames || !out || !ws) return fail(FC_EINVAL, "fc_encode_image: bad argument");
  if (((uintptr_t)frames | (uintptr_t)out | (uintptr_t)ws) & 15) return fail(FC_EINVAL, "fc_encode_image: unaligned pointer");
  const fc_config& c = h->cfg;
  const int vw = c.vision_width, T = h->vtokens(), P = h->patches(), R = c.image_resolution, Kp = h->patch_kp();
  const size_t per = per_item_bytes(h, 0);
  int chunk = std::min(n, planned_chunk(h, 0, n));
  const int split = h->split() ? h->cfg.split_gemm : 0;
  if (carve(nullptr, chunk, T, vw, h->esz, Kp, split).total > ws_bytes) {  // smaller workspace: as many items as fit
    chunk = (int)std::min<size_t>(chunk, ws_bytes / std::max<size_t>(1, per / 2));
    while (chunk > 0 && carve(nullptr, chunk, T, vw, h->esz, Kp, split).total > ws_bytes) --chunk;
  }
  if (chunk <= 0) return fail(FC_ENOMEM, "fc_encode_image: workspace too small (need >= %zu bytes)", per + 2048);
  const int kind = c.precision;
  for (int off = 0; off < n; off += chunk) {
    const int cn = std::min(chunk, n - off);
    const Scratch s = carve(static_cast<char*>(ws), cn, T, vw, h->esz, Kp, split);
    const float* f = frames + (size_t)off * 3 * R * R;
    const int p = c.vision_patch_size;
    if (split == 2 && h->conv_w2 && x2_pass_ok(cn * T, vw) && cn * P < (1 << 23)) {
      // three-product mode: the patch embedding is a plane GEMM too - the frames become x2 rows of the im2col matrix in one pass
      // (in the MLP-hidden buffer, free until the first block), the GEMM's epilogue adds the positional embedding and leaves the
      // class rows out.  3.78 -> 1.7 ms per 2048 frames (fp32-MFMA GEMM with the patch gather in its loader: 0.87 of ITS peak)
      FC_TRY(launch_im2col_x2(f, s.big, x2_row_elems(Kp), cn, R, p, h->sat_flag, st));
      FC_TRY(gemm_x2(h, EPI_PATCH_F32, s.big, h->conv_w2, h->conv_s2, h->zero_bias, s.x, cn * P, vw, Kp, vw, st,
                     h->w("visual.positional_embedding"), P));
    } else if (kind == PREC_F32 && Kp == h->patch_k() && p % 4 == 0 && R % 4 == 0) {
      // fp32 mode: the GEMM's LDS-DMA loader gathers the 16 x 16 x 3 patches straight from the NCHW frames (16-byte
      // pieces of 4 pixels): the frames are read once, no im2col matrix is written to / re-read from HBM
      GemmArgs a{};
      a.A = f; a.W = h->conv_w; a.C = s.x; a.aux = h->w("visual.positional_embedding"); a.alpha = 1.f;
      a.M = cn * P; a.N = vw; a.K = Kp; a.lda = Kp; a.ldw = Kp; a.ldc = vw; a.P = P; a.gR = R; a.gP = p;
      ProfScope ps(h, st, kind, EPI_PATCH_F32, gemm_resolved_tile(kind, EPI_PATCH_F32, a, h->cfg.gemm_tile), a);
      FC_TRY(launch_gemm(kind, EPI_PATCH_F32, a, h->cfg.gemm_tile, st));
    } else {  // bf16 operands (the conversion pass IS the im2col pass) or a patch size that needs K padding (14)
      FC_TRY(launch_im2col(f, s.big, kind, cn, R, p, Kp, st));
      FC_TRY(gemm(h, EPI_PATCH_F32, s.big, h->conv_w, nullptr, s.x, h->w("visual.positional_embedding"), cn * P, vw,
                  Kp, vw, P, st));
    }
    const TowerEntry entry{h->w("visual.class_embedding"), h->w("visual.positional_embedding"),
                           h->w("visual.ln_pre.weight"), h->w("visual.ln_pre.bias")};
    if (split == 2 && x2_pass_ok(cn * T, vw)) {
      FC_TRY(run_blocks_x2(h, h->vis, s, cn, T, vw, h->vheads(), 0, h->w("visual.ln_post.weight"),
                           h->w("visual.ln_post.bias"), nullptr, T, st, &entry));
    } else if (split == 1 && x3_pass_ok(cn * T, vw)) {
      FC_TRY(run_blocks_x3(h, h->vis, s, cn, T, vw, h->vheads(), h->w("visual.ln_post.weight"),
                           h->w("visual.ln_post.bias"), T, st, entry));
    } else {  // (split mode: a pass too small for the pipelined GEMM takes the plain fp32 path)
      FC_TRY(run_blocks(h, h->vis, s, cn, T, vw, h->vheads(), 0, h->w("visual.ln_post.weight"),
                        h->w("visual.ln_post.bias"), nullptr, T, st, &entry));
    }
    FC_TRY(gemm(h, EPI_STORE_F32, s.clsn, h->vproj_t, nullptr, out + (size_t)off * c.embed_dim, nullptr, cn,
                c.embed_dim, vw, c.embed_dim, 0, st));
  }
  return finish_range(h, out, (size_t)n * c.embed_dim, st, "fc_encode_image");
}

int fc_range_strict(fc_handle* h, int32_t on) {
  if (!h) return fail(FC_EINVAL, "fc_range_strict: null handle");
  h->strict_range = on != 0;
  return FC_OK;
}

int fc_range_status(fc_handle* h, fc_stream st, int32_t wait) {
  if (!h) return fail(FC_EINVAL, "fc_range_status: null handle");
  if (!h->sat_flag || !h->sat_host) return FC_OK;   // only split_gemm = 2 writes fp16 planes
  if (wait) {
    if (hipMemcpyAsync(h->sat_host, h->sat_flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return fail(FC_ELAUNCH, "fc_range_status: flag copy");
  }
  return *h->sat_host ? fail(FC_ERANGE, "fc_range_status: a value beyond fp16's range (65504) was met since the weights were packed") : FC_OK;
}

int fc_encode_text(fc_handle* h, const int64_t* ids, int32_t n, float* out, void* ws, size_t ws_bytes, fc_stream st) {
  if (!h) return fail(FC_EINVAL, "fc_encode_text: null handle");
  if (!h->packed) return fail(FC_ESTATE, "fc_encode_text: call fc_pack_weights first");
  if (n == 0) return FC_OK;
  if (n < 0 || !ids || !out || !ws) return fail(FC_EINVAL, "fc_encode_text: bad argument");
  if (((uintptr_t)out | (uintptr_t)ws) & 15) return fail(FC_EINVAL, "fc_encode_text: unaligned pointer");
  const fc_config& c = h->cfg;
  const int tw = c.transformer_width, L = c.context_length;
  // split_gemm 2: one arithmetic per CALL - a call of kTextX2MinRows token rows or more runs the four block GEMMs on the fp16 pipe
  // (all of its passes, also a short last one), a smaller call the fp32 kernels
  const int split = text_x2_call(h, n) ? 2 : 0;
  if (split && h->sat_host && *h->sat_host)
    return fail(FC_ERANGE, "fc_encode_text: an earlier call met an activation (or LayerNorm weights) beyond fp16's range (65504): "
                           "split_gemm = 2 cannot represent this model's values; its results since then are not valid");
  const int carve_split = h->x2_text() ? 2 : 0;   // the layout fc_workspace_bytes sized: the same for both kinds of call
  const size_t per = per_item_bytes(h, 1);
  int chunk = std::min(n, planned_chunk(h, 1, n));
  if (carve(nullptr, chunk, L, tw, h->esz, 0, carve_split).total > ws_bytes) {
    chunk = (int)std::min<size_t>(chunk, ws_bytes / std::max<size_t>(1, per / 2));
    while (chunk > 0 && carve(nullptr, chunk, L, tw, h->esz, 0, carve_split).total > ws_bytes) --chunk;
  }
  if (chunk <= 0) return fail(FC_ENOMEM, "fc_encode_text: workspace too small (need >= %zu bytes)", per + 2048);
  for (int off = 0; off < n; off += chunk) {
    const int cn = std::min(chunk, n - off);
    const Scratch s = carve(static_cast<char*>(ws), cn, L, tw, h->esz, 0, carve_split);
    FC_TRY(launch_text_embed(ids + (size_t)off * L, h->w("token_embedding.weight"), h->w("positional_embedding"), s.x,
                             s.eot, cn, L, tw, c.vocab_size, st));
    if (split) {
      FC_TRY(run_blocks_x2(h, h->txt, s, cn, L, tw, c.transformer_heads, 1, h->w("ln_final.weight"), h->w("ln_final.bias"),
                           s.eot, 0, st, nullptr));
    } else {
      FC_TRY(run_blocks(h, h->txt, s, cn, L, tw, c.transformer_heads, 1, h->w("ln_final.weight"),
                        h->w("ln_final.bias"), s.eot, 0, st));
    }
    FC_TRY(gemm(h, EPI_STORE_F32, s.clsn, h->tproj_t, nullptr, out + (size_t)off * c.embed_dim, nullptr, cn,
                c.embed_dim, tw, c.embed_dim, 0, st));
  }
  if (split) return finish_range(h, out, (size_t)n * c.embed_dim, st, "fc_encode_text");
  FC_DEBUG_SCAN(out, (size_t)n * c.embed_dim, st, "fc_encode_text");
  return FC_OK;
}

int fc_preprocess_u8(const uint8_t* frames, float* out, int32_t n, int32_t H, int32_t W, int32_t R, const float* mean3,
                     const float* std3, fc_stream st) {
  if (!frames || !out || !mean3 || !std3) return fail(FC_EINVAL, "fc_preprocess_u8: null argument");
  return launch_preprocess_u8(frames, out, n, H, W, R, mean3, std3, st);
}
int fc_pool_normalize(const float* e, float* out, int32_t n_clips, int32_t frames, int32_t dim, fc_stream st) {
  return launch_pool_normalize(e, out, n_clips, frames, dim, st);
}
int fc_l2_normalize(const float* in, float* out, int32_t n, int32_t dim, fc_stream st) {
  return launch_l2_normalize(in, out, n, dim, st);
}

int fc_similarity(const float* A, const float* B, int32_t na, int32_t nb, int32_t dim, float alpha, float* out,
                  int32_t ldo, fc_stream st) {
  if (na == 0 || nb == 0) return FC_OK;
  GemmArgs a{};
  a.A = A; a.W = B; a.bias = nullptr; a.C = out; a.aux = nullptr; a.alpha = alpha;
  a.M = na; a.N = nb; a.K = dim; a.lda = dim; a.ldw = dim; a.ldc = ldo; a.P = 0;
  return launch_gemm(PREC_F32, EPI_STORE_F32, a, 1, st);
}
int fc_similarity_ranks(const float* T, const float* V, int32_t nt, int32_t nv, int32_t dim, float alpha,
                        int32_t target_offset, const int32_t* targets, int32_t* ranks, fc_stream st) {
  return launch_similarity_ranks(T, V, nt, nv, dim, alpha, target_offset, targets, ranks, st);
}
int fc_ranks(const float* s, int32_t ld, int32_t n_rows, int32_t n_cols, int32_t off, int32_t* ranks, fc_stream st) {
  return launch_ranks(s, ld, n_rows, n_cols, off, nullptr, ranks, st);
}
int fc_ranks_of(const float* s, int32_t ld, int32_t n_rows, int32_t n_cols, const int32_t* targets, int32_t* ranks,
                fc_stream st) {
  if (!targets) return fail(FC_EINVAL, "fc_ranks_of: targets is null");
  return launch_ranks(s, ld, n_rows, n_cols, 0, targets, ranks, st);
}
int fc_group_mean(const float* in, float* out, int32_t n_groups, int32_t group, int32_t dim, fc_stream st) {
  return launch_group_mean(in, out, n_groups, group, dim, st);
}
int fc_nce_loss(const float* s, int32_t n, float* out, float* ws, fc_stream st) { return launch_nce_loss(s, n, out, ws, st); }
int fc_kd_loss(const float* s, const float* t, int32_t n, float* out, float* ws, fc_stream st) {
  return launch_kd_loss(s, t, n, n, out, ws, st);
}
int fc_kd_loss_rect(const float* s, const float* t, int32_t rows, int32_t cols, float* out, float* ws, fc_stream st) {
  if (!s || !t || !out || !ws) return fail(FC_EINVAL, "fc_kd_loss_rect: null argument");
  return launch_kd_loss(s, t, rows, cols, out, ws, st);
}
int fc_wise(const float* a, const float* b, double w, float* out, size_t n, fc_stream st) {
  return launch_wise(a, b, w, out, n, st);
}

int fc_gemm(int32_t precision, int32_t epilogue, const void* A, const void* W, const float* bias, void* C,
            const float* aux, float alpha, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc,
            int32_t P, int32_t tile, fc_stream st) {
  GemmArgs a{};
  a.A = A; a.W = W; a.bias = bias; a.C = C; a.aux = aux; a.alpha = alpha;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc; a.P = P;
  return launch_gemm(precision, epilogue, a, tile, st);
}
int fc_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t* head_panels, int32_t* tail_units) {
  if (M <= 0 || N <= 0 || K <= 0 || !head_panels || !tail_units) return fail(FC_EINVAL, "fc_gemm_plan: bad argument");
  int hp = 0, ht = 0;
  gemm_tail_plan(M, N, K, &hp, &ht);
  *head_panels = hp;
  *tail_units = ht;
  return FC_OK;
}
int fc_gemm_split2_plan(int32_t M, int32_t N, int32_t compute_units, int32_t* tile_rows, int32_t* workgroups) {
  if (M <= 0 || N <= 0 || !tile_rows || !workgroups) return fail(FC_EINVAL, "fc_gemm_split2_plan: bad argument");
  int rows = 0, wgs = 0;
  gemm_split2_plan(M, N, compute_units, &rows, &wgs);
  *tile_rows = rows;
  *workgroups = wgs;
  return FC_OK;
}
int fc_layernorm(const float* x, int64_t xs, const int32_t* gather, const float* g, const float* b, void* y,
                 int64_t ys, int32_t out_kind, int32_t rows, int32_t D, fc_stream st) {
  return launch_layernorm(x, (long)xs, gather, g, b, y, (long)ys, out_kind, rows, D, st);
}
int fc_add_layernorm(float* x, int64_t xs, const void* delta, int64_t ds, const int32_t* gather, const float* g,
                     const float* b, void* y, int64_t ys, int32_t kind, int32_t rows, int32_t D, int32_t write_x,
                     fc_stream st) {
  return launch_add_layernorm(x, (long)xs, delta, (long)ds, gather, g, b, y, (long)ys, kind, rows, D, write_x, 0, st);
}
int fc_attention(int32_t precision, const void* qkv, void* out, int32_t n_seq, int32_t S, int32_t heads,
                 int32_t causal, fc_stream st) {
  if (precision == KIND_X3) {  // fp32 in, x3 rows out (the sequence lengths the streaming-block kernel serves)
    if (causal) return fail(FC_EINVAL, "fc_attention: the three-plane output is not available for causal attention");
    return launch_attention_x3(qkv, out, n_seq, S, heads, st);
  }
  if (precision == ATTN_SPLIT) {  // fp32 in, x3 rows out, six bf16 products per fp32 product (193..208 tokens)
    if (causal) return fail(FC_EINVAL, "fc_attention: split-fp32 attention is not available for causal attention");
    return launch_attention_split(qkv, out, n_seq, S, heads, st);
  }
  if (precision == ATTN_SPLIT_X2) {  // the same kernel, x2 rows out (two fp16 planes: the operand format of fc_gemm_split2)
    if (causal) return fail(FC_EINVAL, "fc_attention: split-fp32 attention is not available for causal attention");
    return launch_attention_split(qkv, out, n_seq, S, heads, st, KIND_X2, nullptr);
  }
  if (precision == ATTN_SPLIT2) {  // three fp16 products per fp32 product (attention_split2.hip), x2 rows out
    if (causal) return fail(FC_EINVAL, "fc_attention: split-fp32 attention is not available for causal attention");
    return launch_attention_split2(qkv, out, n_seq, S, heads, st, nullptr);
  }
  return launch_attention(precision, qkv, out, n_seq, S, heads, causal, st);
}
int fc_convert(const float* in, void* out, int32_t out_kind, size_t n, fc_stream st) {
  return launch_convert(in, out, out_kind, n, st);
}
int fc_split3(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int32_t K, fc_stream st) {
  if (!in || !out) return fail(FC_EINVAL, "fc_split3: null operand");
  return launch_split3_rows(in, (long)ld_in, out, (long)ld_out, (long)rows, K, st);
}
int fc_gemm_split3(int32_t epilogue, const void* A3, const void* W3, const float* bias, void* C, int32_t M, int32_t N,
                   int32_t K, int32_t lda, int32_t ldw, int32_t ldc, fc_stream st) {
  GemmArgs a{};
  a.A = A3; a.W = W3; a.bias = bias; a.C = C; a.aux = nullptr; a.alpha = 1.f;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc;
  if (!A3 || !W3 || !C) return fail(FC_EINVAL, "fc_gemm_split3: null operand");
  return launch_gemm_split3(epilogue, a, st);
}

int fc_split2(const float* in, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int32_t K, int32_t* sat_flag, fc_stream st) {
  if (!in || !out) return fail(FC_EINVAL, "fc_split2: null operand");
  return launch_split2_rows(in, (long)ld_in, out, (long)ld_out, (long)rows, K, sat_flag, st);
}
int fc_split2_weight(const float* w, int64_t ld_in, void* out, int64_t ld_out, int64_t rows, int32_t K, float* scale2, int32_t* sat_flag,
                     fc_stream st) {
  if (!w || !out || !scale2) return fail(FC_EINVAL, "fc_split2_weight: null operand");
  return launch_split2_weight(w, (long)ld_in, out, (long)ld_out, (long)rows, K, scale2, sat_flag, st);
}
int fc_gemm_split2(int32_t epilogue, const void* A2, const void* W2, const float* scale2, const float* bias, void* C, int32_t M,
                   int32_t N, int32_t K, int32_t lda, int32_t ldw, int32_t ldc, int32_t* sat_flag, int32_t cut, fc_stream st) {
  GemmArgs a{};
  a.A = A2; a.W = W2; a.bias = bias; a.C = C; a.aux = nullptr; a.alpha = 1.f; a.wscale = scale2; a.sat_flag = sat_flag;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldw = ldw; a.ldc = ldc;
  if (!A2 || !W2 || !C || !scale2) return fail(FC_EINVAL, "fc_gemm_split2: null operand");
  if (cut < 0 || cut > 3) return fail(FC_EINVAL, "fc_gemm_split2: cut %d", cut);
  return launch_gemm_split2(epilogue, a, st, cut);
}

int fc_profile_enable(fc_handle* h, int32_t max_records) {
  if (!h || max_records < 0) return fail(FC_EINVAL, "fc_profile_enable: bad argument");
  for (auto e : h->ev) (void)hipEventDestroy(e);
  h->ev.clear();
  h->recs.clear();
  h->prof_cap = 0;
  for (int i = 0; i < 2 * max_records; ++i) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return fail(FC_ELAUNCH, "fc_profile_enable: hipEventCreate failed");
    h->ev.push_back(e);
  }
  h->recs.reserve(max_records);
  h->prof_cap = max_records;
  return FC_OK;
}
int fc_profile_select(fc_handle* h, uint32_t kind_mask, uint32_t epilogue_mask) {
  if (!h) return fail(FC_EINVAL, "fc_profile_select: null handle");
  h->prof_kinds = kind_mask;
  h->prof_epis = epilogue_mask;
  return FC_OK;
}
int fc_profile_reset(fc_handle* h) {
  if (!h) return fail(FC_EINVAL, "fc_profile_reset: null handle");
  h->recs.clear();
  return FC_OK;
}
int fc_profile_read(fc_handle* h, fc_prof_record* out, int32_t max_records) {
  if (!h || !out) return fail(FC_EINVAL, "fc_profile_read: null argument");
  const int n = std::min<int>((int)h->recs.size(), max_records);
  for (int i = 0; i < n; ++i) {
    float ms = -1.f;
    if (hipEventElapsedTime(&ms, h->ev[2 * i], h->ev[2 * i + 1]) != hipSuccess) ms = -1.f;
    h->recs[i].ms = ms;
    out[i] = h->recs[i];
  }
  return n;
}

}  // extern "C"

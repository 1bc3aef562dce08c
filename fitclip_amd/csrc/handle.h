// Internal: the handle behind the C ABI (shared by api.hip: inference, and train.hip: the KD training step).
#pragma once
#include "../../include/fitclip_hip.h"
#include "common.h"

#include <map>
#include <string>
#include <vector>

namespace fc {

struct WeightSlot {
  std::vector<int64_t> shape;
  const float* ptr = nullptr;
  float* grad = nullptr;  // fc_set_grad (training only)
};

struct Block {
  const float *ln1_w, *ln1_b, *in_b, *out_b, *ln2_w, *ln2_b, *fc_b, *proj_b;
  const void *in_w, *out_w, *fc_w, *proj_w;  // element type T of the handle's precision, [N, K]
  // split_gemm: x3 rows [N, 4 K bf16 positions] of the four weights (visual tower only; common.h)
  const void *in_w3 = nullptr, *out_w3 = nullptr, *fc_w3 = nullptr, *proj_w3 = nullptr;
  // split_gemm 2: x2 rows [N, 2 K fp16 positions] of the four weights and their {s, 1 / s} scale pairs (device floats)
  const void *in_w2 = nullptr, *out_w2 = nullptr, *fc_w2 = nullptr, *proj_w2 = nullptr;
  const float *in_s2 = nullptr, *out_s2 = nullptr, *fc_s2 = nullptr, *proj_s2 = nullptr;
  // training: transposed copies [K, N] for the dgrad GEMMs (fc_train_prepare)
  const void *in_wT = nullptr, *out_wT = nullptr, *fc_wT = nullptr, *proj_wT = nullptr;
};

struct Tower {
  std::vector<Block> blocks;
};

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

}  // namespace fc

struct fc_handle {
  fc_config cfg{};
  int esz = 4;  // bytes per activation / GEMM-weight element
  std::vector<std::string> names;
  std::map<std::string, fc::WeightSlot> slots;
  bool packed = false;
  fc::Tower vis, txt;
  const void *conv_w = nullptr, *vproj_t = nullptr, *tproj_t = nullptr;
  // split_gemm 2 with a patch size the im2col -> x2 pass serves: the x2 image of visual.conv1.weight as [width, 3 p^2], its scale
  // pair and a zero bias vector (the convolution has none; the GEMM kernel starts its accumulators from one)
  const void* conv_w2 = nullptr;
  const float *conv_s2 = nullptr, *zero_bias = nullptr;
  bool x2_patch() const {
    return split2() && patch_kp() == patch_k() && cfg.vision_patch_size % 8 == 0 && cfg.image_resolution % 4 == 0 && patch_k() % 64 == 0 &&
           patch_k() >= 128;
  }
  // split_gemm 2 also serves the text tower's four block GEMMs (x2 images of its weights are packed) when its width has the
  // kernels' granularity; which CALLS use them is fc_encode_text's decision (api.hip: kTextX2MinRows)
  bool x2_text() const { return split2() && cfg.transformer_width % 256 == 0 && cfg.transformer_layers > 0; }
  // split_gemm 2: the range flag of the x2 writers - a device int in the packed-weights arena and its pinned host mirror
  // (copied behind every tower call that wrote fp16 planes; allocated by fc_pack_weights, freed by fc_destroy)
  int* sat_flag = nullptr;
  int* sat_host = nullptr;
  bool strict_range = false;  // fc_range_strict: fc_encode_image waits for its own flag copy and returns FC_ERANGE itself
  // training
  bool train_ready = false;
  const float* zeros = nullptr;  // >= 16 KiB of zeros in the training weight arena (null bias / TN tail rows)
  // profiling
  std::vector<hipEvent_t> ev;
  std::vector<fc_prof_record> recs;
  int prof_cap = 0;
  unsigned prof_kinds = ~0u, prof_epis = ~0u;  // fc_profile_select masks

  int vheads() const { return cfg.vision_width / 64; }
  int grid() const { return cfg.image_resolution / cfg.vision_patch_size; }
  int patches() const { return grid() * grid(); }
  int vtokens() const { return patches() + 1; }
  int patch_k() const { return 3 * cfg.vision_patch_size * cfg.vision_patch_size; }
  // the patch-embed GEMM's K: 3 p^2 padded to the kernel's K-tile (64 bf16 / 32 f32 elements); ViT-L/14: 588 -> 640
  int patch_kp() const {
    const int gran = cfg.precision == FC_PREC_BF16 ? 64 : 32;
    return (patch_k() + gran - 1) / gran * gran;
  }
  bool split() const { return cfg.precision == FC_PREC_F32 && cfg.split_gemm != 0; }   // either split-fp32 mode
  bool split2() const { return cfg.precision == FC_PREC_F32 && cfg.split_gemm == 2; }  // two fp16 planes, three products
  const float* w(const std::string& n) const { return slots.at(n).ptr; }
  float* grad(const std::string& n) const { return slots.at(n).grad; }
};

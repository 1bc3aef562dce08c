// Lab: attn_f32_blocks_kernel as a PERSISTENT workgroup (tools/attn_lab_f32.hip).  Two workgroups per CU walk the
// (sequence, head) items round robin; the stream of 64-key blocks does not stop at an item boundary: while the last block of
// item i is multiplied, block 0 of item i + G is on its way (LDS-DMA) and its Q fragments are loaded into the registers the
// last S products have just released.  Same operations in the same order per query tile as the one-shot kernel: bitwise equal.
// ROT: the wave that carries query tiles (w, 8 + w) moves by ROT per item (0 = fixed roles).
#pragma once

namespace fc {
namespace {

template <int NW, int ROT = 0, bool IL = false>
__global__ void __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4)))
attn_f32_persist_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S, int heads, int n_items, int delay) {
  constexpr int BT = 4, BK = BT * 16;
  constexpr int OFF_V = BK * 256, VPIECE = 1024 + 64, BUF = OFF_V + (BK / 4) * VPIECE;
  constexpr int NPIECE = (2 * BK / 4 + NW - 1) / NW;
  constexpr int QPW = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int D = heads * 64;
  const long ld = 3L * D;
  const int r = lane & 15, g = lane >> 4;
  const int nqt = (S + 15) >> 4;
  const int nblk = (S + BK - 1) / BK;
  const int G = gridDim.x;
  int item = blockIdx.x;
  if (item >= n_items) return;
  if (delay > 0 && 2 * (int)blockIdx.x >= G)     // lab: the second workgroup of a CU starts `delay` x 8 k cycles late
    for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(127);
  if (delay < 0) {                  // lab: every workgroup starts at its own offset in [0, 16) x -delay x 64 cycles
    const int n = (int)((blockIdx.x * 2654435761u) >> 28) * -delay;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(1);
  }

  auto item_base = [&](int it) {
    const int seq = it / heads, h = it - seq * heads;
    return qkv + (long)seq * S * ld + h * 64;
  };
  // (addresses: a uniform item pointer + a 32-bit lane offset, rebuilt from an opaque copy of the lane id at every use - hipcc
  // would otherwise keep every piece's 64-bit address alive across the whole item loop)
  auto stage = [&](const float* base, int blk, int buf) {
    int lane_s = lane;
    asm volatile("" : "+v"(lane_s));
    const int prow = lane_s >> 4, pch = lane_s & 15;
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int p = wave + j * NW;
      if (p < 2 * BK / 4) {
        const int isv = p >= BK / 4, piece = p - (isv ? BK / 4 : 0);
        const int row = piece * 4 + prow;
        const int srow = min(blk * BK + row, S - 1);
        const unsigned off = (unsigned)srow * (unsigned)(ld * 4) + (unsigned)(((isv ? pch : pch ^ (row & 15)) << 4) + (isv ? 2 * D : D) * 4);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(base) + off),
                                         (__attribute__((address_space(3))) void*)(smem + buf * BUF + (isv ? OFF_V + piece * VPIECE : piece * 1024)),
                                         16, 0, 0);
      }
    }
  };
  auto load_q = [&](const float* base, int role, int qi, f32x4 (&dst)[4]) {
    int lane_q = lane;
    asm volatile("" : "+v"(lane_q));
    const unsigned off = (unsigned)min((qi * NW + role) * 16 + (lane_q & 15), S - 1) * (unsigned)(ld * 4) + (unsigned)((lane_q >> 4) << 4);
    const char* qrow = reinterpret_cast<const char*>(base) + off;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = *reinterpret_cast<const f32x4*>(qrow + 64 * c);
  };

  int koff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) koff[c] = r * 256 + (((4 * c + g) ^ r) << 4);
  const int voff = OFF_V + g * VPIECE + r * 4;

  const float* base = item_base(item);
  stage(base, 0, 0);
  int role = wave;                     // query tiles role, NW + role
  f32x4 qf[QPW][4];
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi) load_q(base, role, qi, qf[qi]);   // (scaled at the item's first block)
  int gb = 0;            // blocks staged so far by this workgroup: block gb lives in buffer gb & 1
  // The output of an item is stored at the START of the next one (after its first barrier): nothing ever waits for a store that
  // has just been issued - every wait of the kernel is a plain vmcnt(0) one block after the youngest operation.
  f32x4 o[QPW][4];
  float mrun[QPW], lrun[QPW];
  int prole = wave;      // o holds the normalised output of item `pitem`, computed under role prole
  int pitem = item;
#pragma unroll
  for (int qi = 0; qi < QPW; ++qi)
#pragma unroll
    for (int n = 0; n < 4; ++n) o[qi][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto store_out = [&]() {
    const int seq = pitem / heads, h = pitem - seq * heads;
    char* obase = reinterpret_cast<char*>(out + (long)seq * S * D + h * 64);
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      const int qt = qi * NW + prole, query = qt * 16 + (lane_o & 15);
      if (qt < nqt && query < S) {
        char* orow = obase + ((unsigned)query * (unsigned)(D * 4) + (unsigned)((lane_o >> 4) << 4));
#pragma unroll
        for (int n = 0; n < 4; ++n) *reinterpret_cast<f32x4*>(orow + 64 * n) = o[qi][n];
      }
    }
  };

  while (true) {
    const int next_item = item + G;
    const bool has_next = next_item < n_items;
    const float* nbase = has_next ? item_base(next_item) : base;
    const int nrole = ROT ? (role + ROT) & (NW - 1) : role;
    const int qtile[2] = {role, NW + role};
    for (int blk = 0; blk < nblk; ++blk, ++gb) {   // ONE copy of the block body (three peeled copies: 70 KB of code, 10 % slower)
      const bool first = blk == 0, last = blk + 1 == nblk;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the block (and Q), issued a whole block ago
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();   // (raw: __syncthreads() carries a release fence = vmcnt(0), the stores included)
      asm volatile("" ::: "memory");
      // hipcc cannot know that a last block is always followed by a first one: to it the Q loads of a last block may be
      // pending in EVERY block, and its wait for them would sit in front of the S products, behind the LDS-DMA issued below
      // (vmcnt(0): the staging exposed in every block).  An empty asm that "rewrites" Q here, where nothing is in flight,
      // puts that wait where it costs nothing.
#pragma unroll
      for (int qi = 0; qi < QPW; ++qi)
#pragma unroll
        for (int c = 0; c < 4; ++c) asm volatile("" : "+v"(qf[qi][c]));
      if (first) {  // 1 / sqrt(64): exact, so it does not matter that it is applied here and not at the load
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi)
#pragma unroll
          for (int c = 0; c < 4; ++c) qf[qi][c] *= 0.125f;
        store_out();   // (the very first time: zeros to this item's own rows, overwritten in order by its result)
#pragma unroll
        for (int qi = 0; qi < QPW; ++qi) {
          mrun[qi] = kNegInf;
          lrun[qi] = 0.f;
#pragma unroll
          for (int n = 0; n < 4; ++n) o[qi][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      asm volatile("" ::: "memory");
      if (!last || has_next) stage(last ? nbase : base, last ? 0 : blk + 1, (gb + 1) & 1);
      asm volatile("" ::: "memory");   // (hipcc sinks the LDS-DMA to the end of the block otherwise: nothing consumes it)
      const char* kv = smem + (gb & 1) * BUF;
      const char* kfrag[4] = {kv + koff[0], kv + koff[1], kv + koff[2], kv + koff[3]};
      const char* vfrag = kv + voff;
#pragma unroll
      for (int qi = 0; qi < QPW; ++qi) {
        const int qt = qtile[qi];
        f32x4 sT[BT];
        if (qt < nqt) {
          if (IL && blk * BK + BK <= S) {
            // a full block: the chains of two key tiles interleaved (each accumulator still sees its products in the same order)
#pragma unroll
            for (int tp = 0; tp < BT; tp += 2) {
              f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(kfrag[c] + tp * 4096);
                const f32x4 k1 = *reinterpret_cast<const f32x4*>(kfrag[c] + (tp + 1) * 4096);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(k0[e], qf[qi][c][e], a0, 0, 0, 0);
                  a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(k1[e], qf[qi][c][e], a1, 0, 0, 0);
                }
              }
              sT[tp] = a0;
              sT[tp + 1] = a1;
            }
          } else {
#pragma unroll
          for (int t = 0; t < BT; ++t) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (blk * BK + t * 16 < S) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kfrag[c] + t * 4096);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[e], qf[qi][c][e], acc, 0, 0, 0);
              }
            }
            sT[t] = acc;
          }
          }
        }
        if (last && has_next) load_q(nbase, nrole, qi, qf[qi]);  // this slot's Q is dead: the next item's arrives under the P.V products
        if (qt < nqt) {
          float mx = mrun[qi];
#pragma unroll
          for (int t = 0; t < BT; ++t) {
            if (blk * BK + t * 16 + 16 > S) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (blk * BK + t * 16 + 4 * g + e >= S) sT[t][e] = kNegInf;
            }
            mx = fmaxf(fmaxf(mx, fmaxf(sT[t][0], sT[t][1])), fmaxf(sT[t][2], sT[t][3]));
          }
          mx = max_over_lane_groups(mx);
          const float alpha = exp_neg_f32(mrun[qi] - mx);
          mrun[qi] = mx;
          float sum = 0.f;
#pragma unroll
          for (int t = 0; t < BT; ++t) {
            if (blk * BK + t * 16 + 16 > S) {
#pragma unroll
              for (int e = 0; e < 4; ++e) sT[t][e] = exp_neg_f32(sT[t][e] - mx);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) sT[t][e] = exp_neg_finite_f32(sT[t][e] - mx);
            }
            sum += (sT[t][0] + sT[t][1]) + (sT[t][2] + sT[t][3]);
          }
          lrun[qi] = lrun[qi] * alpha + sum;
#pragma unroll
          for (int n = 0; n < 4; ++n) o[qi][n] *= alpha;
#pragma unroll
          for (int t = 0; t < BT; ++t) {
            if (blk * BK + t * 16 >= S) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
              for (int n = 0; n < 4; ++n) {
                const float vf = *reinterpret_cast<const float*>(vfrag + t * 4 * VPIECE + e * 256 + n * 64);
                o[qi][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf, sT[t][e], o[qi][n], 0, 0, 0);
              }
            }
          }
        }
      }
    }
#pragma unroll
    for (int qi = 0; qi < QPW; ++qi) {
      if (qtile[qi] < nqt) {  // wave-uniform
        const float inv = 1.f / sum_over_lane_groups(lrun[qi]);
#pragma unroll
        for (int n = 0; n < 4; ++n) o[qi][n] = o[qi][n] * inv;
      }
    }
    prole = role;
    pitem = item;
    if (!has_next) break;
    item = next_item;
    base = nbase;
    role = nrole;
  }
  store_out();
}

}  // namespace
}  // namespace fc

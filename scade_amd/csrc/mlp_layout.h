// Packed-parameter layout of one 8x256 NeRF (model/run_nerf_helpers.py:193-247)
// as consumed by the fused MFMA kernels.  All offsets in floats.
//
// Forward ("A-fragment") packing of a layer with N outputs and KB k-blocks of 8
// input channels:   Wp[((nt*KB + kb)*64 + lane)*4 + j] =
//        W[nt*32 + (lane&31)][kmap(kb*8 + 4*(lane>>5) + j)]     (0 where padded)
// so that one wave-wide 16-byte load is the A operand of four consecutive
// v_mfma_f32_32x32x2_f32 (k index of MFMA step j, lane half h  <->  channel
// kb*8 + 4h + j; the activation operand uses the same map).
#pragma once
#include <hip/hip_runtime.h>

namespace scade {

constexpr int W = 256;          // hidden width
constexpr int EMB = 57;         // 3 + 3*2*9 positional-encoding channels
constexpr int EMB_PAD = 64;     // padded to 8 k-blocks
constexpr int VIEW_PAD = 8;     // 3 view channels padded to one k-block
constexpr int NLAYER_MFMA = 10; // pts 0..7, feature, views

// layer ids
enum { L_PTS0 = 0, L_PTS5 = 5, L_PTS7 = 7, L_FEAT = 8, L_VIEWS = 9 };

// k-blocks read from the "pre" LDS region (embedding / view pad) and from h
constexpr int kb_pre(int l) { return l == 0 ? 8 : (l == 5 ? 8 : (l == L_VIEWS ? 1 : 0)); }
constexpr int kb_h(int l) { return l == 0 ? 0 : 32; }
constexpr int kb_total(int l) { return kb_pre(l) + kb_h(l); }
constexpr int n_out(int l) { return l == L_VIEWS ? 128 : 256; }
constexpr int w_floats(int l) { return n_out(l) * kb_total(l) * 8; }

constexpr int off_w(int l) {
  int o = 0;
  for (int i = 0; i < l; ++i) o += w_floats(i);
  return o;
}
constexpr int OFF_BIAS = off_w(NLAYER_MFMA);          // 10 bias vectors, 256 floats each slot
constexpr int off_b(int l) { return OFF_BIAS + l * 256; }
constexpr int OFF_WA = OFF_BIAS + NLAYER_MFMA * 256;   // alpha_linear.weight [256]
constexpr int OFF_BA = OFF_WA + 256;                   // alpha_linear.bias  [1] (+3 pad)
constexpr int OFF_WR = OFF_BA + 4;                     // rgb_linear.weight  [3][128]
constexpr int OFF_BR = OFF_WR + 384;                   // rgb_linear.bias    [3] (+1 pad)
constexpr int PACKED_FWD_FLOATS = OFF_BR + 4 + 256;    // + one k-block of slack

// Backward ("transposed") packing used by the dgrad chain.  For the 9 layers whose
// input gradient is needed (pts 1..7, feature, views) only the 256 columns that
// multiply the hidden state h take part:
//   WT[((kt*NB + nb)*64 + lane)*4 + j] = W[nb*8 + 4*(lane>>5) + j][hcol0 + kt*32 + (lane&31)]
// (hcol0 = 57 for pts layer 5, else 0; NB = N/8).  Index t = 0..8 <-> pts 1..7, feature, views.
constexpr int NLAYER_DGRAD = 9;
constexpr int dgrad_layer(int t) { return t < 7 ? t + 1 : (t == 7 ? L_FEAT : L_VIEWS); }
constexpr int dgrad_index(int l) { return l <= 7 ? l - 1 : (l == L_FEAT ? 7 : 8); }
constexpr int wt_floats(int t) { return 256 * n_out(dgrad_layer(t)); }
constexpr int off_wt(int t) {
  int o = 0;
  for (int i = 0; i < t; ++i) o += wt_floats(i);
  return o;
}
constexpr int PACKED_BWD_FLOATS = off_wt(NLAYER_DGRAD) + 256;

// Training workspace written by the forward (floats, P = points of the launch):
//   slots 0..7 : post-ReLU output of pts layer l        [P][256]
//   slot  8    : post-ReLU views hidden (cols 0..127)    [P][256]
//   slot  9    : feature_linear output (no activation)   [P][256]
//   emb        : [P][64] = gamma(x)(57) | 0 0 0 | viewdir(3) | 0
//   alpha_pre  : [P] alpha_linear output before softplus
//   masks      : [8][ceil(P/64)][256] u64, lane-private ReLU sign bits of pts layers 0..7
constexpr int N_ACT_SLOTS = 10;
constexpr int SLOT_VIEWS_H = 8, SLOT_FEAT = 9;
constexpr long acts_slot_off(long P, int s) { return (long)s * P * 256; }
constexpr long acts_emb_off(long P) { return (long)N_ACT_SLOTS * P * 256; }
constexpr long acts_alpha_off(long P) { return acts_emb_off(P) + P * 64; }
// ReLU sign bits of pts layers 0..7: [8][tiles][256 lanes] u64 (2 floats each), 8-byte aligned
constexpr long acts_mask_off(long P) { return (acts_alpha_off(P) + P + 1) / 2 * 2; }
constexpr long acts_floats(long P) { return acts_mask_off(P) + 8L * ((P + 63) / 64) * 256 * 2; }
// dgrad workspace: dZ slots [10][P][256] (same slot numbering: gradient w.r.t. the
// PRE-activation of that layer; slot 9 = d feature) followed by d alpha_pre [P]
constexpr long dz_dalpha_off(long P) { return (long)N_ACT_SLOTS * P * 256; }
constexpr long dz_floats(long P) { return dz_dalpha_off(P) + P; }

// parameter order of the 24 raw tensors handed to scade_mlp_pack
//  0..15 : pts_linears.{0..7}.{weight,bias}
// 16,17  : views_linears.0.{weight,bias}
// 18,19  : feature_linear.{weight,bias}
// 20,21  : alpha_linear.{weight,bias}
// 22,23  : rgb_linear.{weight,bias}
constexpr int N_PARAM_TENSORS = 24;
constexpr int N_PARAM_FLOATS = 589700;

// compute units of the current device (256 on MI355X); 256 when no device is visible (CPU-side
// size queries in the build container)
inline int device_cus() {
  // asked every time (a property lookup, no driver round trip): the library keeps no per-process copy, so a
  // process that drives devices of different sizes sizes every workspace and plan for the CURRENT one
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
    return v;
  return 256;
}

// ReLU sign words of the fp32-layout workspace: ONE 32-bit word per (pts layer, 32-point tile, thread
// of the workgroup that owns the tile), bit (t*4 + q)*4 + i <-> value (n-tile t, row group q, i) of that
// point tile in the thread's accumulator fragment.  Indexed by POINT TILE, not by workgroup, so the
// producers and consumers of a workspace may tile the points differently (64-point workgroups carry two
// words per thread - in registers the u64 `p*32 + (t*4+q)*4 + i` - 32-point workgroups one).
constexpr long relu_word_tiles(long P) { return 2 * ((P + 63) / 64); }
template <int PT>
__device__ __forceinline__ void store_relu_words(float* acts, long P, int layer, int tid,
                                                 unsigned long long bits) {
  unsigned* w = reinterpret_cast<unsigned*>(acts + acts_mask_off(P)) +
                ((size_t)layer * relu_word_tiles(P) + (size_t)blockIdx.x * PT) * 256 + tid;
  w[0] = (unsigned)bits;
  if (PT > 1) w[256] = (unsigned)(bits >> 32);
}
// ``blk``: index of the workgroup among those of ITS network (a launch may cover two networks)
template <int PT>
__device__ __forceinline__ unsigned long long load_relu_words(const float* acts, long P, int layer, int tid, int blk) {
  const unsigned* w = reinterpret_cast<const unsigned*>(acts + acts_mask_off(P)) +
                      ((size_t)layer * relu_word_tiles(P) + (size_t)blk * PT) * 256 + tid;
  unsigned long long b = w[0];
  if (PT > 1) b |= (unsigned long long)w[256] << 32;
  return b;
}

// Point tiles (of 32) per workgroup of the exact forward / dgrad kernels for a launch over P points:
// two (64 points, the throughput shape) unless the launch would not even give every CU its two
// workgroups - then one, which doubles the workgroups of a small batch (BASELINE configs[3]: 128 rays
// per GPU).  (The ReLU words are indexed by point tile, so forward and dgrad need not even agree.)
inline int pick_point_tiles(long P) { return (P + 63) / 64 < 2L * device_cus() ? 1 : 2; }

}  // namespace scade

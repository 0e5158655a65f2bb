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

// Backward ("transposed") packing used by the dgrad chain: for a layer with N
// outputs and K(padded) inputs,  WT[((kt*NB + nb)*64 + lane)*4 + j] =
//        W[nb*8 + 4*(lane>>5) + j][kmap(kt*32 + (lane&31))]
constexpr int wt_floats(int l) { return w_floats(l); }
constexpr int off_wt(int l) { return off_w(l); }
constexpr int PACKED_BWD_FLOATS = off_w(NLAYER_MFMA) + 256;

// parameter order of the 24 raw tensors handed to scade_mlp_pack
//  0..15 : pts_linears.{0..7}.{weight,bias}
// 16,17  : views_linears.0.{weight,bias}
// 18,19  : feature_linear.{weight,bias}
// 20,21  : alpha_linear.{weight,bias}
// 22,23  : rgb_linear.{weight,bias}
constexpr int N_PARAM_TENSORS = 24;
constexpr int N_PARAM_FLOATS = 589700;

}  // namespace scade

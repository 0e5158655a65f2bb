// The four weight packs of the MLP kernels as device functions of (parameter table, output blob, row, block),
// so that the stand-alone pack kernels (scade_mlp_pack, _pack_t, _pack_lp, _pack_t_lp) and the fused
// per-step pack of a Trainer (scade_mlp_pack_step: both networks, forward and transposed layouts, ONE launch -
// mlp_pack_step.hip) share one definition.  Every function handles ONE row (blockIdx.y of the stand-alone
// kernels: a layer, or the bias / head row) with block bx of nbx; blockDim.x = 256.
#pragma once
#include "mlp_tile_lp.h"

namespace scade {

constexpr int PACK_BLOCKS = 64;                 // blocks per row (= LP_NAN_BLOCKS: the census slots)
static_assert(PACK_BLOCKS == LP_NAN_BLOCKS, "one census slot per pack block");
constexpr int PACK_FWD_ROWS = NLAYER_MFMA + 1;  // 10 MFMA layers + biases / heads
constexpr int PACK_T_ROWS = NLAYER_DGRAD;       // 9 transposed layers

__device__ __forceinline__ float canon_nan(float x) { return x != x ? __builtin_nanf("") : x; }

// source column of padded channel k' for layer l, or -1 for zero padding
__device__ __forceinline__ int kmap(int l, int kp) {
  if (l == 0) return kp < EMB ? kp : -1;
  if (l == 5) return kp < EMB_PAD ? (kp < EMB ? kp : -1) : EMB + (kp - EMB_PAD);
  if (l == L_VIEWS) return kp < VIEW_PAD ? (kp < 3 ? W + kp : -1) : kp - VIEW_PAD;
  return kp;
}
__device__ __forceinline__ int k_real(int l) {
  return l == 0 ? EMB : (l == 5 ? EMB + W : (l == L_VIEWS ? W + 3 : W));
}

// ---- exact forward pack (mlp_layout.h), row l: 0..9 MFMA layers, 10 = biases + heads --------------------
__device__ __forceinline__ void pack_fwd_row(const float* const* p, float* __restrict__ packed, int l, int bx, int nbx) {
  const int t0 = bx * 256 + threadIdx.x, stride = nbx * 256;
  if (l < NLAYER_MFMA) {
    const int widx = l < 8 ? 2 * l : (l == L_FEAT ? 18 : 16);
    const float* __restrict__ Wsrc = p[widx];
    const int KB = kb_total(l);
    const int total = w_floats(l);
    const int kr = k_real(l);
    const int off = off_w(l);
    for (int i = t0; i < total; i += stride) {
      const int j = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
      const int kb = blk % KB, nt = blk / KB;
      const int n = nt * 32 + (lane & 31);
      const int src = kmap(l, kb * 8 + 4 * (lane >> 5) + j);
      packed[off + i] = src >= 0 ? canon_nan(Wsrc[(size_t)n * kr + src]) : 0.f;   // NaN weights: positive-signed
    }
  } else {
    for (int i = t0; i < NLAYER_MFMA * 256; i += stride) {
      const int ll = i >> 8, f = i & 255;
      const int bidx = ll < 8 ? 2 * ll + 1 : (ll == L_FEAT ? 19 : 17);
      packed[OFF_BIAS + i] = (ll == L_VIEWS && f >= 128) ? 0.f : canon_nan(p[bidx][f]);
    }
    for (int i = t0; i < 256; i += stride) packed[OFF_WA + i] = p[20][i];
    for (int i = t0; i < 4; i += stride) packed[OFF_BA + i] = i == 0 ? p[21][0] : 0.f;
    for (int i = t0; i < 384; i += stride) packed[OFF_WR + i] = p[22][i];
    for (int i = t0; i < 4; i += stride) packed[OFF_BR + i] = i < 3 ? p[23][i] : 0.f;
    for (int i = t0; i < 256; i += stride) packed[OFF_BR + 4 + i] = 0.f;
  }
}

// ---- exact transposed pack (dgrad chain), row t: dgrad index 0..8 -----------------------------------------
__device__ __forceinline__ void pack_t_row(const float* const* p, float* __restrict__ packedT, int t, int bx, int nbx) {
  const int t0 = bx * 256 + threadIdx.x, stride = nbx * 256;
  const int l = dgrad_layer(t);
  const int widx = l <= 7 ? 2 * l : (l == L_FEAT ? 18 : 16);
  const float* __restrict__ Wsrc = p[widx];
  const int N = n_out(l);
  const int NB = N / 8;
  const int ld = l == 5 ? EMB + W : (l == L_VIEWS ? W + 3 : W);
  const int hcol0 = l == 5 ? EMB : 0;
  const int total = 256 * N;
  const int off = off_wt(t);
  for (int i = t0; i < total; i += stride) {
    const int j = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
    const int nb = blk % NB, kt = blk / NB;
    const int n = nb * 8 + 4 * (lane >> 5) + j;
    const int k = kt * 32 + (lane & 31);
    packedT[off + i] = Wsrc[(size_t)n * ld + hcol0 + k];
  }
  if (t == 0)
    for (int i = t0; i < 256; i += stride) packedT[off_wt(NLAYER_DGRAD) + i] = 0.f;
}

// ---- 16-bit forward pack (mlp_tile_lp.h) + NaN census of the fp32 parameters, row l as pack_fwd_row ---------
__device__ __forceinline__ int kmap16_lp(int l, int kp) {
  // padded channel kp -> source column of layer l's weight, or -1 (zero)
  if (l == 0) return kp < EMB ? kp : -1;
  if (l == 5) return kp < 64 ? (kp < EMB ? kp : -1) : EMB + (kp - 64);
  if (l == L_VIEWS) return kp < 16 ? (kp < 3 ? W + kp : -1) : kp - 16;
  return kp;
}

template <bool BF>
__device__ __forceinline__ void pack_lp_row(const float* const* p, void* packed, int l, int bx, int nbx) {
  typedef typename LP<BF>::T T;
  T* wpk = reinterpret_cast<T*>(packed);
  if (l < NLAYER_MFMA) {
    const int widx = l < 8 ? 2 * l : (l == L_FEAT ? 18 : 16);
    const float* __restrict__ Wsrc = p[widx];
    const int KB = kb16(l);
    const long total = wl_elems(l);
    const int kr = l == 0 ? EMB : (l == 5 ? EMB + W : (l == L_VIEWS ? W + 3 : W));
    const long off = off_wl(l);
    for (long i = (long)bx * 256 + threadIdx.x; i < total; i += (long)nbx * 256) {
      const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
      const long blk = i >> 9;                        // nt*KB + kb
      const int kb = (int)(blk % KB), nt = (int)(blk / KB);
      const int n = nt * 32 + (lane & 31);
      const int src = kmap16_lp(l, kb * 16 + 8 * (lane >> 5) + j);
      wpk[off + i] = (T)(src >= 0 ? Wsrc[(size_t)n * kr + src] : 0.f);
    }
  } else {
    float* tail = reinterpret_cast<float*>(wpk + PACKED_LP_ELEMS);
    const int t0 = bx * 256 + threadIdx.x, stride = nbx * 256;
    for (int i = t0; i < NLAYER_MFMA * 256; i += stride) {
      const int ll = i >> 8, f = i & 255;
      const int bidx = ll < 8 ? 2 * ll + 1 : (ll == L_FEAT ? 19 : 17);
      tail[i] = (ll == L_VIEWS && f >= 128) ? 0.f : p[bidx][f];
    }
    for (int i = t0; i < 256; i += stride) tail[OFF_WA - OFF_BIAS + i] = p[20][i];
    for (int i = t0; i < 4; i += stride) tail[OFF_BA - OFF_BIAS + i] = i == 0 ? p[21][0] : 0.f;
    for (int i = t0; i < 384; i += stride) tail[OFF_WR - OFF_BIAS + i] = p[22][i];
    for (int i = t0; i < 4; i += stride) tail[OFF_BR - OFF_BIAS + i] = i < 3 ? p[23][i] : 0.f;
    for (int i = t0; i < 2 * 64 * 8; i += stride) wpk[off_wl(NLAYER_MFMA) + i] = (T)0.f;
    // (the rest of the tail's slack behind the census slots: every byte of the blob is defined)
    for (int i = LP_NAN_COLOUR + LP_NAN_BLOCKS - OFF_BIAS + t0; i < (int)F16_TAIL_FLOATS; i += stride) tail[i] = 0.f;
    // NaN census of the hidden layers' fp32 parameters (see the forward's alpha head): this block's slice of
    // every tensor, one 0 / NaN float per block and class; gridDim.x = LP_NAN_BLOCKS.  Straight-line: every
    // thread issues its ~24 sixteen-byte loads back to back (index clamped instead of predicated - re-reading
    // an element is harmless for a census) and only then looks at them; as twenty small loops, each waiting for
    // its own loads, this tripled the pack kernel's time (5 -> 15 us).
    int bad_trunk = 0, bad_colour = 0;
    {
      constexpr int NTHR = LP_NAN_BLOCKS * 256;
      f32x4 v[24];
      int n = 0;
#pragma unroll
      for (int t = 0; t < 20; ++t) {
        const int n4 = lp_param_numel(t) / 4;
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(p[t]);
#pragma unroll
        for (int k = 0; k < (n4 + NTHR - 1) / NTHR; ++k) v[n++] = src[min(t0 + k * NTHR, n4 - 1)];
      }
      n = 0;
#pragma unroll
      for (int t = 0; t < 20; ++t) {
        const int n4 = lp_param_numel(t) / 4;
#pragma unroll
        for (int k = 0; k < (n4 + NTHR - 1) / NTHR; ++k) {
          const f32x4 x = v[n++];
          const int bad = (x[0] != x[0]) | (x[1] != x[1]) | (x[2] != x[2]) | (x[3] != x[3]);
          if (t < 16) bad_trunk |= bad; else bad_colour |= bad;
        }
      }
    }
    bad_trunk = __syncthreads_or(bad_trunk);
    bad_colour = __syncthreads_or(bad_colour);
    if (threadIdx.x == 0) {
      tail[LP_NAN_TRUNK - OFF_BIAS + bx] = bad_trunk ? __builtin_nanf("") : 0.f;
      tail[LP_NAN_COLOUR - OFF_BIAS + bx] = bad_colour ? __builtin_nanf("") : 0.f;
    }
  }
}

// ---- 16-bit transposed pack, row t: dgrad index 0..8 --------------------------------------------------------
//   WTL[((kt*NB16 + nb)*64 + lane)*8 + j] = W[nb*16 + 8*(lane>>5) + j][hcol0 + kt*32 + (lane&31)]
constexpr long wtl_elems(int t) { return 256L * n_out(dgrad_layer(t)); }
constexpr long off_wtl(int t) {
  long o = 0;
  for (int i = 0; i < t; ++i) o += wtl_elems(i);
  return o;
}
constexpr long PACKED_T_LP_ELEMS = off_wtl(NLAYER_DGRAD) + 2 * 64 * 8;
// fp32 tail behind the 16-bit planes: the head weights the dgrad kernel multiplies on the VALU
// (alpha_linear.weight [256] | rgb_linear.weight [3][128]) - so that a 16-bit training step needs no
// fp32 forward blob at all
constexpr int TL_WA = 0, TL_WR = 256, PACKED_T_LP_TAIL_FLOATS = 256 + 384;
constexpr long PACKED_T_LP_BYTES = PACKED_T_LP_ELEMS * 2 + PACKED_T_LP_TAIL_FLOATS * 4;

template <bool BF>
__device__ __forceinline__ void pack_t_lp_row(const float* const* p, void* packed, int t, int bx, int nbx) {
  typedef typename LP<BF>::T T;
  T* out = reinterpret_cast<T*>(packed);
  const int l = dgrad_layer(t);
  const int widx = l <= 7 ? 2 * l : (l == L_FEAT ? 18 : 16);
  const float* __restrict__ Wsrc = p[widx];
  const int N = n_out(l);
  const int NB = N / 16;
  const int ld = l == 5 ? EMB + W : (l == L_VIEWS ? W + 3 : W);
  const int hcol0 = l == 5 ? EMB : 0;
  const long total = 256L * N;
  const long off = off_wtl(t);
  for (long i = (long)bx * 256 + threadIdx.x; i < total; i += (long)nbx * 256) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long blk = i >> 9;
    const int nb = (int)(blk % NB), kt = (int)(blk / NB);
    const int n = nb * 16 + 8 * (lane >> 5) + j;
    const int k = kt * 32 + (lane & 31);
    out[off + i] = (T)Wsrc[(size_t)n * ld + hcol0 + k];
  }
  if (t == 0) {
    for (int i = bx * 256 + threadIdx.x; i < 2 * 64 * 8; i += nbx * 256) out[off_wtl(NLAYER_DGRAD) + i] = (T)0.f;
    float* tl = reinterpret_cast<float*>(out + PACKED_T_LP_ELEMS);
    for (int i = bx * 256 + threadIdx.x; i < PACKED_T_LP_TAIL_FLOATS; i += nbx * 256)
      tl[i] = i < TL_WR ? p[20][i] : p[22][i - TL_WR];
  }
}

// ---- split-precision ("f16x3") packs: two fp16 planes in fragment order (mlp_fwd_f16.hip / mlp_bwd_f16.hip) ----------
__device__ __forceinline__ int kmap16(int l, int kp) {
  // padded channel kp -> source column of layer l's weight, or -1 (zero)
  if (l == 0) return kp < EMB ? kp : -1;
  if (l == 5) return kp < 64 ? (kp < EMB ? kp : -1) : EMB + (kp - 64);
  if (l == L_VIEWS) return kp < 16 ? (kp < 3 ? W + kp : -1) : kp - 16;
  return kp;
}
// forward pack, row l: 0..9 MFMA layers, 10 = the fp32 tail (biases, heads) + the slack block
__device__ __forceinline__ void pack_f16_row(const float* const* p, void* packed, int l, int bx, int nbx) {
  _Float16* wpk = reinterpret_cast<_Float16*>(packed);
  if (l < NLAYER_MFMA) {
    const int widx = l < 8 ? 2 * l : (l == L_FEAT ? 18 : 16);
    const float* __restrict__ Wsrc = p[widx];
    const int KB = kb16(l);
    const long total = wh_halves(l) / 2;              // elements per plane pair
    const int kr = l == 0 ? EMB : (l == 5 ? EMB + W : (l == L_VIEWS ? W + 3 : W));
    const long off = off_wh(l);
    for (long i = (long)bx * 256 + threadIdx.x; i < total; i += (long)nbx * 256) {
      const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
      const long blk = i >> 9;                        // (nt*KB + kb)
      const int kb = (int)(blk % KB), nt = (int)(blk / KB);
      const int n = nt * 32 + (lane & 31);
      const int src = kmap16(l, kb * 16 + 8 * (lane >> 5) + j);
      const float w = src >= 0 ? Wsrc[(size_t)n * kr + src] : 0.f;
      _Float16 h, lo;
      split2(w, h, lo);
      const long base = off + (blk * 2) * 512 + lane * 8 + j;
      wpk[base] = h;
      wpk[base + 512] = lo;
    }
  } else {
    float* tail = reinterpret_cast<float*>(wpk + PACKED_F16_HALVES);
    const int t0 = bx * 256 + threadIdx.x, stride = nbx * 256;
    for (int i = t0; i < NLAYER_MFMA * 256; i += stride) {
      const int ll = i >> 8, f = i & 255;
      const int bidx = ll < 8 ? 2 * ll + 1 : (ll == L_FEAT ? 19 : 17);
      tail[i] = (ll == L_VIEWS && f >= 128) ? 0.f : p[bidx][f];
    }
    for (int i = t0; i < 256; i += stride) tail[OFF_WA - OFF_BIAS + i] = p[20][i];
    for (int i = t0; i < 4; i += stride) tail[OFF_BA - OFF_BIAS + i] = i == 0 ? p[21][0] : 0.f;
    for (int i = t0; i < 384; i += stride) tail[OFF_WR - OFF_BIAS + i] = p[22][i];
    for (int i = t0; i < 4; i += stride) tail[OFF_BR - OFF_BIAS + i] = i < 3 ? p[23][i] : 0.f;
    for (int i = t0; i < 2 * 64 * 8; i += stride) wpk[off_wh(NLAYER_MFMA) + i] = (_Float16)0.f;
  }
}
// transposed two-plane pack: for dgrad index t (mlp_layout.h), layer l = dgrad_layer(t):
//   WT16[((kt*NB16 + nb)*2 + plane)*64*8 + lane*8 + j] = split(W[nb*16 + 8*(lane>>5) + j][hcol0 + kt*32 + (lane&31)])
constexpr long wt16_halves(int t) { return (long)256 * n_out(dgrad_layer(t)) * 2; }
constexpr long off_wt16(int t) {
  long o = 0;
  for (int i = 0; i < t; ++i) o += wt16_halves(i);
  return o;
}
constexpr long PACKED_T_F16_HALVES = off_wt16(NLAYER_DGRAD) + 2 * 64 * 8;
__device__ __forceinline__ void pack_t_f16_row(const float* const* p, void* packed_t, int t, int bx, int nbx) {
  _Float16* packed = reinterpret_cast<_Float16*>(packed_t);
  const int l = dgrad_layer(t);
  const int widx = l <= 7 ? 2 * l : (l == L_FEAT ? 18 : 16);
  const float* __restrict__ Wsrc = p[widx];
  const int N = n_out(l);
  const int NB = N / 16;
  const int ld = l == 5 ? EMB + W : (l == L_VIEWS ? W + 3 : W);
  const int hcol0 = l == 5 ? EMB : 0;
  const long total = (long)256 * N;
  const long off = off_wt16(t);
  for (long i = (long)bx * 256 + threadIdx.x; i < total; i += (long)nbx * 256) {
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long blk = i >> 9;
    const int nb = (int)(blk % NB), kt = (int)(blk / NB);
    const int n = nb * 16 + 8 * (lane >> 5) + j;
    const int k = kt * 32 + (lane & 31);
    _Float16 h, lo;
    split2(Wsrc[(size_t)n * ld + hcol0 + k], h, lo);
    const long base = off + blk * 1024 + lane * 8 + j;
    packed[base] = h;
    packed[base + 512] = lo;
  }
  if (t == 0)
    for (int i = bx * 256 + threadIdx.x; i < 2 * 64 * 8; i += nbx * 256) packed[off_wt16(NLAYER_DGRAD) + i] = (_Float16)0.f;
}

// ---- the weight packs of a train step as numbered work items -------------------------------------------------------
// One item = one (network, row, block) of the per-step pack (mlp_pack_step.hip runs them as its grid); the launches that
// OPEN a graph-captured step (scade_stage_inputs_points / scade_gather_batch_points) run them as extra workgroups, so
// the captured step itself starts at the first MLP launch.  fmt: 0 exact (exact + tr = scade_mlp_pack / _pack_t
// layouts), 1 bf16, 2 fp16 (fwd + tr = scade_mlp_pack_lp / _pack_t_lp), 3 split precision (exact, fwd = _pack_f16,
// tr = _pack_t_f16); blob entries may be null (skipped).
struct PackItemsArgs {
  const float* p[2][N_PARAM_TENSORS];
  float* exact[2];
  void* fwd[2];
  void* tr[2];
  int n_nets, fmt;
};
__host__ __device__ constexpr int pack_item_rows(int fmt) {
  return fmt == 3 ? 2 * PACK_FWD_ROWS + PACK_T_ROWS : PACK_FWD_ROWS + PACK_T_ROWS;
}
__host__ __device__ constexpr int pack_item_count(int n_nets, int fmt) { return n_nets * pack_item_rows(fmt) * PACK_BLOCKS; }
__device__ __forceinline__ void pack_item(const PackItemsArgs& a, int item) {      // (item is workgroup-uniform)
  const int bx = item % PACK_BLOCKS, rw = item / PACK_BLOCKS;
  const int rows = pack_item_rows(a.fmt);
  const int net = rw / rows, row = rw % rows;
  const float* const* p = a.p[net];
  if (a.fmt == 3) {
    if (row < PACK_FWD_ROWS) { if (a.exact[net]) pack_fwd_row(p, a.exact[net], row, bx, PACK_BLOCKS); }
    else if (row < 2 * PACK_FWD_ROWS) { if (a.fwd[net]) pack_f16_row(p, a.fwd[net], row - PACK_FWD_ROWS, bx, PACK_BLOCKS); }
    else if (a.tr[net]) pack_t_f16_row(p, a.tr[net], row - 2 * PACK_FWD_ROWS, bx, PACK_BLOCKS);
  } else if (row < PACK_FWD_ROWS) {
    if (a.fmt == 0) { if (a.exact[net]) pack_fwd_row(p, a.exact[net], row, bx, PACK_BLOCKS); }
    else if (a.fwd[net]) { if (a.fmt == 1) pack_lp_row<true>(p, a.fwd[net], row, bx, PACK_BLOCKS); else pack_lp_row<false>(p, a.fwd[net], row, bx, PACK_BLOCKS); }
  } else if (a.tr[net]) {
    if (a.fmt == 0) pack_t_row(p, reinterpret_cast<float*>(a.tr[net]), row - PACK_FWD_ROWS, bx, PACK_BLOCKS);
    else if (a.fmt == 1) pack_t_lp_row<true>(p, a.tr[net], row - PACK_FWD_ROWS, bx, PACK_BLOCKS);
    else pack_t_lp_row<false>(p, a.tr[net], row - PACK_FWD_ROWS, bx, PACK_BLOCKS);
  }
}
// host side: the arguments of a C entry -> PackItemsArgs (pack_format < 0: none; returns false on a missing pointer)
inline bool pack_items_fill(PackItemsArgs& a, int pack_format, int n_nets, const float* const* net_params,
                            float* const* packed_exact, void* const* packed_fwd, void* const* packed_t) {
  a = PackItemsArgs{};
  a.fmt = pack_format; a.n_nets = pack_format < 0 ? 0 : n_nets;
  if (pack_format < 0) return true;
  if (!net_params || n_nets < 1 || n_nets > 2 || pack_format > 3) return false;
  for (int k = 0; k < n_nets; ++k) {
    for (int i = 0; i < N_PARAM_TENSORS; ++i) {
      if (!net_params[k * N_PARAM_TENSORS + i]) return false;
      a.p[k][i] = net_params[k * N_PARAM_TENSORS + i];
    }
    a.exact[k] = packed_exact ? packed_exact[k] : nullptr;
    a.fwd[k] = packed_fwd ? packed_fwd[k] : nullptr;
    a.tr[k] = packed_t ? packed_t[k] : nullptr;
  }
  return true;
}

}  // namespace scade

// Tile-level building blocks shared by the fused MLP forward and dgrad kernels.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace scade {

constexpr int TM = 64;                         // points per workgroup
constexpr int H_FLOATS = TM * W;               // 16384
constexpr int EMB_STRIDE = 60;                 // floats; 240 B rows -> conflict-free b128
constexpr int EMB_FLOATS = TM * EMB_STRIDE + 4;  // 3840 + one zeroed 16-byte pad: the last
                                                // row's k-block 7 reads columns 60..63
constexpr int MLP_LDS_BYTES = (H_FLOATS + EMB_FLOATS) * 4;  // 80912
// the same tile with PT point tiles of 32 (PT = 2 above; PT = 1: the small-batch variant, twice the
// workgroups for launches that would otherwise leave CUs idle)
constexpr int tile_pts(int PT) { return 32 * PT; }
constexpr int h_floats(int PT) { return 32 * PT * W; }
constexpr int mlp_lds_bytes(int PT) { return (h_floats(PT) + 32 * PT * EMB_STRIDE + 4) * 4; }

// float index of 16-byte chunk `chunk` of row `row` in the swizzled h tile
__device__ __forceinline__ int h_idx(int row, int chunk) {
  return row * W + ((chunk ^ (row & 15)) << 2);
}

// ---------------------------------------------------------------------------
// k-loop of one layer.  acc[t][p]: n-tile t of this wave x point-tile p.
//   wp      : this wave's first n-tile, [NT][KB][64] float4, KB = KBP + KBH
//   an      : in  = the layer's k-block 0 A fragments (prefetched by the previous layer)
//             out = k-block 0 of the NEXT layer (wp_next, kb_next k-blocks per n-tile), so
//             the L2 latency of a layer's first weights hides under the previous layer
//   pre     : LDS region for the first KBP k-blocks (row stride PRE_STRIDE floats)
//   hbuf    : swizzled h tile for the remaining KBH k-blocks
// ---------------------------------------------------------------------------
//   binit   : this lane's bias values in accumulator order (load_bias); BINIT = false: start from zero.  The
//             accumulators START from the bias (the same 128 v_mov the zero fill costs) instead of adding it in
//             the epilogue: fp32 MFMAs run on the SIMD's fp32 lanes, so the 128 v_add per layer and wave were
//             1.5 % of the matrix time (round 4)
template <int NT, int KBP, int KBH, int PRE_STRIDE, int PT, bool BINIT>
__device__ __forceinline__ void layer_gemm_b(f32x16 (&acc)[NT][PT], f32x4 (&an)[2],
                                             const f32x4* __restrict__ wp,
                                             const f32x4* __restrict__ wp_next, int kb_next,
                                             const float* pre, const float* hbuf, int lane,
                                             const f32x4 (&binit)[NT][4]) {
  constexpr int KB = KBP + KBH;
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int p = 0; p < PT; ++p)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][p][i] = BINIT ? binit[t][i >> 2][i & 3] : 0.f;

  auto load_b = [&](int kb, f32x4& b0, f32x4& b1) {   // b1: rows 32.. (PT == 2 only)
    if (KBP > 0 && kb < KBP) {
      const float* q = pre + r * PRE_STRIDE + (2 * kb + hh) * 4;
      b0 = *reinterpret_cast<const f32x4*>(q);
      if (PT > 1) b1 = *reinterpret_cast<const f32x4*>(q + 32 * PRE_STRIDE);
    } else {
      const float* q = hbuf + h_idx(r, 2 * (kb - KBP) + hh);
      b0 = *reinterpret_cast<const f32x4*>(q);
      if (PT > 1) b1 = *reinterpret_cast<const f32x4*>(q + 32 * W);
    }
  };
  auto mfma_block = [&](const f32x4 (&a)[2], const f32x4 (&b)[2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int p = 0; p < PT; ++p)
          acc[t][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[p][j], acc[t][p], 0, 0, 0);
  };

  f32x4 bn[2];
  load_b(0, bn[0], bn[1]);

#pragma unroll 2
  for (int kb = 0; kb < KB - 1; ++kb) {
    f32x4 a[2], b[2];
    a[0] = an[0]; a[1] = an[1];
    b[0] = bn[0];
    if (PT > 1) b[1] = bn[1];
#pragma unroll
    for (int t = 0; t < NT; ++t) an[t] = wp[(t * KB + kb + 1) * 64 + lane];
    load_b(kb + 1, bn[0], bn[1]);
    // keep the next block's loads ABOVE this block's MFMAs (hipcc otherwise sinks
    // them below the last use of a[]/b[] to reuse the registers: no prefetch)
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(a, b);
  }
  {  // last k-block: prefetch the next layer's first weights instead
    f32x4 a[2], b[2];
    a[0] = an[0]; a[1] = an[1];
    b[0] = bn[0];
    if (PT > 1) b[1] = bn[1];
    an[0] = wp_next[lane];
    an[1] = wp_next[kb_next * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    mfma_block(a, b);
  }
}
template <int NT, int KBP, int KBH, int PRE_STRIDE, int PT = 2>
__device__ __forceinline__ void layer_gemm(f32x16 (&acc)[NT][PT], f32x4 (&an)[2], const f32x4* __restrict__ wp,
                                           const f32x4* __restrict__ wp_next, int kb_next, const float* pre,
                                           const float* hbuf, int lane) {
  const f32x4 none[NT][4] = {};
  layer_gemm_b<NT, KBP, KBH, PRE_STRIDE, PT, false>(acc, an, wp, wp_next, kb_next, pre, hbuf, lane, none);
}
template <int NT, int KBP, int KBH, int PRE_STRIDE, int PT = 2>
__device__ __forceinline__ void layer_gemm(f32x16 (&acc)[NT][PT], f32x4 (&an)[2], const f32x4* __restrict__ wp,
                                           const f32x4* __restrict__ wp_next, int kb_next, const float* pre,
                                           const float* hbuf, int lane, const f32x4 (&binit)[NT][4]) {
  layer_gemm_b<NT, KBP, KBH, PRE_STRIDE, PT, true>(acc, an, wp, wp_next, kb_next, pre, hbuf, lane, binit);
}

// coalesced copy of the h tile (first ncols columns) to dst[P][256] (full 1-KiB rows)
__device__ __forceinline__ void save_tile(const float* hbuf, float* __restrict__ dst, int p0, int P,
                                          int ncols, int tid, int tm = TM) {
  const int chunks_per_row = ncols >> 2;
  for (int i = tid; i < tm * chunks_per_row; i += 256) {
    const int row = i / chunks_per_row, c = i - row * chunks_per_row;
    if (p0 + row < P)   // streamed once, read by a later kernel: non-temporal
      __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(hbuf + h_idx(row, c)),
                                  reinterpret_cast<f32x4*>(dst + (size_t)(p0 + row) * W + 4 * c));
  }
}

// the same copy for the columns ONE WAVE has just written (NCW columns from column c0, every row of the tile):
// a wave's LDS accesses execute in order, so it may read its own layer_store back without a barrier and its
// rows leave for HBM while the other waves are still in their epilogues.
// Round 4 (the stamp trace of profiles/r04_fwd_layer_trace.txt): fp32 MFMAs execute on the SIMD's fp32 lanes, so every
// VALU instruction of either workgroup of a CU is time the matrix stream does not get - and this copy spent ~10 of
// them per 16-byte chunk (lane map, swizzle, 64-bit row address, bounds compare + exec-mask branch) plus a full LDS
// round trip per chunk.  Now: the lane's chunks are 4 (2) LDS base addresses + immediate offsets (rows 16 apart
// share their swizzle), the stores are buffer stores with a fixed lane offset, the row part in the scalar offset and
// the ragged last tile left to the descriptor's range check (rows >= P are dropped) - no VALU work in the loop, and
// BATCH chunks in flight (the training forward has 16 registers to spare, not 32).
template <int NCW, int PT, int BATCH = 2>
__device__ __forceinline__ void save_tile_wave(const float* hbuf, float* __restrict__ dst, int p0, int P, int c0, int lane) {
  static_assert(NCW == 64 || NCW == 32, "a wave owns 64 columns (32 in the views layer)");
  constexpr int CPR = NCW / 4;                   // 16-byte chunks per row of this wave's columns: 16 or 8
  constexpr int RPI = 64 / CPR;                  // rows per wave instruction: 4 or 8
  constexpr int NB = 16 / RPI;                   // swizzle classes (instructions per 16 rows): 4 or 2
  constexpr int ITERS = 32 * PT / RPI;           // instructions per tile
  // (the lane's addresses are recomputed per call - a dozen VALU instructions - instead of living in registers across
  // the whole kernel, where the compiler would hoist them to: the training forward has none to spare)
  asm volatile("" : "+v"(lane));
  const int lr = lane / CPR, c = (c0 >> 2) + lane % CPR;
  const unsigned char* hb = reinterpret_cast<const unsigned char*>(hbuf);
  const unsigned char* base[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) base[b] = hb + (RPI * b + lr) * (W * 4) + ((c ^ (RPI * b + lr)) << 4);
  // the descriptor starts at the TILE's first row and ends at row P (at most 4096 rows on): nothing in it depends on
  // P * pitch fitting 32 bits (a slot of 4 M points is 4 GiB)
  const unsigned long long pd = reinterpret_cast<unsigned long long>(dst) + (unsigned long long)p0 * 1024ull;
  const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)pd), dhi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)dhi << 32) | dlo), 0,
      __builtin_amdgcn_readfirstlane(tile_rows_left(p0, P) * 1024u), 0x00020000);
  const int voff = lr * 1024 + c * 16;
  constexpr int soff = 0;
  typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
    u32x4_ v[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int it = it0 + j;
      v[j] = *reinterpret_cast<const u32x4_*>(base[it % NB] + (it / NB) * 16 * W * 4);
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int it = it0 + j;
      __builtin_amdgcn_raw_buffer_store_b128(v[j], rs, voff, soff + it * RPI * 1024, 2);     // streamed once: nt
    }
    STORE_DATA_HOLD(v[BATCH - 1]);                      // (common.h: the next batch's LDS address landed in v[BATCH - 1][0])
#pragma unroll
    for (int j = 0; j + 1 < BATCH; ++j) STORE_DATA_PIN(v[j]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace scade

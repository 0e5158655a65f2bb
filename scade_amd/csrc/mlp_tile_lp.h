// Building blocks of the single-plane 16-bit ("lp": fp16 or bf16 operands, fp32 accumulate)
// variant of the fused NeRF MLP - BASELINE.json config 5's "bf16 MFMA path".
//
// Tile: 128 points per workgroup (4 point tiles of 32), 4 waves, every wave owns 64 output
// features (2 n-tiles) of all 128 points -> 8 independent v_mfma_f32_32x32x16_{f16,bf16}
// per 16-channel k-block and 128 accumulator registers per lane.  128 points (not the exact
// kernel's 64) because the weight stream is the scarce resource at this MFMA rate: every
// A fragment fetched from L2 now feeds 4 MFMAs, which halves the bytes per point through the
// vector-memory path (32 B/clk/CU at the full matrix rate instead of 64, the L1 peak).
// LDS: activations [128][256] + embedding [128][64], 16-bit each = 80 KiB -> 2 workgroups/CU.
#pragma once
#include "common.h"
#include "mlp_layout.h"
#include "mlp_tile_f16.h"   // kb16/kbp16/kbh16, kmap helpers, x_idx/e_idx swizzles

namespace scade {

// forces compile-time evaluation of the constexpr layout helpers at their use sites (hipcc
// otherwise emits some of them as real device functions and CALLS them from the kernel)
template <long V> struct CE { static constexpr long v = V; };
typedef float lp_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 lp_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 lp_f16x2 __attribute__((ext_vector_type(2)));

template <bool BF> struct LP;
template <> struct LP<false> {
  typedef _Float16 T;
  typedef _Float16 V8 __attribute__((ext_vector_type(8)));
  typedef _Float16 V4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct LP<true> {
  typedef __bf16 T;
  typedef __bf16 V8 __attribute__((ext_vector_type(8)));
  typedef __bf16 V4 __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ f32x16 mfma(V8 a, V8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

constexpr int LM = 128;                      // points per workgroup
constexpr int LPT = LM / 32;                 // point tiles
constexpr int LXPLANE = LM * W;              // elements of the activation tile
constexpr int LEPLANE = LM * 64;             // elements of the embedding tile
constexpr int LP_LDS_BYTES = (LXPLANE + LEPLANE) * 2;   // 81920
// the same tile with NPT point tiles (NPT = 4 above; NPT = 2: the small-batch variant - twice the
// workgroups for launches that would leave most CUs idle)
constexpr int lp_lds_bytes(int NPT) { return 32 * NPT * (W + 64) * 2; }
// point tiles per workgroup for a launch over P points (forward and dgrad of a step agree: the sign
// words are indexed by workgroup)
inline int lp_pick_point_tiles(long P) { return (P + LM - 1) / LM < device_cus() ? 2 : 4; }

// packed blob: per layer [ntile][kb][64 lanes][8 elements]  (counted in 16-bit elements)
constexpr long wl_elems(int l) { return (long)n_out(l) / 32 * kb16(l) * 64 * 8; }
constexpr long off_wl(int l) {
  long o = 0;
  for (int i = 0; i < l; ++i) o += wl_elems(i);
  return o;
}
constexpr long PACKED_LP_ELEMS = off_wl(NLAYER_MFMA) + 2 * 64 * 8;   // + slack for the prefetch
// NaN census of the hidden layers' fp32 parameters, written by the pack kernel into the (otherwise unused)
// 256-float slack at the end of the fp32 tail: one 0 / NaN float per pack block for the trunk
// (pts_linears.*) and one for the colour branch (feature_linear, views_linears.0); offsets like OFF_*
constexpr int LP_NAN_BLOCKS = 64;
constexpr int LP_NAN_TRUNK = OFF_BR + 4, LP_NAN_COLOUR = LP_NAN_TRUNK + LP_NAN_BLOCKS;
static_assert(LP_NAN_COLOUR + LP_NAN_BLOCKS <= PACKED_FWD_FLOATS, "census slots must fit the tail's slack");
// elements of raw parameter tensor t (order of mlp_layout.h: 0..15 pts w/b, 16/17 views, 18/19 feature)
__host__ __device__ constexpr int lp_param_numel(int t) {
  return (t & 1) ? (t == 17 ? 128 : 256)
                 : (t == 0 ? 256 * EMB : (t == 10 ? 256 * (EMB + W) : (t == 16 ? 128 * (W + 3) : W * W)));
}
constexpr long PACKED_LP_BYTES = PACKED_LP_ELEMS * 2 + F16_TAIL_FLOATS * 4;

// ---- training workspaces of the 16-bit path (BYTE offsets; T = 16-bit element) ------------
// acts: slots [10][P][256] T (same slot numbering as mlp_layout.h) | emb [P][64] T |
//       alpha_pre [P] fp32 | ReLU sign words of pts layers 0..7: [8][tiles][256 lanes][2] u64
constexpr long lp_align(long b) { return (b + 255) / 256 * 256; }
constexpr long lp_tiles(long P) { return (P + LM - 1) / LM; }
constexpr long lp_acts_alpha_byte(long P) { return lp_align((acts_emb_off(P) + P * 64) * 2); }
constexpr long lp_acts_mask_byte(long P) { return lp_align(lp_acts_alpha_byte(P) + P * 4); }
constexpr long lp_acts_bytes(long P) { return lp_acts_mask_byte(P) + 9L * ((P + 63) / 64) * 256 * 16; }   // 8 trunk layers + the views layer; sized for 64-point workgroups
// dz: slots [10][P][256] T, every row multiplied by the launch-wide power of two S | d alpha_pre [P] fp32
constexpr long lp_dz_dalpha_byte(long P) { return lp_align((long)N_ACT_SLOTS * P * 256 * 2); }
constexpr long lp_dz_bytes(long P) { return lp_align(lp_dz_dalpha_byte(P) + P * 4); }

// coalesced copy of the first NCOLS columns of the LDS tile to dst[P][256] (optional per-row
// factor).  Four 16-byte chunks per thread in flight: as a plain loop every iteration exposed the
// LDS latency in front of its store (16 dependent round trips per layer).
template <bool BF, int NCOLS, int NPT = LPT>
__device__ __forceinline__ void save_tile_lp(const typename LP<BF>::T* x, typename LP<BF>::T* __restrict__ dst,
                                             int p0, int P, const float* row_fac, int tid) {
  typedef typename LP<BF>::T T;
  typedef typename LP<BF>::V8 V8;
  constexpr int CPR = NCOLS >> 3;                     // 16-byte chunks per row
  constexpr int ITERS = 32 * NPT * CPR / 256;
  static_assert(ITERS % 4 == 0, "save_tile_lp: batches of four");
#pragma unroll 1      // keep the batches apart: merged, their 16 reads + conversions spill
  for (int it0 = 0; it0 < ITERS; it0 += 4) {
    V8 v[4];
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * (it0 + j);
      const int row = i / CPR, c = i - row * CPR;
      v[j] = *reinterpret_cast<const V8*>(x + x_idx(row, c));
      f[j] = row_fac ? row_fac[row] : 1.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = tid + 256 * (it0 + j);
      const int row = i / CPR, c = i - row * CPR;
      if (row_fac) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = (T)((float)v[j][e] * f[j]);
      }
      if (p0 + row < P) __builtin_nontemporal_store(v[j], reinterpret_cast<V8*>(dst + (size_t)(p0 + row) * W + 8 * c));
    }
  }
}

// (row, chunk) a lane copies in wave-instruction ``it`` of a per-wave tile copy whose rows are CPR 16-byte
// chunks wide (8: 64 columns, 4: 32 columns).  ds_read_b128 is serviced in the fixed 16-lane groups
// {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32); under the x_idx swizzle (slot = chunk ^ (row & 15)) a group is
// conflict free when its rows differ in the row bits ABOVE the chunk bits it spans - rows r, r + 8 for eight
// chunks, r, r + 4, r + 8, r + 12 for four.  The straight mapping (lane / CPR, lane % CPR) put rows r, r + 1
// (.. r + 3) into one group: 2-way conflicts on every read of the tile copies (the difference between the
// training forward's 14.7 % conflict cycles and the inference kernel's 4.7 % in round 2).  Every instruction
// still covers whole rows, so the global stores stay 128-byte (64-byte) row segments.
template <int CPR>
__device__ __forceinline__ void tile_copy_map(int lane, int it, int& row, int& chunk) {
  static_assert(CPR == 8 || CPR == 4, "tile_copy_map: 8 or 4 chunks per row");
  const int q = (lane >> 2) & 7, up = lane >> 5;          // lane quad within the half wave, half wave
  if (CPR == 8) {
    // quads 0..7 of a half wave -> rows {a, b, b, a, b+8, a+8, a+8, b+8}, a = 2 up, b = a + 1
    const int off = (0x98890110 >> (4 * q)) & 15;
    row = 16 * (it >> 1) + 4 * (it & 1) + 2 * up + off;
    chunk = lane & 7;
  } else {
    // quads 0..7 -> rows {a, b, b+4, a+4, b+8, a+8, a+12, b+12}
    const int off = (int)((0xDC895410u >> (4 * q)) & 15u);
    row = 16 * it + 2 * up + off;
    chunk = lane & 3;
  }
}

// the same copy for the NCW columns (from column c0) ONE WAVE has just written, every row of the tile: a
// wave's LDS accesses execute in order, so it reads its own epilogue back without a barrier and its rows
// leave for HBM while the other waves are still in their epilogues (the exact kernels' save_tile_wave)
template <bool BF, int NCW, int NPT = LPT>
__device__ __forceinline__ void save_tile_lp_wave(const typename LP<BF>::T* x, typename LP<BF>::T* __restrict__ dst,
                                                  int p0, int P, const float* row_fac, int c0, int lane) {
  typedef typename LP<BF>::T T;
  typedef typename LP<BF>::V8 V8;
  constexpr int CPR = NCW >> 3;                       // 16-byte chunks per row of this wave's columns
  constexpr int ITERS = 32 * NPT * CPR / 64;
  static_assert(ITERS % 4 == 0, "save_tile_lp_wave: batches of four");
#pragma unroll 1
  for (int it0 = 0; it0 < ITERS; it0 += 4) {
    V8 v[4];
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int row, c;
      tile_copy_map<CPR>(lane, it0 + j, row, c);
      c += c0 >> 3;
      v[j] = *reinterpret_cast<const V8*>(x + x_idx(row, c));
      f[j] = row_fac ? row_fac[row] : 1.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int row, c;
      tile_copy_map<CPR>(lane, it0 + j, row, c);
      c += c0 >> 3;
      if (row_fac) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = (T)((float)v[j][e] * f[j]);
      }
      if (p0 + row < P) __builtin_nontemporal_store(v[j], reinterpret_cast<V8*>(dst + (size_t)(p0 + row) * W + 8 * c));
    }
  }
}

// ---- format code 2: bf16 arithmetic, rows SAVED for the weight gradient as 8-bit e5m2 ("bf8") ------------------
// The 16-bit training step is HBM-bound on the rows it saves (activations written by the forward, dZ rows written
// by the dgrad chain, both read back by the weight gradient: 21 KB per point; measured roofs of this part 4.5
// TB/s written, 6.7 TB/s read).  Format code 2 halves those bytes: the forward / dgrad arithmetic is the bf16
// path's, bit for bit, only the COPY of a tile that leaves for HBM is rounded (RNE) to e5m2 - activations as they
// are, dZ rows under the launch-wide power-of-two loss scale of the fp16 path (e5m2 has fp16's exponent range) -
// and the weight gradient converts the rows back to bf16 while staging them into LDS.  What changes numerically
// is the weight gradient's operands (2 mantissa bits, zero-mean rounding); dgrad and forward are untouched.
// Layout: the workspaces keep the 16-bit offsets; an 8-bit row p of slot s lies at byte acts_slot_off(P, s) * 2 +
// p * 256 (the first half of the slot's region).  One part stays 16-bit: the activation slot of the 128-wide views
// hidden layer - the dgrad kernel derives that layer's ReLU mask from it, so rounding it could turn a tiny positive
// activation into "inactive"; with it 16-bit the dgrad chain of format code 2 is the bf16 path's bit for bit
// (tests/test_gpu_lp.py).  The embedding rows (64 columns; read by the weight gradient only) are saved as fp8 e4m3
// since round 4 - |gamma(x)| <= 1 needs no exponent range, so the byte goes to a third mantissa bit - at byte
// acts_emb_off(P) * 2 + p * 64: the weight gradient's embedding-input jobs contract them on the fp8 MFMA as they
// lie (mixed bf8 x fp8 operands), like the 256-wide jobs.
typedef unsigned lp_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned lp_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lp_pack4_bf8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
  return (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(c, d, w, true);
}
// fp8 e4m3 (OCP): the embedding rows of format code 2 (|gamma(x)|, |viewdir| <= 1; 3 mantissa bits)
__device__ __forceinline__ unsigned lp_pack4_fp8(float a, float b, float c, float d) {
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
}
// 4 e5m2 bytes -> 4 values of T as two packed dwords (exact: every e5m2 value is a bf16 / fp16 value)
template <bool BF>
__device__ __forceinline__ lp_u32x2 lp_unpack4_bf8(unsigned w) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, false), hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)w, true);
  lp_u32x2 r;
  if (BF) {
    r[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, lp_bf16x2));
    r[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, lp_bf16x2));
  } else {
    r[0] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, lp_f16x2));
    r[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, lp_f16x2));
  }
  return r;
}
// save_tile_lp_wave with 8-bit rows: dst8 = byte base of the slot (row pitch 256 bytes), ``fac`` = the ONE
// power-of-two factor of every row (format code 2 runs bf16 arithmetic: no per-point scale; nullptr = 1).
// Round 4: the knock-out of this copy was worth 19 % of the training forward and 25 % of the dgrad chain at 1.3
// TB/s of stores - instruction overhead, not bytes: per 16-byte chunk the compiler recomputed the lane map, the
// swizzled LDS index and a 64-bit row address, and wrapped every store in an exec-mask branch for the ragged
// last tile.  Now a lane's sixteen chunks are two LDS base addresses + immediate offsets (rows 16 apart share
// their swizzle; rows 4 apart differ in one address bit), the stores are buffer stores whose descriptor ends at
// row P (rows past the end are dropped by the range check: no branch) with the lane part of the address fixed
// and the rest in the scalar offset, and the conversion is v_cvt_scalef32_pk_bf8_bf16 (two packed bf16 -> two
// e5m2 bytes of x / scale, RNE: one op per pair where bf16 -> fp32, the factor and v_cvt_pk_bf8_f32 took five;
// bit-identical for power-of-two factors - probed on the part over all bf16 encodings).
template <bool BF, int NCW, int NPT = LPT>
__device__ __forceinline__ void save_tile_lp_wave8(const typename LP<BF>::T* x, unsigned char* __restrict__ dst8,
                                                   int p0, int P, const float* fac, int c0, int lane) {
  static_assert(NCW == 64, "a wave owns 64 columns");
  if constexpr (!BF) return;                          // (format code 2 is bf16 arithmetic; never reached)
  typedef __bf16 V8 __attribute__((ext_vector_type(8)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  constexpr int ITERS = 4 * NPT;                      // 16-byte chunks per lane
  int rl, c;
  tile_copy_map<8>(lane, 0, rl, c);                   // rl in {0..3, 8..11}: rows rl + 4 (it & 1) + 16 (it >> 1)
  c += c0 >> 3;
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(x);
  const unsigned char* a0 = xb + 2 * (rl * W + ((c ^ rl) << 3));            // it even
  const unsigned char* a1 = xb + 2 * ((rl + 4) * W + ((c ^ rl ^ 4) << 3));  // it odd (row & 15 = rl | 4)
  const unsigned long long pd = reinterpret_cast<unsigned long long>(dst8) + (unsigned long long)p0 * 256ull;   // (mlp_tile.h)
  const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)pd), dhi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)dhi << 32) | dlo), 0,
      __builtin_amdgcn_readfirstlane(tile_rows_left(p0, P) * 256u), 0x00020000);
  const int voff = rl * 256 + 8 * c;
  constexpr int soff = 0;
  const float inv = fac ? __builtin_amdgcn_rcpf(fac[0]) : 1.0f;
#pragma unroll
  for (int it0 = 0; it0 < ITERS; it0 += 4) {          // batches of four chunks in flight (all sixteen would spill)
    V8 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = it0 + j;
      v[j] = *reinterpret_cast<const V8*>(((it & 1) ? a1 : a0) + (it >> 1) * 16 * W * 2);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int it = it0 + j;
      s16x2 q0 = {0, 0}, q1 = {0, 0};
      q0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q0, __builtin_shufflevector(v[j], v[j], 0, 1), inv, false);
      q0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q0, __builtin_shufflevector(v[j], v[j], 2, 3), inv, true);
      q1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q1, __builtin_shufflevector(v[j], v[j], 4, 5), inv, false);
      q1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q1, __builtin_shufflevector(v[j], v[j], 6, 7), inv, true);
      const lp_u32x2 o = {__builtin_bit_cast(unsigned, q0), __builtin_bit_cast(unsigned, q1)};
      __builtin_amdgcn_raw_buffer_store_b64(o, rs, voff + (it & 1) * 4 * 256, soff + (it >> 1) * 16 * 256, 2);    // nt
    }
    __builtin_amdgcn_sched_barrier(0);                // keep the batches apart
  }
}

// (common.h: lds_barrier / LDS_SYNC - a workgroup barrier that orders LDS only)
__device__ __forceinline__ void lp_lds_barrier() { lds_barrier(); }
#define LP_SYNC() LDS_SYNC()

// The same copy as a RIDER of the next layer's k-loop (round 4): the tile that was just written stays in LDS as the
// B operand of the next gemm, so its sixteen chunks per lane leave one per k-block - read in block kb, converted
// and stored in block kb + 1 - instead of as a burst of 32 KiB per workgroup in front of the weight fetches of the
// next layer (the stores and the A-fragment loads share the CU's vector-memory path and its in-order counter).
struct NoRider {
  static constexpr bool ON = false;
  static constexpr int NCH = 0;
  __device__ __forceinline__ void begin(int) {}
  __device__ __forceinline__ void read(int) {}
  __device__ __forceinline__ void emit() {}
};
// Whole rows per wave: behind the barrier that precedes the k-loop a wave may read any column of the tile, so wave w
// takes rows [8 NPT w, 8 NPT (w + 1)) with all 256 columns - a store instruction writes two complete 256-byte rows
// (512 contiguous bytes) where the per-wave column slices of the burst copy write 64-byte quarters of eight rows.
// init() keeps wave-uniform state only (SGPRs); begin() derives the lane's part - the gemm calls it behind its
// peeled first k-block, where the registers of the initial accumulator value (the bias vector) have just died.
template <int NPT = LPT, int NCOLS = W>
struct SaveRider8 {
  static constexpr bool ON = true;
  static constexpr int LPR = NCOLS / 8;               // lanes per row (8 columns = 16 bytes of LDS, 8 of HBM each)
  static constexpr int RPI = 64 / LPR;                // rows per store instruction: 2 (256 columns) or 4 (128)
  static constexpr int NCH = 8 * NPT / RPI;           // chunks per lane: the wave's 8 NPT rows
  typedef __bf16 V8 __attribute__((ext_vector_type(8)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const unsigned char* xb;
  int row0;                                           // first row of this wave
  __amdgpu_buffer_rsrc_t rs;
  int voff, soff0, soff_p, c, rlo;
  float inv;
  V8 v;
  __device__ __forceinline__ void init(const __bf16* x, unsigned char* __restrict__ dst8, int p0, int P, const float* fac, int wave) {
    row0 = 8 * NPT * wave;
    xb = reinterpret_cast<const unsigned char*>(x) + row0 * W * 2;
    const unsigned long long pd = reinterpret_cast<unsigned long long>(dst8) + (unsigned long long)p0 * 256ull;   // (mlp_tile.h)
    const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)pd), dhi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
    rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)dhi << 32) | dlo), 0,
                                           __builtin_amdgcn_readfirstlane(tile_rows_left(p0, P) * 256u), 0x00020000);
    soff0 = __builtin_amdgcn_readfirstlane(row0 * 256);
    inv = fac ? __builtin_amdgcn_rcpf(fac[0]) : 1.0f;
  }
  __device__ __forceinline__ void begin(int lane) {
    asm volatile("" : "+v"(lane));                    // (not hoisted above the caller's first k-block)
    c = lane % LPR;
    rlo = lane / LPR;
    voff = rlo * 256 + c * 8;                         // (rows are 256 bytes apart in HBM whatever NCOLS)
  }
  __device__ __forceinline__ void read(int it) {     // it: wave-uniform chunk index; rows row0 + RPI it + {0 .. RPI-1}
    if (it < NCH) {
      const int r = RPI * it + rlo;                   // (row0 is a multiple of 16: the swizzle sees r & 15)
      v = *reinterpret_cast<const V8*>(xb + r * (W * 2) + ((c ^ (r & 15)) << 4));
      soff_p = soff0 + it * RPI * 256;
    }
  }
  __device__ __forceinline__ void emit() {
    s16x2 q0 = {0, 0}, q1 = {0, 0};
    q0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q0, __builtin_shufflevector(v, v, 0, 1), inv, false);
    q0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q0, __builtin_shufflevector(v, v, 2, 3), inv, true);
    q1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q1, __builtin_shufflevector(v, v, 4, 5), inv, false);
    q1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(q1, __builtin_shufflevector(v, v, 6, 7), inv, true);
    const lp_u32x2 o = {__builtin_bit_cast(unsigned, q0), __builtin_bit_cast(unsigned, q1)};
    // nt: with a cached store the dgrad chain measured +10 % and the weight gradient, whose ring then finds the
    // rows' lines dirty in L2, +27 %
    __builtin_amdgcn_raw_buffer_store_b64(o, rs, voff, soff_p, 2);
  }
};

// the same rider for 16-bit saved rows (plain bf16 training: a copy, 512-byte rows): a store instruction writes two
// complete rows = 1 KiB
template <int NPT = LPT>
struct SaveRider16 {
  static constexpr bool ON = true;
  static constexpr int NCH = 4 * NPT;                 // chunks per lane: the wave's 8 NPT rows, two per instruction
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const unsigned char* xb;
  int row0;
  __amdgpu_buffer_rsrc_t rs;
  int voff, soff0, soff_p, c, rlo;
  u4 v;
  __device__ __forceinline__ void init(const __bf16* x, __bf16* __restrict__ dst, int p0, int P, int wave) {
    row0 = 8 * NPT * wave;
    xb = reinterpret_cast<const unsigned char*>(x) + row0 * W * 2;
    const unsigned long long pd = reinterpret_cast<unsigned long long>(dst) + (unsigned long long)p0 * 512ull;   // (mlp_tile.h)
    const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)pd), dhi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
    rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)dhi << 32) | dlo), 0,
                                           __builtin_amdgcn_readfirstlane(tile_rows_left(p0, P) * 512u), 0x00020000);
    soff0 = __builtin_amdgcn_readfirstlane(row0 * 512);
  }
  __device__ __forceinline__ void begin(int lane) {
    asm volatile("" : "+v"(lane));
    c = lane & 31;
    rlo = lane >> 5;
    voff = lane * 16;
  }
  __device__ __forceinline__ void read(int it) {
    if (it < NCH) {
      const int r = 2 * it + rlo;
      v = *reinterpret_cast<const u4*>(xb + r * (W * 2) + ((c ^ (r & 15)) << 4));
      soff_p = soff0 + it * 2 * 512;
    }
  }
  __device__ __forceinline__ void emit() {
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, soff_p, 2);
    STORE_DATA_HOLD(v);                                 // (common.h: read() of the next chunk follows)
  }
};

// ReLU sign bits of one layer: 4 x 32-bit words per lane.  The 64 packed dwords a lane produces
// per layer are numbered d = ((t*4 + q)*4 + p)*2 + j (n-tile t, row group q, point tile p, value
// pair j); word d >> 4 holds, at bit (d & 15), the sign of the dword's LOW half and at bit
// 16 + (d & 15) the sign of its HIGH half - so both sides handle two values per instruction:
//   forward : m = v_pk_min_u16(w, 1) (0/1 per half of the ReLU'd pair);  word |= m << (d & 15)
//   backward: m = (word >> (d & 15)) & 0x00010001;  w = v_pk_mul_lo_u16(w, m)
__device__ __forceinline__ unsigned sign_pair(unsigned w_relu) {
  unsigned m;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(w_relu), "v"(0x00010001u));
  return m;
}
__device__ __forceinline__ unsigned mask_pair(unsigned w, unsigned word, int k) {
  const unsigned m = (word >> k) & 0x00010001u;
  unsigned r;
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(m));
  return r;
}

template <bool BF>
struct AFragL { typename LP<BF>::V8 t0, t1; };
// NS rotating A-fragment register sets: k-block kb of a layer entered with rotation ROT lives in set
// (ROT + kb) % NS; the sets of blocks kb+1 .. kb+NS-1 are in flight while block kb is multiplied
template <bool BF, int NS>
struct AFragN { AFragL<BF> s[NS]; };
template <bool BF>
using AFrag3 = AFragN<BF, 3>;

// acc[t][p] = cinit[t] + W[n-tile t] * act[point tile p] over the layer's k-blocks (cinit == nullptr:
// zero).  The first k-block is peeled so that the initial value rides in as the MFMA's C operand
// (the lane's bias vector, or the inline constant 0) instead of 128 v_mov + 128 v_add per layer.
// A (weights, L2 latency): fetched NS-1 k-blocks ahead into the rotating sets of AFragN - across the
// layer boundary too (the last NS-1 blocks prefetch blocks 0 .. NS-2 of the next layer; the caller
// enters the next layer with rotation (ROT + KB) % NS).  B (activations, LDS): every fragment is
// reloaded in place for the next block right after the two MFMAs that consume it were issued.
template <bool BF, int NT, int KBP, int KBH, bool PRE_VIEW, int ROT, int NS = 3, int NPT = LPT, class RID = NoRider>
__device__ __forceinline__ void layer_gemm_lp(f32x16 (&acc)[NT][NPT], AFragN<BF, NS>& A,
                                              const typename LP<BF>::V8* __restrict__ wp,
                                              const typename LP<BF>::V8* __restrict__ wp_next, int kb_next,
                                              const typename LP<BF>::T* e, const typename LP<BF>::T* x,
                                              int lane, const f32x16* cinit, RID& rid) {
  typedef typename LP<BF>::V8 V8;
  constexpr int KB = KBP + KBH;
  const int r = lane & 31, hh = lane >> 5;

#define LOAD_BL(KBX, PX, B)                                                         \
  {                                                                                 \
    const int kb_ = (KBX);                                                          \
    if (KBP > 0 && kb_ < KBP) {                                                     \
      if (PRE_VIEW) B = *reinterpret_cast<const V8*>(e + ((PX)*32 + r) * 16 + hh * 8); \
      else B = *reinterpret_cast<const V8*>(e + e_idx((PX)*32 + r, 2 * kb_ + hh));  \
    } else {                                                                        \
      B = *reinterpret_cast<const V8*>(x + x_idx((PX)*32 + r, 2 * (kb_ - KBP) + hh)); \
    }                                                                               \
  }
#define MFMA2(PX, AS, B)                                                \
  acc[0][PX] = LP<BF>::mfma(AS.t0, B, acc[0][PX]);                      \
  if (NT > 1) acc[NT - 1][PX] = LP<BF>::mfma(AS.t1, B, acc[NT - 1][PX]);
#define MFMA2_FIRST(PX, AS, B)                                          \
  acc[0][PX] = LP<BF>::mfma(AS.t0, B, c00);                             \
  if (NT > 1) acc[NT - 1][PX] = LP<BF>::mfma(AS.t1, B, c01);
  // fetch k-block KBX+NS-1 (of this layer, or one of the first of the next) into set DST
#define FETCH_A(KBX, DST)                                               \
  {                                                                     \
    const int blk_ = (KBX) + NS - 1;                                    \
    if (blk_ < KB) {                                                    \
      DST.t0 = wp[blk_ * 64 + lane];                                    \
      if (NT > 1) DST.t1 = wp[(KB + blk_) * 64 + lane];                 \
    } else {                                                            \
      DST.t0 = wp_next[(blk_ - KB) * 64 + lane];                        \
      DST.t1 = wp_next[(kb_next + blk_ - KB) * 64 + lane];              \
    }                                                                   \
  }
#define KBLOCK(KBX, R)                                                  \
  {                                                                     \
    FETCH_A(KBX, A.s[((R) + NS - 1) % NS])                              \
    const int kn = (KBX) + 1 < KB ? (KBX) + 1 : (KBX);                  \
    __builtin_amdgcn_sched_barrier(0);                                  \
    MFMA2(0, A.s[R], b0) LOAD_BL(kn, 0, b0)                             \
    if constexpr (RID::ON) { if ((KBX) >= 2 && (KBX) - 2 < RID::NCH) rid.emit(); } \
    __builtin_amdgcn_sched_barrier(0);                                  \
    MFMA2(1, A.s[R], b1) LOAD_BL(kn, 1, b1)                             \
    if constexpr (RID::ON) rid.read((KBX) - 1);                         \
    __builtin_amdgcn_sched_barrier(0);                                  \
    if constexpr (NPT > 2) {                                            \
      MFMA2(2, A.s[R], b2) LOAD_BL(kn, 2, b2)                           \
      __builtin_amdgcn_sched_barrier(0);                                \
      MFMA2(3, A.s[R], b3) LOAD_BL(kn, 3, b3)                           \
      __builtin_amdgcn_sched_barrier(0);                                \
    }                                                                   \
  }

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const f32x16 c00 = cinit ? cinit[0] : zero16;
  const f32x16 c01 = cinit ? cinit[NT - 1] : zero16;
  V8 b0, b1, b2, b3;
  LOAD_BL(0, 0, b0) LOAD_BL(0, 1, b1)
  if constexpr (NPT > 2) { LOAD_BL(0, 2, b2) LOAD_BL(0, 3, b3) }
  {   // peeled k-block 0
    FETCH_A(0, A.s[(ROT + NS - 1) % NS])
    const int kn = 1 < KB ? 1 : 0;
    __builtin_amdgcn_sched_barrier(0);
    // point tile 0 last: its accumulator can then take over the registers of the initial value
    MFMA2_FIRST(1, A.s[ROT], b1) LOAD_BL(kn, 1, b1)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NPT > 2) {
      MFMA2_FIRST(2, A.s[ROT], b2) LOAD_BL(kn, 2, b2)
      __builtin_amdgcn_sched_barrier(0);
      MFMA2_FIRST(3, A.s[ROT], b3) LOAD_BL(kn, 3, b3)
      __builtin_amdgcn_sched_barrier(0);
    }
    MFMA2_FIRST(0, A.s[ROT], b0) LOAD_BL(kn, 0, b0)
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (RID::ON) rid.begin(lane);
  // a real loop over groups of NS k-blocks (one body per register set); never fully unrolled: ten
  // layers of straight-line k-loops would not fit the instruction cache
  static_assert(NS >= 3 && NS <= 6, "layer_gemm_lp: three to six A sets");
  int kb = 1;
#pragma unroll 1
  for (; kb + NS - 1 < KB; kb += NS) {
    KBLOCK(kb, (ROT + 1) % NS)
    KBLOCK(kb + 1, (ROT + 2) % NS)
    KBLOCK(kb + 2, (ROT + 3) % NS)
    if constexpr (NS > 3) KBLOCK(kb + 3, (ROT + 4) % NS)
    if constexpr (NS > 4) KBLOCK(kb + 4, (ROT + 5) % NS)
    if constexpr (NS > 5) KBLOCK(kb + 5, (ROT + 6) % NS)
  }
  constexpr int REM = (KB - 1) % NS;
  if constexpr (REM >= 1) KBLOCK(kb, (ROT + 1) % NS)
  if constexpr (REM >= 2) KBLOCK(kb + 1, (ROT + 2) % NS)
  if constexpr (REM >= 3) KBLOCK(kb + 2, (ROT + 3) % NS)
  if constexpr (REM >= 4) KBLOCK(kb + 3, (ROT + 4) % NS)
  if constexpr (REM >= 5) KBLOCK(kb + 4, (ROT + 5) % NS)
  if constexpr (RID::ON) {        // chunk kb - 1 is read in k-block kb and leaves in k-block kb + 1: what is left
    if constexpr (KB >= 2 && KB - 2 < RID::NCH) rid.emit();
#pragma unroll
    for (int c = KB - 1; c < RID::NCH; ++c) { rid.read(c); rid.emit(); }
  }
#undef KBLOCK
#undef FETCH_A
#undef LOAD_BL
#undef MFMA2
#undef MFMA2_FIRST
}

template <bool BF, int NT, int KBP, int KBH, bool PRE_VIEW, int ROT, int NS = 3, int NPT = LPT>
__device__ __forceinline__ void layer_gemm_lp(f32x16 (&acc)[NT][NPT], AFragN<BF, NS>& A,
                                              const typename LP<BF>::V8* __restrict__ wp,
                                              const typename LP<BF>::V8* __restrict__ wp_next, int kb_next,
                                              const typename LP<BF>::T* e, const typename LP<BF>::T* x,
                                              int lane, const f32x16* cinit) {
  NoRider none;
  layer_gemm_lp<BF, NT, KBP, KBH, PRE_VIEW, ROT, NS, NPT, NoRider>(acc, A, wp, wp_next, kb_next, e, x, lane, cinit, none);
}

// this lane's bias values in accumulator order: cb[t][4q+i] = bias[(ntile0+t)*32 + 8q + 4*(lane>>5) + i]
template <int NT>
__device__ __forceinline__ void load_bias16(f32x16 (&cb)[NT], const float* __restrict__ bias, int ntile0,
                                            int lane) {
  const int hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(bias + (ntile0 + t) * 32 + 8 * q + 4 * hh);
      cb[t][4 * q + 0] = v[0]; cb[t][4 * q + 1] = v[1]; cb[t][4 * q + 2] = v[2]; cb[t][4 * q + 3] = v[3];
    }
}

// round two fp32 values to T (RNE) and pack them into one dword; with RELU the ReLU is applied to
// the PACKED pair as a signed 16-bit integer max with 0 (v_pk_max_i16: a negative float, including
// -0, is a negative int16, a positive float a positive one) - one VALU op per two values.
// The conversion is a 2-vector __builtin_convertvector, which hipcc selects as ONE v_cvt_pk_{bf16,f16}_f32
// (scalar casts get converted one by one and merged with v_perm_b32).  It must NOT be inline asm: its inputs
// are MFMA accumulators, and a VALU read of an MFMA result needs software wait states on gfx950 (no hardware
// interlock) that the compiler inserts only in front of instructions it selected itself - while it is free to
// hoist an asm statement right behind the last MFMA of the k-loop.  Round 3's first dgrad epilogue order made
// it do exactly that: the first two chunks of the LAST point tile came out stale, differently from run to
// run, in the bf16 build only (fp16's schedule happened to differ); the round-2 epilogues had been correct by
// the luck of their schedules.  The integer ReLU stays asm: its input is the conversion's ordinary VALU result.
// v_permlane32_swap of two dwords whose producers may be INLINE ASM (the integer ReLU of pack2, mask_pair).
// gfx950 needs two wait states between a VALU write of a VGPR and a v_permlane*_swap that reads it; hipcc's
// hazard recognizer inserts them for instructions it selected itself but under-counts behind an asm statement
// (one wait state instead of two in a minimal test).  The s_nop rides in an asm that reads both operands and
// "modifies" one, which orders it behind their producers and in front of the swap.
__device__ __forceinline__ void lp_swap_halves(unsigned a, unsigned b, unsigned& lo, unsigned& hi) {
  asm("s_nop 1" : "+v"(a) : "v"(b));      // data dependences only (not volatile): a, b ready -> nop -> swap
  const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  lo = sw[0];
  hi = sw[1];
}

template <bool BF, bool RELU>
__device__ __forceinline__ unsigned pack2(float y0, float y1) {
  const lp_f32x2 y = {y0, y1};
  unsigned w;
  if (BF) w = __builtin_bit_cast(unsigned, __builtin_convertvector(y, lp_bf16x2));
  else w = __builtin_bit_cast(unsigned, __builtin_convertvector(y, lp_f16x2));
  if (RELU) asm("v_pk_max_i16 %0, %1, 0" : "=v"(w) : "v"(w));
  return w;
}

}  // namespace scade

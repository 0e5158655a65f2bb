// Building blocks shared by the split-precision (f16x3) forward and dgrad kernels: the
// two-plane fp16 representation x ~= h + l*2^-11, LDS plane indexing, and the k-loop.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace scade {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int HM = 64;                       // points per workgroup
constexpr int XPLANE = HM * W;               // halves per activation plane
constexpr int EPLANE = HM * 64;              // halves per embedding plane
constexpr int F16_LDS_BYTES = (2 * XPLANE + 2 * EPLANE) * 2;   // 81920
constexpr float LSCALE = 2048.0f, LINV = 1.0f / 2048.0f;

// k-blocks of 16 channels
constexpr int kbp16(int l) { return l == 0 ? 4 : (l == 5 ? 4 : (l == L_VIEWS ? 1 : 0)); }
constexpr int kbh16(int l) { return l == 0 ? 0 : 16; }
constexpr int kb16(int l) { return kbp16(l) + kbh16(l); }
// packed blob: per layer [ntile][kb][plane 2][64 lanes][8 halves]  (counted in halves)
constexpr long wh_halves(int l) { return (long)n_out(l) / 32 * kb16(l) * 2 * 64 * 8; }
constexpr long off_wh(int l) {
  long o = 0;
  for (int i = 0; i < l; ++i) o += wh_halves(i);
  return o;
}
constexpr long PACKED_F16_HALVES = off_wh(NLAYER_MFMA) + 2 * 64 * 8;   // + slack block
// fp32 tail (biases + head weights) reuses the fp32 blob layout after the MFMA weights
constexpr long F16_TAIL_FLOATS = PACKED_FWD_FLOATS - OFF_BIAS;
constexpr long PACKED_F16_BYTES = PACKED_F16_HALVES * 2 + F16_TAIL_FLOATS * 4;

__device__ __forceinline__ void split2(float x, _Float16& h, _Float16& l) {
  h = (_Float16)x;
  l = (_Float16)((x - (float)h) * LSCALE);
}

// The same split for FOUR values at once, bit-identical to split2, with hand-selected instructions:
// v_cvt_pk_f16_f32 for a pair of high parts, then v_fma_mixlo/hi_f16 (h * -2048 + 2048 x in fp32 - exact -
// rounded to fp16 into the low / high half), which read the fp16 h in place.  hipcc's selection for split2
// converts h back to fp32, subtracts, scales and converts again: three conversion-class VALU ops per value
// (they issue at a fraction of the plain VALU rate) instead of one and a half.
__device__ __forceinline__ void split4(const float (&x)[4], half4& vh, half4& vl) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2 h, l;
  const float m = -LSCALE;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float s0 = x[2 * k] * LSCALE, s1 = x[2 * k + 1] * LSCALE;
    unsigned hk, lk;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hk) : "v"(x[2 * k]), "v"(x[2 * k + 1]));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lk) : "v"(hk), "s"(m), "v"(s0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lk) : "v"(hk), "s"(m), "v"(s1));
    h[k] = hk; l[k] = lk;
  }
  vh = __builtin_bit_cast(half4, h);
  vl = __builtin_bit_cast(half4, l);
}

// halves index of the 8-half chunk c of row r in an activation plane / embedding plane
__device__ __forceinline__ int x_idx(int row, int c) { return row * W + ((c ^ (row & 15)) << 3); }
__device__ __forceinline__ int e_idx(int row, int c) { return row * 64 + ((c ^ ((row >> 1) & 7)) << 3); }

// A fragments of one k-block for the wave's (up to) two n-tiles, both planes.  Plain named
// members (no arrays passed by reference): hipcc (ROCm 7.2) mis-allocates registers for the
// array-reference form of this loop on gfx950 (address temporaries land in a live operand).
struct AFrag { half8 t0h, t0l, t1h, t1l; };

// A rider of the k-loop (round 5, after the 16-bit kernels' SaveRider8): the tile the PREVIOUS layer wrote stays in LDS
// as this gemm's B operand, so its copy to HBM leaves one chunk per pair of k-blocks - read in one, stored in the
// next - instead of as a burst between the epilogue and the next layer's first weight fetches.
struct NoRiderH {
  static constexpr bool ON = false;
  static constexpr int NCH = 0;
  __device__ __forceinline__ void begin(int) {}
  __device__ __forceinline__ void read(int) {}
  __device__ __forceinline__ void emit(int) {}
};

template <int NT, int KBP, int KBH, bool PRE_VIEW, class RID = NoRiderH>
__device__ __forceinline__ void layer_gemm_h(f32x16 (&acc0)[NT][2], f32x16 (&acc1)[NT][2], AFrag& an,
                                             const half8* __restrict__ wp,
                                             const half8* __restrict__ wp_next, int kb_next,
                                             const _Float16* eh, const _Float16* el,
                                             const _Float16* xh, const _Float16* xl, int lane,
                                             const f32x16* cinit, RID& rid) {
  // acc0 + acc1/2048 = cinit + W x.  The first k-block is peeled so that the initial value (the
  // lane's bias vector, or nothing) rides in as the C operand of acc0's first MFMA and acc1
  // starts from the inline constant 0: no zero-fill, no bias add in the epilogue.
  constexpr int KB = KBP + KBH;
  const int r = lane & 31, hh = lane >> 5;
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const f32x16 c00 = cinit ? cinit[0] : zero16;
  const f32x16 c01 = cinit ? cinit[NT - 1] : zero16;

  // activation fragment (both planes) of point tile p for k-block kb
#define LOAD_B(KBX, PX, BH, BL)                                                              \
  {                                                                                          \
    const int kb_ = (KBX);                                                                   \
    if (KBP > 0 && kb_ < KBP) {                                                              \
      if (PRE_VIEW) {                                                                        \
        BH = *reinterpret_cast<const half8*>(eh + ((PX)*32 + r) * 16 + hh * 8);               \
        BL = *reinterpret_cast<const half8*>(el + ((PX)*32 + r) * 16 + hh * 8);               \
      } else {                                                                               \
        const int o_ = e_idx((PX)*32 + r, 2 * kb_ + hh);                                     \
        BH = *reinterpret_cast<const half8*>(eh + o_);                                       \
        BL = *reinterpret_cast<const half8*>(el + o_);                                       \
      }                                                                                      \
    } else {                                                                                 \
      const int o_ = x_idx((PX)*32 + r, 2 * (kb_ - KBP) + hh);                               \
      BH = *reinterpret_cast<const half8*>(xh + o_);                                         \
      BL = *reinterpret_cast<const half8*>(xl + o_);                                         \
    }                                                                                        \
  }
// six MFMAs of one point tile: the two dependent updates of each acc1 are kept >= 3 MFMAs apart
#define MFMA6(PX, A, VH, VL)                                                                   \
  acc0[0][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t0h, VH, acc0[0][PX], 0, 0, 0);        \
  acc1[0][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t0h, VL, acc1[0][PX], 0, 0, 0);        \
  if (NT > 1) {                                                                                \
    acc0[NT - 1][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t1h, VH, acc0[NT - 1][PX], 0, 0, 0); \
    acc1[NT - 1][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t1h, VL, acc1[NT - 1][PX], 0, 0, 0); \
  }                                                                                            \
  acc1[0][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t0l, VH, acc1[0][PX], 0, 0, 0);        \
  if (NT > 1)                                                                                  \
    acc1[NT - 1][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t1l, VH, acc1[NT - 1][PX], 0, 0, 0);

  // register plan: A (weights, L2 latency) is fetched a whole k-block ahead; B (LDS) ping-pongs
  // between the two point tiles inside the block: b1 of this block loads under the p=0 MFMAs,
  // b0 of the next block under the p=1 MFMAs.
#define MFMA6_FIRST(PX, A, VH, VL)                                                             \
  acc0[0][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t0h, VH, c00, 0, 0, 0);                \
  acc1[0][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t0h, VL, zero16, 0, 0, 0);             \
  if (NT > 1) {                                                                                \
    acc0[NT - 1][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t1h, VH, c01, 0, 0, 0);        \
    acc1[NT - 1][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t1h, VL, zero16, 0, 0, 0);     \
  }                                                                                            \
  acc1[0][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t0l, VH, acc1[0][PX], 0, 0, 0);        \
  if (NT > 1)                                                                                  \
    acc1[NT - 1][PX] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A.t1l, VH, acc1[NT - 1][PX], 0, 0, 0);
  // the k-loop at raised priority (dropped again behind it): a SIMD's other wave is the other workgroup's, and when
  // that one is in its VALU-bound epilogue this one's MFMAs should win the issue slot (forward -1.3 %, dgrad -0.8 %)
  __builtin_amdgcn_s_setprio(1);
  half8 b0h, b0l, b1h, b1l;
  LOAD_B(0, 0, b0h, b0l)
  LOAD_B(0, 1, b1h, b1l)
  {   // peeled k-block 0 (KB >= 4 everywhere).  Point tile 1 first: tile 0's accumulator can then
      // take over the registers of the initial value.
    const AFrag a = an;
    an.t0h = wp[(1 * 2 + 0) * 64 + lane];
    an.t0l = wp[(1 * 2 + 1) * 64 + lane];
    if (NT > 1) {
      an.t1h = wp[((KB + 1) * 2 + 0) * 64 + lane];
      an.t1l = wp[((KB + 1) * 2 + 1) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);
    MFMA6_FIRST(1, a, b1h, b1l)
    __builtin_amdgcn_sched_barrier(0);
    MFMA6_FIRST(0, a, b0h, b0l)
    __builtin_amdgcn_sched_barrier(0);
    LOAD_B(1, 0, b0h, b0l)
  }
  if (RID::ON) {                   // behind the peeled block: the registers of the initial value have just died
    rid.begin(lane);
    rid.read(0);
  }
#undef MFMA6_FIRST
#define KBLOCK_H(KBX)                                                   \
  {                                                                     \
    const AFrag a = an;                                                 \
    if ((KBX) + 1 < KB) {                                               \
      an.t0h = wp[(((KBX) + 1) * 2 + 0) * 64 + lane];                   \
      an.t0l = wp[(((KBX) + 1) * 2 + 1) * 64 + lane];                   \
      if (NT > 1) {                                                     \
        an.t1h = wp[((KB + (KBX) + 1) * 2 + 0) * 64 + lane];            \
        an.t1l = wp[((KB + (KBX) + 1) * 2 + 1) * 64 + lane];            \
      }                                                                 \
    } else { /* last k-block: the next layer's first weights */         \
      an.t0h = wp_next[lane];                                           \
      an.t0l = wp_next[64 + lane];                                      \
      an.t1h = wp_next[(kb_next * 2 + 0) * 64 + lane];                  \
      an.t1l = wp_next[(kb_next * 2 + 1) * 64 + lane];                  \
    }                                                                   \
    LOAD_B(KBX, 1, b1h, b1l)                                            \
    __builtin_amdgcn_sched_barrier(0);                                  \
    MFMA6(0, a, b0h, b0l)                                               \
    __builtin_amdgcn_sched_barrier(0);                                  \
    LOAD_B((KBX) + 1 < KB ? (KBX) + 1 : (KBX), 0, b0h, b0l)             \
    __builtin_amdgcn_sched_barrier(0);                                  \
    MFMA6(1, a, b1h, b1l)                                               \
  }
  // a real loop over PAIRS of k-blocks (the A registers alternate by renaming inside the pair);
  // never fully unrolled: ten layers of straight-line k-loops would not fit the instruction cache
  int kb = 1, ri = 0;
#pragma unroll 1
  for (; kb + 1 < KB; kb += 2) {
    KBLOCK_H(kb)
    if (RID::ON) {
      rid.emit(ri);
      rid.read(ri + 1);
      ++ri;
    }
    KBLOCK_H(kb + 1)
  }
  if ((KB - 1) & 1) KBLOCK_H(kb)
  if (RID::ON) {
    for (; ri < RID::NCH; ++ri) {     // (KB = 16: the eighth chunk)
      rid.emit(ri);
      rid.read(ri + 1);
    }
  }
#undef KBLOCK_H
#undef LOAD_B
#undef MFMA6
  __builtin_amdgcn_s_setprio(0);
}
template <int NT, int KBP, int KBH, bool PRE_VIEW>
__device__ __forceinline__ void layer_gemm_h(f32x16 (&acc0)[NT][2], f32x16 (&acc1)[NT][2], AFrag& an,
                                             const half8* __restrict__ wp,
                                             const half8* __restrict__ wp_next, int kb_next,
                                             const _Float16* eh, const _Float16* el,
                                             const _Float16* xh, const _Float16* xl, int lane,
                                             const f32x16* cinit = nullptr) {
  NoRiderH none;
  layer_gemm_h<NT, KBP, KBH, PRE_VIEW, NoRiderH>(acc0, acc1, an, wp, wp_next, kb_next, eh, el, xh, xl, lane, cinit, none);
}

// x = h + l * 2^-11 for the two halves of a packed pair, one v_fma_mix_f32 each (it reads the fp16 operands in
// place; l * 2^-11 is exact, so this is the same single rounding as convert, convert, scale, add - which is
// what hipcc selects under -ffp-contract=off: two conversion-class VALU ops per value instead of none)
__device__ __forceinline__ void join2(unsigned hp, unsigned lp, float& x0, float& x1) {
  const float inv = LINV;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(x0) : "v"(lp), "s"(inv), "v"(hp));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(x1) : "v"(lp), "s"(inv), "v"(hp));
}

// coalesced fp32 copy of the two-plane tile (first ncols columns, optional per-row scale) to
// dst[P][256] as full rows
__device__ __forceinline__ void save_tile_h(const _Float16* xh, const _Float16* xl, float* __restrict__ dst,
                                            int p0, int P, int ncols, const float* row_scale, int tid) {
  const int cpr = ncols >> 3;                           // 8-half chunks per row
  for (int i = tid; i < HM * cpr; i += 256) {
    const int row = i / cpr, c = i - row * cpr;
    if (p0 + row < P) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 vh = *reinterpret_cast<const u32x4*>(xh + x_idx(row, c));
      const u32x4 vl = *reinterpret_cast<const u32x4*>(xl + x_idx(row, c));
      const float sc = row_scale ? row_scale[row] : 1.0f;
      float x[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) join2(vh[k], vl[k], x[2 * k], x[2 * k + 1]);
      const f32x4 o0 = {x[0] * sc, x[1] * sc, x[2] * sc, x[3] * sc}, o1 = {x[4] * sc, x[5] * sc, x[6] * sc, x[7] * sc};
      float* o = dst + (size_t)(p0 + row) * W + 8 * c;
      __builtin_nontemporal_store(o0, reinterpret_cast<f32x4*>(o));       // streamed once: non-temporal
      __builtin_nontemporal_store(o1, reinterpret_cast<f32x4*>(o + 4));
    }
  }
}

// ---- 24-bit saved rows ("f16x3" training, round 5) ----------------------------------------------------------
// The rows a training forward / dgrad chain saves for the weight gradient were fp32 (1 KiB per point and layer):
// the step's weight gradient reads 4.2 GB of them per fine launch AT THE MEMORY SYSTEM'S RATE, so the
// lever is bytes.  A row now leaves AS THE KERNELS HOLD IT IN LDS - the two fp16 planes x ~= h + l * 2^-11 - with the
// low plane rounded (RNE) to 8 bits: an e5m2 number IS the upper byte of an fp16 number, so
//   h   [P][256] fp16 at byte 0       of the row's old slot (the MFMA operand of the weight gradient as it lies)
//   l8  [P][256] e5m2 at byte P * 512 of the slot           (l to 3 significant bits: x to 14, relative error
//                                                            <= 2^-14, zero-mean)
// 768 of the slot's 1024 bytes per point.  The weight gradient then needs NO split on the VALU (the 88
// conversion-class operations per wave and stage that were its compute side): a byte widens to fp16 by one
// v_perm_b32 per pair, which also does the point-pair transposition of the fragment.  dZ rows stay in the dgrad
// chain's PER-POINT scaled domain (gh, gl as they lie; fp16 has no range for the true 1e-8 .. 1e-3 gradients) and
// 1 / s_p is kept per point (rows24_invs_byte: in the unused quarter of dZ slot 0); the weight gradient multiplies
// the pair of a fragment by (S / s_p) as packed fp16 - a power of two.  "f16x3-dgrad" (exact weight gradient) keeps
// fp32 rows.  A first form of this round - fp32 rounded to its upper 24 bits, reassembled by v_perm_b32 in front of
// the unchanged split - cut the forward and dgrad by 7 % each and made the weight gradient 27 % SLOWER (the split's
// conversions + 48 perms + twice the LDS reads: compute-bound): measured, replaced by this form.
constexpr long rows24_l8_byte(long P) { return P * 512; }
constexpr long rows24_invs_byte(long P) { return P * 768; }     // [P] fp32 inside dZ slot 0's unused quarter
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
// the e5m2 bytes of four packed halves (two dwords of the l plane), RNE
__device__ __forceinline__ unsigned l8_of4(unsigned l01, unsigned l23) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  s16x2 q = {0, 0};
  q = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(q, __builtin_bit_cast(half2v, l01), 1.0f, false);
  q = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(q, __builtin_bit_cast(half2v, l23), 1.0f, true);
  return __builtin_bit_cast(unsigned, q);
}
// x = h + l * 2^-11 from an fp16 h (low 16 bits of hbits) and an e5m2 byte (low 8 bits of lbyte)
__device__ __forceinline__ float r24_value(unsigned hbits, unsigned lbyte) {
  float x;
  const unsigned l16 = lbyte << 8;
  const float inv = LINV;
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(x) : "v"(l16), "s"(inv), "v"(hbits));
  return x;
}
// four values of a row from global memory: columns [c4, c4 + 4) of point pt
__device__ __forceinline__ f32x4 r24_load4(const float* slot, long P, size_t pt, int c4) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const unsigned char* b = reinterpret_cast<const unsigned char*>(slot);
  const u32x2 h = *reinterpret_cast<const u32x2*>(b + pt * 512 + c4 * 2);
  const unsigned m = *reinterpret_cast<const unsigned*>(b + rows24_l8_byte(P) + pt * 256 + c4);
  return f32x4{r24_value(h[0] & 0xffffu, m & 0xffu), r24_value(h[0] >> 16, (m >> 8) & 0xffu),
               r24_value(h[1] & 0xffffu, (m >> 16) & 0xffu), r24_value(h[1] >> 16, m >> 24)};
}
// four values given as their planes (vh, vl: the pair the kernels write to LDS) -> the row
__device__ __forceinline__ void r24_store4(float* slot, long P, size_t pt, int c4, const half4& vh, const half4& vl) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned char* b = reinterpret_cast<unsigned char*>(slot);
  const u32x2 h = __builtin_bit_cast(u32x2, vh), l = __builtin_bit_cast(u32x2, vl);
  *reinterpret_cast<u32x2*>(b + pt * 512 + c4 * 2) = h;
  *reinterpret_cast<unsigned*>(b + rows24_l8_byte(P) + pt * 256 + c4) = l8_of4(l[0], l[1]);
}

// The 24-bit row copy of a [64][256] tile as a rider (layer_gemm_h): wave w takes the whole rows [16 w, 16 w + 16)
// - behind the barrier that precedes the k-loop a wave may read any column - so that one store instruction writes two
// complete h rows (1 KiB) and one two complete l8 rows (512 bytes); eight chunks per lane, one per pair of k-blocks.
struct SaveRiderH {
  static constexpr bool ON = true;
  static constexpr int NCH = 8;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const _Float16* xh;
  const _Float16* xl;
  __amdgpu_buffer_rsrc_t rs, rm;
  int row0, voff, voffm, c, rlo;
  u32x4 vh, vl;
  __device__ __forceinline__ void init(const _Float16* xh_, const _Float16* xl_, float* __restrict__ dst, int p0, int P, int wave) {
    xh = xh_; xl = xl_;
    row0 = 16 * wave;
    const unsigned long long pd = reinterpret_cast<unsigned long long>(dst) + (unsigned long long)p0 * 512ull;
    const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)pd), dhi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
    rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)dhi << 32) | dlo), 0,
                                           __builtin_amdgcn_readfirstlane(tile_rows_left(p0, P) * 512u), 0x00020000);
    const unsigned long long pm = reinterpret_cast<unsigned long long>(dst) + (unsigned long long)rows24_l8_byte(P) +
                                  (unsigned long long)p0 * 256ull;
    const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)pm), mhi = __builtin_amdgcn_readfirstlane((unsigned)(pm >> 32));
    rm = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)mhi << 32) | mlo), 0,
                                           __builtin_amdgcn_readfirstlane(tile_rows_left(p0, P) * 256u), 0x00020000);
  }
  __device__ __forceinline__ void begin(int lane) {
    asm volatile("" : "+v"(lane));                      // (not hoisted above the caller's first k-block)
    c = lane & 31;
    rlo = lane >> 5;
    voff = rlo * 512 + c * 16;
    voffm = rlo * 256 + c * 8;
  }
  __device__ __forceinline__ void read(int it) {       // it: wave-uniform
    if (it < NCH) {
      const int r = row0 + 2 * it + rlo;
      const int o = x_idx(r, c);
      vh = *reinterpret_cast<const u32x4*>(xh + o);
      vl = *reinterpret_cast<const u32x4*>(xl + o);
    }
  }
  __device__ __forceinline__ void emit(int it) {
    if (it < NCH) {
      const u32x2 om = {l8_of4(vl[0], vl[1]), l8_of4(vl[2], vl[3])};
      const int r = __builtin_amdgcn_readfirstlane(row0 + 2 * it);
      __builtin_amdgcn_raw_buffer_store_b128(vh, rs, voff, r * 512, 2);          // streamed once: nt
      __builtin_amdgcn_raw_buffer_store_b64(om, rm, voffm, r * 256, 2);
      STORE_DATA_HOLD(vh);             // common.h: read(it + 1)'s row index landed in vh[0] one slot behind the store
    }
  }
};

// the same copy for the NCW columns (from column c0) ONE WAVE has just written (see save_tile_wave in mlp_tile.h), in
// its round-4 form: the lane's chunks are two LDS base addresses + immediates (rows 16 apart share their swizzle),
// the stores are buffer stores with a fixed lane offset, the row part in the scalar offset and the ragged last tile
// left to the descriptor's range check - no address arithmetic, no exec-mask branch per chunk; two chunks in flight
template <int NCW, bool R24 = false>
__device__ __forceinline__ void save_tile_h_wave(const _Float16* xh, const _Float16* xl, float* __restrict__ dst,
                                                 int p0, int P, int c0, const float* row_scale, int lane) {
  static_assert(NCW == 64 || NCW == 32, "a wave owns 64 columns (32 in the views layer)");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  constexpr int CPR = NCW >> 3;                         // 8-half chunks per row of this wave's columns: 8 or 4
  constexpr int RPI = 64 / CPR;                         // rows per wave instruction: 8 or 16
  constexpr int NB = 16 / RPI;                          // swizzle classes: 2 or 1
  constexpr int ITERS = HM / RPI;
  asm volatile("" : "+v"(lane));                        // (addresses recomputed per call, not kept live across the kernel)
  const int lr = lane / CPR, c = (c0 >> 3) + lane % CPR;
  int off[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) off[b] = (RPI * b + lr) * W + ((c ^ (RPI * b + lr)) << 3);      // x_idx(row, c) in halves
  constexpr unsigned PITCH = R24 ? 512u : 1024u;        // bytes per row of the (first) plane
  const unsigned long long pd = reinterpret_cast<unsigned long long>(dst) + (unsigned long long)p0 * PITCH;   // (mlp_tile.h)
  const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)pd), dhi = __builtin_amdgcn_readfirstlane((unsigned)(pd >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)dhi << 32) | dlo), 0,
      __builtin_amdgcn_readfirstlane(tile_rows_left(p0, P) * PITCH), 0x00020000);
  // 24-bit rows: the l8 plane behind the h plane of the slot (rows24_l8_byte)
  const unsigned long long pm = reinterpret_cast<unsigned long long>(dst) + (unsigned long long)rows24_l8_byte(P) +
                                (unsigned long long)p0 * 256ull;
  const unsigned mlo = __builtin_amdgcn_readfirstlane((unsigned)pm), mhi = __builtin_amdgcn_readfirstlane((unsigned)(pm >> 32));
  const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)mhi << 32) | mlo), 0,
      __builtin_amdgcn_readfirstlane(R24 ? tile_rows_left(p0, P) * 256u : 0u), 0x00020000);
  const int voff = R24 ? lr * 512 + c * 16 : lr * 1024 + c * 32;
  const int voffm = lr * 256 + c * 8;
  constexpr int soff = 0;
#pragma unroll
  for (int it0 = 0; it0 < ITERS; it0 += 2) {
    u32x4 vh[2], vl[2];
    float sc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int it = it0 + j, o = off[it % NB] + (it / NB) * 16 * W;
      vh[j] = *reinterpret_cast<const u32x4*>(xh + o);
      vl[j] = *reinterpret_cast<const u32x4*>(xl + o);
      sc[j] = (!R24 && row_scale) ? row_scale[it * RPI + lr] : 1.0f;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int it = it0 + j;
      if (R24) {
        // the planes as they lie (the dgrad chain's rows stay in its per-point scaled domain: row_scale is not applied)
        const u32x2 om = {l8_of4(vl[j][0], vl[j][1]), l8_of4(vl[j][2], vl[j][3])};
        __builtin_amdgcn_raw_buffer_store_b128(vh[j], rs, voff, soff + it * RPI * 512, 2);      // streamed once: nt
        __builtin_amdgcn_raw_buffer_store_b64(om, rm, voffm, soff + it * RPI * 256, 2);
        STORE_DATA_HOLD(vh[j]);                         // (common.h)
      } else {
        float x[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) join2(vh[j][k], vl[j][k], x[2 * k], x[2 * k + 1]);
        u32x4 o0 = {__float_as_uint(x[0] * sc[j]), __float_as_uint(x[1] * sc[j]), __float_as_uint(x[2] * sc[j]), __float_as_uint(x[3] * sc[j])};
        u32x4 o1 = {__float_as_uint(x[4] * sc[j]), __float_as_uint(x[5] * sc[j]), __float_as_uint(x[6] * sc[j]), __float_as_uint(x[7] * sc[j])};
        __builtin_amdgcn_raw_buffer_store_b128(o0, rs, voff, soff + it * RPI * 1024, 2);          // streamed once: nt
        __builtin_amdgcn_raw_buffer_store_b128(o1, rs, voff + 16, soff + it * RPI * 1024, 2);
        STORE_DATA_HOLD(o1);                            // (common.h: the next chunk's join wrote o0[0] one slot behind)
        STORE_DATA_PIN(o0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace scade

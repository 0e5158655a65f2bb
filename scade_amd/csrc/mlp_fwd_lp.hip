// Single-plane 16-bit variant of the fused NeRF MLP forward (opt-in; fp16 or bf16 operands,
// fp32 accumulate, fp32 biases / heads / outputs) - the "bf16 MFMA path" of BASELINE.json
// config 5.  Replaces the same reference chain as mlp_fwd.hip
// (run_scade_scannet.py:48-63, model/run_nerf_helpers.py:142-172, 223-247).
//
// Activations live in LDS as ONE 16-bit plane (h = round(x)), weights are rounded once at
// pack time; every product is a single v_mfma_f32_32x32x16_{f16,bf16} (2.5 PFLOP/s dense).
// Tile shape and the reasons for it: mlp_tile_lp.h.  Accuracy is that of ordinary mixed
// precision (relative 2^-11 per operand for fp16, 2^-8 for bf16), NOT the 1e-4 parity bar:
// the tests hold this path to a PSNR / relative-L2 bound instead and it is never the default.
#include "mlp_tile_lp.h"
#include "mlp_pack.h"

namespace scade {

// forward layer order: pts 0..7, feature, views
constexpr int fwd_rot(int l, int ns) {
  int k = 0;
  for (int i = 0; i < l; ++i) k += kb16(i);
  return k % ns;
}

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct MlpLpArgs {
  const void* packed;     // PACKED_LP_BYTES
  const float* in;        // mode 0: x [P,60];  mode 1: pts [P,3]
  const float* viewdirs;
  const float* bb;
  float* out;             // [P,4]
  unsigned char* acts;    // optional training workspace (mlp_tile_lp.h: lp_acts_bytes(P))
  int P, S, vd_stride;
};

// acc already holds W x + bias (the bias rode in as the first MFMA's C operand): ReLU, round, store.
// bits[t] bit (q*4+p)*4+i <-> value (t,q,p,i) > 0: the (feature, point) map the dgrad kernel's
// output fragment uses too (only computed for the training variant)
template <bool BF, int NT, bool RELU, bool BITS, int NTN, int NPT = LPT>
__device__ __forceinline__ void layer_store_lp(const f32x16 (&acc)[NT][NPT], int ntile0,
                                               typename LP<BF>::T* x, int lane,
                                               unsigned (&bits)[4], f32x16 (&cb)[2],
                                               const float* __restrict__ bias_next, int ntile0_next) {
  const int r = lane & 31, hh = lane >> 5;
  bits[0] = bits[1] = bits[2] = bits[3] = 0u;
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < 2; ++m) {       // row groups q = 2m, 2m + 1: two 16-byte chunks of 8 features
        u32x2 v[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = 2 * m + qq;
          v[qq][0] = pack2<BF, RELU>(acc[t][p][4 * q + 0], acc[t][p][4 * q + 1]);
          v[qq][1] = pack2<BF, RELU>(acc[t][p][4 * q + 2], acc[t][p][4 * q + 3]);
          if (RELU && BITS) {   // sign words (mlp_tile_lp.h): dword d = ((t*4+q)*4+p)*2 + j
            const int d0 = ((t * 4 + q) * 4 + p) * 2;
            bits[d0 >> 4] |= sign_pair(v[qq][0]) << (d0 & 15);
            bits[d0 >> 4] |= sign_pair(v[qq][1]) << ((d0 & 15) + 1);
          }
        }
        // A lane holds features 8q + 4hh .. + 3 of both chunks, i.e. HALF of each 16-byte chunk; as two
        // ds_write_b64 the 16 rows of a lane group share 8 slots of a 128-byte bank line (2-way conflict,
        // 26 % of this kernel's LDS cycles in round 1).  v_permlane32_swap trades the halves between lane
        // r and lane r + 32: the lower lane ends up with ALL of chunk 2m, the upper one with all of chunk
        // 2m + 1, and each stores one conflict-free ds_write_b128.
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned lo, hi;      // (lp_swap_halves: the swap behind the wait states its asm-produced operands need)
          lp_swap_halves(v[0][j], v[1][j], lo, hi);
          w[j] = lo;            // lower lanes: own chunk 2m low half;  upper lanes: the lower lane's chunk 2m+1 low half
          w[2 + j] = hi;        // lower lanes: the upper lane's chunk 2m high half;  upper lanes: own chunk 2m+1 high half
        }
        const int row = p * 32 + r;
        const int c = (ntile0 + t) * 4 + 2 * m + hh;
        *reinterpret_cast<u32x4*>(x + x_idx(row, c)) = w;
      }
    // the next layer's bias values are fetched into the registers point tile 0 just vacated; the
    // rest of this epilogue hides their latency
    if (p == 0 && NTN > 0) {
      if (NTN == 2) load_bias16<2>(cb, bias_next, ntile0_next, lane);
      else load_bias16<1>(reinterpret_cast<f32x16(&)[1]>(cb), bias_next, ntile0_next, lane);
    }
  }
}

__device__ __forceinline__ bool lp_nonfinite(float v) { return !(fabsf(v) <= 3.4028234664e38f); }   // NaN or +-inf

// -DFL_TRACE (variant build; tools/probe_dgrad_trace.py --fwd): core-clock stamps of sixteen consecutive workgroups of a
// 1536-tile launch's third round at the phase boundaries of the format-code-2 training forward
#ifdef FL_TRACE
#ifndef FL_NPT      // which launch is stamped: -DFL_NPT=2 -DFL_GRID=128 -DFL_FIRST=0 = the coarse pass of a 128-ray shard
#define FL_NPT 4
#define FL_GRID 1536
#define FL_FIRST 1100
#endif
#ifndef FL_SAVE     // 2: the format-code-2 training forward; 0: the inference forward (TRACE_RENDER=1)
#define FL_SAVE 2
#endif
__device__ unsigned long long fl_trace[16 * 4 * 48];
#define FL_STAMP(I)                                                                                        \
  if (SAVE == FL_SAVE && NPT == FL_NPT && gridDim.x == FL_GRID && blockIdx.x >= FL_FIRST && blockIdx.x < FL_FIRST + 16 && lane == 0)  \
    fl_trace[((blockIdx.x - FL_FIRST) * 4 + wave) * 48 + (I)] = clock64();
#define FL_STAMP_RT(I)                                                                                     \
  if (SAVE == FL_SAVE && NPT == FL_NPT && gridDim.x == FL_GRID && blockIdx.x >= FL_FIRST && blockIdx.x < FL_FIRST + 16 && lane == 0)  \
    fl_trace[((blockIdx.x - FL_FIRST) * 4 + wave) * 48 + (I)] = wall_clock64();
#else
#define FL_STAMP(I)
#define FL_STAMP_RT(I)
#endif
// SAVE: 0 inference; 1 training, 16-bit rows saved; 2 training, 8-bit (e5m2) rows saved (format code 2)
template <bool BF, int MODE, int SAVE, int NPT>
__global__ __launch_bounds__(256, 2) void mlp_fwd_lp_kernel(MlpLpArgs a) {
  constexpr int LM = 32 * NPT, LPT = NPT, LXPLANE = LM * W;   // this workgroup's tile (shadow the 128-point default)
  typedef typename LP<BF>::T T;
  typedef typename LP<BF>::V8 V8;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
  T* x = reinterpret_cast<T*>(lds16);
  T* e = x + LXPLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  FL_STAMP(0)
  FL_STAMP_RT(46)
  const int p0 = blockIdx.x * LM;
  const int P = a.P;
  const T* __restrict__ wpk = reinterpret_cast<const T*>(a.packed);
  const float* __restrict__ tail = reinterpret_cast<const float*>(wpk + PACKED_LP_ELEMS);
#define TAIL(off) (tail + (int)CE<(off) - OFF_BIAS>::v)
  // (the NaN census of the hidden layers' parameters - pack kernel; used by the heads - is read in front of the
  // alpha head: read here, its wait sat in front of the prologue)

  // ---- prologue: embedding tile [128][64] (57 real channels, zero padded) ------------
  {
    float cx = 0.f, cy = 0.f, cz = 0.f, sc = 1.f;
    if (MODE == 1) { cx = a.bb[0]; cy = a.bb[1]; cz = a.bb[2]; sc = a.bb[3]; }
    for (int i = tid; i < LM * 7; i += 256) {
      const int row = i / 7, c = 57 + (i - row * 7);
      e[e_idx(row, c >> 3) + (c & 7)] = (T)0.f;
    }
    if (MODE == 0) {
      // x rows already hold gamma(x): copy channel pairs (s==0: raw, s>=1: sin/cos of freq s-1)
      for (int i = tid; i < LM * 30; i += 256) {
        const int row = i / 30, rem = i - row * 30;
        const int c = rem / 10, s = rem - c * 10;
        const int pt = min(p0 + row, P - 1);
        const float* xr = a.in + (size_t)pt * 60;
        if (s == 0) {
          e[e_idx(row, 0) + c] = (T)xr[c];
        } else {
          const int col0 = 3 + 6 * (s - 1) + c, col1 = col0 + 3;
          e[e_idx(row, col0 >> 3) + (col0 & 7)] = (T)xr[col0];
          e[e_idx(row, col1 >> 3) + (col1 & 7)] = (T)xr[col1];
        }
      }
    } else {
      // one thread per (point, coordinate): all nine octaves from ONE exact range reduction each.
      // sin(pi x 2^k) = sin(2 pi t) with t = x 2^(k-1) revolutions; the scaling by 2^(k-1) and the
      // fract are exact in fp32, so the hardware v_sin_f32 / v_cos_f32 (argument in revolutions,
      // ~1e-6 absolute) sees an exact argument.  The reference rounds x*pi_f32 first (up to 5e-5 rad
      // off at the top octave); both are far inside the 16-bit rounding of this path.
      // (both items of a thread are fetched before the first sine: the loop form waited for each load in turn)
      constexpr int NIT = (LM * 3 + 255) / 256;
      float xin[NIT];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int i = min(tid + 256 * k, LM * 3 - 1);
        const int row = i / 3, c = i - row * 3;
        xin[k] = a.in[(size_t)min(p0 + row, P - 1) * 3 + c];
      }
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int i = tid + 256 * k;
        if (i < LM * 3) {
          const int row = i / 3, c = i - row * 3;
          const float ctr = c == 0 ? cx : (c == 1 ? cy : cz);
          const float xv = (xin[k] - ctr) * sc;
          e[e_idx(row, 0) + c] = (T)xv;
          float t = 0.5f * xv;
#pragma unroll
          for (int o = 0; o < 9; ++o) {
            const float fr = __builtin_amdgcn_fractf(t);
            const int col0 = 3 + 6 * o + c, col1 = col0 + 3;
            e[e_idx(row, col0 >> 3) + (col0 & 7)] = (T)__builtin_amdgcn_sinf(fr);
            e[e_idx(row, col1 >> 3) + (col1 & 7)] = (T)__builtin_amdgcn_cosf(fr);
            t += t;
          }
        }
      }
    }
  }
  LP_SYNC();
  FL_STAMP(1)
  T* actsT = reinterpret_cast<T*>(a.acts);
  if (SAVE) {   // emb rows [64]: gamma(57) | 0 0 0 | view(3) | 0 - one 16-byte chunk per item
    T* eo = actsT + acts_emb_off(P);
    unsigned char* eo8 = a.acts + acts_emb_off(P) * 2;        // format code 2: fp8 e4m3 rows, 64 bytes per point
    // (a lane keeps its chunk column ch = tid & 7 for all its rows; the view directions of the ch == 7 lanes are
    // fetched for every lane up front - a load inside the divergent block is waited for at the block's end, which
    // chained LM / 32 memory latencies)
    constexpr int EIT = LM * 8 / 256;
    const int ch = tid & 7;
    float vd[EIT][3];
#pragma unroll
    for (int k = 0; k < EIT; ++k) {
      const int pt = min(p0 + (tid >> 3) + 32 * k, P - 1);
#pragma unroll
      for (int c = 0; c < 3; ++c)
        vd[k][c] = MODE == 0 ? a.in[(size_t)pt * 60 + 57 + c] : a.viewdirs[(size_t)(pt / a.S) * a.vd_stride + c];
    }
#pragma unroll
    for (int k = 0; k < EIT; ++k) {
      const int row = (tid >> 3) + 32 * k;
      const int pt = p0 + row;
      V8 v = *reinterpret_cast<const V8*>(e + e_idx(row, ch));   // columns 57..63 of the tile are zero
      if (ch == 7) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[4 + c] = (T)vd[k][c];
      }
      if (pt < P) {
        if (SAVE == 2) {
          lp_u32x2 o;
          if (ch == 0) {
            // columns 0..2 are the raw normalised coordinate (x - centre) * scale: unbounded for a point outside the
            // scene box, where every other column is a sine / cosine / unit-vector component.  e4m3 ends at +-448:
            // kept finite here (the weight gradient of such a point sees the clamped coordinate)
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = (T)__builtin_fminf(__builtin_fmaxf((float)v[c], -448.0f), 448.0f);
          }
          o[0] = lp_pack4_fp8((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
          o[1] = lp_pack4_fp8((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
          __builtin_nontemporal_store(o, reinterpret_cast<lp_u32x2*>(eo8 + (size_t)pt * 64 + 8 * ch));
        } else {
          __builtin_nontemporal_store(v, reinterpret_cast<V8*>(eo + (size_t)pt * 64 + 8 * ch));
        }
      }
    }
  }

  f32x16 acc[2][LPT];
  f32x16 cb[2];           // this lane's bias values of the NEXT layer, loaded one epilogue ahead
  unsigned bits[4];
  // inference: weights fetched three k-blocks ahead; the training variant has no registers left
  // for a fourth set and stays at two
  constexpr int NS = 4;
  AFragN<BF, NS> A;
  const int nt0 = wave * 2;
#define WLBASE(L) (reinterpret_cast<const V8*>(wpk + CE<off_wl(L)>::v) + ((L) == L_VIEWS ? wave : nt0) * (int)CE<kb16(L) * 64>::v)

  // rotation of the A register sets on entry of layer L = (k-blocks of all earlier layers) % 3
#define FROT(L) ((int)CE<fwd_rot(L, NS)>::v)
  // (the 16-bit rows of plain bf16 training ride the same way: SaveRider16)
  typename std::conditional<SAVE == 2 && BF, SaveRider8<NPT>,
                            typename std::conditional<SAVE == 1 && BF, SaveRider16<NPT>, NoRider>::type>::type rid;
  constexpr bool RID16 = SAVE == 1 && BF;
#define PTS_LAYER_L(L, LNEXT, KBP)                                                              \
  {                                                                                             \
    if constexpr ((SAVE == 2 || RID16) && (L) > 0) {   /* the copy of layer L-1's tile rides in this k-loop */ \
      if constexpr (RID16) rid.init(x, actsT + acts_slot_off(P, (L) > 0 ? (L)-1 : 0), p0, P, wave);   \
      else rid.init(x, a.acts + acts_slot_off(P, (L) > 0 ? (L)-1 : 0) * 2, p0, P, nullptr, wave);   \
      layer_gemm_lp<BF, 2, KBP, kbh16(L), false, FROT(L), NS, NPT>(acc, A, WLBASE(L), WLBASE(LNEXT), \
                                                          (int)CE<kb16(LNEXT)>::v, e, x, lane, cb, rid); \
    } else {                                                                                    \
      layer_gemm_lp<BF, 2, KBP, kbh16(L), false, FROT(L), NS, NPT>(acc, A, WLBASE(L), WLBASE(LNEXT), \
                                                          (int)CE<kb16(LNEXT)>::v, e, x, lane, cb); \
    }                                                                                           \
    FL_STAMP(3 + 4 * (L))                                                                 \
    LP_SYNC();                                                                            \
    FL_STAMP(4 + 4 * (L))                                                                 \
    layer_store_lp<BF, 2, true, SAVE != 0, 2, NPT>(acc, nt0, x, lane, bits, cb, TAIL(off_b(LNEXT)), nt0); \
    if (SAVE) {                                                                                \
      u32x4 mw_ = {bits[0], bits[1], bits[2], bits[3]};                                         \
      reinterpret_cast<u32x4*>(a.acts + lp_acts_mask_byte(P))[                                  \
          ((size_t)(L)*gridDim.x + blockIdx.x) * 256 + tid] = mw_;                              \
    }                                                                                           \
    FL_STAMP(5 + 4 * (L))                                                                       \
    if (SAVE == 1 && !RID16) save_tile_lp_wave<BF, 64, NPT>(x, actsT + acts_slot_off(P, L), p0, P, nullptr, 64 * wave, lane); \
    LP_SYNC();                                                                            \
    FL_STAMP(6 + 4 * (L))                                                                 \
  }

  A.s[0].t0 = WLBASE(0)[lane];
  A.s[0].t1 = WLBASE(0)[(int)CE<kb16(0) * 64>::v + lane];
  A.s[1].t0 = WLBASE(0)[64 + lane];
  A.s[1].t1 = WLBASE(0)[(int)CE<kb16(0) * 64>::v + 64 + lane];
  if constexpr (NS == 4) {
    A.s[2].t0 = WLBASE(0)[128 + lane];
    A.s[2].t1 = WLBASE(0)[(int)CE<kb16(0) * 64>::v + 128 + lane];
  }
  load_bias16<2>(cb, TAIL(off_b(0)), nt0, lane);
  FL_STAMP(2)
  PTS_LAYER_L(0, 1, 4)
  PTS_LAYER_L(1, 2, 0)
  PTS_LAYER_L(2, 3, 0)
  PTS_LAYER_L(3, 4, 0)
  PTS_LAYER_L(4, 5, 0)
  PTS_LAYER_L(5, 6, 4)

  // embedding tile is dead: view pad [128][16] (3 real channels)
  for (int i = tid; i < LM * 4; i += 256) {
    const int row = i >> 2, c = i & 3;
    const int pt = min(p0 + row, P - 1);
    float v = 0.f;
    if (c < 3) v = MODE == 0 ? a.in[(size_t)pt * 60 + 57 + c] : a.viewdirs[(size_t)(pt / a.S) * a.vd_stride + c];
    e[row * 16 + c] = (T)v;
#pragma unroll
    for (int j = 1; j < 4; ++j) e[row * 16 + 4 * j + c] = (T)0.f;
  }
  // bf16: the alpha / colour heads run on the MFMA (below); their fp32 weights, split into bf16 high + low parts,
  // go to a small table behind the view pad: [k-block][rows: high.., low.., one zero row][16 k]
  T* tab_a = e + LM * 16;                // alpha:  16 k-blocks x 3 rows
  T* tab_r = tab_a + 16 * 3 * 16;        // colour:  8 k-blocks x 7 rows
  if constexpr (BF) {
    auto split = [](float w, T& h, T& l) {
      h = (T)w;
      const float hf = (float)h;
      l = (fabsf(hf) <= 3.0e38f) ? (T)(w - hf) : (T)0.f;    // (inf: no inf - inf)
    };
    {
      T h, l;
      split(TAIL(OFF_WA)[tid], h, l);
      T* t = tab_a + (tid >> 4) * 48 + (tid & 15);
      t[0] = h; t[16] = l; t[32] = (T)0.f;
    }
    for (int i = tid; i < 384; i += 256) {
      const int c = i >> 7, k = i & 127;
      T h, l;
      split(TAIL(OFF_WR)[i], h, l);
      T* t = tab_r + (k >> 4) * 112 + (k & 15);
      t[c * 16] = h; t[(3 + c) * 16] = l;
      if (c == 0) t[96] = (T)0.f;
    }
  }
  FL_STAMP(35)

  PTS_LAYER_L(6, 7, 0)
  PTS_LAYER_L(7, L_FEAT, 0)
#undef PTS_LAYER_L

  // ---- alpha head on the VALU (fp32 weights) ------------------------------------------
  // The packed 16-bit ReLU (v_pk_max_i16) does not carry NaN the way torch.relu does, so a poisoned
  // point is decided from its INPUTS: a non-finite coordinate makes the reference's whole output row
  // NaN (sin(inf) = NaN feeds every feature), a non-finite view direction its colour.  The
  // reference's isnan/isinf scan (run_scade_scannet.py:747-749) then still sees it.  The six input
  // floats are fetched here so that their latency hides under the alpha dot products.
  // A NaN in a HIDDEN layer's parameters cannot be trusted to survive this path either (a negative-signed
  // NaN is a negative int16 to the packed ReLU), so the pack kernel takes a census of the fp32 parameters and
  // leaves it in the tail as 0 / NaN floats (LP_NAN_TRUNK: pts_linears - the reference then returns NaN in
  // all four outputs of every point; LP_NAN_COLOUR: feature_linear / views_linears - NaN colour, finite
  // density); nan_trunk / nan_colour are read right below.  NaN head
  // parameters need no help: the heads are fp32 arithmetic.
  bool nan_trunk, nan_colour;            // NaN census of the hidden layers' parameters: two wave-uniform flags
  {
    const float ft = TAIL(LP_NAN_TRUNK)[lane], fc = TAIL(LP_NAN_COLOUR)[lane];
    nan_trunk = __builtin_amdgcn_readfirstlane(__any(ft != ft)) != 0;
    nan_colour = __builtin_amdgcn_readfirstlane(__any(fc != fc)) != 0;
  }
  float alpha[LM / 64];
  int badf[LM / 64];
  // bf16: alpha_pre as 16 MFMAs per wave - wave w owns point tile w (row 32 w + (lane & 31)), B fragments are the
  // gemm's own reads of the activation tile, A fragments come from the table (rows 0 / 1 = high / low parts, every
  // other row the zero row): alpha = acc row 0 + acc row 1, in the lanes of the wave's lower half.  (The VALU form
  // below - fp16 - is 64 dependent FMAs per row block and lane.)
  float alpha_v = 0.f;
  int bad_v = 0;
  const bool mine = wave < NPT;          // (64-point workgroups: waves 2, 3 have no point tile)
  if constexpr (BF) {
    const int r = lane & 31, hh = lane >> 5;
    const int row = (mine ? wave : 0) * 32 + r;
    const size_t ptc = (size_t)min(p0 + row, P - 1);
    float q0, q1, q2, v0, v1, v2;
    if (MODE == 1) {
      const float* q = a.in + ptc * 3;
      const float* vd = a.viewdirs + (ptc / a.S) * a.vd_stride;
      q0 = q[0]; q1 = q[1]; q2 = q[2]; v0 = vd[0]; v1 = vd[1]; v2 = vd[2];
    } else {
      const float* q = a.in + ptc * 60;
      q0 = q[0]; q1 = q[1]; q2 = q[2]; v0 = q[57]; v1 = q[58]; v2 = q[59];
    }
    if (mine) {
      const T* ta = tab_a + (r < 2 ? r : 2) * 16 + 8 * hh;
      f32x16 ha = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const V8 bfr = *reinterpret_cast<const V8*>(x + x_idx(row, 2 * kb + hh));
        const V8 afr = *reinterpret_cast<const V8*>(ta + kb * 48);
        ha = LP<BF>::mfma(afr, bfr, ha);
      }
      alpha_v = (ha[0] + ha[1]) + TAIL(OFF_BA)[0];
    }
    const bool badp = lp_nonfinite(q0) | lp_nonfinite(q1) | lp_nonfinite(q2) | nan_trunk;
    bad_v = badp | lp_nonfinite(v0) | lp_nonfinite(v1) | lp_nonfinite(v2) | nan_colour;
    if (badp) alpha_v = __builtin_nanf("");
    if (SAVE && mine && hh == 0 && p0 + row < P)
      reinterpret_cast<float*>(a.acts + lp_acts_alpha_byte(P))[p0 + row] = alpha_v;
  } else {
    const float* wa = TAIL(OFF_WA);
    // this lane's 64 alpha weights (its four-column-chunk stride is the same for every row block): loaded once
    f32x4 waq[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      waq[2 * i] = *reinterpret_cast<const f32x4*>(wa + (i * 4 + (tid & 3)) * 8);
      waq[2 * i + 1] = *reinterpret_cast<const f32x4*>(wa + (i * 4 + (tid & 3)) * 8 + 4);
    }
#pragma unroll
    for (int rb = 0; rb < LM / 64; ++rb) {
      const int row = rb * 64 + (tid >> 2), sub = tid & 3;
      const size_t ptc = (size_t)min(p0 + row, P - 1);
      float q0, q1, q2, v0, v1, v2;
      if (MODE == 1) {
        const float* q = a.in + ptc * 3;
        const float* vd = a.viewdirs + (ptc / a.S) * a.vd_stride;
        q0 = q[0]; q1 = q[1]; q2 = q[2]; v0 = vd[0]; v1 = vd[1]; v2 = vd[2];
      } else {
        // x rows hold gamma(x) = [x, sin, cos, ...]: a non-finite coordinate shows in the raw columns
        const float* q = a.in + ptc * 60;
        q0 = q[0]; q1 = q[1]; q2 = q[2]; v0 = q[57]; v1 = q[58]; v2 = q[59];
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c = i * 4 + sub;
        const V8 v = *reinterpret_cast<const V8*>(x + x_idx(row, c));
#pragma unroll
        for (int j = 0; j < 8; ++j) s = fmaf((float)v[j], waq[2 * i + (j >> 2)][j & 3], s);
      }
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      alpha[rb] = s + TAIL(OFF_BA)[0];
      const bool badp = lp_nonfinite(q0) | lp_nonfinite(q1) | lp_nonfinite(q2) | nan_trunk;
      badf[rb] = badp | lp_nonfinite(v0) | lp_nonfinite(v1) | lp_nonfinite(v2) | nan_colour;
      if (badp) alpha[rb] = __builtin_nanf("");
      if (SAVE && sub == 0 && p0 + row < P)
        reinterpret_cast<float*>(a.acts + lp_acts_alpha_byte(P))[p0 + row] = alpha[rb];
    }
  }

  // ---- feature_linear ------------------------------------------------------------------
  FL_STAMP(36)
  if constexpr (SAVE == 2) rid.init(x, a.acts + acts_slot_off(P, 7) * 2, p0, P, nullptr, wave);
  if constexpr (RID16) rid.init(x, actsT + acts_slot_off(P, 7), p0, P, wave);
  layer_gemm_lp<BF, 2, 0, 16, false, FROT(L_FEAT), NS, NPT>(acc, A, WLBASE(L_FEAT), WLBASE(L_VIEWS), 0, e, x, lane, cb, rid);
  FL_STAMP(37)
  LP_SYNC();
  FL_STAMP(38)
  layer_store_lp<BF, 2, false, false, 1, NPT>(acc, nt0, x, lane, bits, cb, TAIL(off_b(L_VIEWS)), wave);
  FL_STAMP(39)
  if (SAVE == 1 && !RID16) save_tile_lp_wave<BF, 64, NPT>(x, actsT + acts_slot_off(P, SLOT_FEAT), p0, P, nullptr, 64 * wave, lane);
  LP_SYNC();
  FL_STAMP(40)

  // ---- views layer: [view pad | feature] -> 128, ReLU --------------------------------
  {
    f32x16 av[1][LPT];
    if constexpr (SAVE == 2) rid.init(x, a.acts + acts_slot_off(P, SLOT_FEAT) * 2, p0, P, nullptr, wave);
    if constexpr (RID16) rid.init(x, actsT + acts_slot_off(P, SLOT_FEAT), p0, P, wave);
    layer_gemm_lp<BF, 1, 1, 16, true, FROT(L_VIEWS), NS, NPT>(av, A, WLBASE(L_VIEWS), WLBASE(L_VIEWS), 0, e, x, lane, cb, rid);
    FL_STAMP(41)
    LP_SYNC();
    FL_STAMP(42)
    layer_store_lp<BF, 1, true, SAVE != 0, 0, NPT>(av, wave, x, lane, bits, cb, nullptr, 0);
    if (SAVE) {   // sign words of the views layer (index 8): the dgrad's head epilogue masks with them
      u32x4 mw_ = {bits[0], bits[1], 0u, 0u};
      reinterpret_cast<u32x4*>(a.acts + lp_acts_mask_byte(P))[((size_t)8 * gridDim.x + blockIdx.x) * 256 + tid] = mw_;
    }
    FL_STAMP(43)
    // (format code 2 keeps THIS slot 16-bit: the 128-wide views hidden layer is what the dgrad kernel derives the
    // views ReLU mask from - an activation below e5m2's range must not read as "inactive" - and 256 bytes per
    // point either way)
    if (SAVE) save_tile_lp_wave<BF, 32, NPT>(x, actsT + acts_slot_off(P, SLOT_VIEWS_H), p0, P, nullptr, 32 * wave, lane);
    LP_SYNC();
    FL_STAMP(44)
  }
#undef WLBASE

  // ---- rgb head + softplus ---------------------------------------------------------------
  if constexpr (BF) {
    // 8 MFMAs per wave on the views tile: table rows 0..2 = high parts of the three colour rows, 3..5 = low parts
    // (as A rows 0..2 and 8..10: both land in the lower half-wave's registers 0..2 and 4..6), row 6 = zeros
    const int r = lane & 31, hh = lane >> 5;
    const int row = (mine ? wave : 0) * 32 + r;
    if (mine) {
      const int tr = r < 3 ? r : ((r >= 8 && r < 11) ? r - 5 : 6);
      const T* tb = tab_r + tr * 16 + 8 * hh;
      f32x16 hr = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        const V8 bfr = *reinterpret_cast<const V8*>(x + x_idx(row, 2 * kb + hh));
        const V8 afr = *reinterpret_cast<const V8*>(tb + kb * 112);
        hr = LP<BF>::mfma(afr, bfr, hr);
      }
      if (hh == 0 && p0 + row < P) {
        const float* br = TAIL(OFF_BR);
        const float al = alpha_v;
        const float bx = al * 10.f;
        const float sp = bx > 20.f ? al : log1pf(expf(bx)) / 10.f;
        f32x4 o = {(hr[0] + hr[4]) + br[0], (hr[1] + hr[5]) + br[1], (hr[2] + hr[6]) + br[2], sp};
        if (bad_v) {          // poisoned inputs (see the alpha head): colour NaN; density already is
          const float qn = __builtin_nanf("");
          o[0] = o[1] = o[2] = qn;
        }
        *reinterpret_cast<f32x4*>(a.out + (size_t)(p0 + row) * 4) = o;
      }
    }
  } else {
    const float* wr = TAIL(OFF_WR);
    const float* br = TAIL(OFF_BR);
    // this lane's 3 x 32 colour weights: loaded once for all its row blocks (the accumulators' registers are free)
    f32x4 wq[3][8];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wq[o][2 * i] = *reinterpret_cast<const f32x4*>(wr + o * 128 + (i * 4 + (tid & 3)) * 8);
        wq[o][2 * i + 1] = *reinterpret_cast<const f32x4*>(wr + o * 128 + (i * 4 + (tid & 3)) * 8 + 4);
      }
    const float br0 = br[0], br1 = br[1], br2 = br[2];
#pragma unroll
    for (int rb = 0; rb < LM / 64; ++rb) {
      const int row = rb * 64 + (tid >> 2), sub = tid & 3;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = i * 4 + sub;
        const V8 v = *reinterpret_cast<const V8*>(x + x_idx(row, c));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xv = (float)v[j];
          s0 = fmaf(xv, wq[0][2 * i + (j >> 2)][j & 3], s0);
          s1 = fmaf(xv, wq[1][2 * i + (j >> 2)][j & 3], s1);
          s2 = fmaf(xv, wq[2][2 * i + (j >> 2)][j & 3], s2);
        }
      }
      s0 += __shfl_xor(s0, 1, 64); s0 += __shfl_xor(s0, 2, 64);
      s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
      s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
      if (sub == 0 && p0 + row < P) {
        const float al = alpha[rb];
        const float bx = al * 10.f;
        const float sp = bx > 20.f ? al : log1pf(expf(bx)) / 10.f;
        f32x4 o = {s0 + br0, s1 + br1, s2 + br2, sp};
        if (badf[rb]) {       // poisoned inputs (see the alpha head): colour NaN; density already is
          const float qn = __builtin_nanf("");
          o[0] = o[1] = o[2] = qn;
        }
        *reinterpret_cast<f32x4*>(a.out + (size_t)(p0 + row) * 4) = o;
      }
    }
  }
  FL_STAMP(45)
  FL_STAMP_RT(47)
#undef TAIL
}

// ---------------------------------------------------------------------------
// pack: fp32 parameters -> one 16-bit plane in fragment order + fp32 tail
// ---------------------------------------------------------------------------
struct PackLpArgs {
  const float* p[N_PARAM_TENSORS];
  void* packed;
};

template <bool BF>
__global__ void mlp_pack_lp_kernel(PackLpArgs a) { pack_lp_row<BF>(a.p, a.packed, blockIdx.y, blockIdx.x, gridDim.x); }

}  // namespace scade

using namespace scade;

#ifdef FL_TRACE
extern "C" int scade_debug_fl_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scade::fl_trace), sizeof(unsigned long long) * 16 * 4 * 48);
}
#endif
extern "C" long scade_mlp_packed_lp_bytes(void) { return PACKED_LP_BYTES; }
extern "C" long scade_mlp_acts_lp_bytes(long P) { return lp_acts_bytes(P); }

extern "C" int scade_mlp_pack_lp(const float* const* params, void* packed, int bf16, void* stream) {
  SCADE_REQUIRE(params && packed, -1, "scade_mlp_pack_lp: null pointer");
  PackLpArgs a;
  for (int i = 0; i < N_PARAM_TENSORS; ++i) {
    SCADE_REQUIRE(params[i], -1, "scade_mlp_pack_lp: params[%d] is null", i);
    a.p[i] = params[i];
  }
  a.packed = packed;
  if (bf16) hipLaunchKernelGGL(mlp_pack_lp_kernel<true>, dim3(PACK_BLOCKS, PACK_FWD_ROWS), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(mlp_pack_lp_kernel<false>, dim3(PACK_BLOCKS, PACK_FWD_ROWS), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_mlp_pack_lp");
}

template <bool BF, int MODE, int SAVE, int NPT>
static int launch_lp_pt(const MlpLpArgs& a, hipStream_t s) {
  static unsigned long long attr_set = 0;   // one bit per device ordinal
  auto kern = mlp_fwd_lp_kernel<BF, MODE, SAVE, NPT>;
  if (scade_attr_needed(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lp_lds_bytes(NPT));
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_fwd_lp: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(attr_set);
  }
  hipLaunchKernelGGL(kern, dim3((a.P + 32 * NPT - 1) / (32 * NPT)), dim3(256), lp_lds_bytes(NPT), s, a);
  return scade_check_launch("scade_mlp_fwd_lp");
}
template <bool BF, int MODE, int SAVE>
static int launch_lp(const MlpLpArgs& a, hipStream_t s) {
  return lp_pick_point_tiles(a.P) == 2 ? launch_lp_pt<BF, MODE, SAVE, 2>(a, s)
                                       : launch_lp_pt<BF, MODE, SAVE, 4>(a, s);
}

extern "C" int scade_mlp_fwd_lp(const void* packed_lp, int bf16, int mode, const float* in,
                                const float* viewdirs, int vd_stride, const float* bb, int P, int S,
                                float* out, void* acts, void* stream) {
  if (P == 0) return 0;
  SCADE_REQUIRE(packed_lp && in && out, -1, "scade_mlp_fwd_lp: null pointer");
  SCADE_REQUIRE(mode == 0 || mode == 1, -2, "scade_mlp_fwd_lp: mode must be 0 or 1");
  SCADE_REQUIRE(bf16 >= 0 && bf16 <= 2, -2, "scade_mlp_fwd_lp: format 0 (fp16), 1 (bf16) or 2 (bf16, 8-bit saved rows)");
  if (mode == 1) {
    SCADE_REQUIRE(viewdirs && bb && vd_stride >= 3, -1, "scade_mlp_fwd_lp: mode 1 needs viewdirs and bb");
    SCADE_REQUIRE(S > 0 && P % S == 0, -2, "scade_mlp_fwd_lp: P must be a multiple of S");
  }
  MlpLpArgs a{packed_lp, in, viewdirs, bb, out, reinterpret_cast<unsigned char*>(acts), P, S, vd_stride};
  hipStream_t s = (hipStream_t)stream;
  if (bf16 == 2 && acts) return mode ? launch_lp<true, 1, 2>(a, s) : launch_lp<true, 0, 2>(a, s);
  const int sel = (bf16 ? 4 : 0) + (mode ? 2 : 0) + (acts ? 1 : 0);
  switch (sel) {
    case 0: return launch_lp<false, 0, 0>(a, s);
    case 1: return launch_lp<false, 0, 1>(a, s);
    case 2: return launch_lp<false, 1, 0>(a, s);
    case 3: return launch_lp<false, 1, 1>(a, s);
    case 4: return launch_lp<true, 0, 0>(a, s);
    case 5: return launch_lp<true, 0, 1>(a, s);
    case 6: return launch_lp<true, 1, 0>(a, s);
    default: return launch_lp<true, 1, 1>(a, s);
  }
}

// Split-precision (f16x3) variant of the dgrad chain of mlp_bwd.hip (opt-in training mode).
//
// Same structure as mlp_dgrad_kernel -- 64-point tile, 4 waves, the gradient tile walks the 9
// layers backwards in LDS, transposed weight pack as the A operand, ReLU masks from the lane-
// private sign words, every layer's dZ written to HBM as fp32 rows for the (exact fp32) wgrad --
// but the tile is carried as two fp16 planes (x ~= h + l*2^-11) and every product is three
// v_mfma_f32_32x32x16_f16 (see mlp_fwd_f16.hip).
//
// Gradients are small (1e-8 .. 1e-3) and would underflow fp16, so each POINT's gradient chain
// is scaled by its own power of two s_p = 2^(-4 - exponent(max|g_out[p]|, |d alpha_pre[p]|)):
// the chain is linear in g_out[p] and the product only mixes features, never points, so the
// scale factors out exactly and is removed when the fp32 rows are stored.
#include "mlp_pack.h"       // (mlp_tile_f16.h + the pack rows shared with scade_mlp_pack_step)
#include <type_traits>
#include "mlp_wgrad.h"
#include "mlp_reduce.h"

namespace scade {

struct PackTF16Args {
  const float* p[N_PARAM_TENSORS];
  _Float16* packed;
};

__global__ void mlp_pack_t_f16_kernel(PackTF16Args a) { pack_t_f16_row(a.p, a.packed, blockIdx.y, blockIdx.x, gridDim.x); }

__global__ void zero_word_kernel(unsigned int* w0, unsigned int* w1) {
  if (threadIdx.x == 0) { *w0 = 0u; if (w1) *w1 = 0u; }
}

struct MlpDgradF16Args {
  const float* packed;       // fp32 forward pack (rgb / alpha head weights)
  const _Float16* packedT;   // transposed two-plane pack
  const float* acts;
  const float* g_out;        // [P,4]
  float* dz;                 // dz_floats(P)
  unsigned int* gmax;        // launch-wide max of |g_out| (float bits; zeroed by zero_word_kernel before the launch)
  int P;
};

// one or two network calls in one launch (the coarse + fine NeRF of a train step): workgroups [0, tiles0) take n[0]
struct MlpDgradF16Args2 {
  MlpDgradF16Args n[2];
  int tiles0;
};

template <bool MASK, bool ADD_ALPHA>
__device__ __forceinline__ void dgrad_store_h(const f32x16 (&acc0)[2][2], const f32x16 (&acc1)[2][2],
                                              int ktile0, _Float16* gh, _Float16* gl,
                                              unsigned long long bits, const float* __restrict__ w_a,
                                              const float* dal_scaled, int lane) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = (ktile0 + t) * 32 + 8 * q + 4 * hh;
      f32x4 wa = {0.f, 0.f, 0.f, 0.f};
      if (ADD_ALPHA) wa = *reinterpret_cast<const f32x4*>(w_a + f);
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = p * 32 + r;
        half4 vh, vl;
        float xs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = fmaf(acc1[t][p][4 * q + i], LINV, acc0[t][p][4 * q + i]);
          if (ADD_ALPHA) x = x + wa[i] * dal_scaled[row];
          // (the mask as a sign-extended bit field ANDed onto the value: two operations, not and + compare + select)
          if (MASK) {
            const unsigned wd = (unsigned)(bits >> (p * 32));
            const int m = __builtin_amdgcn_sbfe((int)wd, (t * 4 + q) * 4 + i, 1);
            x = __uint_as_float(__float_as_uint(x) & (unsigned)m);
          }
          xs[i] = x;
        }
        split4(xs, vh, vl);
        const int o = x_idx(row, f >> 3) + (f & 7);
        *reinterpret_cast<half4*>(gh + o) = vh;
        *reinterpret_cast<half4*>(gl + o) = vl;
      }
    }
}

// R24: the dZ rows leave as 24-bit rows (mlp_tile_f16.h) and the saved views-layer rows (the ReLU mask of the heads)
// are read as such - the split-precision weight gradient follows; false: fp32 rows for the exact weight gradient
template <bool R24>
__global__ __launch_bounds__(256, 2) void mlp_dgrad_f16_kernel(MlpDgradF16Args2 aa) {
  const bool second = (int)blockIdx.x >= aa.tiles0;           // wave-uniform
  const MlpDgradF16Args& a = second ? aa.n[1] : aa.n[0];
  const int bid = (int)blockIdx.x - (second ? aa.tiles0 : 0);
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
  _Float16* gh = ldsh;
  _Float16* gl = ldsh + XPLANE;
  float* dal = reinterpret_cast<float*>(ldsh + 2 * XPLANE);   // [64] d alpha_pre * scale
  float* inv_s = dal + 64;                                    // [64] 1/scale of the point
  float* wmx = inv_s + 64;                                    // [4] per-wave max |g_out|

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p0 = bid * HM;
  const int P = a.P;
  const float* __restrict__ pk = a.packed;
  const _Float16* __restrict__ pt_ = a.packedT;
  const float* __restrict__ acts = a.acts;
  float* __restrict__ dz = a.dz;
  auto mask_of = [&](int layer) { return load_relu_words<2>(acts, P, layer, tid, bid); };

  // ---- heads: d alpha_pre, per-point scale, dZ of the views layer ------------------------
  {
    // (round 4: no guarded loads - the compiler closes a divergent block with a wait for its loads, which chained
    // ten memory latencies per tile; rows past P read row P - 1 and are zeroed by selects.  Same arithmetic.)
    const int row = tid >> 2, sub = tid & 3;
    const int pt = p0 + row;
    const bool ok = pt < P;
    const size_t ptc = (size_t)min(pt, P - 1);
    const float* hv = acts + acts_slot_off(P, SLOT_VIEWS_H);
    f32x4 g = *reinterpret_cast<const f32x4*>(a.g_out + ptc * 4);
    const float apre = acts[acts_alpha_off(P) + ptc];
    f32x4 mks[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      mks[i] = R24 ? r24_load4(hv, P, ptc, (i * 4 + sub) * 4) : *reinterpret_cast<const f32x4*>(hv + ptc * W + (i * 4 + sub) * 4);
    if (!ok) g = f32x4{0.f, 0.f, 0.f, 0.f};
    float da = 0.f;
    if (ok) {
      const float bx = apre * 10.f;
      da = bx > 20.f ? g[3] : g[3] / (1.f + expf(-bx));
    }
    const float m = fmaxf(fmaxf(fabsf(g[0]), fabsf(g[1])), fmaxf(fabsf(g[2]), fabsf(da)));
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
      int e;
      frexpf(m, &e);                       // m = f * 2^e, f in [0.5, 1)
      s = ldexpf(1.f, min(-4 - e, 96));    // denormal gradients: keep the scale finite
    }
    if (sub == 0) {
      if (ok) dz[dz_dalpha_off(P) + pt] = da;
      dal[row] = da * s;
      inv_s[row] = 1.f / s;
      // 24-bit rows stay in the per-point scaled domain: the weight gradient needs 1 / s_p (mlp_tile_f16.h)
      if (R24 && ok) reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(dz) + rows24_invs_byte(P))[pt] = 1.f / s;
    }
    {   // launch-wide max for the weight-gradient kernel's dZ scale: one atomic per WORKGROUP (after the
        // barrier below; 12,000 per-wave atomics on one address were a serial tail of the launch)
      float wm = (m < 3.0e38f) ? m : 0.f;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o, 64));
      if (lane == 0) wmx[wave] = wm;
    }
    const float* wr = pk + OFF_WR;
    float* dzv = dz + acts_slot_off(P, SLOT_VIEWS_H);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int chunk = i * 4 + sub;                       // 4-float chunk of the 128 columns
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + chunk * 4);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + 128 + chunk * 4);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 256 + chunk * 4);
      const f32x4 mk = mks[i];
      f32x4 v;
      half4 vh, vl;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = g[0] * w0[j] + g[1] * w1[j] + g[2] * w2[j];
        v[j] = (mk[j] > 0.f && ok) ? d : 0.f;
        _Float16 h, l;
        split2(v[j] * s, h, l);
        vh[j] = h; vl[j] = l;
      }
      const int o = x_idx(row, chunk >> 1) + (chunk & 1) * 4;
      *reinterpret_cast<half4*>(gh + o) = vh;
      *reinterpret_cast<half4*>(gl + o) = vl;
      if (ok) {
        if (R24) r24_store4(dzv, P, (size_t)pt, chunk * 4, vh, vl);
        else *reinterpret_cast<f32x4*>(dzv + (size_t)pt * W + chunk * 4) = v;
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    const float wm = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
    if (wm > 0.f) atomicMax(a.gmax, __float_as_uint(wm));
  }

  f32x16 acc0[2][2], acc1[2][2];
  AFrag an;
  const int kt0 = wave * 2;
  // this wave's k-tile pair of dgrad index T: [kt][NB16][2][64] half8
#define WT16(T, NB) (reinterpret_cast<const half8*>(pt_ + off_wt16(T)) + kt0 * (NB) * 128)
  an.t0h = WT16(8, 8)[lane];
  an.t0l = WT16(8, 8)[64 + lane];
  an.t1h = WT16(8, 8)[(8 * 2 + 0) * 64 + lane];
  an.t1l = WT16(8, 8)[(8 * 2 + 1) * 64 + lane];

  // R24: the 24-bit copy of a layer's dZ tile rides in the NEXT gemm's k-loop (SaveRiderH, mlp_tile_f16.h); the last
  // tile (slot 0) and the fp32-row mode keep the burst copy
  SaveRiderH rid;
  // ---- views layer: d feature = Wv[:, :256]^T dZv  (reduction over 128 = 8 k16-blocks) ----
  layer_gemm_h<2, 0, 8, false>(acc0, acc1, an, WT16(8, 8), WT16(7, 16), 16, gh, gl, gh, gl, lane);
  __syncthreads();
  dgrad_store_h<false, false>(acc0, acc1, kt0, gh, gl, 0ull, nullptr, dal, lane);
  if (!R24) save_tile_h_wave<64, false>(gh, gl, dz + acts_slot_off(P, SLOT_FEAT), p0, P, 64 * wave, inv_s, lane);
  __syncthreads();

  // ---- feature layer: d h7 = Wf^T d feature + w_alpha * d alpha_pre, mask h7 ---------------
  unsigned long long mbits = mask_of(7);
  if constexpr (R24) {
    rid.init(gh, gl, dz + acts_slot_off(P, SLOT_FEAT), p0, P, wave);
    layer_gemm_h<2, 0, 16, false, SaveRiderH>(acc0, acc1, an, WT16(7, 16), WT16(6, 16), 16, gh, gl, gh, gl, lane, nullptr, rid);
  } else {
    layer_gemm_h<2, 0, 16, false>(acc0, acc1, an, WT16(7, 16), WT16(6, 16), 16, gh, gl, gh, gl, lane);
  }
  __syncthreads();
  dgrad_store_h<true, true>(acc0, acc1, kt0, gh, gl, mbits, pk + OFF_WA, dal, lane);
  if (!R24) save_tile_h_wave<64, false>(gh, gl, dz + acts_slot_off(P, 7), p0, P, 64 * wave, inv_s, lane);
  __syncthreads();

#define DGRAD_LAYER_H(L)                                                                         \
  mbits = mask_of((L)-1);                                                                        \
  if constexpr (R24) {                                                                           \
    rid.init(gh, gl, dz + acts_slot_off(P, L), p0, P, wave);                                     \
    layer_gemm_h<2, 0, 16, false, SaveRiderH>(acc0, acc1, an, WT16((L)-1, 16), WT16((L) > 1 ? (L)-2 : 0, 16), 16, \
                                              gh, gl, gh, gl, lane, nullptr, rid);               \
  } else {                                                                                       \
    layer_gemm_h<2, 0, 16, false>(acc0, acc1, an, WT16((L)-1, 16), WT16((L) > 1 ? (L)-2 : 0, 16), 16, \
                                  gh, gl, gh, gl, lane);                                         \
  }                                                                                              \
  __syncthreads();                                                                               \
  dgrad_store_h<true, false>(acc0, acc1, kt0, gh, gl, mbits, nullptr, dal, lane);                \
  if (!R24 || (L) == 1)                                                                          \
    save_tile_h_wave<64, R24>(gh, gl, dz + acts_slot_off(P, (L)-1), p0, P, 64 * wave, inv_s, lane); \
  __syncthreads();

  DGRAD_LAYER_H(7)
  DGRAD_LAYER_H(6)
  DGRAD_LAYER_H(5)
  DGRAD_LAYER_H(4)
  DGRAD_LAYER_H(3)
  DGRAD_LAYER_H(2)
  DGRAD_LAYER_H(1)
#undef DGRAD_LAYER_H
#undef WT16
}

constexpr int DGRAD_F16_LDS_BYTES = 2 * XPLANE * 2 + 128 * 4 + 16;   // + the four waves' gradient maxima

// ---------------------------------------------------------------------------
// B2': split-precision weight gradient.  dW[n][k] = sum_points dZ[p][n] * In[p][k] contracts over POINTS,
// so both MFMA operands need 8 consecutive points of ONE column per lane, while both matrices lie
// point-major in HBM.  Second design (the first loaded 4-byte columns into registers, one 16-point stage
// ahead: 32 KB per CU in flight, latency-bound at 2.4 TB/s):
//   * the fp32 tiles go HBM -> LDS by LDS-DMA as they lie (1-KiB point rows), into a ring of 4 slots of 16
//     points: three stages = 96 KB per CU in flight, no staging registers, rows past the chunk end arrive
//     as zeros (buffer range check) - the ring of mlp_wgrad2.hip;
//   * the fragments are gathered with the OUTPUT-ROW PERMUTATION of that kernel: lane r reads 8 (dZ) / 16
//     (input) consecutive bytes of each of its 8 points - features 2r, 2r+1 (4r..4r+3), the operands of two
//     (four) different output tiles - so MFMA row r of tile t is feature 2r + t, column r of tile u is input
//     4r + u; 16 conflict-free LDS reads per 24 MFMAs, and the split into fp16 planes happens on the VALU
//     behind the read (each value is split by the 2 / 4 waves that use it: 88 hand-selected VALU ops per
//     wave and stage - v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16 - beside 24 MFMAs).  Knock-outs: the memory
//     system alone would take 0.53 ms over the train step's launch mix, the compute phases alone 0.57 ms
//     (the MFMAs themselves are free: the conversion-class VALU ops run at a fraction of the plain VALU
//     rate and both waves of a SIMD do them in lock step), the kernel 0.63 ms;
//   x = h + l' with l' = fp16(x - h) UNSCALED (may be fp16-subnormal: absolute error <= 2^-25 of the operand
//   scale), so ONE fp32 accumulator set (128 VGPRs for the wave's 64 x 128 outputs) takes h*h + h*l' + l'*h.
//   dZ is multiplied by ONE power of two per launch (from max|g_out|, found by the dgrad kernel) so its
//   large entries sit near 2^8; the partial is scaled back exactly.
// The VALU riders (bias / alpha head / view columns) use the exact fp32 values of the published slot; their
// per-point scalars (d alpha, view direction) ride the ring as 4-byte LDS-DMA pieces.
// ---------------------------------------------------------------------------
constexpr int HW_PT = 16;                                 // points per ring slot = one k16 block
constexpr int HW_D = 6;                                   // ring slots: three PAIRS of stages (see the stage loop)
// bytes of a slot: dZ h [16][256] fp16 | dZ l8 [16][256] e5m2 | input h | input l8 (KW = 64: the fp32 embedding
// rows [16][64] instead) | d alpha [16] fp32 | view dirs [16][3] fp32 | 1 / s_p [16] fp32
constexpr int HW_DZ_HI = 0, HW_DZ_MID = HW_PT * 512, HW_IN_HI = HW_PT * 768, HW_IN_MID = HW_PT * 1280;
constexpr int HW_SCAL = HW_PT * 1536;                     // d alpha [16] | view dirs [16][3] | 1 / s_p [16]
constexpr int HW_SLOT = HW_SCAL + 320;                    // 24,896
constexpr int WGRAD_F16_RED_BYTES = (2 * 256 * 4 + 8 + 4 * 256 + 4) * 4;   // the riders' reduction scratch
constexpr int WGRAD_F16_LDS_BYTES = HW_D * HW_SLOT > WGRAD_F16_RED_BYTES ? HW_D * HW_SLOT : WGRAD_F16_RED_BYTES;   // 124,480

struct WgradF16Args {
  WgradArgs w;
  const float* gmax;    // device scalar: max |g_out| of the launch (float bits)
};

typedef __amdgpu_buffer_rsrc_t hw_rsrc_t;
__device__ __forceinline__ hw_rsrc_t hw_make_rsrc(const float* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// workgroup barrier that orders LDS only (lgkmcnt(0), never vmcnt(0): the ring stays in flight); a raw
// s_barrier behind an explicit wait - a local-address-space release fence would be turned into
// s_waitcnt vmcnt(0) by hipcc whenever an LDS-DMA is outstanding and no LDS read is
__device__ __forceinline__ void hw_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

#ifdef HW_TRACE
// -DHW_TRACE (variant build, SCADE_AB_FLAGS): per workgroup and wave of the LAST launch, sums over the ring stages of
// {wait for the stage's data + barrier, DMA issue, compute, stages} on the 100 MHz wall clock (tools/probe_wgrad_f16_trace.py)
__device__ unsigned long long hw_dbg[4096 * 8 * 4];
#define HW_T() wall_clock64()
#endif
// RID: the job's rider as a COMPILE-TIME choice (0 none, WF_ALPHA the alpha head, WF_VIEWCOLS the view columns): as
// run-time branches inside the stage loop the two rare riders cost EVERY 256-wide job its schedule (+6 ... +9 % on the
// launch mix with the alpha head's dot products behind a wave-uniform flag)
template <int KW, int RID = 0>
__device__ __forceinline__ void wgrad_f16_job(const WgradArgs& a, const WgradJob& jb, float* lds_f,
                                              int c0, int c1, float S, float* __restrict__ out) {
  constexpr int NKT = KW == 256 ? 4 : 1;
  constexpr int D = HW_D, PT = HW_PT;
  // LDS-DMA instructions per wave and stage: the tiles, 1 / s_p, and the rider scalars only in the job that has the rider
  constexpr int NI = (KW == 256 ? 3 : 2) + 1 + (RID == WF_ALPHA ? 1 : 0) + (RID == WF_VIEWCOLS ? 1 : 0);
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  unsigned char* lds = reinterpret_cast<unsigned char*>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int n0 = (wave >> 1) * 64;
  const int k0 = (wave & 1) * (KW / 2);
  const bool active = n0 < jb.n_rows;
  const int P = a.P;
  const int npts = c1 - c0;
  // 24-bit rows (mlp_tile_f16.h): the h plane of a slot at its float offset, the l8 plane P * 512 bytes on; the dZ
  // rows are in the dgrad chain's per-point scaled domain, 1 / s_p per point in dZ slot 0's unused quarter
  const unsigned char* dzb = reinterpret_cast<const unsigned char*>(a.dz + jb.dz_off);
  const unsigned char* inb = reinterpret_cast<const unsigned char*>(a.acts + jb.in_off);
  const hw_rsrc_t rah = hw_make_rsrc(reinterpret_cast<const float*>(dzb + (size_t)c0 * 512), (unsigned)npts * 512u);
  const hw_rsrc_t ram = hw_make_rsrc(reinterpret_cast<const float*>(dzb + rows24_l8_byte(P) + (size_t)c0 * 256), (unsigned)npts * 256u);
  const hw_rsrc_t rbh = KW == 256 ? hw_make_rsrc(reinterpret_cast<const float*>(inb + (size_t)c0 * 512), (unsigned)npts * 512u)
                                  : hw_make_rsrc(a.acts + jb.in_off + (size_t)c0 * 64, (unsigned)npts * 256u);   // fp32 embedding rows
  const hw_rsrc_t rbm = hw_make_rsrc(reinterpret_cast<const float*>(inb + rows24_l8_byte(P) + (size_t)c0 * 256),
                                     KW == 256 ? (unsigned)npts * 256u : 0u);
  const hw_rsrc_t rd = hw_make_rsrc(a.dz + dz_dalpha_off(P) + c0, (unsigned)npts * 4u);
  const hw_rsrc_t rs = hw_make_rsrc(reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.dz) + rows24_invs_byte(P)) + c0,
                                    (unsigned)npts * 4u);
  const hw_rsrc_t rv = hw_make_rsrc(a.acts + acts_emb_off(P) + (size_t)c0 * 64, (unsigned)npts * 256u);

  f32x16 acc[2][NKT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NKT; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

  // ---- LDS-DMA of stage st into slot sl.  The LDS image of an instruction is lane-linear (64 x 16 bytes), its SOURCE
  // is free per lane - so the tiles land TILE-MAJOR, laid out for the transposing reads of compute():
  //   h  tile [16 points][256 cols] fp16: byte (p >> 3) * 4096 + ((p >> 2) & 1) * 2048 + (c >> 5) * 256 + (p & 3) * 64
  //      + (c & 31) * 2 - a 256-byte block is [4 points][32 columns], what one ds_read_b64_tr_b16 of a 32-lane pass
  //      gathers (one LDS row: conflict-free); 32-column tiles 256 bytes apart, the point quads 2 KiB (immediates);
  //   l8 tile [16 points][256 cols] e5m2: byte (p >> 3) * 2048 + (c >> 6) * 512 + (p & 7) * 64
  //      + ((((c >> 5) & 1) ^ ((p & 7) >> 2)) * 32) + (c & 31) - [8 points][64 B] per pair of 32-column tiles, the two
  //      tiles' halves swapped in points 4..7, so the 8 x 32 bytes one ds_read_b64_tr_b8 pass gathers cover the 16
  //      16-byte slots of an LDS row once.
  // Every quad of lanes still fetches 64 contiguous bytes of one row, every 128-byte line is consumed by one
  // instruction.  Wave w moves point quad (w >> 1) x column half (w & 1) of both h tiles (1 KiB each) and, the l8
  // tiles, waves 0-3 the dZ tile's and waves 4-7 the input tile's point half (w4 >> 1) x column half (w4 & 1) - and
  // the rider scalars of two points (1 / s_p in every job, d alpha / the view
  // directions only in the job with that rider: RID is a template argument).  The same count in every stage, also past the chunk end (zeros, no traffic),
  // so the vmcnt below is a compile-time constant.
  const int dma_h_voff = ((lane >> 2) & 3) * 512 + (4 * (wave & 1) + (lane >> 4)) * 64 + (lane & 3) * 16;
  const int dma_h_lds = (wave >> 1) * 2048 + (wave & 1) * 1024;
  const int dfj = (lane >> 2) & 7;
  const int dma_l_voff = dfj * 256 + (2 * (wave & 1) + (lane >> 5)) * 64 + ((((lane >> 1) & 1) ^ (dfj >> 2)) * 32) + (lane & 1) * 16;
  const int dma_l_lds = ((wave & 3) >> 1) * 2048 + (wave & 1) * 1024;
  auto issue = [&](int st, int sl) {
    unsigned char* slot = lds + sl * HW_SLOT;
    const int grow = st * PT + 2 * wave;                      // (rider scalars: two points per wave)
    const int growh = st * PT + 4 * (wave >> 1);              // h: this wave's point quad
    const int w4 = wave & 3, grow8 = st * PT + 8 * (w4 >> 1); // l8: this wave's point half
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rah, (lds_ptr_t)(slot + HW_DZ_HI + dma_h_lds), 16, dma_h_voff, growh * 512, 0, 2);
    if (KW == 256) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rbh, (lds_ptr_t)(slot + HW_IN_HI + dma_h_lds), 16, dma_h_voff, growh * 512, 0, 2);
      if (wave < 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ram, (lds_ptr_t)(slot + HW_DZ_MID + dma_l_lds), 16, dma_l_voff, grow8 * 256, 0, 2);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbm, (lds_ptr_t)(slot + HW_IN_MID + dma_l_lds), 16, dma_l_voff, grow8 * 256, 0, 2);
    } else {
      // (the fp32 embedding rows [16][64] stay row-major, four rows per instruction on waves 4-7)
      const int grow4 = st * PT + 4 * w4;
      if (wave < 4)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ram, (lds_ptr_t)(slot + HW_DZ_MID + dma_l_lds), 16, dma_l_voff, grow8 * 256, 0, 2);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rbh, (lds_ptr_t)(slot + HW_IN_HI + 4 * w4 * 256), 16, lane * 16, grow4 * 256, 0, 2);
    }
    if (RID == WF_ALPHA && lane < 2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(slot + HW_SCAL + 4 * 2 * wave), 4, lane * 4, grow * 4, 0, 0);
    if (lane < 2)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(slot + HW_SCAL + 256 + 4 * 2 * wave), 4, lane * 4, grow * 4, 0, 0);
    if (RID == WF_VIEWCOLS && lane < 6)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (lds_ptr_t)(slot + HW_SCAL + 64 + 4 * 6 * wave), 4,
                                               (lane / 3) * 256 + (60 + lane % 3) * 4, grow * 256, 0, 0);
  };

  // ---- riders.  Bias: db[n] = sum over points of dZ[p][n] is a dot product of the dZ fragment the lane has just built
  // (eight points of feature 2r + t, already scaled by S / s_p) with ones: v_dot2_f32_f16, fp32 accumulate, in
  // compute() on the waves of the first k half - no LDS read, no conversion.  (As a separate pass over the published
  // slot - per column and point two LDS reads, a widening and three conversion-class operations, 512 threads x 8
  // values per stage - the riders were 18 % of the kernel: knock-out 693 -> 568 us; four columns x two points per
  // thread with v_fma_mix_f32 on the planes: 674.)
  // View columns / alpha head (one job each): thread = (column tid & 255, point half tid >> 8) on reassembled values,
  // behind one wave-uniform branch per stage.
  float bacc[2] = {0.f, 0.f};
  float vc0 = 0.f, vc1 = 0.f, vc2 = 0.f;
  const int col = tid & 255, ph = __builtin_amdgcn_readfirstlane(tid >> 8);
  constexpr bool want_view = RID == WF_VIEWCOLS, want_alpha = KW == 256 && RID == WF_ALPHA;
  const bool bias_wave = (wave & 1) == 0;
  auto riders_add = [&](int sl) {
    if (!want_view) return;
    const unsigned char* slot = lds + sl * HW_SLOT;
    const float* isv = reinterpret_cast<const float*>(slot + HW_SCAL + 256);
    if (want_view) {
      const float* vw = reinterpret_cast<const float*>(slot + HW_SCAL + 64) + 3 * ph * (PT / 2);
#pragma unroll
      for (int q = 0; q < PT / 2; ++q) {
        const int p = ph * (PT / 2) + q;
        // (the tile-major images of issue())
        const unsigned char* hp = slot + HW_DZ_HI + (p >> 3) * 4096 + ((p >> 2) & 1) * 2048 + (col >> 5) * 256 + (p & 3) * 64 + (col & 31) * 2;
        const unsigned char* lp = slot + HW_DZ_MID + (p >> 3) * 2048 + (col >> 6) * 512 + (p & 7) * 64 +
                                  ((((col >> 5) & 1) ^ ((p & 7) >> 2)) * 32) + (col & 31);
        const float x = r24_value(*reinterpret_cast<const unsigned short*>(hp), *lp) * isv[p];   // the true dZ value
        vc0 = fmaf(x, vw[3 * q + 0], vc0); vc1 = fmaf(x, vw[3 * q + 1], vc1); vc2 = fmaf(x, vw[3 * q + 2], vc2);
      }
    }
  };
  // alpha head (the feature job): d w_alpha[k] = sum_p d alpha[p] In[p][k] is a dot product of the INPUT fragments the
  // lane has just built with the points' d alpha (x S, split h + l' like every other operand): v_dot2_f32_f16 on the two
  // waves of the first feature block, which cover the 256 input columns between them
  float aacc[4] = {0.f, 0.f, 0.f, 0.f}, dal_acc = 0.f;

  // ---- fragments: lane (r, hh) holds column r of a 32-column tile for its 8 points p = 8 hh + j, as the MFMA wants
  // them - and gfx950's transposing LDS reads deliver exactly that from the tile-major images of issue(): one
  // ds_read_b64_tr_b16 = four points of the h plane (two MFMA pairs, no VALU), one ds_read_b64_tr_b8 = eight points of
  // the l8 plane (a byte lands in the upper half of its fp16 by one v_perm_b32 per pair).  MFMA row m of tile t is
  // feature n0 + 32 t + m, column r of tile u is input k0 + 32 u + r.  The dZ pairs take the points' S / s_p (powers
  // of two, packed fp16 multiplies); the 2^-11 of BOTH cross terms sits on the dZ side (ahs = ah * 2^-11 against the
  // raw l pairs of the input, al * 2^-11 against its h pairs), so the input fragments need no arithmetic at all.
  // Per stage and wave 18 LDS reads + 48 VALU operations beside the 24 MFMAs (the round-5 first form, row-major
  // images read a dword per point and transposed by v_perm_b32: 32 + 80).  The fp32 embedding rows of the KW = 64 jobs
  // keep the split (v_cvt_pk_f16_f32 + v_fma_mixlo/hi_f16).
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef s16x4 __attribute__((address_space(3))) * tr16_ptr;
  typedef i32x2 __attribute__((address_space(3))) * tr8_ptr;
  auto split_pair = [](float x0, float x1, unsigned& h, unsigned& l) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x1));
  };
  auto pk_mul = [](unsigned a, half2v k) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(half2v, a) * k); };
  const half2v k_l = {(_Float16)LINV, (_Float16)LINV};
  // the lane's gather addresses inside a slot (lane i of a 16-lane group supplies the address of row i >> 2 (i >> 1),
  // 8-byte piece i & 3 (i & 1) of its group's 16 columns; it receives column i)
  const int i16 = lane & 15, cbit = (lane >> 4) & 1;
  const int la_h = hh * 4096 + (i16 >> 2) * 64 + cbit * 32 + (i16 & 3) * 8;
  const int la_8 = hh * 2048 + (i16 >> 1) * 64 + cbit * 16 + (i16 & 1) * 8, sw8 = (i16 >> 3) * 32;
  const int a_h = HW_DZ_HI + la_h + (n0 >> 5) * 256;                     // tile t: + 256 t; points 4..7: + 2048
  const int a_8 = HW_DZ_MID + la_8 + (n0 >> 6) * 512;                    // tile t: + ((32 t) ^ sw8)
  const int b_h = HW_IN_HI + la_h + (k0 >> 5) * 256;
  const int b_8 = HW_IN_MID + la_8 + (k0 >> 6) * 512;                    // tile u: + 512 (u >> 1) + ((32 (u & 1)) ^ sw8)
  auto compute = [&](int sl) {
    const unsigned char* slot = lds + sl * HW_SLOT;
    auto tr16 = [&](int off) { return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr16_ptr)(slot + off))); };
    auto tr8 = [&](int off) { return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr8_b64_v2i32((tr8_ptr)(slot + off))); };
    // the four fp16 pairs (value = byte << 8) of eight e5m2 bytes
    auto widen = [](u32x2 v) {
      return u32x4{__builtin_amdgcn_perm(0u, v[0], 0x010c000cu), __builtin_amdgcn_perm(0u, v[0], 0x030c020cu),
                   __builtin_amdgcn_perm(0u, v[1], 0x010c000cu), __builtin_amdgcn_perm(0u, v[1], 0x030c020cu)};
    };
    // every LDS read of the stage first (nothing in them depends on the points' factors): the factors' own read and
    // arithmetic then run under the gathers' latency instead of in front of it
    u32x2 ra_lo[2], ra_hi[2], ra_8[2], rb_lo[NKT], rb_hi[NKT], rb_8[NKT];
    float be[8];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      ra_lo[t] = tr16(a_h + 256 * t);
      ra_hi[t] = tr16(a_h + 256 * t + 2048);
      ra_8[t] = tr8(a_8 + ((32 * t) ^ sw8));
    }
    if (NKT == 4) {
#pragma unroll
      for (int u = 0; u < NKT; ++u) {
        rb_lo[u] = tr16(b_h + 256 * u);
        rb_hi[u] = tr16(b_h + 256 * u + 2048);
        rb_8[u] = tr8(b_8 + 512 * (u >> 1) + ((32 * (u & 1)) ^ sw8));
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) be[j] = reinterpret_cast<const float*>(slot + HW_IN_HI)[(8 * hh + j) * 64 + k0 + r];
    }
    // (S / s_p) of the lane's point pairs, and the same times 2^-11 for the cross terms
    half2v kp[4], kl[4];
    {
      const f32x4* isv = reinterpret_cast<const f32x4*>(slot + HW_SCAL + 256) + 2 * hh;
      const f32x4 i0 = isv[0], i1 = isv[1];
      // (a point without gradient has s_p = 1 and rows of zeros: S alone overflows fp16, and inf x 0 is NaN -
      // any finite factor serves such a row; a point at the launch maximum has S / s_p = 2^12)
      auto kf = [&](float is) { return (_Float16)fminf(is * S, 32768.0f); };
      kp[0] = half2v{kf(i0[0]), kf(i0[1])};
      kp[1] = half2v{kf(i0[2]), kf(i0[3])};
      kp[2] = half2v{kf(i1[0]), kf(i1[1])};
      kp[3] = half2v{kf(i1[2]), kf(i1[3])};
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) kl[j2] = kp[j2] * k_l;
    }
    u32x4 ahp[2], ahs[2], alp[2], bhp[NKT], blp[NKT];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const u32x4 l4 = widen(ra_8[t]);
      const unsigned hv[4] = {ra_lo[t][0], ra_lo[t][1], ra_hi[t][0], ra_hi[t][1]};
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        const unsigned lw = l4[j2];
        ahp[t][j2] = pk_mul(hv[j2], kp[j2]);
        ahs[t][j2] = pk_mul(hv[j2], kl[j2]);
        alp[t][j2] = pk_mul(lw, kl[j2]);
      }
    }
    if (NKT == 4) {
#pragma unroll
      for (int u = 0; u < NKT; ++u) {
        bhp[u] = u32x4{rb_lo[u][0], rb_lo[u][1], rb_hi[u][0], rb_hi[u][1]};
        blp[u] = widen(rb_8[u]);                                               // raw l: its 2^-11 is in ahs
      }
    } else {
#pragma unroll
      for (int j2 = 0; j2 < 4; ++j2) {
        unsigned h, l;
        split_pair(be[2 * j2], be[2 * j2 + 1], h, l);
        bhp[0][j2] = h; blp[0][j2] = l;                                        // l' = x - h at its true scale
      }
    }
    if (bias_wave) {
      auto dot_ones = [](unsigned w, float c) {
        const half2v ones = {(_Float16)1.0f, (_Float16)1.0f};
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, w), ones, c, false);
      };
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j2 = 0; j2 < 4; ++j2) {
          // (through scalars: hipcc lowers __builtin_bit_cast of a vector-element subscript as element 0 in unrolled
          // loops - DESIGN.md toolchain notes)
          const unsigned wh = ahp[t][j2], wl = alp[t][j2];
          bacc[t] = dot_ones(wh, bacc[t]);
          bacc[t] = dot_ones(wl, bacc[t]);
        }
    }
    if (NKT == 4 && want_alpha) {
      // (spread over ALL waves - wave pair q = wave >> 1 takes point pair j2 = q of every lane's eight points: on the
      // two waves of the first feature block alone the 48 dot products per stage made those two the last at every
      // barrier, +6 % on the whole launch mix)
      const float* dav = reinterpret_cast<const float*>(slot + HW_SCAL) + 8 * hh;
      auto dot2 = [](unsigned a, unsigned b, float c) {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, a), __builtin_bit_cast(half2v, b), c, false);
      };
      const int j2 = wave >> 1;                       // wave-uniform
      const float d0 = dav[2 * j2], d1 = dav[2 * j2 + 1];
      if (r == 0 && (wave & 1) == 0) dal_acc += d0 + d1;   // (lanes 0 and 32 of the even waves: every point once)
      unsigned dh, dl;
      split_pair(d0 * S, d1 * S, dh, dl);
      const unsigned dhs = pk_mul(dh, k_l);           // (the input's l pairs are raw: their 2^-11 rides here)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // (bhp / blp are indexed by a wave-uniform j2: selects, not a dynamic register index)
        unsigned bhw = bhp[NKT > u ? u : 0][0], blw = blp[NKT > u ? u : 0][0];
#pragma unroll
        for (int jj = 1; jj < 4; ++jj) {
          bhw = j2 == jj ? bhp[NKT > u ? u : 0][jj] : bhw;
          blw = j2 == jj ? blp[NKT > u ? u : 0][jj] : blw;
        }
        aacc[u] = dot2(bhw, dh, aacc[u]);
        aacc[u] = dot2(bhw, dl, aacc[u]);
        aacc[u] = dot2(blw, dhs, aacc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < NKT; ++u)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const half8 ah8 = __builtin_bit_cast(half8, ahp[t]), al8 = __builtin_bit_cast(half8, alp[t]);
        const half8 as8 = __builtin_bit_cast(half8, NKT == 4 ? ahs[t] : ahp[t]);
        const half8 bh8 = __builtin_bit_cast(half8, bhp[u]), bl8 = __builtin_bit_cast(half8, blp[u]);
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah8, bh8, acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as8, bl8, acc[t][u], 0, 0, 0);
        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al8, bh8, acc[t][u], 0, 0, 0);
      }
  };

  // ---- the ring runs in PAIRS of stages: one barrier per 32 points.  Pair pi's two slots have landed (each wave's
  // vmcnt leaves its pieces of pair pi + 1 in flight) -> barrier -> both stages are contracted -> pair pi + 2 is issued
  // into the slots of pair pi - 1, which every wave left behind at this pair's barrier.  Measured on the step's launch
  // mix, same box: 16-point stages with a barrier each and the issue in front of the compute 544 us (ring depth 4 / 5 /
  // 6 alike), the issue behind the compute 525-532 (an LDS-DMA instruction costs its wave 35-50 ns of issue - the
  // back-pressure of a memory system that is kept full - and in front of the compute that sits on the path to every
  // wave's first MFMA), pairs with the issue behind 515-517, in front 709, between the two stages 676.  Kernel stamps
  // (-DHW_TRACE, tools/probe_wgrad_f16_trace.py): a SIMD's second wave (waves 4-7) finishes a stage ~450 ns behind
  // its first - the MFMA pipe serves the older wave first (s_setprio on waves 4-7 swaps the roles, alternating it per
  // stage evens them out; the pair takes 3.3 us either way) - and waves 0-3 spend that at the barrier.  Giving those four
  // waves ALL the DMA issue (they have the slack) was built and is slower, 556-576 us: the stream a wave can keep in
  // flight is bounded per wave, four issuers move 3.6 TB/s where eight move 5.
  const int ns = (npts + PT - 1) / PT;
  const int np = (ns + 1) >> 1;                    // (an odd last stage's partner is all range-check zeros)
#pragma unroll
  for (int st = 0; st < 4; ++st) issue(st, st);
#ifdef HW_TRACE
  unsigned long long tw = 0, ti = 0, tc = 0;
#endif
  // (the loop is written once per value of the wave-uniform `active`: with the test inside it, each stage's
  // contraction was a basic block of its own and nothing of the second stage could be scheduled under the first)
  auto ring = [&](auto ACT) {
    int slp = 0;
    for (int pi = 0; pi < np; ++pi) {
#ifdef HW_TRACE
      const unsigned long long t0 = HW_T();
#endif
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NI) : "memory");
      hw_barrier();
#ifdef HW_TRACE
      const unsigned long long t1 = HW_T();
#endif
      if (decltype(ACT)::value) compute(slp);
      riders_add(slp);
      if (decltype(ACT)::value) compute(slp + 1);
      riders_add(slp + 1);
#ifdef HW_TRACE
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t2 = HW_T();
#endif
      const int sn = slp >= 2 ? slp - 2 : slp + 4;
      issue(2 * pi + 4, sn);
      issue(2 * pi + 5, sn + 1);
#ifdef HW_TRACE
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t3 = HW_T();
      tw += t1 - t0; tc += t2 - t1; ti += t3 - t2;
#endif
      slp = slp == 4 ? 0 : slp + 2;
    }
  };
  if (active) ring(std::true_type{}); else ring(std::false_type{});
#ifdef HW_TRACE
  if (lane == 0) {
    unsigned long long* d = hw_dbg + ((size_t)(blockIdx.x & 4095) * 8 + wave) * 4;
    d[0] = tw; d[1] = ti; d[2] = tc; d[3] = (unsigned long long)np | ((unsigned long long)KW << 32);
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land on the reduction scratch below
  hw_barrier();

  // ---- write the partial (dZ scale removed): MFMA row m of tile t = feature n0 + 32 t + m, column r of tile u =
  // input k0 + 32 u + r (a row's 32 lanes store 128 contiguous bytes)
  const float invS = 1.0f / S;
  if (active) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int m = (i & 3) + 8 * (i >> 2) + 4 * hh;
        const int n = n0 + 32 * t + m;
        if (n < jb.n_rows) {
          float* dst = out + jb.w_off + (size_t)n * jb.ld + jb.kcol0 + k0 + r;
#pragma unroll
          for (int u = 0; u < NKT; ++u)
            if (k0 + 32 * u + r < jb.kvalid) dst[32 * u] = acc[t][u][i] * invS;
        }
      }
  }
  // bias: the two point halves (lanes r and r + 32) of a feature meet by a lane exchange; dZ scale removed
  if (active && bias_wave && (jb.flags & WF_BIAS)) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float b = bacc[t] + __shfl_xor(bacc[t], 32, 64);
      const int n = n0 + 32 * t + r;
      if (hh == 0 && n < jb.n_rows) out[jb.b_off + n] = b * invS;
    }
  }
  // riders: the two point halves of a column are combined through LDS
  float* red = lds_f;    // [2][256][4] alpha | view cols, [2] d alpha
  red[(ph * 256 + col) * 4 + 0] = 0.f;
  float* reda = red + 2 * 256 * 4 + 8;            // alpha head: [4 wave pairs][256 columns] | [4] d alpha sums
  if (NKT == 4 && want_alpha) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v = aacc[u] + __shfl_xor(aacc[u], 32, 64);
      if (hh == 0) reda[(wave >> 1) * 256 + k0 + 32 * u + r] = v;
    }
    const float dsum = dal_acc + __shfl_xor(dal_acc, 32, 64);
    if ((wave & 1) == 0 && lane == 0) reda[4 * 256 + (wave >> 1)] = dsum;
  }
  red[(ph * 256 + col) * 4 + 1] = vc0;
  red[(ph * 256 + col) * 4 + 2] = vc1;
  red[(ph * 256 + col) * 4 + 3] = vc2;
  __syncthreads();
  if (tid < 256) {
    const float* r0 = red + (size_t)tid * 4, *r1 = red + (size_t)(256 + tid) * 4;
    if (NKT == 4 && want_alpha) {
      out[jb.aux_off + tid] = ((reda[tid] + reda[256 + tid]) + (reda[512 + tid] + reda[768 + tid])) * invS;
      if (tid == 0) out[jb.aux_off + 256] = (reda[1024] + reda[1025]) + (reda[1026] + reda[1027]);   // d b_alpha (unscaled fp32)
    }
    if ((jb.flags & WF_VIEWCOLS) && tid < 128) {
      out[jb.w_off + (size_t)tid * jb.ld + 256] = r0[1] + r1[1];
      out[jb.w_off + (size_t)tid * jb.ld + 257] = r0[2] + r1[2];
      out[jb.w_off + (size_t)tid * jb.ld + 258] = r0[3] + r1[3];
    }
  }
}

// one or two network calls per launch: a 1-D grid, the blocks of net[0] first (job-major, chunk fastest)
struct WgradF16Args2 {
  WgradF16Args n[2];
  int gx[2];          // chunks per job of each network
  int blocks0;        // gx[0] * n[0].w.njobs
};
__global__ __launch_bounds__(512, 2) void mlp_wgrad_f16_kernel(WgradF16Args2 faa) {
  extern __shared__ __attribute__((aligned(16))) float ldsw[];
  const bool second = (int)blockIdx.x >= faa.blocks0;        // wave-uniform
  const WgradF16Args& fa = second ? faa.n[1] : faa.n[0];
  const int bl = (int)blockIdx.x - (second ? faa.blocks0 : 0);
  const int gx = second ? faa.gx[1] : faa.gx[0];
  const int by = bl / gx, bx = bl - by * gx;
  const WgradArgs& a = fa.w;
  const WgradJob& jb = a.jobs[by];
  const int c0 = bx * a.chunk;
  const int c1 = min(a.P, c0 + a.chunk);
  float* out = a.partial + (size_t)bx * N_PARAM_FLOATS;
  // one power-of-two scale for dZ: max|g_out| * S in [2^7, 2^8)
  float S = 1.f;
  const float m = fa.gmax[0];
  if (m > 0.f && m < 3.0e38f) {
    int e;
    frexpf(m, &e);
    S = ldexpf(1.f, min(8 - e, 96));
  }
  if (jb.flags & WF_RGB) {
    wgrad_rgb_job<true>(a, jb, ldsw, c0, c1, out);
  } else if (jb.kw == 256) {
    if (jb.flags & WF_ALPHA) wgrad_f16_job<256, WF_ALPHA>(a, jb, ldsw, c0, c1, S, out);
    else if (jb.flags & WF_VIEWCOLS) wgrad_f16_job<256, WF_VIEWCOLS>(a, jb, ldsw, c0, c1, S, out);
    else wgrad_f16_job<256>(a, jb, ldsw, c0, c1, S, out);
  } else {
    wgrad_f16_job<64>(a, jb, ldsw, c0, c1, S, out);
  }
}

// the chunk partials of one or two network calls summed in one launch (blockIdx.y = network; the order of
// wgrad_reduce4_kernel: same bits)
struct Reduce4PairArgs {
  const float* partial[2];
  int nchunks[2];
  float* grad[2];
};
__global__ void wgrad_reduce4_pair_kernel(Reduce4PairArgs a) {
  const int net = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N_PARAM_FLOATS / 4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(a.partial[net]) + i;
  const int nchunks = a.nchunks[net];
  constexpr size_t ST = N_PARAM_FLOATS / 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int c = 0;
  for (; c + 4 <= nchunks; c += 4) {
    s0 += p[(size_t)c * ST]; s1 += p[(size_t)(c + 1) * ST];
    s2 += p[(size_t)(c + 2) * ST]; s3 += p[(size_t)(c + 3) * ST];
  }
  for (; c < nchunks; ++c) s0 += p[(size_t)c * ST];
  reinterpret_cast<f32x4*>(a.grad[net])[i] = (s0 + s1) + (s2 + s3);
}

}  // namespace scade

using namespace scade;

extern "C" long scade_mlp_packed_t_f16_bytes(void) { return PACKED_T_F16_HALVES * 2; }

extern "C" int scade_mlp_pack_t_f16(const float* const* params, void* packed_t_f16, void* stream) {
  SCADE_REQUIRE(params && packed_t_f16, -1, "scade_mlp_pack_t_f16: null pointer");
  PackTF16Args a;
  for (int i = 0; i < N_PARAM_TENSORS; ++i) {
    SCADE_REQUIRE(params[i], -1, "scade_mlp_pack_t_f16: params[%d] is null", i);
    a.p[i] = params[i];
  }
  a.packed = reinterpret_cast<_Float16*>(packed_t_f16);
  hipLaunchKernelGGL(mlp_pack_t_f16_kernel, dim3(64, NLAYER_DGRAD), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_mlp_pack_t_f16");
}

// n = 1 or 2 network calls: one zeroing launch, one dgrad launch, one weight-gradient launch, one reduce (the
// exact-weight-gradient mode wgrad_f16 = 0 runs the exact kernel's own launches per network)
static int launch_bwd_f16(int n, const float* const* packed, const void* const* packed_t_f16, const float* const* acts,
                          const float* const* g_out, const int* P, int wgrad_f16, float* const* workspace,
                          float* const* grad_flat, hipStream_t s, ReduceDesc* defer = nullptr) {
  static unsigned long long attr_set = 0;   // one bit per device ordinal
  if (scade_attr_needed(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_f16_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, DGRAD_F16_LDS_BYTES);
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd_f16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_f16_kernel<true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, DGRAD_F16_LDS_BYTES);
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd_f16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(attr_set);
  }
  float* dz[2] = {nullptr, nullptr};
  float* partial[2] = {nullptr, nullptr};
  unsigned int* gmax[2] = {nullptr, nullptr};
  int tiles[2] = {0, 0};
  MlpDgradF16Args2 d{};
  for (int i = 0; i < n; ++i) {
    // workspace: [dz | partial | gmax]
    dz[i] = workspace[i];
    partial[i] = workspace[i] + dz_floats(P[i]);
    gmax[i] = reinterpret_cast<unsigned int*>(partial[i] + (size_t)pick_chunks(P[i]) * N_PARAM_FLOATS);
    tiles[i] = (P[i] + HM - 1) / HM;
    d.n[i] = MlpDgradF16Args{packed[i], reinterpret_cast<const _Float16*>(packed_t_f16[i]), acts[i], g_out[i], dz[i],
                             gmax[i], P[i]};
  }
  d.tiles0 = tiles[0];
  // a one-thread KERNEL, not hipMemsetAsync: inside a captured train step that is a memset node, and memset
  // nodes of back-to-back graph replays are not ordered against their neighbouring kernels on this ROCm (the
  // dgrad workgroups' atomicMax raced the reset: see lp_gmax_kernel in mlp_bwd_lp.hip)
  hipLaunchKernelGGL(zero_word_kernel, dim3(1), dim3(64), 0, s, gmax[0], gmax[1]);
  if (int e = scade_check_launch("scade_mlp_bwd_f16(zero)")) return e;
  // wgrad_f16: the whole backward works on 24-bit saved rows (the forward was run with mode + 2); else on fp32 rows
  const dim3 dgrid(tiles[0] + tiles[1]);
  if (wgrad_f16) hipLaunchKernelGGL(mlp_dgrad_f16_kernel<true>, dgrid, dim3(256), DGRAD_F16_LDS_BYTES, s, d);
  else hipLaunchKernelGGL(mlp_dgrad_f16_kernel<false>, dgrid, dim3(256), DGRAD_F16_LDS_BYTES, s, d);
  if (int e = scade_check_launch("scade_mlp_bwd_f16(dgrad)")) return e;
  if (!wgrad_f16) {
    for (int i = 0; i < n; ++i)
      if (int e = scade_launch_wgrad(acts[i], dz[i], g_out[i], P[i], partial[i], grad_flat[i], s)) return e;
    return 0;
  }
  static unsigned long long wattr = 0;   // one bit per device ordinal
  if (scade_attr_needed(wattr)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wgrad_f16_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_F16_LDS_BYTES);
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd_f16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(wattr);
  }
  WgradF16Args2 fa{};
  Reduce4PairArgs r{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    fa.gx[i] = build_wgrad_jobs(fa.n[i].w, acts[i], dz[i], g_out[i], partial[i], P[i], HW_PT);
    fa.n[i].gmax = reinterpret_cast<const float*>(gmax[i]);
    if (i == 0) fa.blocks0 = fa.gx[0] * fa.n[0].w.njobs;
    blocks += fa.gx[i] * fa.n[i].w.njobs;
    r.partial[i] = partial[i]; r.nchunks[i] = fa.gx[i]; r.grad[i] = grad_flat ? grad_flat[i] : nullptr;
  }
  hipLaunchKernelGGL(mlp_wgrad_f16_kernel, dim3(blocks), dim3(512), WGRAD_F16_LDS_BYTES, s, fa);
  if (int e = scade_check_launch("scade_mlp_bwd_f16(wgrad)")) return e;
  if (defer) {            // summed by scade_step_finish, inside the optimizer's launch
    *defer = ReduceDesc{{partial[0], partial[1]}, {r.nchunks[0], r.nchunks[1]}, {}};
    return 0;
  }
  hipLaunchKernelGGL(wgrad_reduce4_pair_kernel, dim3(WGRAD_REDUCE_BLOCKS, n), dim3(256), 0, s, r);
  return scade_check_launch("scade_mlp_bwd_f16(reduce)");
}

extern "C" int scade_mlp_bwd_f16(const float* packed, const void* packed_t_f16, const float* acts,
                                 const float* g_out, int P, int wgrad_f16, float* workspace,
                                 float* grad_flat, void* stream) {
  SCADE_REQUIRE(P > 0, -2, "scade_mlp_bwd_f16: P must be positive");
  SCADE_REQUIRE(packed && packed_t_f16 && acts && g_out && workspace && grad_flat, -1,
                "scade_mlp_bwd_f16: null pointer");
  return launch_bwd_f16(1, &packed, &packed_t_f16, &acts, &g_out, &P, wgrad_f16, &workspace, &grad_flat,
                        (hipStream_t)stream);
}

// the split-precision backward of TWO network calls (the coarse + fine NeRF of a train step) in one launch sequence:
// every pointer argument is a host array of two; workspace[i] as scade_mlp_bwd_f16 wants it for P[i]
extern "C" int scade_mlp_bwd_f16_2(const float* const* packed, const void* const* packed_t_f16, const float* const* acts,
                                   const float* const* g_out, const int* P, int wgrad_f16, float* const* workspace,
                                   float* const* grad_flat, void* stream) {
  SCADE_REQUIRE(packed && packed_t_f16 && acts && g_out && P && workspace && grad_flat, -1,
                "scade_mlp_bwd_f16_2: null pointer");
  for (int i = 0; i < 2; ++i) {
    SCADE_REQUIRE(P[i] > 0, -2, "scade_mlp_bwd_f16_2: P[%d] must be positive", i);
    SCADE_REQUIRE(packed[i] && packed_t_f16[i] && acts[i] && g_out[i] && workspace[i] && grad_flat[i], -1,
                  "scade_mlp_bwd_f16_2: null pointer in entry %d", i);
  }
  return launch_bwd_f16(2, packed, packed_t_f16, acts, g_out, P, wgrad_f16, workspace, grad_flat, (hipStream_t)stream);
}

// scade_mlp_bwd_f16_2 (wgrad_f16 = 1) WITHOUT its reduce launch: see scade_mlp_bwd2_deferred
extern "C" int scade_mlp_bwd_f16_2_deferred(const float* const* packed, const void* const* packed_t_f16,
                                            const float* const* acts, const float* const* g_out, const int* P,
                                            float* const* workspace, void* reduce_desc, void* stream) {
  SCADE_REQUIRE(packed && packed_t_f16 && acts && g_out && P && workspace && reduce_desc, -1,
                "scade_mlp_bwd_f16_2_deferred: null pointer");
  for (int i = 0; i < 2; ++i) {
    SCADE_REQUIRE(P[i] > 0, -2, "scade_mlp_bwd_f16_2_deferred: P[%d] must be positive", i);
    SCADE_REQUIRE(packed[i] && packed_t_f16[i] && acts[i] && g_out[i] && workspace[i], -1,
                  "scade_mlp_bwd_f16_2_deferred: null pointer in entry %d", i);
  }
  return launch_bwd_f16(2, packed, packed_t_f16, acts, g_out, P, 1, workspace, nullptr, (hipStream_t)stream,
                        reinterpret_cast<ReduceDesc*>(reduce_desc));
}

#ifdef HW_TRACE
extern "C" int scade_debug_hw(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scade::hw_dbg), sizeof(unsigned long long) * 4096 * 8 * 4);
}
#endif

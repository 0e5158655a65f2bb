// Backward of the single-plane 16-bit ("lp": fp16 / bf16 operands, fp32 accumulate) NeRF MLP -
// the training half of BASELINE.json config 5's "bf16 MFMA path".  Same mathematics as
// mlp_bwd.hip (autograd of model/run_nerf_helpers.py:223-247), three launches:
//
//   G  lp_gmax_kernel      max|g_out| of the launch -> one power-of-two loss scale S
//   B1 mlp_dgrad_lp_kernel 128-point tile, the gradient tile walks the 9 layers backwards in LDS
//                          (one 16-bit plane, per-POINT power-of-two scale s_p so small gradients
//                          survive fp16; the chain is linear in g_out[p] and never mixes points, so
//                          s_p factors out exactly); every layer's dZ goes to HBM as 16-bit rows
//                          multiplied by S (s_p removed, S applied: one exact power-of-two factor)
//   B2 mlp_wgrad_lp_kernel dW[n][k] = sum_p dZ[p][n] In[p][k]: both operands are point-major rows
//                          in HBM but the MFMA contracts over points, so tiles are copied row-major
//                          into LDS (16-byte chunks, no transposition on the way in) and the MFMA
//                          fragments are gathered with gfx950's transposing LDS read
//                          ds_read_b64_tr_b16 (4 points x 16 columns per 16-lane group).
//                          HBM-bound by design: 1 KB per point-layer at the 16-bit MFMA rate.
//   B3 wgrad_reduce        sum of the per-chunk partials, S removed.
#include "mlp_tile_lp.h"
#include "mlp_pack.h"
#include "mlp_wgrad.h"
#include "mlp_reduce.h"

namespace scade {

struct PackTLpArgs {
  const float* p[N_PARAM_TENSORS];
  void* packed;
};

template <bool BF>
__global__ void mlp_pack_t_lp_kernel(PackTLpArgs a) { pack_t_lp_row<BF>(a.p, a.packed, blockIdx.y, blockIdx.x, gridDim.x); }

// ---------------------------------------------------------------------------
// G: launch-wide max |g_out| (finite values only) -> LP_GMAX_SLOTS per-workgroup maxima; every consumer
// workgroup takes the max of the slots itself (lp_read_gmax, one 16-byte load per lane + a wave max).
// No atomic and nothing to zero first: the former form (hipMemsetAsync of one word + atomicMax) put a MEMSET
// NODE into the captured train step, and memset nodes of back-to-back graph replays are not ordered against
// their neighbouring kernels on this ROCm - a replay queued behind another could read the word before or after
// its own reset, i.e. take the previous step's maximum or zero (tools/soak_train.py: the fp16 / 8-bit-save
// precisions stalled at 5x the loss of the others once the host ran far enough ahead).
// ---------------------------------------------------------------------------
constexpr int LP_GMAX_SLOTS = 256;          // == the block size of lp_gmax_kernel, >= its grid
static_assert(N_PARAM_FLOATS % 4 == 0, "the slots sit behind the partial rows and are read as f32x4");

struct LpGmaxArgs {
  const float* g[2];      // [P,4] of one network, or of the two networks of a joint launch
  long n[2];              // floats
  float* slots[2];
  int blocks0;            // workgroups [0, blocks0) take g[0], the rest g[1]
  const float* alpha_pre[2];   // [P] the forward's saved alpha_linear outputs (see below), or null
};
// What enters the dgrad chain through the density channel is NOT g_out[p][3] but d alpha_pre = g_out[p][3] *
// sigmoid(10 alpha_pre[p]) (the backward of softplus(beta = 10), model/run_nerf_helpers.py:242) - and the two can be
// ten decades apart: the last sample of a ray has delta = 1e10 (run_scade_scannet.py:515), so for a near-empty sample
// d alpha / d sigma = delta exp(-sigma delta) ~ 1e10 while sigmoid(10 alpha_pre) ~ 1e-9; an importance sampler's
// empty bin (den < 1e-5 -> 1, helpers:371) does the same.  Round 6 (tests/test_gpu_config5.py, the full-size bucket
// comparison): ONE such entry (1.5e3 against a 99.99th percentile of 4e-5) set the launch-wide scale of a 1024-ray
// step, every 8-bit dZ row of the fine network underflowed e5m2 and its weight gradient came out as exact zeros -
// silently, on one batch in two at K = 40.  The maximum is therefore taken over the EFFECTIVE gradient: the colour
// channels as they are and the density channel behind its sigmoid, evaluated as the dgrad heads evaluate it.
__device__ __forceinline__ float lp_effective_g3(float g3, float alpha_pre) {
  const float bx = alpha_pre * 10.f;
  return bx > 20.f ? g3 : g3 / (1.f + expf(-bx));
}
__global__ __launch_bounds__(LP_GMAX_SLOTS) void lp_gmax_kernel(LpGmaxArgs a) {
  __shared__ float part[LP_GMAX_SLOTS / 64];
  const bool second = (int)blockIdx.x >= a.blocks0;                  // wave-uniform
  const float* __restrict__ g = second ? a.g[1] : a.g[0];
  const long n = second ? a.n[1] : a.n[0];
  float* __restrict__ slots = second ? a.slots[1] : a.slots[0];
  const int bx = (int)blockIdx.x - (second ? a.blocks0 : 0);
  const int nb = second ? (int)gridDim.x - a.blocks0 : a.blocks0;
  float m = 0.f;
  const f32x4* g4 = reinterpret_cast<const f32x4*>(g);        // 16-byte loads (g_out is [P,4])
  const float* __restrict__ ap = second ? a.alpha_pre[1] : a.alpha_pre[0];
  for (long i = (long)bx * blockDim.x + threadIdx.x; i < n / 4; i += (long)nb * blockDim.x) {
    f32x4 v = g4[i];
    if (ap) v[3] = lp_effective_g3(v[3], ap[i]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x = fabsf(v[j]);
      if (x < 3.0e38f) m = fmaxf(m, x);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) slots[bx] = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
  // the slots no workgroup of this network owns
  if (bx == 0 && (int)threadIdx.x >= nb) slots[threadIdx.x] = 0.f;
}

// max of the slots, wave-uniform (every wave of a consumer reads the 1 KB itself: an L2 hit, no LDS, no barrier)
__device__ __forceinline__ float lp_read_gmax(const float* __restrict__ slots) {
  const f32x4 v = reinterpret_cast<const f32x4*>(slots)[threadIdx.x & 63];
  float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m)));
}

// loss scale: max|g_out| * S in [2^5, 2^6)  (dZ entries can exceed max|g_out| by the layer gains;
// 2^6 leaves three decades of fp16 headroom and nine below before the subnormal floor)
__device__ __forceinline__ float lp_loss_scale(float m) {
  if (!(m > 0.f) || !(m < 3.0e38f)) return 1.f;
  int e;
  frexpf(m, &e);
  return ldexpf(1.f, min(6 - e, 96));   // denormal max: keep the scale (and S / s_p) finite
}

// ---------------------------------------------------------------------------
// B1: dgrad chain
// ---------------------------------------------------------------------------
struct MlpDgradLpArgs {
  const float* packed;          // unused (the head weights ride in the tail of packedT)
  const void* packedT;          // transposed 16-bit pack
  const unsigned char* acts;    // lp_acts_bytes(P)
  const float* g_out;           // [P,4]
  unsigned char* dz;            // lp_dz_bytes(P)
  const float* gmax;
  int P;
};
// one launch may walk the tiles of TWO networks (see MlpDgradArgs2 in mlp_bwd.hip): workgroups [0, tiles0)
// belong to n[0], the rest to n[1]; tiles1 = workgroups of n[1].  The sign words are indexed by the workgroup
// of the network's OWN forward launch, so each network keeps its forward's tiling (both must use the same NPT).
struct MlpDgradLpArgs2 {
  MlpDgradLpArgs n[2];
  int tiles0, tiles1;
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool BF, bool MASK, bool ADD_ALPHA, int NPT = LPT>
__device__ __forceinline__ void dgrad_store_lp(const f32x16 (&acc)[2][NPT], int ktile0, typename LP<BF>::T* g,
                                               const unsigned (&bits)[4],
                                               const float* __restrict__ w_a, const float* dal_scaled,
                                               int lane) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int m = 0; m < 2; ++m) {       // row groups q = 2m, 2m + 1: two 16-byte chunks of 8 features
      f32x4 wa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      if (ADD_ALPHA) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq)
          wa[qq] = *reinterpret_cast<const f32x4*>(w_a + (ktile0 + t) * 32 + 8 * (2 * m + qq) + 4 * hh);
      }
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        const int row = p * 32 + r;
        u32x2 v[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = 2 * m + qq;
          float y[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            y[i] = acc[t][p][4 * q + i];
            if (ADD_ALPHA) y[i] = y[i] + wa[qq][i] * dal_scaled[row];
          }
          v[qq][0] = pack2<BF, false>(y[0], y[1]);
          v[qq][1] = pack2<BF, false>(y[2], y[3]);
          if (MASK) {   // sign words (mlp_tile_lp.h): dword d = ((t*4+q)*4+p)*2 + j
            const int d0 = ((t * 4 + q) * 4 + p) * 2;
            v[qq][0] = mask_pair(v[qq][0], bits[d0 >> 4], d0 & 15);
            v[qq][1] = mask_pair(v[qq][1], bits[d0 >> 4], (d0 & 15) + 1);
          }
        }
        // as in the forward's layer_store_lp: a lane holds HALF (features 8q + 4hh .. +3) of each of the two
        // 16-byte chunks; stored as ds_write_b64 the 16 rows of a lane group share 8 slots of a bank line
        // (2-way conflict: 30 % of this kernel's LDS cycles in round 2).  v_permlane32_swap trades the halves
        // between lane r and lane r + 32 - the lower lane ends up with all of chunk 2m, the upper one with
        // all of chunk 2m + 1 - and each stores one conflict-free ds_write_b128.
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned lo, hi;
          lp_swap_halves(v[0][j], v[1][j], lo, hi);
          w[j] = lo;
          w[2 + j] = hi;
        }
        *reinterpret_cast<u32x4*>(g + x_idx(row, (ktile0 + t) * 4 + 2 * m + hh)) = w;
      }
    }
}

constexpr int dgrad_lp_lds_bytes(int NPT) { return 32 * NPT * W * 2 + 2 * 32 * NPT * 4; }

// -DDG_TRACE (variant build; tools/probe_dgrad_trace.py): core-clock stamps of sixteen consecutive workgroups of the
// launch's third round at the phase boundaries of every gemm of the chain, + their HW_ID / XCC_ID words
#ifdef DG_TRACE
__device__ unsigned long long dg_trace[16 * 4 * 9 * 6];
__device__ unsigned dg_hwid[16 * 2];
__device__ unsigned long long dg_heads[16 * 4 * 6];
#define DH_STAMP(I)                                                                                \
  if (S8 && NPT == 4 && blockIdx.x >= 1300 && blockIdx.x < 1316 && lane == 0)                       \
    dg_heads[((blockIdx.x - 1300) * 4 + wave) * 6 + (I)] = clock64();
#define DT_STAMP(G, I)                                                                             \
  if (S8 && NPT == 4 && blockIdx.x >= 1300 && blockIdx.x < 1316 && lane == 0)                       \
    dg_trace[(((blockIdx.x - 1300) * 4 + wave) * 9 + (G)) * 6 + (I)] = clock64();
#else
#define DT_STAMP(G, I)
#define DH_STAMP(I)
#endif
// S8: format code 2 - the saved rows (activations read, dZ written) are 8-bit e5m2 (mlp_tile_lp.h); the dZ rows
// then carry the launch-wide loss scale like the fp16 path's (e5m2 has fp16's exponent range)
template <bool BF, int NPT, bool S8 = false>
__global__ __launch_bounds__(256, 2) void mlp_dgrad_lp_kernel(MlpDgradLpArgs2 aa) {
  constexpr int LM = 32 * NPT, LPT = NPT, LXPLANE = LM * W;   // this workgroup's tile (shadow the 128-point default)
  const bool second = (int)blockIdx.x >= aa.tiles0;                 // wave-uniform: scalar selects
  const MlpDgradLpArgs& a = second ? aa.n[1] : aa.n[0];
  const int blk = (int)blockIdx.x - (second ? aa.tiles0 : 0);
  const int ntiles = second ? aa.tiles1 : aa.tiles0;
  typedef typename LP<BF>::T T;
  typedef typename LP<BF>::V4 V4;
  typedef typename LP<BF>::V8 V8;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
  T* g = reinterpret_cast<T*>(lds16);
  float* dal = reinterpret_cast<float*>(lds16 + LXPLANE);   // [128] d alpha_pre * s_p
  float* fac = dal + LM;                                    // [128] S / s_p

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  DT_STAMP(0, 4)          // (kernel entry)
  DH_STAMP(0)
  const int p0 = blk * LM;
  const int P = a.P;
  const T* __restrict__ pt_ = reinterpret_cast<const T*>(a.packedT);
  const float* __restrict__ tl = reinterpret_cast<const float*>(pt_ + PACKED_T_LP_ELEMS);
  const T* __restrict__ actsT = reinterpret_cast<const T*>(a.acts);
  T* __restrict__ dzT = reinterpret_cast<T*>(a.dz);
  const float* __restrict__ alpha_pre = reinterpret_cast<const float*>(a.acts + lp_acts_alpha_byte(P));
  float* __restrict__ dalpha = reinterpret_cast<float*>(a.dz + lp_dz_dalpha_byte(P));
  const u32x4* __restrict__ masks = reinterpret_cast<const u32x4*>(a.acts + lp_acts_mask_byte(P));
  // bf16 carries fp32's exponent range: the launch-wide loss scale (and the two launches that find it) is an
  // fp16 matter; a power-of-two scale commutes with every rounding here, so S = 1 gives the same bits
  const float S = (BF && !S8) ? 1.f : lp_loss_scale(lp_read_gmax(a.gmax));
  unsigned char* __restrict__ dz8 = a.dz;
  const unsigned char* __restrict__ acts8 = a.acts;
  DH_STAMP(1)

  // ---- heads: d alpha_pre, per-point scale, dZ of the views layer ------------------------
  // (Round 4: the phase trace - tools/probe_dgrad_trace.py, profiles/r04_lp_phase_trace.txt - showed this section at
  // 32 k cycles = 23 % of a workgroup's life; 8 k now.  DESIGN.md section 3.1c.)
  constexpr bool MFMA_HEADS = BF;
  if constexpr (MFMA_HEADS) {
    // bf16 formats: dZv[p][c] = mask * sum_k go[p][k] wr[k][c] is a K = 3 GEMM - ONE 32x32x16 MFMA per point tile
    // with both operands split into bf16 high + low parts in the 16 k-slots (hi*hi + lo*hi + hi*lo: 2^-17 relative,
    // far inside the bf16 rounding of the result): wave w owns the 32 views columns [32 w, 32 w + 32), operands are
    // built in registers from the fp32 head weights and the fp32 output gradient (no staging), the epilogue is the
    // chain's own (mask on the packed pair, v_permlane32_swap, ds_write_b128).  4 MFMAs and ~150 VALU
    // instructions per lane where the VALU form needed ~600.
    const int r = lane & 31, hh = lane >> 5;
    const float* wr = tl + TL_WR;
    const int ncol = 32 * wave + r;
    const float wv0 = wr[ncol], wv1 = wr[128 + ncol], wv2 = wr[256 + ncol];
    f32x4 gq[NPT];
    // ReLU mask: the forward's sign words of the views layer (same (wave, lane, register) map as this MFMA's output)
    const u32x4 hw = masks[((size_t)8 * ntiles + blk) * 256 + tid];
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      const size_t pt = (size_t)min(p0 + 32 * p + r, P - 1);       // (rows past P: zeroed below)
      gq[p] = *reinterpret_cast<const f32x4*>(a.g_out + pt * 4);
    }
    float go3r = 0.f, apr = 0.f;                         // d alpha_pre: lane `tid` owns row `tid`
    if (tid < LM) {
      const int pt = min(p0 + tid, P - 1);
      go3r = a.g_out[(size_t)pt * 4 + 3];
      apr = alpha_pre[pt];
    }
    DH_STAMP(2)
    auto split = [](float x, T& h, T& l) {
      h = (T)x;
      const float hf = (float)h;
      l = (fabsf(hf) <= 3.0e38f) ? (T)(x - hf) : (T)0.f;   // (inf: no inf - inf)
    };
    T wh0, wl0, wh1, wl1, wh2, wl2;
    split(wv0, wh0, wl0); split(wv1, wh1, wl1); split(wv2, wh2, wl2);
    const T z = (T)0.f;
    V8 af;
    if (hh == 0) af = V8{wh0, wh1, wh2, wh0, wh1, wh2, z, z};
    else af = V8{wl0, wl1, wl2, z, z, z, z, z};
    f32x16 hacc[NPT];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < NPT; ++p) {
      const bool ok = p0 + 32 * p + r < P;
      T gh0, gl0, gh1, gl1, gh2, gl2;
      split(ok ? gq[p][0] : 0.f, gh0, gl0); split(ok ? gq[p][1] : 0.f, gh1, gl1); split(ok ? gq[p][2] : 0.f, gh2, gl2);
      V8 bfr;
      if (hh == 0) bfr = V8{gh0, gh1, gh2, gl0, gl1, gl2, z, z};
      else bfr = V8{gh0, gh1, gh2, z, z, z, z, z};
      hacc[p] = LP<BF>::mfma(af, bfr, zero16);
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        u32x2 v[2];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = 2 * m + qq;
          v[qq][0] = pack2<BF, false>(hacc[p][4 * q + 0], hacc[p][4 * q + 1]);
          v[qq][1] = pack2<BF, false>(hacc[p][4 * q + 2], hacc[p][4 * q + 3]);
          const int d0 = (q * 4 + p) * 2;                  // sign words (mlp_tile_lp.h), n-tile 0 of this wave
          v[qq][0] = mask_pair(v[qq][0], d0 < 16 ? hw[0] : hw[1], d0 & 15);
          v[qq][1] = mask_pair(v[qq][1], d0 < 16 ? hw[0] : hw[1], (d0 & 15) + 1);
        }
        u32x4 w;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          unsigned lo, hi;
          lp_swap_halves(v[0][j], v[1][j], lo, hi);
          w[j] = lo;
          w[2 + j] = hi;
        }
        *reinterpret_cast<u32x4*>(g + x_idx(p * 32 + r, wave * 4 + 2 * m + hh)) = w;
      }
    // (format code 2: the 8-bit copy of this tile rides in the views gemm below; plain bf16: S = 1, the rows leave
    // as they are - a wave saves the 32 columns it wrote)
    if (!S8) save_tile_lp_wave<BF, 32, NPT>(g, dzT + acts_slot_off(P, SLOT_VIEWS_H), p0, P, nullptr, 32 * wave, lane);
    if (tid < LM) {
      const int pt = p0 + tid;
      const float bx = apr * 10.f;
      float da = bx > 20.f ? go3r : go3r / (1.f + expf(-bx));
      if (pt < P) dalpha[pt] = da; else da = 0.f;
      dal[tid] = da;
      fac[tid] = S;
    }
    DH_STAMP(3)
  } else {
    // fp16 (per-point power-of-two scale s, launch-wide loss scale S) on the VALU: a lane owns ONE 8-column chunk of
    // the 128 columns for all its rows (the 24 head weights stay in registers, a wave instruction covers four whole
    // 256-byte rows) and every load is issued before the first use (no `if (ok)` around the loads: the compiler
    // closes a divergent block with a wait for its loads, which chained eight HBM latencies; rows past P read row
    // P - 1 and are zeroed by a select).  The mask is derived from the saved 16-bit views activations.
    constexpr int HIT = LM / 16;                          // rows per lane: row = 16 it + (tid >> 4)
    const int chunk = tid & 15, r16 = tid >> 4;
    const float* wr = tl + TL_WR;
    f32x4 w0[2], w1[2], w2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      w0[h] = *reinterpret_cast<const f32x4*>(wr + chunk * 8 + 4 * h);
      w1[h] = *reinterpret_cast<const f32x4*>(wr + 128 + chunk * 8 + 4 * h);
      w2[h] = *reinterpret_cast<const f32x4*>(wr + 256 + chunk * 8 + 4 * h);
    }
    const T* hv = actsT + acts_slot_off(P, SLOT_VIEWS_H);
    T* dzv = dzT + acts_slot_off(P, SLOT_VIEWS_H);
    V8 mk[HIT];
    f32x4 go[HIT];
    float ap[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int pt = min(p0 + it * 16 + r16, P - 1);
      mk[it] = *reinterpret_cast<const V8*>(hv + (size_t)pt * W + chunk * 8);
      go[it] = *reinterpret_cast<const f32x4*>(a.g_out + (size_t)pt * 4);
      ap[it] = alpha_pre[pt];
    }
    DH_STAMP(2)
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int row = it * 16 + r16, pt = p0 + row;
      const bool ok = pt < P;
      if (!ok) {                                           // rows past P: zero gradient, zero mask
        go[it] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) mk[it][j] = (T)0.f;
      }
      const float bx = ap[it] * 10.f;
      float da = bx > 20.f ? go[it][3] : go[it][3] / (1.f + expf(-bx));
      if (!ok) da = 0.f;
      const float m = fmaxf(fmaxf(fabsf(go[it][0]), fabsf(go[it][1])), fmaxf(fabsf(go[it][2]), fabsf(da)));
      // per-point power-of-two scale: an fp16 matter like S.  (bf16 carries fp32's exponent range and a power of
      // two commutes with every rounding of the chain: s = 1 gives the same bits, and the dZ rows then leave LDS as
      // plain copies.  CAVEAT: "same bits" holds while every value of the chain stays in bf16's NORMAL range - a
      // point whose output gradient is below ~1e-30 produces dZ values under 2^-126, which flush to bf16
      // subnormals / zero where a per-point scale would have kept them normal.  Such gradients are 22 decades under
      // Adam's eps (1e-8) and change no update; tests/test_gpu_lp.py pins the behaviour (finite, -> 0).)
      float s = 1.f;
      if (m > 0.f && m < 3.0e38f) {
        int e;
        frexpf(m, &e);
        s = ldexpf(1.f, min(-4 - e, 96));    // denormal gradients: keep the scale finite
      }
      if (chunk == 0) {
        if (ok) dalpha[pt] = da;
        dal[row] = da * s;
        fac[row] = S / s;
      }
      V8 vs, vS;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        // (fused: three ops per value instead of five; the result is rounded to 16 bit right below)
        const float d = __builtin_fmaf(go[it][2], w2[j >> 2][j & 3], __builtin_fmaf(go[it][1], w1[j >> 2][j & 3], go[it][0] * w0[j >> 2][j & 3]));
        const float v = (float)mk[it][j] > 0.f ? d : 0.f;
        vs[j] = (T)(v * s);
        vS[j] = (T)(v * S);
      }
      *reinterpret_cast<V8*>(g + x_idx(row, chunk)) = vs;
      if (ok) *reinterpret_cast<V8*>(dzv + (size_t)pt * W + chunk * 8) = vS;
    }
    DH_STAMP(3)
  }
  LP_SYNC();
  DH_STAMP(4)

  f32x16 acc[2][LPT];
  unsigned mb[4] = {0u, 0u, 0u, 0u};
  constexpr int DNS = 4;               // A register sets: weights fetched three k-blocks ahead (round 4: the dgrad has the registers)
  AFragN<BF, DNS> A;
  const int kt0 = wave * 2;
  auto load_mask = [&](int layer) {
    const u32x4 mw = masks[((size_t)layer * ntiles + blk) * 256 + tid];
    mb[0] = mw[0]; mb[1] = mw[1]; mb[2] = mw[2]; mb[3] = mw[3];
  };
  // this wave's k-tile pair of dgrad index TT: [kt][NB16][64] V8
#define WTL(TT, NB) (reinterpret_cast<const V8*>(pt_ + CE<off_wtl(TT)>::v) + kt0 * (NB) * 64)
#pragma unroll
  for (int j = 0; j < DNS - 1; ++j) {
    A.s[j].t0 = WTL(8, 8)[j * 64 + lane];
    A.s[j].t1 = WTL(8, 8)[(8 + j) * 64 + lane];
  }
  // rotation of the A register sets on entry of the n-th gemm of the chain: views (8 k-blocks),
  // feature (16), then layers 7..1 (16 each)
#define DROT(N) ((N) == 0 ? 0 : (8 + 16 * ((N)-1)) % DNS)

  // ---- views layer: d feature = Wv[:, :256]^T dZv  (reduction over 128 = 8 k16-blocks) ----
#ifdef DG_TRACE
  if (S8 && NPT == 4 && blockIdx.x >= 1300 && blockIdx.x < 1316 && tid == 0) {
    dg_hwid[(blockIdx.x - 1300) * 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    dg_hwid[(blockIdx.x - 1300) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
  DT_STAMP(0, 0)
  if constexpr (S8 && MFMA_HEADS) {   // the 8-bit copy of the views dZ tile (128 columns) rides in this k-loop
    SaveRider8<NPT, 128> rid0;
    rid0.init(g, dz8 + acts_slot_off(P, SLOT_VIEWS_H) * 2, p0, P, fac, wave);
    layer_gemm_lp<BF, 2, 0, 8, false, DROT(0), DNS, NPT>(acc, A, WTL(8, 8), WTL(7, 16), 16, g, g, lane, nullptr, rid0);
  } else {
    layer_gemm_lp<BF, 2, 0, 8, false, DROT(0), DNS, NPT>(acc, A, WTL(8, 8), WTL(7, 16), 16, g, g, lane, nullptr);
  }
  DT_STAMP(0, 1)
  LP_SYNC();
  DT_STAMP(0, 2)
  dgrad_store_lp<BF, false, false, NPT>(acc, kt0, g, mb, nullptr, dal, lane);
  DT_STAMP(0, 3)
  constexpr bool RID16 = BF && !S8;      // plain bf16: the 16-bit copies ride too (SaveRider16; S = 1, no factor)
  if (!S8 && !RID16) save_tile_lp_wave<BF, 64, NPT>(g, dzT + acts_slot_off(P, SLOT_FEAT), p0, P, BF ? nullptr : fac, 64 * wave, lane);
  LP_SYNC();

  // format code 2: the 8-bit copy of the tile just written leaves as a rider of the NEXT gemm's k-loop
  // (SaveRider8, mlp_tile_lp.h); only the last tile of the chain is saved as a burst
  typename std::conditional<S8 && BF, SaveRider8<NPT>,
                            typename std::conditional<RID16, SaveRider16<NPT>, NoRider>::type>::type rid;
#define RID_INIT(SLOT)                                                                        \
  if constexpr (S8 && BF) rid.init(g, dz8 + acts_slot_off(P, SLOT) * 2, p0, P, fac, wave);    \
  if constexpr (RID16) rid.init(g, dzT + acts_slot_off(P, SLOT), p0, P, wave);

  // ---- feature layer: d h7 = Wf^T d feature + w_alpha * d alpha_pre, mask h7 ---------------
  load_mask(7);
  RID_INIT(SLOT_FEAT)
  DT_STAMP(0, 5) DT_STAMP(1, 0)
  layer_gemm_lp<BF, 2, 0, 16, false, DROT(1), DNS, NPT>(acc, A, WTL(7, 16), WTL(6, 16), 16, g, g, lane, nullptr, rid);
  DT_STAMP(1, 1)
  LP_SYNC();
  DT_STAMP(1, 2)
  dgrad_store_lp<BF, true, true, NPT>(acc, kt0, g, mb, tl + TL_WA, dal, lane);
  DT_STAMP(1, 3)
  if (!S8 && !RID16) save_tile_lp_wave<BF, 64, NPT>(g, dzT + acts_slot_off(P, 7), p0, P, BF ? nullptr : fac, 64 * wave, lane);
  LP_SYNC();

#define DGRAD_LAYER_L(L)                                                                            \
  load_mask((L)-1);                                                                                 \
  RID_INIT(L)                                                                                       \
  DT_STAMP(8 - (L), 5) DT_STAMP(9 - (L), 0)                                                         \
  layer_gemm_lp<BF, 2, 0, 16, false, DROT(9 - (L)), DNS, NPT>(acc, A, WTL((L)-1, 16), WTL((L) > 1 ? (L)-2 : 0, 16), 16, \
                                                    g, g, lane, nullptr, rid);                      \
  DT_STAMP(9 - (L), 1)                                                                              \
  LP_SYNC();                                                                                  \
  DT_STAMP(9 - (L), 2)                                                                              \
  dgrad_store_lp<BF, true, false, NPT>(acc, kt0, g, mb, nullptr, dal, lane);                             \
  DT_STAMP(9 - (L), 3)                                                                              \
  if (S8 && (L) == 1) save_tile_lp_wave8<BF, 64, NPT>(g, dz8 + acts_slot_off(P, 0) * 2, p0, P, fac, 64 * wave, lane);      \
  else if (!S8 && (!RID16 || (L) == 1)) save_tile_lp_wave<BF, 64, NPT>(g, dzT + acts_slot_off(P, (L)-1), p0, P, BF ? nullptr : fac, 64 * wave, lane);    \
  LP_SYNC();

  DGRAD_LAYER_L(7)
  DGRAD_LAYER_L(6)
  DGRAD_LAYER_L(5)
  DGRAD_LAYER_L(4)
  DGRAD_LAYER_L(3)
  DGRAD_LAYER_L(2)
  DGRAD_LAYER_L(1)
  DT_STAMP(8, 5)
#undef DGRAD_LAYER_L
#undef RID_INIT
#undef DROT
#undef WTL
}

// ---------------------------------------------------------------------------
// B2: weight gradient
// ---------------------------------------------------------------------------
struct WgradLpJob {
  long dz_off;       // element offset of the dZ matrix (row stride 256) in the dz workspace
  long in_off;       // element offset of the input matrix in the acts workspace
                     // (both filled in by the kernel from the slots below and ITS network's point count)
  int dz_slot;       // workspace slot of the dZ matrix
  int in_slot;       // workspace slot of the input matrix; -1 = the embedding rows
  int in_stride;     // 256 (activation slot) or 64 (emb)
  int kw;            // tile width in k: 256 or 64
  int n_rows;        // valid output rows (256 or 128)
  int w_off;         // flat-gradient offset of the weight tensor
  int ld;            // its row length
  int kcol0;         // output column of input column kfirst
  int kfirst;        // input columns [kfirst, kvalid) are written
  int kvalid;
  int b_off;         // flat-gradient offset of the bias (WF_BIAS)
  int flags;
  int aux_off;       // WF_ALPHA: offset of alpha weight (bias follows at +256)
};

// per-network part of a launch (one network, or the coarse and the fine NeRF of a train step: chunks
// are two runs of entries in the launch's plan, see WgradLpPlan)
struct WgradLpNet {
  const unsigned char* acts;
  const unsigned char* dz;
  const float* g_out;   // [P,4] (rgb head)
  float* partial;       // [lp_ws_rows()][N_PARAM_FLOATS]: row k = segment k of the job that owns the element
  const float* gmax;
  int P;
};
// Balanced persistent launch (round 3).  The work of a launch is the list of (network, job) ENTRIES, each a run of
// 32-point stages weighted by what a stage of that job costs (bytes per stage: the kernel is bound by the memory
// system); workgroup w of the NWG = one-per-CU workgroups owns the weighted positions [w W / NWG, (w + 1) W / NWG)
// of the concatenated list, i.e. a contiguous stage range of one entry and, where it crosses an entry boundary, of
// the next one or two; the boundaries are laid out on the host so that every workgroup carries the same stage cost
// PLUS a fixed cost per segment it opens (prologue latency and the partial-row stores: ~16 stages of a 256 x 256
// job - at 128 rays per GPU that is a third of a workgroup's whole budget).  Against the former grid of (chunk, job) workgroups handed out by the dispatcher: every CU
// streams the same number of bytes (4.4 part-filled rounds before), and there are NWG + entries segments instead
// of chunks x jobs workgroups - a third of the fp32 partial rows to write (their stores keep a CU's read stream
// idle: 15 % of the kernel) and to reduce.  Segment k of an entry writes partial row k (= w - first workgroup of
// the entry); the reduce kernel sums, per parameter, the rows of the job that owns it.
constexpr int LP_MAX_ENTRIES = 2 * MAX_WGRAD_JOBS;
constexpr int LP_MAX_WG = 256;
struct WgradLpPlan {
  int cum[LP_MAX_ENTRIES + 1];   // weighted position of entry e = net * njobs + job (cum[n] = W)
  int bound[LP_MAX_WG + 1];      // workgroup w owns the weighted positions [bound[w], bound[w + 1])
  short first_wg[LP_MAX_ENTRIES];   // workgroup that owns position cum[e]
  unsigned char weight[MAX_WGRAD_JOBS];   // cost of one stage of a job
  int nst[2];                    // stages per network
  int nwg;
  int chunk, gx0, gx1;           // chunk > 0: the small-launch grid instead (see lp_build_plan) - workgroup id =
                                 // job * (gx0 + gx1) + chunk index, chunks [0, gx0) of `chunk` points are net[0]'s
};
struct WgradLpArgs {
  WgradLpJob jobs[MAX_WGRAD_JOBS];
  WgradLpNet net[2];
  WgradLpPlan plan;
  int njobs;
};

constexpr int WL_PT = 32;                          // points per stage = two k16 blocks
constexpr int WL_PITCH = 288;                      // elements per LDS row: 256 + 32 pad (576 B: the four
                                                   // rows of a transposing read land 16 banks apart)
constexpr int WL_TILE = WL_PT * WL_PITCH;          // elements per operand tile
constexpr int WL_STAGE = 2 * WL_TILE;              // dZ tile + input tile
constexpr int WL_LDS_BYTES = 3 * WL_STAGE * 2;         // triple buffered: 110592
constexpr int W8_S = 64;                               // format code 2 (wgrad_lp8_dma_job): points per ring slot,
constexpr int W8_D = 3;                                 // ring slots,
constexpr int W8_SLOT = 2 * W8_S * 256 + 256;          // bytes per slot: dZ [64][256] | input [64][256] | d alpha [64] fp32
constexpr int WGRAD_LP_LDS_BYTES = W8_D * W8_SLOT > WL_LDS_BYTES ? W8_D * W8_SLOT : WL_LDS_BYTES;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// MFMA operand fragment (32 columns x 16 points, lane = column lane&31, points 8*(lane>>5)..+7)
// gathered from a row-major [point][column] LDS tile with two transposing reads
template <bool BF>
__device__ __forceinline__ typename LP<BF>::V8 tr_frag(const typename LP<BF>::T* tile, int row0, int col0,
                                                       int lane) {
  const int i = lane & 15, cb = (lane >> 4) & 1, hh = lane >> 5;
  const typename LP<BF>::T* p = tile + (row0 + 8 * hh + (i >> 2)) * WL_PITCH + col0 + 16 * cb + 4 * (i & 3);
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * WL_PITCH));
  s16x8 v;
  v.lo = lo; v.hi = hi;
  return __builtin_bit_cast(typename LP<BF>::V8, v);
}

template <bool BF>
struct WStage {
  typename LP<BF>::V8 a0, a1, b0, b1;
  float d0, d1;      // d alpha_pre of the two rows (WF_ALPHA job)
};

template <bool BF, int KW>
__device__ __forceinline__ void wgrad_lp_job(const WgradLpNet& a, const WgradLpJob& jb, typename LP<BF>::T* lds,
                                             int c0, int c1, float invS, float* __restrict__ out) {
  typedef typename LP<BF>::T T;
  typedef typename LP<BF>::V8 V8;
  constexpr int NKT = KW == 256 ? 4 : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int n0 = (wave >> 1) * 64;
  const int k0 = (wave & 1) * (KW / 2);
  const bool active = n0 < jb.n_rows;
  const int P = a.P;
  const T* __restrict__ dzm = reinterpret_cast<const T*>(a.dz) + jb.dz_off;
  const T* __restrict__ inm = reinterpret_cast<const T*>(a.acts) + jb.in_off;
  const float* __restrict__ dalp = reinterpret_cast<const float*>(a.dz + lp_dz_dalpha_byte(P));
  // staging item of this thread: 8-column chunk cc of rows rr and rr+16 of the stage
  const int cc = tid & 31, rr = tid >> 5;
  const V8 zero8 = __builtin_bit_cast(V8, s16x8{0, 0, 0, 0, 0, 0, 0, 0});

  f32x16 acc[2][NKT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NKT; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;
  float bias_acc[8], alpha_acc[8], dal_acc = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { bias_acc[j] = 0.f; alpha_acc[j] = 0.f; }

  // Loads are UNCONDITIONAL (row index clamped to the last point, zeroed at commit time): a
  // wave-uniform or per-lane branch around them would make the outstanding-load count path
  // dependent and force s_waitcnt vmcnt(0) at every commit, i.e. no prefetch at all.
  const bool want_alpha = KW == 256 && (jb.flags & WF_ALPHA);
  auto issue = [&](WStage<BF>& s, int pt0) {
    const int pa = min(pt0 + rr, P - 1), pb = min(pt0 + rr + 16, P - 1);
    s.a0 = __builtin_nontemporal_load(reinterpret_cast<const V8*>(dzm + (size_t)pa * 256 + 8 * cc));   // read once
    s.a1 = __builtin_nontemporal_load(reinterpret_cast<const V8*>(dzm + (size_t)pb * 256 + 8 * cc));
    if (KW == 256) {
      s.b0 = __builtin_nontemporal_load(reinterpret_cast<const V8*>(inm + (size_t)pa * 256 + 8 * cc));
      s.b1 = __builtin_nontemporal_load(reinterpret_cast<const V8*>(inm + (size_t)pb * 256 + 8 * cc));
      s.d0 = dalp[want_alpha ? pa : 0];
      s.d1 = dalp[want_alpha ? pb : 0];
    } else {
      const int pe = min(pt0 + ((tid >> 3) & 31), P - 1);
      s.b0 = *reinterpret_cast<const V8*>(inm + (size_t)pe * 64 + 8 * (tid & 7));
    }
  };
  auto commit = [&](const WStage<BF>& s, int pt0, int buf) {
    T* st = lds + buf * WL_STAGE;
    // whole stages (all but a segment's last) skip the row-validity selects: a wave-uniform branch
    const bool full = pt0 + WL_PT <= c1;
    const bool va = full || pt0 + rr < c1, vb = full || pt0 + rr + 16 < c1;
    V8 a0 = s.a0, a1 = s.a1;
    if (!full) { a0 = va ? s.a0 : zero8; a1 = vb ? s.a1 : zero8; }
    *reinterpret_cast<V8*>(st + rr * WL_PITCH + 8 * cc) = a0;
    *reinterpret_cast<V8*>(st + (rr + 16) * WL_PITCH + 8 * cc) = a1;
    V8 b0 = zero8, b1 = zero8;
    if (KW == 256) {
      b0 = s.b0; b1 = s.b1;
      if (!full) { b0 = va ? s.b0 : zero8; b1 = vb ? s.b1 : zero8; }
      *reinterpret_cast<V8*>(st + WL_TILE + rr * WL_PITCH + 8 * cc) = b0;
      *reinterpret_cast<V8*>(st + WL_TILE + (rr + 16) * WL_PITCH + 8 * cc) = b1;
    } else if (tid < 256) {
      b0 = pt0 + (tid >> 3) < c1 ? s.b0 : zero8;
      *reinterpret_cast<V8*>(st + WL_TILE + (tid >> 3) * WL_PITCH + 8 * (tid & 7)) = b0;
    }
    // fp32 riders on the values in flight
    if (jb.flags & WF_BIAS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bias_acc[j] += (float)a0[j] + (float)a1[j];
    }
    if (want_alpha) {
      const float d0 = va ? s.d0 : 0.f, d1 = vb ? s.d1 : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        alpha_acc[j] = fmaf(d1, (float)b1[j], fmaf(d0, (float)b0[j], alpha_acc[j]));
      if (cc == 0) dal_acc += d0 + d1;
    }
  };
  auto compute = [&](int buf) {
    if (!active) return;
    const T* st = lds + buf * WL_STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const V8 a0 = tr_frag<BF>(st, 16 * kk, n0, lane);
      const V8 a1 = tr_frag<BF>(st, 16 * kk, n0 + 32, lane);
#pragma unroll
      for (int u = 0; u < NKT; ++u) {
        const V8 b = tr_frag<BF>(st + WL_TILE, 16 * kk, k0 + 32 * u, lane);
        acc[0][u] = LP<BF>::mfma(a0, b, acc[0][u]);
        acc[1][u] = LP<BF>::mfma(a1, b, acc[1][u]);
      }
    }
  };

  // three register stages (96 KB per CU) in flight ahead of a triple LDS buffer: at the top of
  // stage s LDS buffer s%3 holds stage s and the register sets hold stages s+1, s+2, s+3
  WStage<BF> r0, r1, r2;
  issue(r0, c0);
  issue(r1, c0 + WL_PT);
  issue(r2, c0 + 2 * WL_PT);
  commit(r0, c0, 0);
  issue(r0, c0 + 3 * WL_PT);
  __syncthreads();
#define WL_STEP(RN, BUF, BUFN, K)                                   \
  if (pt0 + ((K) + 1) * WL_PT < c1) commit(RN, pt0 + ((K) + 1) * WL_PT, BUFN); \
  issue(RN, pt0 + ((K) + 4) * WL_PT);                               \
  compute(BUF);                                                     \
  __syncthreads();                                                  \
  if (pt0 + ((K) + 1) * WL_PT >= c1) break;
  for (int pt0 = c0;; pt0 += 3 * WL_PT) {
    WL_STEP(r1, 0, 1, 0)
    WL_STEP(r2, 1, 2, 1)
    WL_STEP(r0, 2, 0, 2)
  }
#undef WL_STEP

  // ---- write the partial (loss scale removed) -------------------------------------------
  if (active) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < NKT; ++u) {
        const int k = k0 + 32 * u + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = n0 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (n < jb.n_rows && k >= jb.kfirst && k < jb.kvalid)
            out[jb.w_off + (size_t)n * jb.ld + jb.kcol0 + (k - jb.kfirst)] = acc[t][u][i] * invS;
        }
      }
  }
  // riders: the 16 row groups of a column are combined through LDS
  float* red = reinterpret_cast<float*>(lds);      // [16][256]
  if (jb.flags & WF_BIAS) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rr * 256 + 8 * cc + j] = bias_acc[j];
    __syncthreads();
    if (tid < jb.n_rows) {
      float s = 0.f;
      for (int g = 0; g < 16; ++g) s += red[g * 256 + tid];
      out[jb.b_off + tid] = s * invS;
    }
    __syncthreads();
  }
  if (KW == 256 && (jb.flags & WF_ALPHA)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rr * 256 + 8 * cc + j] = alpha_acc[j];
    if (cc == 0) red[16 * 256 + rr] = dal_acc;
    __syncthreads();
    if (tid < 256) {
      float s = 0.f;
      for (int g = 0; g < 16; ++g) s += red[g * 256 + tid];
      out[jb.aux_off + tid] = s;
    }
    if (tid == 0) {
      float s = 0.f;
      for (int g = 0; g < 16; ++g) s += red[16 * 256 + g];
      out[jb.aux_off + 256] = s;
    }
  }
}

// ---- format code 2 (8-bit e5m2 dZ / activation rows, mlp_tile_lp.h): the 256-wide jobs contract the rows AS THEY LIE
// IN HBM, on the fp8-family MFMA.  The round-3 design (deleted in round 5; git history) passed every byte through registers (load -> v_perm widening to fp16 -> ds_write ->
// transposing 16-bit read -> fp16 MFMA) and runs commit (830 cycles, VALU) | MFMA (950) | barrier (500) in series per
// 32-point stage.  Here:
//  * the rows go HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), 1 KiB = four 256-byte rows per wave
//    instruction: no staging registers, no commit phase, no VALU work at all; rows past the segment end arrive as
//    zeros through the descriptor's range check;
//  * the MFMA operands are gathered straight from the byte rows with gfx950's ds_read_b64_tr_b8 (a 16-lane group
//    reads an [8 points][16 columns] block, lane j receives column j's eight bytes; lane map probed on the part) -
//    half the LDS bytes of the 16-bit image, no widening;
//  * the contraction is v_mfma_scale_f32_32x32x64_f8f6f4 on bf8 (e5m2) operands with unit block scales (E8M0 127):
//    64 points per instruction at TWICE the rate of the 16-bit / plain fp8 MFMAs (measured 4.5-4.9 PFLOP/s against
//    2.0-2.3; bit-exact products - every e5m2 x e5m2 product and the fp32 accumulation are what the fp16 MFMA gave).
//    A lane's 32 k-values are points 32 (lane >> 5) .. + 31 of the stage (any k order works as long as both
//    operands use the same one).  The first ring version contracted 32-point stages on v_mfma_f32_32x32x16_bf8_bf8
//    and sat at 0.73-0.77 us per stage where the memory side alone gives 0.52-0.57 (knock-outs: without MFMAs 0.52,
//    without fragment reads 0.71, with the MFMAs replaced by s_sleep of their length 0.66): a stage's MFMA time
//    (1152 cycles per SIMD) plus ~600 cycles around the barrier were as long as the stage's memory time, and two
//    bottlenecks of equal length behind one barrier add up.  At twice the rate and half the barriers per point the
//    MFMA side is 40 % of the memory time;
//  * LDS-DMA writes a lane-linear image (no row padding possible inside an instruction), and rows 256 bytes apart
//    share their banks, so the SOURCE is swizzled instead: the lane that fills 16-byte slot c of row r fetches
//    logical chunk c ^ 2 (r & 7).  A transposing read's eight rows then sit in eight different slots, and the two
//    column chunks of a 32-lane pass in the odd / even ones: SQ_LDS_BANK_CONFLICT 0.2 %;
//  * the bias rider is one more MFMA per stage against a constant all-ones operand (every column of its tile is the
//    row sum; wave (m, kw) does n-tile kw of its 64 features); the alpha-head rider (one job) reads the published
//    input rows, two 16-byte pieces per thread and stage, its d alpha scalars ride the ring as 4-byte DMAs;
//  * ring of W8_D slots of 64 points, one barrier per stage in the MIDDLE of the stage's MFMAs: the B fragments
//    of k-tiles 2, 3 are read in front of it (under the MFMAs of k-tiles 0, 1), the next stage's B 0, 1 fragments
//    behind it (under the MFMAs of k-tiles 2, 3) and its A fragments last, into the registers those MFMAs free: 48
//    fragment registers beside the 144 accumulators (a second A set spilled), the A reads' latency is the only
//    one exposed - on a pipe that is 40 % busy.
// What the memory system gives this access pattern through LDS-DMA, no compute (tools/probe_dma.hip): 6.8-7.0 TB/s
// with 256 workgroups, swizzle and nt included, any ring depth from 3 slots - a CU does not keep more than ~2 slots
// of this size in flight however many are queued, which is why deeper rings measured the same.
static_assert(W8_D * W8_SLOT <= WGRAD_LP_LDS_BYTES && WGRAD_LP_LDS_BYTES <= 160 * 1024, "the ring lives in the kernel's LDS allocation");
static_assert(33 * 256 * 4 <= WGRAD_LP_LDS_BYTES, "alpha rider's reduction scratch");

#ifdef WL_DBG
__device__ unsigned long long wl_dbg2[4 * 512];     // per workgroup, its LAST ring job: {loop start, loop end, stages, after epilogue}
#endif
typedef __amdgpu_buffer_rsrc_t lp_rsrc_t;
__device__ __forceinline__ lp_rsrc_t lp_make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <bool ALPHA>
__device__ __forceinline__ void wgrad_lp8_dma_job(const WgradLpNet& a, const WgradLpJob& jb, unsigned char* lds,
                                                  int c0, int c1, float invS, float* __restrict__ out) {
  constexpr int S = W8_S, D = W8_D;
  constexpr int NI = ALPHA ? 5 : 4;                     // LDS-DMA instructions per wave and stage
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef i32x2 __attribute__((address_space(3))) * tr_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int m = wave >> 1, kw = wave & 1;
  const int n0 = m * 64, k0 = kw * 128;
  const bool active = n0 < jb.n_rows;
  const int P = a.P;
  const int npts = c1 - c0;
  const lp_rsrc_t ra = lp_make_rsrc(a.dz + jb.dz_off * 2 + (size_t)c0 * 256, (unsigned)npts * 256u);
  const lp_rsrc_t rb = lp_make_rsrc(a.acts + jb.in_off * 2 + (size_t)c0 * 256, (unsigned)npts * 256u);
  const lp_rsrc_t rd = lp_make_rsrc(a.dz + lp_dz_dalpha_byte(P) + (size_t)c0 * 4, (unsigned)npts * 4u);

  f32x16 acc[2][4], accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    accb[i] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[t][u][i] = 0.f;
  }

  // ---- LDS-DMA of stage st into slot sl: wave w moves rows 8w .. 8w+7 of both tiles (two instructions each; lane L
  // of instruction e fills slot L & 15 of row 8w + 4e + (L >> 4) with the row's logical chunk (L & 15) ^ 2 (row & 7)).
  // Read once: nt.
  // (the 128-row views layer's dZ rows are 128 bytes wide: the lanes of the upper column chunks point past the
  // descriptor's end - zeros, no memory traffic)
  const int lc0 = (lane & 15) ^ (2 * (lane >> 4)), lc1 = (lane & 15) ^ (2 * (4 + (lane >> 4)));
  const bool narrow = jb.n_rows <= 128;
  const int voff0 = narrow && lc0 >= 8 ? 0x40000000 : (lane >> 4) * 256 + (lc0 << 4);          // rows 8w + 0..3
  const int voff1 = narrow && lc1 >= 8 ? 0x40000000 : (lane >> 4) * 256 + (lc1 << 4);          // rows 8w + 4..7
  const int voffb0 = (lane >> 4) * 256 + (lc0 << 4), voffb1 = (lane >> 4) * 256 + (lc1 << 4);   // the input rows: all 256 bytes
  auto issue = [&](int st, int sl) {
    unsigned char* slot = lds + sl * W8_SLOT;
    const int grow = st * S + 8 * wave;                 // row of the segment (past its end: zeros)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(slot + 8 * wave * 256), 16, voff0, grow * 256, 0, 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(slot + (8 * wave + 4) * 256), 16, voff1, (grow + 4) * 256, 0, 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(slot + S * 256 + 8 * wave * 256), 16, voffb0, grow * 256, 0, 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(slot + S * 256 + (8 * wave + 4) * 256), 16, voffb1, (grow + 4) * 256, 0, 2);
    if (ALPHA) {
      if (lane < 8)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_ptr_t)(slot + 2 * S * 256 + 8 * wave * 4), 4, lane * 4, grow * 4, 0, 0);
    }
  };

  // ---- fragments: lane (group g = lane >> 4, i = lane & 15) supplies the address of 8 bytes of row 32 (lane >> 5) +
  // 8 q + (i >> 1) - half (i & 1) of the 16-column chunk (g & 1) of its tile - and receives, in registers 2q, 2q+1 of
  // the operand, column lane & 31 of the tile at those eight points
  struct AF { i32x8 t[2]; };      // A (dZ) fragments of the wave's two n-tiles
  struct BF2 { i32x8 u[2]; };     // B (input) fragments of two of its four k-tiles
  const int fj = (lane & 15) >> 1, gb = (lane >> 4) & 1;
  const int lane_row = (32 * hh + fj) * 256 + (lane & 1) * 8;
  // (fragment t of wave (m, kw) is n-tile t ^ kw of its 64 features: the tile whose bias it sums is always t = 0)
  const int fa = lane_row + (((4 * m + 2 * kw + gb) ^ (2 * fj)) << 4);        // fragment 0 (fragment 1: ^ 32)
  const int fb = S * 256 + lane_row + (((8 * kw + gb) ^ (2 * fj)) << 4);      // k-tile 0 (k-tile u: ^ 32 u)
  auto read8 = [&](i32x8& f, const unsigned char* p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((tr_ptr_t)(p + q * 8 * 256));
      f[2 * q] = v[0];
      f[2 * q + 1] = v[1];
    }
  };
  auto read_A = [&](AF& f, int sl) {
    if (!active) return;
    const unsigned char* slot = lds + sl * W8_SLOT;
    read8(f.t[0], slot + fa);
    read8(f.t[1], slot + (fa ^ 32));
  };
  auto read_B = [&](BF2& f, int sl, int pair) {
    if (!active) return;
    const unsigned char* slot = lds + sl * W8_SLOT;
    read8(f.u[0], slot + (fb ^ (64 * pair)));
    read8(f.u[1], slot + (fb ^ (64 * pair + 32)));
  };
  constexpr int UNIT = 0x7F7F7F7F;                      // E8M0 block scales 2^0
  const i32x8 ones = {0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C};   // e5m2 1.0
#define W8_MFMA(A_, B_, C_) C_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A_, B_, C_, 1, 1, 0, UNIT, 0, UNIT)
  auto mfma_pair = [&](const AF& A, const BF2& B, int pair) {
    if (!active) return;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t) W8_MFMA(A.t[t], B.u[j], acc[t][2 * pair + j]);
    if (pair == 0) W8_MFMA(A.t[0], ones, accb);
  };
#undef W8_MFMA

  // ---- alpha-head rider: thread (rows rr, rr + 32; slot pc) holds the 16 columns of logical chunk pc ^ 2 (rr & 7)
  const int rr = tid >> 4, pc = tid & 15;
  float alpha_acc[16], dal_acc = 0.f, rd_d[2] = {0.f, 0.f};
  lp_u32x4 rd_w[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
#pragma unroll
  for (int j = 0; j < 16; ++j) alpha_acc[j] = 0.f;
  auto riders_read = [&](int sl) {
    if (!ALPHA) return;
    const unsigned char* slot = lds + sl * W8_SLOT;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      rd_w[h] = *reinterpret_cast<const lp_u32x4*>(slot + S * 256 + (rr + 32 * h) * 256 + pc * 16);
      rd_d[h] = *reinterpret_cast<const float*>(slot + 2 * S * 256 + (rr + 32 * h) * 4);
    }
  };
  auto riders_add = [&]() {
    if (!ALPHA) return;
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f2 lo = __builtin_amdgcn_cvt_pk_f32_bf8((int)rd_w[h][q], false), hi = __builtin_amdgcn_cvt_pk_f32_bf8((int)rd_w[h][q], true);
        alpha_acc[4 * q + 0] = fmaf(rd_d[h], lo[0], alpha_acc[4 * q + 0]);
        alpha_acc[4 * q + 1] = fmaf(rd_d[h], lo[1], alpha_acc[4 * q + 1]);
        alpha_acc[4 * q + 2] = fmaf(rd_d[h], hi[0], alpha_acc[4 * q + 2]);
        alpha_acc[4 * q + 3] = fmaf(rd_d[h], hi[1], alpha_acc[4 * q + 3]);
      }
      if (pc == 0) dal_acc += rd_d[h];
    }
  };

  // ---- pipeline.  Stages past the segment are issued too (zeros, no memory traffic): the vmcnt stays a constant.
  const int ns = (npts + S - 1) / S;
#pragma unroll
  for (int st = 0; st < D; ++st) issue(st, st);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
  lp_lds_barrier();
  AF A = {};
  BF2 B01 = {}, B23 = {};
  read_A(A, 0);
  read_B(B01, 0, 0);
  int sl = 0;
#ifdef WL_DBG
  if (tid == 0) { wl_dbg2[4 * blockIdx.x] = wall_clock64(); wl_dbg2[4 * blockIdx.x + 2] = (unsigned long long)ns; }
#endif
  for (int st = 0; st < ns; ++st) {
    const int sl1 = sl + 1 == D ? 0 : sl + 1;
    read_B(B23, sl, 1);
    riders_read(sl);
    __builtin_amdgcn_sched_barrier(0);
    mfma_pair(A, B01, 0);
    __builtin_amdgcn_sched_barrier(0);
    riders_add();
    // stage st+1: this wave's pieces have landed (D-2 younger stages stay in flight) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NI) : "memory");
    lp_lds_barrier();                             // ... and so have everybody's; slot sl is read out
    issue(st + D, sl);
    read_B(B01, sl1, 0);                          // (its registers are free; A's are not until the MFMAs below are out)
    __builtin_amdgcn_sched_barrier(0);
    mfma_pair(A, B23, 1);
    __builtin_amdgcn_sched_barrier(0);
    read_A(A, sl1);
    sl = sl1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land on the scratch below / after the job
#ifdef WL_DBG
  if (tid == 0) wl_dbg2[4 * blockIdx.x + 1] = wall_clock64();
#endif

  // ---- write the partial (loss scale removed)
  if (active) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = k0 + 32 * u + r;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = n0 + 32 * (t ^ kw) + (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (n < jb.n_rows && k >= jb.kfirst && k < jb.kvalid)
            out[jb.w_off + (size_t)n * jb.ld + jb.kcol0 + (k - jb.kfirst)] = acc[t][u][i] * invS;
        }
      }
    if (r == 0) {                                 // every column of the bias tile holds the row sums
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + 32 * kw + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (n < jb.n_rows) out[jb.b_off + n] = accb[i] * invS;
      }
    }
  }
#ifdef WL_DBG
  __syncthreads();
  if (tid == 0) wl_dbg2[4 * blockIdx.x + 3] = wall_clock64();
#endif
  if (ALPHA) {
    __syncthreads();                              // every wave is through its reads and its DMAs: the ring is free
    float* red = reinterpret_cast<float*>(lds);   // [32 rows][256] (+ 32)
    const int lc = pc ^ (2 * (rr & 7));
#pragma unroll
    for (int j = 0; j < 16; ++j) red[rr * 256 + 16 * lc + j] = alpha_acc[j];
    if (pc == 0) red[32 * 256 + rr] = dal_acc;
    __syncthreads();
    if (tid < 256) {
      float sum = 0.f;
      for (int g = 0; g < 32; ++g) sum += red[g * 256 + tid];
      out[jb.aux_off + tid] = sum;
    }
    if (tid == 0) {
      float sum = 0.f;
      for (int g = 0; g < 32; ++g) sum += red[32 * 256 + g];
      out[jb.aux_off + 256] = sum;
    }
  }
}

// The embedding-input jobs (layer 0, the skip block of layer 5, the view-direction columns of the views layer:
// 256 or 128 features x 64 embedding columns) on the same ring: dZ rows as above, the fp8 e4m3 embedding rows (64
// bytes per point, mlp_tile_lp.h) as the B operand of the same MFMA (mixed bf8 x fp8).  Wave (m, kw) owns 64 features
// x embedding columns 32 kw .. + 31: two accumulators, two MFMAs per stage (+ the bias rider of layer 0), so the
// fragments are simply read behind the barrier - the job is all memory (20 KB per 64-point stage).  The 64-byte rows
// need their own source swizzle: rows 4 apart share their banks, so rows 4..7 of every eight fetch their chunks
// c ^ 2 - a transposing read's 8 rows x 2 chunks then cover 256 different bytes of the bank line.
__device__ __forceinline__ void wgrad_lp8_dma_emb_job(const WgradLpNet& a, const WgradLpJob& jb, unsigned char* lds,
                                                      int c0, int c1, float invS, float* __restrict__ out) {
  constexpr int S = W8_S, D = W8_D, NI = 3;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  typedef int i32x8 __attribute__((ext_vector_type(8)));
  typedef i32x2 __attribute__((address_space(3))) * tr_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int m = wave >> 1, kw = wave & 1;
  const int n0 = m * 64;
  const bool active = n0 < jb.n_rows;
  const int P = a.P;
  const int npts = c1 - c0;
  const lp_rsrc_t ra = lp_make_rsrc(a.dz + jb.dz_off * 2 + (size_t)c0 * 256, (unsigned)npts * 256u);
  const lp_rsrc_t rb = lp_make_rsrc(a.acts + acts_emb_off(P) * 2 + (size_t)c0 * 64, (unsigned)npts * 64u);

  f32x16 acc[2], accb;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; accb[i] = 0.f; }

  const int lc0 = (lane & 15) ^ (2 * (lane >> 4)), lc1 = (lane & 15) ^ (2 * (4 + (lane >> 4)));
  const bool narrow = jb.n_rows <= 128;            // (the views layer's dZ rows: 128 bytes wide, see wgrad_lp8_dma_job)
  const int voff0 = narrow && lc0 >= 8 ? 0x40000000 : (lane >> 4) * 256 + (lc0 << 4);
  const int voff1 = narrow && lc1 >= 8 ? 0x40000000 : (lane >> 4) * 256 + (lc1 << 4);
  // embedding rows 8w .. 8w+7 (lanes 0-31: row 8w + (L >> 2), slot L & 3 <- chunk (L & 3) ^ 2 ((row >> 2) & 1))
  const int voffe = (lane >> 2) * 64 + (((lane & 3) ^ (2 * ((lane >> 4) & 1))) << 4);
  auto issue = [&](int st, int sl) {
    unsigned char* slot = lds + sl * W8_SLOT;
    const int grow = st * S + 8 * wave;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(slot + 8 * wave * 256), 16, voff0, grow * 256, 0, 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(slot + (8 * wave + 4) * 256), 16, voff1, (grow + 4) * 256, 0, 2);
    if (lane < 32)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(slot + S * 256 + 8 * wave * 64), 16, voffe, grow * 64, 0, 2);
  };
  const int fj = (lane & 15) >> 1, gb = (lane >> 4) & 1;
  const int fa = (32 * hh + fj) * 256 + (lane & 1) * 8 + (((4 * m + 2 * kw + gb) ^ (2 * fj)) << 4);   // fragment 0 = n-tile kw
  const int fe = S * 256 + (32 * hh + fj) * 64 + (lane & 1) * 8 + (((2 * kw + gb) ^ (2 * (fj >> 2))) << 4);
  constexpr int UNIT = 0x7F7F7F7F;
  const i32x8 ones = {0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C, 0x3C3C3C3C};
  const bool want_bias = (jb.flags & WF_BIAS) != 0;

  const int ns = (npts + S - 1) / S;
#pragma unroll
  for (int st = 0; st < D; ++st) issue(st, st);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
  lp_lds_barrier();
  int sl = 0;
  for (int st = 0; st < ns; ++st) {
    if (active) {
      const unsigned char* slot = lds + sl * W8_SLOT;
      i32x8 a0, a1, b;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const i32x2 v0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((tr_ptr_t)(slot + fa + q * 8 * 256));
        const i32x2 v1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((tr_ptr_t)(slot + (fa ^ 32) + q * 8 * 256));
        const i32x2 vb = __builtin_amdgcn_ds_read_tr8_b64_v2i32((tr_ptr_t)(slot + fe + q * 8 * 64));
        a0[2 * q] = v0[0]; a0[2 * q + 1] = v0[1];
        a1[2 * q] = v1[0]; a1[2 * q + 1] = v1[1];
        b[2 * q] = vb[0]; b[2 * q + 1] = vb[1];
      }
      acc[0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, b, acc[0], 1, 0, 0, UNIT, 0, UNIT);    // bf8 x fp8
      acc[1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a1, b, acc[1], 1, 0, 0, UNIT, 0, UNIT);
      if (want_bias) accb = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a0, ones, accb, 1, 1, 0, UNIT, 0, UNIT);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NI) : "memory");
    lp_lds_barrier();
    issue(st + D, sl);
    sl = sl + 1 == D ? 0 : sl + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if (active) {
    const int k = 32 * kw + r;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + 32 * (t ^ kw) + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (n < jb.n_rows && k >= jb.kfirst && k < jb.kvalid)
          out[jb.w_off + (size_t)n * jb.ld + jb.kcol0 + (k - jb.kfirst)] = acc[t][i] * invS;
      }
    if (want_bias && r == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + 32 * kw + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (n < jb.n_rows) out[jb.b_off + n] = accb[i] * invS;
      }
    }
  }
}

// rgb head: dW_r[c][k] = sum_pt g[pt][c] * hv[pt][k], db_r[c] = sum_pt g[pt][c].
// A thread owns 8 columns (one 16-byte load per point) of every 32nd point, four points in flight:
// this job is pure load latency, and as the LAST job of the table its workgroups set the kernel's end.
template <bool BF, bool S8 = false>
__device__ __forceinline__ void wgrad_rgb_lp_job(const WgradLpNet& a, const WgradLpJob& jb, float* lds,
                                                 int c0, int c1, float* __restrict__ out) {
  typedef typename LP<BF>::T T;
  typedef typename LP<BF>::V8 V8;
  const int tid = threadIdx.x;
  const int k8 = tid & 15, pl = tid >> 4;        // 16 column groups x 32 point lanes
  const T* __restrict__ hv = reinterpret_cast<const T*>(a.acts) + jb.in_off + 8 * k8;
  const unsigned char* __restrict__ hv8 = a.acts + jb.in_off * 2 + 8 * k8;     // format code 2: 8-bit rows
  const int P = a.P;
  float s[3][8], b[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[c][j] = 0.f;
  // software-pipelined: the eight loads of the NEXT 128 points are in flight while this thread's four points
  // are accumulated (one round trip per 128 points left each of these workgroups a straggler of the balanced
  // launch: the job is pure load latency and runs beside 250 streaming workgroups)
  struct Set { V8 h[4]; f32x4 g[4]; };
  auto fetch = [&](Set& st, int pt0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pt = min(pt0 + 32 * q, P - 1);
      if (S8) {
        const lp_u32x2 w8 = *reinterpret_cast<const lp_u32x2*>(hv8 + (size_t)pt * 256);
        const lp_u32x2 q0 = lp_unpack4_bf8<BF>(w8[0]), q1 = lp_unpack4_bf8<BF>(w8[1]);
        st.h[q] = __builtin_bit_cast(V8, lp_u32x4{q0[0], q0[1], q1[0], q1[1]});
      } else {
        st.h[q] = *reinterpret_cast<const V8*>(hv + (size_t)pt * 256);
      }
      st.g[q] = *reinterpret_cast<const f32x4*>(a.g_out + (size_t)pt * 4);
    }
  };
  auto accumulate = [&](const Set& st, int pt0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (pt0 + 32 * q < c1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = (float)st.h[q][j];
          s[0][j] = fmaf(st.g[q][0], x, s[0][j]); s[1][j] = fmaf(st.g[q][1], x, s[1][j]);
          s[2][j] = fmaf(st.g[q][2], x, s[2][j]);
        }
        b[0] += st.g[q][0]; b[1] += st.g[q][1]; b[2] += st.g[q][2];
      }
    }
  };
  {
    Set s0, s1;
    int pt0 = c0 + pl;
    fetch(s0, pt0);                         // (clamped rows beyond c1 are loaded and never accumulated)
    for (; pt0 < c1; pt0 += 256) {
      fetch(s1, pt0 + 128);
      accumulate(s0, pt0);
      fetch(s0, pt0 + 256);
      accumulate(s1, pt0 + 128);
    }
  }
  float* red = lds;                               // [32 point lanes][3][128] + [32][4]
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[(pl * 3 + c) * 128 + 8 * k8 + j] = s[c][j];
  float* redb = red + 32 * 3 * 128;
  if (k8 == 0) { redb[pl * 4 + 0] = b[0]; redb[pl * 4 + 1] = b[1]; redb[pl * 4 + 2] = b[2]; }
  __syncthreads();
  if (tid < 384) {
    float t = 0.f;
    for (int p = 0; p < 32; ++p) t += red[p * 384 + tid];
    out[jb.w_off + tid] = t;
  }
  if (tid < 3) {
    float t = 0.f;
    for (int p = 0; p < 32; ++p) t += redb[p * 4 + tid];
    out[jb.b_off + tid] = t;
  }
}

#ifdef WL_DBG
// -DWL_DBG (a variant build: SCADE_AB_FLAGS, scade_amd/build.py): per workgroup {start, stages, entries, end} on the
// 100 MHz wall clock, read back with scade_debug_wl - how the per-stage weights of lp_job_weights() were measured
__device__ unsigned long long wl_dbg[4 * 512];
#endif
template <bool BF, bool S8 = false>
__global__ __launch_bounds__(512, 2) void mlp_wgrad_lp_kernel(WgradLpArgs aa) {
  typedef typename LP<BF>::T T;
  extern __shared__ __attribute__((aligned(16))) unsigned short ldsw16[];
  const WgradLpPlan& pl = aa.plan;
  const int w = (int)blockIdx.x;
#ifdef WL_DBG
  int dbg_k = 1;
  if (threadIdx.x == 0) { wl_dbg[4 * w] = wall_clock64(); wl_dbg[4 * w + 1] = wl_dbg[4 * w + 2] = 0; }
#endif
  const bool grid_mode = pl.chunk > 0;                  // small launches: one (job, chunk) per workgroup
  const int gx = pl.gx0 + pl.gx1;
  const int gj = grid_mode ? w / gx : 0, gb = grid_mode ? w - gj * gx : 0;
  const long b0 = grid_mode ? 0 : pl.bound[w], b1 = grid_mode ? 0 : pl.bound[w + 1];
  const int e_first = grid_mode ? gj + (gb >= pl.gx0 ? aa.njobs : 0) : 0;
  const int e_last = grid_mode ? e_first + 1 : 2 * aa.njobs;
  bool first = true;
  for (int e = e_first; e < e_last; ++e) {
    const int ce = pl.cum[e], ce1 = pl.cum[e + 1];
    if (!grid_mode && (ce1 == ce || (long)ce >= b1 || (long)ce1 <= b0)) continue;      // empty entry / not this workgroup's
    const bool second = e >= aa.njobs;                                 // wave-uniform: scalar selects
    const WgradLpNet& a = second ? aa.net[1] : aa.net[0];
    const int j = e - (second ? aa.njobs : 0);
    const int wj = pl.weight[j], nst = second ? pl.nst[1] : pl.nst[0];
    // stage(x) = clamp(floor((x - cum) / weight), 0, nst): the same function on both sides of every boundary
    const long x0 = b0 - ce, x1 = b1 - ce;
    const int s0 = x0 <= 0 ? 0 : (int)min((long)nst, x0 / wj);
    const int s1 = (int)min((long)nst, x1 / wj);
    WgradLpJob jb = aa.jobs[j];
    jb.dz_off = acts_slot_off(a.P, jb.dz_slot);
    jb.in_off = jb.in_slot >= 0 ? acts_slot_off(a.P, jb.in_slot) : acts_emb_off(a.P);
    const int gc = gb - (second ? pl.gx0 : 0);          // grid mode: chunk index within the network
    const int c0 = grid_mode ? gc * pl.chunk : s0 * WL_PT;
    int c1 = grid_mode ? min(a.P, c0 + pl.chunk)
                       : min(a.P, max(s1, s0) * WL_PT);       // an empty segment (c1 == c0) still writes its zero row
    float* out = a.partial + (size_t)(grid_mode ? gc : w - pl.first_wg[e]) * N_PARAM_FLOATS;
    const float invS = (BF && !S8) ? 1.0f : 1.0f / lp_loss_scale(lp_read_gmax(a.gmax));
    if (!first) __syncthreads();                        // the previous segment's riders still read the LDS
    first = false;
    if (jb.flags & WF_RGB) {
      wgrad_rgb_lp_job<BF, false>(a, jb, reinterpret_cast<float*>(ldsw16), c0, c1, out);   // (its input slot is 16-bit in every format)
    } else if (S8) {
      if (jb.kw == 256) {
        if (jb.flags & WF_ALPHA) wgrad_lp8_dma_job<true>(a, jb, reinterpret_cast<unsigned char*>(ldsw16), c0, c1, invS, out);
        else wgrad_lp8_dma_job<false>(a, jb, reinterpret_cast<unsigned char*>(ldsw16), c0, c1, invS, out);
      } else {
        wgrad_lp8_dma_emb_job(a, jb, reinterpret_cast<unsigned char*>(ldsw16), c0, c1, invS, out);
      }
    } else if (jb.kw == 256) {
      wgrad_lp_job<BF, 256>(a, jb, reinterpret_cast<T*>(ldsw16), c0, c1, invS, out);
    } else {
      wgrad_lp_job<BF, 64>(a, jb, reinterpret_cast<T*>(ldsw16), c0, c1, invS, out);
    }
#ifdef WL_DBG
    if (threadIdx.x == 0) { wl_dbg[4 * w + 1] += (unsigned long long)((c1 - c0 + 31) / 32); wl_dbg[4 * w + 2] = wl_dbg[4 * w + 2] * 100 + e + 1; }
    ++dbg_k;
#endif
  }
#ifdef WL_DBG
  if (threadIdx.x == 0) wl_dbg[4 * w + 3] = wall_clock64();
#endif
}
#ifdef WL_DBG
}  // namespace scade
extern "C" int scade_debug_wl(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out + 4 * 512, HIP_SYMBOL(scade::wl_dbg2), sizeof(unsigned long long) * 4 * 512) != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scade::wl_dbg), sizeof(unsigned long long) * 4 * 512);
}
namespace scade {
#endif

// (mlp_reduce.h: LP_OFF, lp_param_job, reduce_rows4 - shared with scade_step_finish)
static_assert(MAX_WGRAD_JOBS == REDUCE_MAX_JOBS, "the reduce descriptor holds one row count per job");

// sum of the per-segment partial rows: element x of network n = sum over the rows [0, nseg[n][job(x)]) of
// partial[n]; one float4 per thread
struct ReduceLpArgs {
  const float* partial[2];
  float* grad[2];
  unsigned char nseg[2][MAX_WGRAD_JOBS];
  int uniform[2];      // > 0: every job of the network has this many rows (the small-launch grid): no lookup
};
__global__ static void wgrad_lp_reduce_kernel(ReduceLpArgs r) {
  const bool second = blockIdx.x >= WGRAD_REDUCE_BLOCKS;
  const int i = (blockIdx.x - (second ? WGRAD_REDUCE_BLOCKS : 0)) * 256 + threadIdx.x;
  if (i >= N_PARAM_FLOATS / 4) return;
  reinterpret_cast<f32x4*>(second ? r.grad[1] : r.grad[0])[i] =
      reduce_rows4(second ? r.partial[1] : r.partial[0], second ? r.nseg[1] : r.nseg[0],
                   second ? r.uniform[1] : r.uniform[0], i);
}

// fills the job table (13 jobs; identical for every network: offsets are slots)
static void build_wgrad_lp_jobs(WgradLpArgs& w) {
  int off[N_PARAM_TENSORS + 1];
  param_offsets(off);
  int nj = 0;
  constexpr int EMB_SLOT = -1;
  auto add = [&](int dzs, int ins, int in_stride, int kw, int nrows, int woff, int ld, int kcol0, int kfirst,
                 int kvalid, int boff, int flags, int aux) {
    WgradLpJob& j = w.jobs[nj++];
    j.dz_off = 0; j.in_off = 0; j.dz_slot = dzs; j.in_slot = ins; j.in_stride = in_stride; j.kw = kw; j.n_rows = nrows;
    j.w_off = woff; j.ld = ld; j.kcol0 = kcol0; j.kfirst = kfirst; j.kvalid = kvalid; j.b_off = boff; j.flags = flags;
    j.aux_off = aux;
  };
  for (int l = 1; l <= 7; ++l) {
    const int ld = l == 5 ? 313 : 256, kc0 = l == 5 ? 57 : 0;
    add(l, l - 1, 256, 256, 256, off[2 * l], ld, kc0, 0, 256, off[2 * l + 1], WF_BIAS, 0);
  }
  add(SLOT_FEAT, 7, 256, 256, 256, off[18], 256, 0, 0, 256, off[19], WF_BIAS | WF_ALPHA, off[20]);
  add(SLOT_VIEWS_H, SLOT_FEAT, 256, 256, 128, off[16], 259, 0, 0, 256, off[17], WF_BIAS, 0);
  add(0, EMB_SLOT, 64, 64, 256, off[0], 57, 0, 0, 57, off[1], WF_BIAS, 0);
  add(5, EMB_SLOT, 64, 64, 256, off[10], 313, 0, 0, 57, 0, 0, 0);
  // view-direction columns of views_linears.0: emb columns 60..62 -> weight columns 256..258
  add(SLOT_VIEWS_H, EMB_SLOT, 64, 64, 128, off[16], 259, 256, 60, 63, 0, 0, 0);
  add(0, SLOT_VIEWS_H, 256, 0, 0, off[22], 128, 0, 0, 0, off[23], WF_RGB, 0);
  w.njobs = nj;
}
// Per-stage cost of the jobs in build_wgrad_lp_jobs' order, in 32nds of a 256 x 256 layer's: MEASURED per
// workgroup with the kernel's own timestamps (tools/scratch: -DWL_DBG) under the running mix, 1024-ray step -
// bf16 16 / 16.9 / 13.4 / 10.3 / 9.2 / 9.1 / 3.7 (layer, feature + alpha rider, views, the three embedding-input
// jobs, rgb head; in 16ths), 8-bit rows 16 / 17.7 / 12.6 / 9.8 / 8.8 / 8.5 / 3.8, fp16 as bf16.  Bytes per stage
// alone would say 16 / 16 / 16 / 10 / 10 / 10 / 8.
static constexpr int LP_JOB_WEIGHTS[7] = {32, 35, 26, 20, 18, 17, 8};
// format code 2 since round 4: every job but the rgb head runs on the LDS-DMA ring at the memory system's rate for
// one workgroup per CU (~25 GB/s per CU with all 256 streaming), and so does the rgb head's register pipeline - a
// stage costs its BYTES: 16 KB per 32 points for a layer, + the d alpha scalars, 12 KB for the 128-row views
// layer (its dZ rows are 128 bytes wide), 10 KB for the embedding-input jobs (8 KB of dZ + 2 KB of fp8 embedding
// rows: measured 10.4 / 16), 8.7 KB for the rgb head (16-bit views rows + g_out: measured 8.5 - 10.4 / 16).
static constexpr int LP8_JOB_WEIGHTS[7] = {32, 34, 24, 21, 21, 17, 21};
static int lp_job_weight(int j, bool s8) {
  const int* w = s8 ? LP8_JOB_WEIGHTS : LP_JOB_WEIGHTS;
  return j < 7 ? w[0] : w[j - 6];      // jobs 0..6: layers 1..7; 7 feature, 8 views, 9..11 embedding jobs, 12 rgb
}
// partial rows a network's workspace holds: no entry is cut into more segments than this (its share of the
// workgroups, rounded up, plus the two it may share with its neighbours)
static int lp_rows_bound(bool s8) {
  WgradLpArgs w{};
  build_wgrad_lp_jobs(w);
  int sum = 0, mx = 0;
  for (int j = 0; j < w.njobs; ++j) {
    const int v = lp_job_weight(j, s8);
    sum += v;
    mx = v > mx ? v : mx;
  }
  return (device_cus() * mx + sum - 1) / sum + 6;    // (+ slack: the fixed per-segment costs shorten the segments)
}
// rows every workspace holds (scade_mlp_bwd_lp_workspace_bytes): either format
static int lp_ws_rows() {
  const int a = lp_rows_bound(false), b = lp_rows_bound(true);
  return a > b ? a : b;
}
// fixed cost of one segment of job j (its prologue round trip + the stores of its partial row), in the units of
// lp_job_weights(): 80 + 460 x (elements written / 65536); the 90 us the partial stores cost the 1024-ray launch
// (knock-out) were 20 us per 256 x 256 segment = 15-19 of its stages
static int lp_job_fixed(const WgradLpJob& j) {
  const long elems = (j.flags & WF_RGB) ? 3 * 128 + 3 : (long)j.n_rows * (j.kvalid - j.kfirst) + ((j.flags & WF_BIAS) ? j.n_rows : 0);
  return 80 + (int)(460 * elems / 65536);
}
// fills w.plan for networks of P[0] and P[1] (0 = absent) points; nseg[n][j] = partial rows of job j of network n
static int lp_build_plan(WgradLpArgs& w, const int* P, bool s8, unsigned char nseg[2][MAX_WGRAD_JOBS]) {
  WgradLpPlan& pl = w.plan;
  const int nj = w.njobs, ne = 2 * nj;
  long cum = 0;
  for (int n = 0; n < 2; ++n) {
    pl.nst[n] = (P[n] + WL_PT - 1) / WL_PT;
    for (int j = 0; j < nj; ++j) {
      pl.cum[n * nj + j] = (int)cum;
      cum += (long)pl.nst[n] * lp_job_weight(j, s8);
      SCADE_REQUIRE(cum < (1L << 30), -2, "scade_mlp_bwd_lp: %d + %d points are more than one launch takes", P[0], P[1]);
    }
  }
  pl.cum[ne] = (int)cum;
  int fixed[MAX_WGRAD_JOBS];
  for (int j = 0; j < nj; ++j) {
    pl.weight[j] = (unsigned char)lp_job_weight(j, s8);
    fixed[j] = lp_job_fixed(w.jobs[j]);
  }
  const long W = cum;
  pl.chunk = pl.gx0 = pl.gx1 = 0;
  // Small launches keep the round-2 grid of (job, chunk) workgroups handed out by the dispatcher: below ~4 stages
  // per CU and job the stage costs are latency, not bytes (the weights above do not hold), and a third more
  // workgroups than CUs lets one's partial-row stores overlap another's loop (128 rays per GPU: 0.325 ms graphed
  // bf16 step against 0.333 with the balanced plan; 1024 rays: 1.45 against 1.37, 4096 rays: 5.5 against 5.2).
  constexpr long LP_GRID_BELOW = 100000;
  if ((long)P[0] + P[1] < LP_GRID_BELOW) {
    const long Pt = (long)P[0] + P[1];
    // ONE round: chunks x jobs <= CUs (19 chunks on 256 CUs; 28 chunks = 1.4 rounds measured 3 % slower on the
    // 128- and 256-ray graph steps, 14 chunks 3 % slower too), chunks never shorter than 512 points
    int nchunks = device_cus() / nj;
    const long nmax = Pt / 512 > 1 ? Pt / 512 : 1;
    nchunks = nchunks < 1 ? 1 : nchunks > nmax ? (int)nmax : nchunks;
    const int rows_max = lp_ws_rows();
    nchunks = nchunks > rows_max - 1 ? rows_max - 1 : nchunks;
    int chunk = (int)((Pt + nchunks - 1) / nchunks);
    chunk = (chunk + WL_PT - 1) / WL_PT * WL_PT;
    pl.chunk = chunk;
    pl.gx0 = (P[0] + chunk - 1) / chunk;
    pl.gx1 = (P[1] + chunk - 1) / chunk;
    SCADE_REQUIRE(pl.gx0 <= rows_max && pl.gx1 <= rows_max, -9, "scade_mlp_bwd_lp: internal: %d / %d chunks exceed the %d partial rows",
                  pl.gx0, pl.gx1, rows_max);
    pl.nwg = (pl.gx0 + pl.gx1) * nj;
    for (int j = 0; j < nj; ++j) { nseg[0][j] = (unsigned char)pl.gx0; nseg[1][j] = (unsigned char)pl.gx1; }
    return 0;
  }
  pl.nwg = device_cus() < LP_MAX_WG ? device_cus() : LP_MAX_WG;
  // greedy walk with per-workgroup budget C: a segment costs its job's fixed part plus its stages; the smallest
  // C that places all W positions (bisection: the walk is monotone in C) balances the launch
  auto walk = [&](long C, bool store) {
    long pos = 0;
    int e = 0;
    for (int g = 0; g < pl.nwg; ++g) {
      if (store) pl.bound[g] = (int)pos;
      long budget = C;
      while (pos < W) {
        while (pl.cum[e + 1] <= pos) ++e;
        const long f = fixed[e % nj], wj = pl.weight[e % nj];
        if (budget < f + wj) break;                       // not worth opening (or continuing into) this entry
        budget -= f;
        long take = pl.cum[e + 1] - pos;
        if (take > budget) take = budget / wj * wj;       // whole stages
        pos += take;
        budget -= take;
        if (pos < pl.cum[e + 1]) break;
      }
    }
    if (store) pl.bound[pl.nwg] = (int)W;
    return pos >= W;
  };
  // (never so small that an entry is cut into more segments than the workspace has rows: tiny launches then
  // simply leave the last workgroups without work)
  const int rows = lp_ws_rows();
  long lo = 1, hi = W + 4096;
  for (int e = 0; e < ne; ++e) {
    const long need = fixed[e % nj] + ((long)pl.cum[e + 1] - pl.cum[e] + rows - 4) / (rows - 3) + pl.weight[e % nj];
    lo = need > lo ? need : lo;
  }
  hi = hi > lo ? hi : lo;
  while (lo < hi) {
    const long mid = (lo + hi) / 2;
    if (walk(mid, false)) hi = mid; else lo = mid + 1;
  }
  walk(lo, true);
  for (int e = 0; e < ne; ++e) {
    const long elo = pl.cum[e], ehi = pl.cum[e + 1];
    if (ehi == elo) { pl.first_wg[e] = 0; nseg[e / nj][e % nj] = 0; continue; }
    int a = 0, b = 0;
    for (int g = 0; g < pl.nwg; ++g) {
      if (pl.bound[g] <= elo) a = g;                      // a = max { g : bound[g] <= lo }
      if (pl.bound[g] < ehi) b = g;                       // b = max { g : bound[g] <  hi }
    }
    SCADE_REQUIRE(b - a + 1 <= rows, -9, "scade_mlp_bwd_lp: internal: %d segments of entry %d exceed the %d partial rows",
                  b - a + 1, e, rows);
    pl.first_wg[e] = (short)a;
    nseg[e / nj][e % nj] = (unsigned char)(b - a + 1);
  }
  return 0;
}

}  // namespace scade

using namespace scade;

extern "C" long scade_mlp_packed_t_lp_bytes(void) { return PACKED_T_LP_BYTES; }

extern "C" long scade_mlp_bwd_lp_workspace_bytes(int P) {
  // [dz | partial rows | max|g_out| slots]; the row count covers every format and every partner network
  return lp_dz_bytes(P) + (long)lp_ws_rows() * N_PARAM_FLOATS * 4 + LP_GMAX_SLOTS * 4;
}

extern "C" int scade_mlp_pack_t_lp(const float* const* params, void* packed_t_lp, int bf16, void* stream) {
  SCADE_REQUIRE(params && packed_t_lp, -1, "scade_mlp_pack_t_lp: null pointer");
  PackTLpArgs a;
  for (int i = 0; i < N_PARAM_TENSORS; ++i) {
    SCADE_REQUIRE(params[i], -1, "scade_mlp_pack_t_lp: params[%d] is null", i);
    a.p[i] = params[i];
  }
  a.packed = packed_t_lp;
  if (bf16) hipLaunchKernelGGL(mlp_pack_t_lp_kernel<true>, dim3(PACK_BLOCKS, PACK_T_ROWS), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(mlp_pack_t_lp_kernel<false>, dim3(PACK_BLOCKS, PACK_T_ROWS), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_mlp_pack_t_lp");
}

template <bool BF, bool S8>
static int lp_bwd_set_attr() {
  static unsigned long long attr_set = 0;   // one bit per device ordinal
  if (scade_attr_needed(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_lp_kernel<BF, 4, S8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, dgrad_lp_lds_bytes(4));
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd_lp: hipFuncSetAttribute: %s", hipGetErrorString(e));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_lp_kernel<BF, 2, S8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, dgrad_lp_lds_bytes(2));
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd_lp: hipFuncSetAttribute: %s", hipGetErrorString(e));
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wgrad_lp_kernel<BF, S8>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_LP_LDS_BYTES);
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd_lp: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(attr_set);
  }
  return 0;
}

// the launch-wide loss scale (fp16 rows, and the 8-bit rows of format code 2; plain bf16 uses S = 1 and never
// reads gmax)
static int lp_gmax_blocks(int P) {
  const long want = (4L * P + LP_GMAX_SLOTS * 16 - 1) / (LP_GMAX_SLOTS * 16);       // 16 values per thread
  return (int)(want < LP_GMAX_SLOTS ? want : LP_GMAX_SLOTS);
}
// one launch for one network (P[1] = 0) or for both networks of a joint launch
static int lp_launch_gmax(const float* const* g_out, const int* P, float* const* gmax, const unsigned char* const* acts,
                          hipStream_t s) {
  LpGmaxArgs a{};
  int blocks = 0;
  for (int i = 0; i < 2 && P[i] > 0; ++i) {
    a.g[i] = g_out[i]; a.n[i] = 4L * P[i]; a.slots[i] = gmax[i];
    a.alpha_pre[i] = reinterpret_cast<const float*>(acts[i] + lp_acts_alpha_byte(P[i]));
    if (i == 0) a.blocks0 = lp_gmax_blocks(P[0]);
    blocks += lp_gmax_blocks(P[i]);
  }
  hipLaunchKernelGGL(lp_gmax_kernel, dim3(blocks), dim3(LP_GMAX_SLOTS), 0, s, a);
  return scade_check_launch("scade_mlp_bwd_lp(gmax)");
}

template <bool BF, bool S8>
static int launch_bwd_lp(const float* packed, const void* packed_t, const unsigned char* acts, const float* g_out,
                         int P, unsigned char* ws, float* grad_flat, hipStream_t s) {
  if (int e = lp_bwd_set_attr<BF, S8>()) return e;
  unsigned char* dz = ws;
  float* partial = reinterpret_cast<float*>(ws + lp_dz_bytes(P));
  float* gmax = partial + (size_t)lp_ws_rows() * N_PARAM_FLOATS;
  if (!BF || S8) {
    const float* gs[2] = {g_out, nullptr};
    const int Pg[2] = {P, 0};
    float* gm[2] = {gmax, nullptr};
    const unsigned char* ac[2] = {acts, nullptr};
    if (int e = lp_launch_gmax(gs, Pg, gm, ac, s)) return e;
  }
  // same point tiling as the forward that wrote the sign words of this workspace
  const int npt = lp_pick_point_tiles(P);
  const int tiles = (P + 32 * npt - 1) / (32 * npt);
  MlpDgradLpArgs2 d{{{packed, packed_t, acts, g_out, dz, gmax, P}, {}}, tiles, 0};
  if (npt == 2)
    hipLaunchKernelGGL((mlp_dgrad_lp_kernel<BF, 2, S8>), dim3(tiles), dim3(256), dgrad_lp_lds_bytes(2), s, d);
  else
    hipLaunchKernelGGL((mlp_dgrad_lp_kernel<BF, 4, S8>), dim3(tiles), dim3(256), dgrad_lp_lds_bytes(4), s, d);
  if (int e = scade_check_launch("scade_mlp_bwd_lp(dgrad)")) return e;
  WgradLpArgs w{};
  build_wgrad_lp_jobs(w);
  w.net[0] = WgradLpNet{acts, dz, g_out, partial, gmax, P};
  ReduceLpArgs r{};
  const int Ps[2] = {P, 0};
  if (int e = lp_build_plan(w, Ps, S8, r.nseg)) return e;
  r.uniform[0] = w.plan.chunk > 0 ? w.plan.gx0 : 0;
  hipLaunchKernelGGL((mlp_wgrad_lp_kernel<BF, S8>), dim3(w.plan.nwg), dim3(512), WGRAD_LP_LDS_BYTES, s, w);
  if (int e = scade_check_launch("scade_mlp_bwd_lp(wgrad)")) return e;
  r.partial[0] = partial; r.grad[0] = grad_flat;
  hipLaunchKernelGGL(wgrad_lp_reduce_kernel, dim3(WGRAD_REDUCE_BLOCKS), dim3(256), 0, s, r);
  return scade_check_launch("scade_mlp_bwd_lp(reduce)");
}

// two networks: ONE dgrad launch, ONE weight-gradient launch, ONE reduce (scade_mlp_bwd_lp2)
// phases: bit 0 = loss-scale maxima + the joint dgrad chain of both networks, bit 1 / bit 2 = weight gradient + reduce
// of network 0 / 1.  Bits 1 and 2 together = ONE balanced weight-gradient launch over both networks; one of them = that
// network's own balanced launch (a sharded step starts the gradient exchange of the first network behind its reduce
// and lets it run under the second network's weight gradient).
// gmax_pre (nullable): per network the LP_GMAX_SLOTS maxima of |g_out| already computed by the caller's loss
// launch (scade_ray_tail_train writes them) - the lp_gmax launch is skipped; defer (nullable): leave the partial
// rows unreduced and describe them (scade_step_finish sums them inside the optimizer's launch).
template <bool BF, bool S8>
static int launch_bwd_lp2(const void* const* packed_t, const void* const* acts, const float* const* g_out,
                          const int* P, void* const* wsv, float* const* grad_flat, int phases, hipStream_t s,
                          const float* const* gmax_pre = nullptr, ReduceDesc* defer = nullptr) {
  if (int e = lp_bwd_set_attr<BF, S8>()) return e;
  const int npt = lp_pick_point_tiles(P[0]);
  SCADE_REQUIRE(lp_pick_point_tiles(P[1]) == npt, -3,
                "scade_mlp_bwd_lp2: the forwards of the two launches tiled their points differently (P = %d, %d); "
                "use scade_mlp_bwd_lp twice", P[0], P[1]);
  MlpDgradLpArgs2 d{};
  WgradLpArgs w{};
  build_wgrad_lp_jobs(w);
  float* partial[2];
  float* gmaxs[2];
  for (int i = 0; i < 2; ++i) {
    unsigned char* ws = reinterpret_cast<unsigned char*>(wsv[i]);
    const unsigned char* ac = reinterpret_cast<const unsigned char*>(acts[i]);
    partial[i] = reinterpret_cast<float*>(ws + lp_dz_bytes(P[i]));
    float* gmax = gmax_pre && gmax_pre[i] ? const_cast<float*>(gmax_pre[i])
                                          : partial[i] + (size_t)lp_ws_rows() * N_PARAM_FLOATS;
    gmaxs[i] = gmax;
    d.n[i] = MlpDgradLpArgs{nullptr, packed_t[i], ac, g_out[i], ws, gmax, P[i]};
    w.net[i] = WgradLpNet{ac, ws, g_out[i], partial[i], gmax, P[i]};
  }
  if (phases & 1) {
    if ((!BF || S8) && !(gmax_pre && gmax_pre[0] && gmax_pre[1])) {      // both networks' maxima in one launch
      const unsigned char* ac[2] = {reinterpret_cast<const unsigned char*>(acts[0]), reinterpret_cast<const unsigned char*>(acts[1])};
      if (int e = lp_launch_gmax(g_out, P, gmaxs, ac, s)) return e;
    }
    d.tiles0 = (P[0] + 32 * npt - 1) / (32 * npt);
    d.tiles1 = (P[1] + 32 * npt - 1) / (32 * npt);
    if (npt == 2)
      hipLaunchKernelGGL((mlp_dgrad_lp_kernel<BF, 2, S8>), dim3(d.tiles0 + d.tiles1), dim3(256), dgrad_lp_lds_bytes(2), s, d);
    else
      hipLaunchKernelGGL((mlp_dgrad_lp_kernel<BF, 4, S8>), dim3(d.tiles0 + d.tiles1), dim3(256), dgrad_lp_lds_bytes(4), s, d);
    if (int e = scade_check_launch("scade_mlp_bwd_lp2(dgrad)")) return e;
  }
  if ((phases & 6) == 6) {
    ReduceLpArgs r{};
    if (int e = lp_build_plan(w, P, S8, r.nseg)) return e;
    r.uniform[0] = w.plan.chunk > 0 ? w.plan.gx0 : 0;
    r.uniform[1] = w.plan.chunk > 0 ? w.plan.gx1 : 0;
    hipLaunchKernelGGL((mlp_wgrad_lp_kernel<BF, S8>), dim3(w.plan.nwg), dim3(512), WGRAD_LP_LDS_BYTES, s, w);
    if (int e = scade_check_launch("scade_mlp_bwd_lp2(wgrad)")) return e;
    if (defer) {
      *defer = ReduceDesc{{partial[0], partial[1]}, {r.uniform[0], r.uniform[1]}, {}};
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < MAX_WGRAD_JOBS; ++j) defer->nseg[i][j] = r.nseg[i][j];
      return 0;
    }
    for (int i = 0; i < 2; ++i) { r.partial[i] = partial[i]; r.grad[i] = grad_flat[i]; }
    hipLaunchKernelGGL(wgrad_lp_reduce_kernel, dim3(2 * WGRAD_REDUCE_BLOCKS), dim3(256), 0, s, r);
    return scade_check_launch("scade_mlp_bwd_lp2(reduce)");
  }
  for (int i = 0; i < 2; ++i) {
    if (!(phases & (2 << i))) continue;
    WgradLpArgs w1{};
    build_wgrad_lp_jobs(w1);
    w1.net[0] = w.net[i];
    ReduceLpArgs r{};
    const int Ps[2] = {P[i], 0};
    if (int e = lp_build_plan(w1, Ps, S8, r.nseg)) return e;
    r.uniform[0] = w1.plan.chunk > 0 ? w1.plan.gx0 : 0;
    hipLaunchKernelGGL((mlp_wgrad_lp_kernel<BF, S8>), dim3(w1.plan.nwg), dim3(512), WGRAD_LP_LDS_BYTES, s, w1);
    if (int e = scade_check_launch("scade_mlp_bwd_lp2(wgrad, one network)")) return e;
    r.partial[0] = partial[i]; r.grad[0] = grad_flat[i];
    hipLaunchKernelGGL(wgrad_lp_reduce_kernel, dim3(WGRAD_REDUCE_BLOCKS), dim3(256), 0, s, r);
    if (int e = scade_check_launch("scade_mlp_bwd_lp2(reduce, one network)")) return e;
  }
  return 0;
}

// The launch plan of the 16-bit weight gradient for networks of P[0] and P[1] (0 = absent) points, as the kernel
// receives it (host code only: no device needed) - for tests and tools.  Outputs: info[0] = workgroups, info[1] =
// jobs per network, info[2] = points per stage, info[3] = partial rows of a workspace, info[4..6] = chunk, gx0,
// gx1 of the small-launch grid (chunk = 0: the balanced plan); bound[257], cum[33], first_wg[32], nseg[32],
// weight[16] (unused entries zero).
extern "C" int scade_mlp_wgrad_lp_plan(const int* P, int s8, int* info, int* bound, int* cum, int* first_wg,
                                       int* nseg, int* weight) {
  SCADE_REQUIRE(P && info && bound && cum && first_wg && nseg && weight, -1, "scade_mlp_wgrad_lp_plan: null pointer");
  SCADE_REQUIRE(P[0] > 0 && P[1] >= 0, -2, "scade_mlp_wgrad_lp_plan: P[0] must be positive, P[1] non-negative");
  WgradLpArgs w{};
  build_wgrad_lp_jobs(w);
  unsigned char ns[2][MAX_WGRAD_JOBS] = {};
  if (int e = lp_build_plan(w, P, s8 != 0, ns)) return e;
  info[0] = w.plan.nwg; info[1] = w.njobs; info[2] = WL_PT; info[3] = lp_ws_rows();
  info[4] = w.plan.chunk; info[5] = w.plan.gx0; info[6] = w.plan.gx1;
  for (int i = 0; i <= LP_MAX_WG; ++i) bound[i] = w.plan.chunk > 0 ? 0 : (i <= w.plan.nwg ? w.plan.bound[i] : 0);
  for (int i = 0; i <= LP_MAX_ENTRIES; ++i) cum[i] = i <= 2 * w.njobs ? w.plan.cum[i] : 0;
  for (int i = 0; i < LP_MAX_ENTRIES; ++i) {
    first_wg[i] = i < 2 * w.njobs ? w.plan.first_wg[i] : 0;
    nseg[i] = i < 2 * w.njobs ? ns[i / w.njobs][i % w.njobs] : 0;
  }
  for (int j = 0; j < MAX_WGRAD_JOBS; ++j) weight[j] = j < w.njobs ? w.plan.weight[j] : 0;
  return 0;
}

// point tiles (of 32) per workgroup the 16-bit forward / dgrad use for a launch over P points
extern "C" int scade_mlp_lp_point_tiles(int P) { return lp_pick_point_tiles(P); }

// workspace of one network of a joint launch over networks of P and P_other points (scade_mlp_bwd_lp2); never
// smaller than scade_mlp_bwd_lp_workspace_bytes(P)
extern "C" long scade_mlp_bwd_lp2_workspace_bytes(int P, int P_other) {
  (void)P_other;       // the partial rows of a network are bounded by the workgroup count, not by its partner
  return scade_mlp_bwd_lp_workspace_bytes(P);
}

extern "C" int scade_mlp_bwd_lp2_phases(const void* const* packed_t_lp, int bf16, const void* const* acts,
                                        const float* const* g_out, const int* P, void* const* workspace,
                                        float* const* grad_flat, int phases, void* stream) {
  SCADE_REQUIRE(packed_t_lp && acts && g_out && P && workspace && grad_flat, -1, "scade_mlp_bwd_lp2: null pointer");
  SCADE_REQUIRE(phases > 0 && phases <= 7, -2, "scade_mlp_bwd_lp2: phases is a mask of bits 0..2");
  for (int i = 0; i < 2; ++i) {
    SCADE_REQUIRE(P[i] > 0, -2, "scade_mlp_bwd_lp2: P[%d] must be positive", i);
    SCADE_REQUIRE(packed_t_lp[i] && acts[i] && g_out[i] && workspace[i] && grad_flat[i], -1,
                  "scade_mlp_bwd_lp2: null pointer in entry %d", i);
  }
  hipStream_t s = (hipStream_t)stream;
  SCADE_REQUIRE(bf16 >= 0 && bf16 <= 2, -2, "scade_mlp_bwd_lp2: format 0 (fp16), 1 (bf16) or 2 (bf16, 8-bit saved rows)");
  if (bf16 == 2) return launch_bwd_lp2<true, true>(packed_t_lp, acts, g_out, P, workspace, grad_flat, phases, s);
  return bf16 ? launch_bwd_lp2<true, false>(packed_t_lp, acts, g_out, P, workspace, grad_flat, phases, s)
              : launch_bwd_lp2<false, false>(packed_t_lp, acts, g_out, P, workspace, grad_flat, phases, s);
}

// scade_mlp_bwd_lp2 for a train step that (a) has the loss-scale maxima of both networks' output gradients already
// (gmax_pre[i]: LP_GMAX_SLOTS floats whose maximum is max|g_out[i]| over the finite entries, written by
// scade_ray_tail_train; NULL array or entry: computed here as usual) and (b) does not exchange its gradient between
// ranks (reduce_desc != NULL: no reduce launch, the partial rows are described for scade_step_finish; the workspaces
// must stay alive until then).  With both NULL this is scade_mlp_bwd_lp2.
extern "C" int scade_mlp_bwd_lp2_deferred(const void* const* packed_t_lp, int bf16, const void* const* acts,
                                          const float* const* g_out, const int* P, void* const* workspace,
                                          float* const* grad_flat, const float* const* gmax_pre, void* reduce_desc,
                                          void* stream) {
  SCADE_REQUIRE(packed_t_lp && acts && g_out && P && workspace && (grad_flat || reduce_desc), -1, "scade_mlp_bwd_lp2_deferred: null pointer");
  for (int i = 0; i < 2; ++i) {
    SCADE_REQUIRE(P[i] > 0, -2, "scade_mlp_bwd_lp2_deferred: P[%d] must be positive", i);
    SCADE_REQUIRE(packed_t_lp[i] && acts[i] && g_out[i] && workspace[i] && (reduce_desc || grad_flat[i]), -1,
                  "scade_mlp_bwd_lp2_deferred: null pointer in entry %d", i);
  }
  hipStream_t s = (hipStream_t)stream;
  ReduceDesc* defer = reinterpret_cast<ReduceDesc*>(reduce_desc);
  SCADE_REQUIRE(bf16 >= 0 && bf16 <= 2, -2, "scade_mlp_bwd_lp2_deferred: format 0 (fp16), 1 (bf16) or 2 (bf16, 8-bit saved rows)");
  if (bf16 == 2) return launch_bwd_lp2<true, true>(packed_t_lp, acts, g_out, P, workspace, grad_flat, 7, s, gmax_pre, defer);
  return bf16 ? launch_bwd_lp2<true, false>(packed_t_lp, acts, g_out, P, workspace, grad_flat, 7, s, gmax_pre, defer)
              : launch_bwd_lp2<false, false>(packed_t_lp, acts, g_out, P, workspace, grad_flat, 7, s, gmax_pre, defer);
}

extern "C" int scade_mlp_bwd_lp2(const void* const* packed_t_lp, int bf16, const void* const* acts,
                                 const float* const* g_out, const int* P, void* const* workspace,
                                 float* const* grad_flat, void* stream) {
  return scade_mlp_bwd_lp2_phases(packed_t_lp, bf16, acts, g_out, P, workspace, grad_flat, 7, stream);
}

extern "C" int scade_mlp_bwd_lp(const float* packed, const void* packed_t_lp, int bf16, const void* acts,
                                const float* g_out, int P, void* workspace, float* grad_flat, void* stream) {
  SCADE_REQUIRE(P > 0, -2, "scade_mlp_bwd_lp: P must be positive");
  (void)packed;   // not read: the head weights are the fp32 tail of packed_t_lp
  SCADE_REQUIRE(packed_t_lp && acts && g_out && workspace && grad_flat, -1,
                "scade_mlp_bwd_lp: null pointer");
  const unsigned char* ac = reinterpret_cast<const unsigned char*>(acts);
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  hipStream_t s = (hipStream_t)stream;
  SCADE_REQUIRE(bf16 >= 0 && bf16 <= 2, -2, "scade_mlp_bwd_lp: format 0 (fp16), 1 (bf16) or 2 (bf16, 8-bit saved rows)");
  if (bf16 == 2) return launch_bwd_lp<true, true>(packed, packed_t_lp, ac, g_out, P, ws, grad_flat, s);
  return bf16 ? launch_bwd_lp<true, false>(packed, packed_t_lp, ac, g_out, P, ws, grad_flat, s)
              : launch_bwd_lp<false, false>(packed, packed_t_lp, ac, g_out, P, ws, grad_flat, s);
}

#ifdef DG_TRACE
extern "C" int scade_debug_dg_heads(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scade::dg_heads), sizeof(unsigned long long) * 16 * 4 * 6);
}
extern "C" int scade_debug_dg_trace(unsigned long long* out, unsigned* hwid) {
  if (hipMemcpyFromSymbol(hwid, HIP_SYMBOL(scade::dg_hwid), sizeof(unsigned) * 32) != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scade::dg_trace), sizeof(unsigned long long) * 16 * 4 * 9 * 6);
}
#endif

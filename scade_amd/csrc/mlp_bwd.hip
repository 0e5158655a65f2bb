// Backward of the fused NeRF MLP for gfx950: what autograd does for
// NeRF.forward / run_network in the reference (model/run_nerf_helpers.py:223-247
// reversed: 24 addmm-backward pairs, relu/softplus/cat backward), as two kernels.
//
//  B1  mlp_dgrad_kernel  -- same 64-point tile / 4-wave structure as the forward: the
//      gradient tile lives in LDS (swizzled [64][256]) and walks the layers backwards,
//      dX^T[k][point] = W^T[k][n] * dZ^T[n][point] on v_mfma_f32_32x32x2_f32 with a
//      TRANSPOSED weight pack as the A operand (same layer_gemm as the forward).  The
//      ReLU masks come from the activations the forward saved; every layer's
//      pre-activation gradient dZ is written to HBM for B2.
//  B2  mlp_wgrad_kernel  -- dW[n][k] = sum_points dZ[point][n] * In[point][k]: the
//      reduction runs over POINTS, so a workgroup (8 waves) owns a full 256x256 weight
//      gradient in registers (128 accumulator VGPRs/lane) for one layer and one chunk of
//      points, streams dZ / In tiles of 32 points through LDS, and writes a per-chunk
//      partial; bias, alpha-head, view-column and rgb-head gradients ride along on the
//      VALU.  wgrad_reduce4_kernel (mlp_wgrad.h) sums the chunk partials (deterministic order)
//      into one flat gradient in PyTorch parameter layout.
#include "mlp_tile.h"
#include "mlp_pack.h"
#include "mlp_wgrad.h"
#include "mlp_reduce.h"

namespace scade {

// ---------------------------------------------------------------------------
// transposed pack
// ---------------------------------------------------------------------------
struct PackTArgs {
  const float* p[N_PARAM_TENSORS];
  float* packedT;
};

__global__ void mlp_pack_t_kernel(PackTArgs a) { pack_t_row(a.p, a.packedT, blockIdx.y, blockIdx.x, gridDim.x); }

// ---------------------------------------------------------------------------
// B1: dgrad chain
// ---------------------------------------------------------------------------
struct MlpDgradArgs {
  const float* packed;    // forward pack (rgb / alpha head weights)
  const float* packedT;   // transposed pack
  const float* acts;      // forward workspace
  const float* g_out;     // [P,4]
  float* dz;              // dz_floats(P)
  int P;
};
// One launch may walk the gradient tiles of TWO networks (the coarse and the fine NeRF of a train step: their
// backward chains are independent, and one launch of tiles0 + tiles1 workgroups fills the rounds of two
// workgroups per CU that two small launches leave half empty): workgroups [0, tiles0) belong to n[0], the
// rest to n[1].  A single-network launch sets tiles0 to its whole grid.
struct MlpDgradArgs2 {
  MlpDgradArgs n[2];
  int tiles0;
};

// write the wave's [2 k-tiles x 64 points] gradient block in place to LDS: optional
// alpha-head term, optional ReLU mask from the lane-private sign bits the forward saved
template <bool MASK, bool ADD_ALPHA, int PT = 2>
__device__ __forceinline__ void dgrad_store(const f32x16 (&acc)[2][PT], int ktile0, float* hbuf,
                                            unsigned long long bits, const float* __restrict__ w_a,
                                            const float* dalpha_lds, int lane) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = (ktile0 + t) * 32 + 8 * q + 4 * hh;
      f32x4 wa = {0.f, 0.f, 0.f, 0.f};
      if (ADD_ALPHA) wa = *reinterpret_cast<const f32x4*>(w_a + f);
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const int row = p * 32 + r;
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[t][p][4 * q + i];
        if (ADD_ALPHA) {
          const float da = dalpha_lds[row];
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = v[i] + wa[i] * da;
        }
        if (MASK) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            v[i] = ((bits >> (p * 32 + (t * 4 + q) * 4 + i)) & 1ull) ? v[i] : 0.f;
        }
        *reinterpret_cast<f32x4*>(hbuf + h_idx(row, f >> 2)) = v;
      }
    }
}

template <int PT>
__global__ __launch_bounds__(256, 2) void mlp_dgrad_kernel(MlpDgradArgs2 aa) {
  constexpr int TM = tile_pts(PT);     // points of this workgroup (shadows the 64-point default)
  const bool second = (int)blockIdx.x >= aa.tiles0;                 // wave-uniform: scalar selects
  const MlpDgradArgs& a = second ? aa.n[1] : aa.n[0];
  const int blk = (int)blockIdx.x - (second ? aa.tiles0 : 0);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* hbuf = lds;
  float* dal = lds + h_floats(PT);   // d alpha_pre of the tile's points

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p0 = blk * TM;
  const int P = a.P;
  const float* __restrict__ pk = a.packed;
  const float* __restrict__ pt_ = a.packedT;
  const float* __restrict__ acts = a.acts;
  float* __restrict__ dz = a.dz;
  auto mask_of = [&](int layer) { return load_relu_words<PT>(acts, P, layer, tid, blk); };

  // ---- heads: d alpha_pre, dZ of the views layer (rgb head + ReLU mask) ----------
  // Round 4: a lane owns ONE 16-byte chunk of the 128 views columns for all its rows (its twelve head weights stay
  // in registers; a wave instruction covers two whole 512-byte rows) and every load of the section is issued before
  // the first use.  The earlier form - one row per four lanes, a guarded load per chunk - made the compiler close
  // each guarded block with a wait for its load: ten memory latencies in a chain per tile.  Same arithmetic, same
  // bits.
  {
    constexpr int HIT = TM / 8;                           // rows per lane: row = 8 it + (tid >> 5)
    const int chunk = tid & 31, r8 = tid >> 5;
    const float* wr = pk + OFF_WR;
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + chunk * 4);
    const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + 128 + chunk * 4);
    const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 256 + chunk * 4);
    const float* hv = acts + acts_slot_off(P, SLOT_VIEWS_H);
    float* dzv = dz + acts_slot_off(P, SLOT_VIEWS_H);
    f32x4 g[HIT], m[HIT];
    float g3r = 0.f, apr = 0.f;                           // d alpha_pre: lane `tid` owns row `tid`
    if (tid < TM) {
      const int pt = min(p0 + tid, P - 1);
      g3r = a.g_out[(size_t)pt * 4 + 3];
      apr = acts[acts_alpha_off(P) + pt];
    }
#pragma unroll
    for (int it = 0; it < HIT; ++it) {                    // (rows past P read row P - 1 and are zeroed by a select)
      const int pt = min(p0 + it * 8 + r8, P - 1);
      g[it] = *reinterpret_cast<const f32x4*>(a.g_out + (size_t)pt * 4);
      m[it] = *reinterpret_cast<const f32x4*>(hv + (size_t)pt * W + chunk * 4);
    }
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
      const int row = it * 8 + r8, pt = p0 + row;
      const bool ok = pt < P;
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float d = g[it][0] * w0[j] + g[it][1] * w1[j] + g[it][2] * w2[j];
        v[j] = (m[it][j] > 0.f && ok) ? d : 0.f;
      }
      *reinterpret_cast<f32x4*>(hbuf + h_idx(row, chunk)) = v;
      if (ok) *reinterpret_cast<f32x4*>(dzv + (size_t)pt * W + chunk * 4) = v;
    }
    if (tid < TM) {
      const int pt = p0 + tid;
      // softplus(x, beta=10)' = sigmoid(10 x)  (1 beyond the linear threshold 10x > 20)
      const float bx = apr * 10.f;
      float da = bx > 20.f ? g3r : g3r / (1.f + expf(-bx));
      if (pt < P) dz[dz_dalpha_off(P) + pt] = da; else da = 0.f;
      dal[tid] = da;
    }
  }
  __syncthreads();

  f32x16 acc[2][PT];
  f32x4 an[2];
  const int kt0 = wave * 2;
  // transposed-pack base of this wave for dgrad index T (NB = reduction blocks per k-tile)
#define WTBASE(T, NB) (reinterpret_cast<const f32x4*>(pt_ + off_wt(T)) + kt0 * (NB) * 64)
  an[0] = WTBASE(8, 16)[lane];
  an[1] = WTBASE(8, 16)[16 * 64 + lane];

  // ---- views layer: d feature = Wv[:, :256]^T dZv  (reduction over 128) --------
  layer_gemm<2, 0, 16, EMB_STRIDE, PT>(acc, an, WTBASE(8, 16), WTBASE(7, 32), 32, hbuf, hbuf, lane);
  __syncthreads();
  dgrad_store<false, false, PT>(acc, kt0, hbuf, 0ull, nullptr, dal, lane);
  save_tile_wave<64, PT>(hbuf, dz + acts_slot_off(P, SLOT_FEAT), p0, P, 64 * wave, lane);
  __syncthreads();

  // ---- feature layer: d h7 = Wf^T d feature + w_alpha * d alpha_pre, mask h7 -----
  unsigned long long mbits = mask_of(7);
  layer_gemm<2, 0, 32, EMB_STRIDE, PT>(acc, an, WTBASE(7, 32), WTBASE(6, 32), 32, hbuf, hbuf, lane);
  __syncthreads();
  dgrad_store<true, true, PT>(acc, kt0, hbuf, mbits, pk + OFF_WA, dal, lane);
  save_tile_wave<64, PT>(hbuf, dz + acts_slot_off(P, 7), p0, P, 64 * wave, lane);
  __syncthreads();

  // ---- pts layers 7..1: dZ_{l-1} = (W_l^T dZ_l) masked by h_{l-1} > 0 -------------
#define DGRAD_LAYER(L)                                                                         \
  mbits = mask_of((L)-1);                                                                      \
  layer_gemm<2, 0, 32, EMB_STRIDE, PT>(acc, an, WTBASE((L)-1, 32), WTBASE((L) > 1 ? (L)-2 : 0, 32), \
                                   32, hbuf, hbuf, lane);                                      \
  __syncthreads();                                                                             \
  dgrad_store<true, false, PT>(acc, kt0, hbuf, mbits, nullptr, dal, lane);                         \
  save_tile_wave<64, PT>(hbuf, dz + acts_slot_off(P, (L)-1), p0, P, 64 * wave, lane);             \
  __syncthreads();

  DGRAD_LAYER(7)
  DGRAD_LAYER(6)
  DGRAD_LAYER(5)
  DGRAD_LAYER(4)
  DGRAD_LAYER(3)
  DGRAD_LAYER(2)
  DGRAD_LAYER(1)
#undef DGRAD_LAYER
#undef WTBASE
}

}  // namespace scade

// ===========================================================================
// C ABI
// ===========================================================================
using namespace scade;

extern "C" long scade_mlp_packed_t_floats(void) { return PACKED_BWD_FLOATS; }

extern "C" int scade_mlp_pack_t(const float* const* params, float* packed_t, void* stream) {
  SCADE_REQUIRE(params && packed_t, -1, "scade_mlp_pack_t: null pointer");
  PackTArgs a;
  for (int i = 0; i < N_PARAM_TENSORS; ++i) {
    SCADE_REQUIRE(params[i], -1, "scade_mlp_pack_t: params[%d] is null", i);
    a.p[i] = params[i];
  }
  a.packedT = packed_t;
  hipLaunchKernelGGL(mlp_pack_t_kernel, dim3(PACK_BLOCKS, PACK_T_ROWS), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_mlp_pack_t");
}

namespace scade {   // mlp_wgrad2.hip
int pick_chunks_v2(int P);
void wgrad2_joint_chunking(const int* P, int& chunk, int& gx0, int& gx1);
}
int scade_launch_wgrad2(const float* const* acts, const float* const* dz, const float* const* g_out, const int* P,
                        float* const* partial, float* const* grad_flat, hipStream_t s, scade::ReduceDesc* defer = nullptr);

// chunk count of the exact weight-gradient kernel (mlp_wgrad2.hip)
extern "C" int scade_mlp_bwd_chunks(int P) { return pick_chunks_v2(P); }

extern "C" long scade_mlp_bwd_workspace_floats(int P) {
  // one workspace serves the exact kernel and the split-precision one (which chunks by pick_chunks)
  const int nc = pick_chunks(P) > pick_chunks_v2(P) ? pick_chunks(P) : pick_chunks_v2(P);
  return dz_floats(P) + (long)nc * N_PARAM_FLOATS + 4;   // + launch-wide max slot (f16x3 mode)
}

// workspace of network 0 of a joint launch over networks of P and P_other points (scade_mlp_bwd2): its dZ
// rows + the per-chunk partials under the JOINT chunking (never smaller than scade_mlp_bwd_workspace_floats(P),
// so the same buffer also serves a separate launch)
extern "C" long scade_mlp_bwd2_workspace_floats(int P, int P_other) {
  int Ps[2] = {P, P_other}, chunk, gx0, gx1;
  wgrad2_joint_chunking(Ps, chunk, gx0, gx1);
  const long joint = dz_floats(P) + (long)gx0 * N_PARAM_FLOATS + 4;
  const long alone = scade_mlp_bwd_workspace_floats(P);
  return joint > alone ? joint : alone;
}

template <int PT>
static int launch_dgrad(const MlpDgradArgs2& d, int tiles, hipStream_t s) {
  static unsigned long long attr_set = 0;   // one bit per device ordinal
  if (scade_attr_needed(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_dgrad_kernel<PT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, mlp_lds_bytes(PT));
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(attr_set);
  }
  hipLaunchKernelGGL(mlp_dgrad_kernel<PT>, dim3(tiles), dim3(256), mlp_lds_bytes(PT), s, d);
  return scade_check_launch("scade_mlp_bwd(dgrad)");
}

extern "C" int scade_mlp_bwd(const float* packed, const float* packed_t, const float* acts,
                             const float* g_out, int P, float* workspace, float* grad_flat,
                             void* stream) {
  SCADE_REQUIRE(P > 0, -2, "scade_mlp_bwd: P must be positive");
  SCADE_REQUIRE(packed && packed_t && acts && g_out && workspace && grad_flat, -1,
                "scade_mlp_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  float* dz = workspace;
  float* partial = workspace + dz_floats(P);

  // same point tiling as the forward that wrote the ReLU words of this workspace
  const int pt = pick_point_tiles(P);
  const int tiles = (P + tile_pts(pt) - 1) / tile_pts(pt);
  MlpDgradArgs2 d{{{packed, packed_t, acts, g_out, dz, P}, {}}, tiles};
  if (int e = pt == 1 ? launch_dgrad<1>(d, tiles, s) : launch_dgrad<2>(d, tiles, s)) return e;

  return scade_launch_wgrad(acts, dz, g_out, P, partial, grad_flat, s);
}

// The backward of TWO network calls (the coarse and the fine NeRF of one train step) as ONE dgrad launch, ONE
// weight-gradient launch and ONE reduce: same arithmetic per network as two scade_mlp_bwd calls (bitwise - a
// workgroup's work does not depend on its neighbours), but the joint grid fills whole rounds of two workgroups
// per CU, which matters for the 128-ray shards of a strongly scaled batch, and three launches go.
static int mlp_bwd2_impl(const float* const* packed, const float* const* packed_t, const float* const* acts,
                         const float* const* g_out, const int* P, float* const* workspace,
                         float* const* grad_flat, int phases, ReduceDesc* defer, void* stream) {
  SCADE_REQUIRE(packed && packed_t && acts && g_out && P && workspace && (grad_flat || defer), -1, "scade_mlp_bwd2: null pointer");
  SCADE_REQUIRE(phases > 0 && phases <= 7, -2, "scade_mlp_bwd2: phases is a mask of bits 0..2");
  for (int i = 0; i < 2; ++i) {
    SCADE_REQUIRE(P[i] > 0, -2, "scade_mlp_bwd2: P[%d] must be positive", i);
    SCADE_REQUIRE(packed[i] && packed_t[i] && acts[i] && g_out[i] && workspace[i] && (defer || grad_flat[i]), -1,
                  "scade_mlp_bwd2: null pointer in entry %d", i);
  }
  hipStream_t s = (hipStream_t)stream;
  if (phases & 1) {
    // the ReLU words are indexed by 32-point tile, not by workgroup, so the point tiling of this launch is free:
    // it is chosen for the JOINT grid (64-point workgroups as soon as both networks together give every CU two)
    const int pt = pick_point_tiles((long)P[0] + P[1]);
    const int t0 = (P[0] + tile_pts(pt) - 1) / tile_pts(pt), t1 = (P[1] + tile_pts(pt) - 1) / tile_pts(pt);
    MlpDgradArgs2 d{{{packed[0], packed_t[0], acts[0], g_out[0], workspace[0], P[0]},
                     {packed[1], packed_t[1], acts[1], g_out[1], workspace[1], P[1]}}, t0};
    if (int e = pt == 1 ? launch_dgrad<1>(d, t0 + t1, s) : launch_dgrad<2>(d, t0 + t1, s)) return e;
  }
  float* partial[2] = {workspace[0] + dz_floats(P[0]), workspace[1] + dz_floats(P[1])};
  const float* dz[2] = {workspace[0], workspace[1]};
  if ((phases & 6) == 6) return scade_launch_wgrad2(acts, dz, g_out, P, partial, grad_flat, s, defer);
  // one network's weight gradient on its own (scade_mlp_bwd2_workspace_floats covers the separate launch too)
  for (int i = 0; i < 2; ++i)
    if (phases & (2 << i))
      if (int e = scade_launch_wgrad(acts[i], dz[i], g_out[i], P[i], partial[i], grad_flat[i], s)) return e;
  return 0;
}

extern "C" int scade_mlp_bwd2_phases(const float* const* packed, const float* const* packed_t, const float* const* acts,
                                     const float* const* g_out, const int* P, float* const* workspace,
                                     float* const* grad_flat, int phases, void* stream) {
  SCADE_REQUIRE(grad_flat, -1, "scade_mlp_bwd2: null pointer");
  return mlp_bwd2_impl(packed, packed_t, acts, g_out, P, workspace, grad_flat, phases, nullptr, stream);
}

// scade_mlp_bwd2 WITHOUT its last launch: the weight gradient's partial rows stay in the workspaces and *reduce_desc
// (64 bytes, host memory) describes them for scade_step_finish, which sums them inside the optimizer's launch - for
// train steps whose gradient is not exchanged between ranks.  The workspaces must stay alive until that launch.
extern "C" int scade_mlp_bwd2_deferred(const float* const* packed, const float* const* packed_t, const float* const* acts,
                                       const float* const* g_out, const int* P, float* const* workspace,
                                       void* reduce_desc, void* stream) {
  SCADE_REQUIRE(reduce_desc, -1, "scade_mlp_bwd2_deferred: null descriptor");
  return mlp_bwd2_impl(packed, packed_t, acts, g_out, P, workspace, nullptr, 7, reinterpret_cast<ReduceDesc*>(reduce_desc), stream);
}

extern "C" int scade_mlp_bwd2(const float* const* packed, const float* const* packed_t, const float* const* acts,
                              const float* const* g_out, const int* P, float* const* workspace,
                              float* const* grad_flat, void* stream) {
  return scade_mlp_bwd2_phases(packed, packed_t, acts, g_out, P, workspace, grad_flat, 7, stream);
}

// Fused positional-encoding + 8x256 NeRF MLP forward for gfx950 (CDNA4).
//
// Replaces, in one launch, the reference's run_network -> Embedder.embed ->
// batchify -> NeRF.forward chain (run_scade_scannet.py:48-63,
// model/run_nerf_helpers.py:142-172, 223-247), which the reference issues as
// ~60 eager ATen ops materialising a [P,60] embedding in HBM.
//
// Design (MI355X-first):
//  * one workgroup = 4 wavefronts (one per SIMD) owns a tile of 64 points and
//    walks all 12 layers with the activations resident in LDS (64 KiB h-tile,
//    XOR-swizzled, + 15 KiB embedding tile = 79 KiB  ->  2 workgroups / CU);
//  * layers are computed TRANSPOSED, out^T[feature][point] = W[feature][k] *
//    act^T[k][point], on v_mfma_f32_32x32x2_f32 (exact fp32): the weight is the
//    A operand, read straight from L2 in a pre-packed fragment order (every
//    wave owns 64 output features, so nothing about W is shared inside the
//    workgroup and it never touches LDS); the activation is the B operand read
//    from LDS with ds_read_b128; both operands are k-contiguous so one 16-byte
//    read feeds 4 MFMAs; the accumulator layout then gives every lane 4
//    consecutive features of ONE point -> ds_write_b128 epilogue, in place;
//  * skip-concat (layer 5) and view-concat are extra k-blocks, never a concat;
//  * the 256->1 and 128->3 heads are VALU dot products.
#include "mlp_tile.h"
#include "mlp_pack.h"

namespace scade {

struct MlpFwdArgs {
  const float* packed;    // PACKED_FWD_FLOATS
  const float* in;        // mode 0: x [P,60];  mode 1: pts [P,3]
  const float* viewdirs;  // mode 1: [P/S,3]
  const float* bb;        // mode 1: {cx,cy,cz,scale}
  float* out;             // [P,4]
  float* acts;            // optional training workspace (mlp_layout.h: acts_floats(P))
  int P;
  int S;                  // samples per ray (mode 1)
  int vd_stride;          // row stride of viewdirs (mode 1)
};

// bias (+ReLU) and in-place write of the wave's [NT*32 features] x [64 points]
// this lane's 16 bias values per n-tile (issued before the k-loop, consumed in layer_store)
__device__ __forceinline__ float relu_keep_nan(float x) {
  const int b = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, b > 0 ? b : 0);
}
// positive-signed quiet NaN for any NaN (the sign of a propagated NaN is otherwise the input's)
// ... and for +-inf: an infinite coordinate makes the reference's whole row NaN (sin(inf) = NaN meets every
// feature of layer 0) while inf - inf inside an MFMA yields a default NaN of either sign, which the integer
// relu would not carry reliably - so it enters the layers as +NaN already
__device__ __forceinline__ float canon_nonfinite(float x) { return fabsf(x) <= 3.4028234664e38f ? x : __builtin_nanf(""); }

template <int NT>
__device__ __forceinline__ void load_bias(f32x4 (&bv)[NT][4], const float* __restrict__ bias,
                                          int ntile0, int lane) {
  const int hh = lane >> 5;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      bv[t][q] = *reinterpret_cast<const f32x4*>(bias + (ntile0 + t) * 32 + 8 * q + 4 * hh);
}

// returns this lane's ReLU sign bits: bit p*32+(t*4+q)*4+i <-> value (t,q,p,i) > 0, the
// same (t,q,p,i) -> (feature, point) map the dgrad kernel uses for its output fragment
template <int NT, bool RELU, int PT = 2>
__device__ __forceinline__ unsigned long long layer_store(const f32x16 (&acc)[NT][PT], int ntile0,
                                                          float* hbuf, int lane) {
  const int r = lane & 31, hh = lane >> 5;
  unsigned long long bits = 0ull;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = (ntile0 + t) * 32 + 8 * q + 4 * hh;
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = acc[t][p][4 * q + i];         // (the accumulator started from the bias: layer_gemm)
          // relu that PROPAGATES NaN like torch.relu (v_max_f32 would return 0 for a NaN input and
          // hide a poisoned point from the reference's isnan/isinf scan, run_scade_scannet.py:747-749):
          // ONE integer max on the float's bits - negative floats (and -0) are negative integers and
          // become 0, a positive-signed NaN is a large positive integer and survives.  NaNs enter the
          // layers positive-signed: the prologue canonicalises the embedding, the hardware's own
          // default NaN (inf - inf in an MFMA) is +qNaN.
          v[i] = RELU ? relu_keep_nan(x) : x;
          if (RELU && x > 0.f) bits |= 1ull << (p * 32 + (t * 4 + q) * 4 + i);
        }
        const int row = p * 32 + r;
        *reinterpret_cast<f32x4*>(hbuf + h_idx(row, f >> 2)) = v;
      }
    }
  return bits;
}

#ifdef FWD_TRACE
__device__ unsigned long long fwd_trace[4 * 4 * 24];   // [inference wg 0/1, training wg 0/1][wave][layer 1..4][6 stamps]
__device__ unsigned long long fwd_sect[4 * 16];        // [kernel x wg]: wave 0's stamps at the section boundaries
#define FS_STAMP(I) if (PT == 2 && (blockIdx.x == 1300 || blockIdx.x == 1301) && threadIdx.x == 0) \
    fwd_sect[((SAVE ? 2 : 0) + (blockIdx.x - 1300)) * 16 + (I)] = clock64();
#else
#define FS_STAMP(I)
#endif
template <int MODE, bool SAVE, int PT>
__global__ __launch_bounds__(256, 2) void mlp_fwd_kernel(MlpFwdArgs a) {
  constexpr int TM = tile_pts(PT);     // points of this workgroup (shadows the 64-point default)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* hbuf = lds;
  float* ebuf = lds + h_floats(PT);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p0 = blockIdx.x * TM;
  const int P = a.P;
  const float* __restrict__ pk = a.packed;

  FS_STAMP(0)
  // ---------------- prologue: embedding tile [64][60] (+ zeroed tail pad) ----
  if (tid < 4) ebuf[TM * EMB_STRIDE + tid] = 0.f;
  if (MODE == 0) {
    // x rows are 60 contiguous floats == the tile layout; columns 57..59 carry
    // the view direction and meet zero weights in layers 0/5.
    const int nvalid = min(TM, P - p0);
    const f32x4* src = reinterpret_cast<const f32x4*>(a.in + (size_t)p0 * 60);
    for (int i = tid; i < TM * 15; i += 256) {
      const int row = i / 15;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < nvalid) v = src[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = canon_nonfinite(v[c]);
      reinterpret_cast<f32x4*>(ebuf)[i] = v;
    }
  } else {
    const float cx = a.bb[0], cy = a.bb[1], cz = a.bb[2], sc = a.bb[3];
    // one thread per (point, coordinate): ONE load, the raw slot and all nine octaves (round 4: the item loop over
    // (point, coordinate, slot) re-read x ten times through a chain of eight dependent memory latencies and called
    // the library's sincosf - ~3x the VALU work of sincos_cw, and VALU instructions are matrix time in this kernel)
    if (tid < TM * 3) {
      const int row = tid / 3, c = tid - row * 3;
      const int pt = min(p0 + row, P - 1);
      const float ctr = c == 0 ? cx : (c == 1 ? cy : cz);
      const float x = (a.in[(size_t)pt * 3 + c] - ctr) * sc;   // run_scade_scannet.py:52
      float* e = ebuf + row * EMB_STRIDE;
      e[c] = canon_nonfinite(x);
      e[57 + c] = 0.f;
      // helpers:165  p_fn(x * np.pi * freq): fl32(x*pi_f32) * 2^k (exact)
      const float t = x * 3.14159274101257324f;
      if (sincos_cw_ok(t * 256.f)) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          float sn, cs;
          sincos_cw(t * (float)(1 << k), sn, cs);
          e[3 + 6 * k + c] = sn;
          e[6 + 6 * k + c] = cs;
        }
      } else {                             // far outside the scene box, or poisoned: the library (one copy of it)
#pragma unroll 1
        for (int k = 0; k < 9; ++k) {
          float sn, cs;
          sincosf(t * (float)(1 << k), &sn, &cs);
          e[3 + 6 * k + c] = canon_nan(sn);
          e[6 + 6 * k + c] = canon_nan(cs);
        }
      }
    }
  }
  __syncthreads();
  if (SAVE) {   // emb rows [64]: gamma(57) | 0 0 0 | view(3) | 0   (coalesced 256-B rows, 16-byte chunks)
    float* eo = a.acts + acts_emb_off(P);
    for (int i = tid; i < TM * 16; i += 256) {
      const int row = i >> 4, c4 = i & 15;
      const int pt = p0 + row;
      if (pt < P) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c4 < 15) {
          v = *reinterpret_cast<const f32x4*>(ebuf + row * EMB_STRIDE + 4 * c4);
          if (c4 == 14) { v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }       // tile columns 57..59: padding / view dir
        } else {
#pragma unroll
          for (int c = 0; c < 3; ++c)
            v[c] = MODE == 0 ? a.in[(size_t)pt * 60 + 57 + c]
                             : a.viewdirs[(size_t)(pt / a.S) * a.vd_stride + c];
        }
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(eo + (size_t)pt * 64 + 4 * c4));
      }
    }
  }

  f32x16 acc[2][PT];
  f32x4 an[2], bias[2][4];
  const int nt0 = wave * 2;
  // weight base of this wave for MFMA layer L (views layer: one n-tile per wave)
#define WBASE(L) (reinterpret_cast<const f32x4*>(pk + off_w(L)) + \
                  ((L) == L_VIEWS ? wave : nt0) * kb_total(L) * 64)

  // -DFWD_TRACE (a variant build: SCADE_AB_FLAGS, scade_amd/build.py; tools/probe_fwd_trace.py): core-clock stamps of
  // two workgroups of the launch's third round at the phase boundaries of layers 1..4 -
  // {k-loop start, k-loop end, barrier passed, epilogue done, tile copy issued, second barrier passed}
#ifdef FWD_TRACE
#define FT_STAMP(L, I)                                                                                      \
  if (PT == 2 && (L) >= 1 && (L) <= 4 && (blockIdx.x == 1300 || blockIdx.x == 1301) && lane == 0)              \
    fwd_trace[(((SAVE ? 2 : 0) + (blockIdx.x - 1300)) * 4 + wave) * 24 + ((L) - 1) * 6 + (I)] = clock64();
#else
#define FT_STAMP(L, I)
#endif
#define PTS_LAYER(L, LNEXT, KBP, PRE)                                                              \
  {                                                                                                \
    FT_STAMP(L, 0)                                                                                 \
    layer_gemm<2, KBP, kb_h(L), EMB_STRIDE, PT>(acc, an, WBASE(L), WBASE(LNEXT), kb_total(LNEXT),  \
                                                PRE, hbuf, lane, bias);                            \
    /* the NEXT layer's bias: its registers are free since the accumulators took this layer's, and an */ \
    /* epilogue, a copy and two barriers hide the fetch */                                         \
    load_bias<2>(bias, pk + off_b(LNEXT), nt0, lane);                                              \
    FT_STAMP(L, 1)                                                                                 \
    __syncthreads();                                                                               \
    FT_STAMP(L, 2)                                                                                 \
    const unsigned long long bits_ = layer_store<2, true, PT>(acc, nt0, hbuf, lane);               \
    FT_STAMP(L, 3)                                                                                 \
    if (SAVE) store_relu_words<PT>(a.acts, P, L, tid, bits_);                                      \
    if (SAVE) save_tile_wave<64, PT>(hbuf, a.acts + acts_slot_off(P, L), p0, P, 64 * wave, lane);    \
    FT_STAMP(L, 4)                                                                                 \
    __syncthreads();                                                                               \
    FT_STAMP(L, 5)                                                                                 \
  }

  FS_STAMP(1)
  an[0] = WBASE(0)[lane];
  an[1] = WBASE(0)[kb_total(0) * 64 + lane];
  load_bias<2>(bias, pk + off_b(0), nt0, lane);
  PTS_LAYER(0, 1, 8, ebuf)
  FS_STAMP(2)
  PTS_LAYER(1, 2, 0, ebuf)
  FS_STAMP(3)
  PTS_LAYER(2, 3, 0, ebuf)
  FS_STAMP(4)
  PTS_LAYER(3, 4, 0, ebuf)
  FS_STAMP(5)
  PTS_LAYER(4, 5, 0, ebuf)
  FS_STAMP(6)
  PTS_LAYER(5, 6, 8, ebuf)
  FS_STAMP(7)

  // embedding tile is dead now: reuse its head as the view pad [64][8]
  {
    const int row = tid >> 2, c = tid & 3;  // 64 rows x 4 (3 used)
    const int pt = min(p0 + row, P - 1);
    float v = 0.f;
    if (c < 3) v = MODE == 0 ? a.in[(size_t)pt * 60 + 57 + c] : a.viewdirs[(size_t)(pt / a.S) * a.vd_stride + c];
    if (PT == 2 || row < TM) {
      ebuf[row * VIEW_PAD + c] = canon_nonfinite(v);
      ebuf[row * VIEW_PAD + 4 + c] = 0.f;
    }
  }
  // (visibility of the view pad is covered by the barriers of layers 6/7)

  PTS_LAYER(6, 7, 0, ebuf)
  FS_STAMP(8)
  PTS_LAYER(7, L_FEAT, 0, ebuf)
  FS_STAMP(9)
#undef PTS_LAYER

  // ---------------- alpha head: 256 -> 1 on the VALU -------------------------
  // (PT == 1: the upper half of the workgroup repeats rows 0..31 and does not store)
  const bool head_store = PT == 2 || tid < 128;
  float alpha;
  {
    const int row = PT == 2 ? tid >> 2 : (tid >> 2) & 31, sub = tid & 3;
    const float* wa = pk + OFF_WA;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int chunk = i * 4 + sub;
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hbuf + h_idx(row, chunk));
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wa + chunk * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s = fmaf(hv[j], wv[j], s);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    alpha = s + pk[OFF_BA];
    if (SAVE && sub == 0 && head_store && p0 + row < P) a.acts[acts_alpha_off(P) + p0 + row] = alpha;
  }

  FS_STAMP(10)
  // ---------------- feature_linear: 256 -> 256, no activation ----------------
  layer_gemm<2, 0, 32, EMB_STRIDE, PT>(acc, an, WBASE(L_FEAT), WBASE(L_VIEWS), 0, ebuf, hbuf, lane, bias);
  f32x4 biasv[1][4];
  load_bias<1>(biasv, pk + off_b(L_VIEWS), wave, lane);
  __syncthreads();
  layer_store<2, false, PT>(acc, nt0, hbuf, lane);
  if (SAVE) save_tile_wave<64, PT>(hbuf, a.acts + acts_slot_off(P, SLOT_FEAT), p0, P, 64 * wave, lane);
  __syncthreads();

  FS_STAMP(11)
  // ---------------- views_linears[0]: [view pad | feature] -> 128, ReLU ------
  {
    f32x16 accv[1][PT];
    // (an[1] is unused by the one-tile views layer; the trailing prefetch re-reads block 0)
    layer_gemm<1, 1, 32, VIEW_PAD, PT>(accv, an, WBASE(L_VIEWS), WBASE(L_VIEWS), 0, ebuf, hbuf, lane, biasv);
    __syncthreads();
    layer_store<1, true, PT>(accv, wave, hbuf, lane);
    if (SAVE) save_tile_wave<32, PT>(hbuf, a.acts + acts_slot_off(P, SLOT_VIEWS_H), p0, P, 32 * wave, lane);
    __syncthreads();
  }

#undef WBASE
  FS_STAMP(12)
  // ---------------- rgb head 128 -> 3, softplus(alpha, beta=10) --------------
  {
    const int row = PT == 2 ? tid >> 2 : (tid >> 2) & 31, sub = tid & 3;
    const float* wr = pk + OFF_WR;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int chunk = i * 4 + sub;
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hbuf + h_idx(row, chunk));
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + chunk * 4);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(wr + 128 + chunk * 4);
      const f32x4 w2 = *reinterpret_cast<const f32x4*>(wr + 256 + chunk * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0 = fmaf(hv[j], w0[j], s0);
        s1 = fmaf(hv[j], w1[j], s1);
        s2 = fmaf(hv[j], w2[j], s2);
      }
    }
    s0 += __shfl_xor(s0, 1, 64); s0 += __shfl_xor(s0, 2, 64);
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
    if (sub == 0 && head_store && p0 + row < P) {
      // F.softplus(alpha, beta=10): x if 10x > 20 else log1p(exp(10x))/10
      const float bx = alpha * 10.f;
      const float sp = bx > 20.f ? alpha : log1pf(expf(bx)) / 10.f;
      f32x4 o = {s0 + pk[OFF_BR + 0], s1 + pk[OFF_BR + 1], s2 + pk[OFF_BR + 2], sp};
      *reinterpret_cast<f32x4*>(a.out + (size_t)(p0 + row) * 4) = o;
    }
  }
  FS_STAMP(13)
}

// ---------------------------------------------------------------------------
// parameter packing
// ---------------------------------------------------------------------------
struct PackArgs {
  const float* p[N_PARAM_TENSORS];
  float* packed;
};

__global__ void mlp_pack_kernel(PackArgs a) { pack_fwd_row(a.p, a.packed, blockIdx.y, blockIdx.x, gridDim.x); }

}  // namespace scade

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using namespace scade;
#ifdef FWD_TRACE
extern "C" int scade_debug_fwd_trace(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out + 4 * 4 * 24, HIP_SYMBOL(scade::fwd_sect), sizeof(unsigned long long) * 4 * 16) != hipSuccess) return -1;
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(scade::fwd_trace), sizeof(unsigned long long) * 4 * 4 * 24);
}
#endif

extern "C" long scade_mlp_packed_floats(void) { return PACKED_FWD_FLOATS; }

extern "C" int scade_mlp_pack(const float* const* params, float* packed, void* stream) {
  SCADE_REQUIRE(params && packed, -1, "scade_mlp_pack: null pointer");
  PackArgs a;
  for (int i = 0; i < N_PARAM_TENSORS; ++i) {
    SCADE_REQUIRE(params[i], -1, "scade_mlp_pack: params[%d] is null", i);
    a.p[i] = params[i];
  }
  a.packed = packed;
  hipLaunchKernelGGL(mlp_pack_kernel, dim3(PACK_BLOCKS, PACK_FWD_ROWS), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_mlp_pack");
}

template <int MODE, bool SAVE, int PT>
static int launch_fwd_pt(const MlpFwdArgs& a, hipStream_t s) {
  static unsigned long long attr_set = 0;   // one bit per device ordinal
  auto kern = mlp_fwd_kernel<MODE, SAVE, PT>;
  if (scade_attr_needed(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, mlp_lds_bytes(PT));
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(attr_set);
  }
  const int grid = (a.P + tile_pts(PT) - 1) / tile_pts(PT);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), mlp_lds_bytes(PT), s, a);
  return scade_check_launch("scade_mlp_fwd");
}
template <int MODE, bool SAVE>
static int launch_fwd(const MlpFwdArgs& a, hipStream_t s) {
  return pick_point_tiles(a.P) == 1 ? launch_fwd_pt<MODE, SAVE, 1>(a, s) : launch_fwd_pt<MODE, SAVE, 2>(a, s);
}

extern "C" int scade_mlp_fwd(const float* packed, int mode, const float* in, const float* viewdirs,
                             int vd_stride, const float* bb, int P, int S, float* out, float* acts,
                             void* stream) {
  if (P == 0) return 0;
  SCADE_REQUIRE(packed && in && out, -1, "scade_mlp_fwd: null pointer");
  SCADE_REQUIRE(mode == 0 || mode == 1, -2, "scade_mlp_fwd: mode must be 0 (x[P,60]) or 1 (pts[P,3])");
  SCADE_REQUIRE(P > 0, -2, "scade_mlp_fwd: P < 0");
  if (mode == 1) {
    SCADE_REQUIRE(viewdirs && bb && vd_stride >= 3, -1, "scade_mlp_fwd: mode 1 needs viewdirs (stride >= 3) and bb");
    SCADE_REQUIRE(S > 0 && P % S == 0, -2, "scade_mlp_fwd: P (%d) must be a multiple of S (%d)", P, S);
  }
  MlpFwdArgs a{packed, in, viewdirs, bb, out, acts, P, S, vd_stride};
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) return acts ? launch_fwd<0, true>(a, s) : launch_fwd<0, false>(a, s);
  return acts ? launch_fwd<1, true>(a, s) : launch_fwd<1, false>(a, s);
}

extern "C" int scade_mlp_lds_bytes(void) { return MLP_LDS_BYTES; }
extern "C" long scade_mlp_acts_floats(long P) { return acts_floats(P); }

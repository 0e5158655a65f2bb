// Split-precision inference variant of the fused NeRF MLP forward (opt-in).
//
// The exact kernel (mlp_fwd.hip) is bound by the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32,
// 157 TFLOP/s).  Here every fp32 value x is carried as TWO fp16 numbers
//        x  ~=  h + l * 2^-11,    h = fp16(x),   l = fp16((x - h) * 2^11)
// (22 significant bits; the 2^11 pre-scale keeps l out of the fp16 subnormal range), and
//        x*w ~=  h_x h_w  +  (h_x l_w + l_x h_w) * 2^-11
// is evaluated with THREE v_mfma_f32_32x32x16_f16 into two fp32 accumulators (the dropped
// l*l term and the split residuals are ~2^-21 relative per product): fp32-class results at
// 3/16 of the fp32-MFMA cost.  Same tile structure as mlp_fwd.hip: 64 points per workgroup,
// activations resident in LDS as two fp16 planes (64 KiB, same bytes as the fp32 tile),
// weights streamed from L2 in a packed two-plane fragment order, transposed product so the
// epilogue writes 4 consecutive features of one point.
//
// Measured (round 1): 0.62 ms / 196,608 points = 375 algorithmic TFLOP/s (2.7x the exact kernel,
// 2.4x the fp32-MFMA peak).  PMC: matrix pipe 54 % busy; the limiter is the vector-memory path
// streaming 2.4 MB of weights per 64-point tile (two workgroups per CU, ~40 B/clk/CU at full
// MFMA rate).  A variant with 8 waves per workgroup (one n-tile per wave, 4 waves per SIMD,
// weights two k-blocks ahead) measured 10 % SLOWER (LDS fragment traffic doubles) - kept out.
//
// Validity: |activation| and |weight| < 65504 (fp16 range).  Inference only (no saved
// activations); training uses the exact fp32 kernels.
#include "mlp_pack.h"       // (mlp_tile_f16.h + the pack rows shared with scade_mlp_pack_step)

namespace scade {

struct MlpF16Args {
  const void* packed;     // PACKED_F16_BYTES
  const float* in;        // mode 0: x [P,60];  mode 1: pts [P,3]
  const float* viewdirs;
  const float* bb;
  float* out;             // [P,4]
  float* acts;            // optional training workspace (mlp_layout.h), same contents as the exact kernel's
  int P, S, vd_stride;
};

// this lane's bias values in accumulator order: cb[t][4q+i] = bias[(ntile0+t)*32 + 8q + 4*(lane>>5) + i]
template <int NT>
__device__ __forceinline__ void load_bias16_h(f32x16 (&cb)[2], const float* __restrict__ bias, int ntile0,
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

// acc0 + acc1/2048 already holds W x + bias (the bias rode in as the first MFMA's C operand).
// Returns the lane's ReLU sign bits in the exact kernel's format (mlp_fwd.hip layer_store).  After
// point tile 0 the NEXT layer's bias values are fetched (NTN n-tiles) into the vacated registers.
template <int NT, bool RELU, int NTN>
__device__ __forceinline__ unsigned long long layer_store_h(const f32x16 (&acc0)[NT][2],
                                                            const f32x16 (&acc1)[NT][2], int ntile0,
                                                            _Float16* xh, _Float16* xl, int lane,
                                                            f32x16 (&cb)[2],
                                                            const float* __restrict__ bias_next,
                                                            int ntile0_next) {
  const int r = lane & 31, hh = lane >> 5;
  // ReLU + sign bits in THREE operations per value (round 5; two compares, two selects, half an OR and the wait states
  // between compare and select before: the epilogue's arithmetic is 29 % of this kernel and the other workgroup's
  // MFMAs do not run under it - knock-out): v_maximum3_f32(x, 0, 0) is gfx950's NaN-PROPAGATING
  // maximum (torch.relu semantics), "positive" is then "the result's bits are not zero" (v_min_u32 against 1) and the
  // bit is shifted in; value n = (t*4 + q)*4 + i of point tile p ends at bit 31 - n (15 - n with one n-tile): one
  // v_bfrev_b32 per point tile restores the exact kernel's format (bit n).  (A NaN now sets its bit: it poisons the row
  // either way.)
  unsigned w[2] = {0u, 0u};
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f = (ntile0 + t) * 32 + 8 * q + 4 * hh;
        half4 vh, vl;
        float xs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float x = fmaf(acc1[t][p][4 * q + i], LINV, acc0[t][p][4 * q + i]);
          if (RELU) {
            x = __builtin_elementwise_maximum(x, 0.0f);
            unsigned b;
            asm("v_min_u32 %0, 1, %1" : "=v"(b) : "v"(__float_as_uint(x)));
            w[p] = (w[p] << 1) | b;
          }
          xs[i] = x;
        }
        split4(xs, vh, vl);
        const int row = p * 32 + r;
        const int o = x_idx(row, f >> 3) + (f & 7);
        *reinterpret_cast<half4*>(xh + o) = vh;
        *reinterpret_cast<half4*>(xl + o) = vl;
      }
    if (p == 0 && NTN > 0) {
      if (NTN == 2) load_bias16_h<2>(cb, bias_next, ntile0_next, lane);
      else load_bias16_h<1>(cb, bias_next, ntile0_next, lane);
    }
  }
  constexpr int SH = NT == 1 ? 16 : 0;
  return (unsigned long long)(__builtin_bitreverse32(w[0]) >> SH) |
         ((unsigned long long)(__builtin_bitreverse32(w[1]) >> SH) << 32);
}

template <int MODE, int SAVE>
__global__ __launch_bounds__(256, 2) void mlp_fwd_f16_kernel(MlpF16Args a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
  _Float16* xh = ldsh;
  _Float16* xl = ldsh + XPLANE;
  _Float16* eh = ldsh + 2 * XPLANE;
  _Float16* el = eh + EPLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p0 = blockIdx.x * HM;
  const int P = a.P;
  const _Float16* __restrict__ wpk = reinterpret_cast<const _Float16*>(a.packed);
  const float* __restrict__ tail = reinterpret_cast<const float*>(wpk + PACKED_F16_HALVES);
  // fp32 tail uses the fp32 blob's offsets relative to OFF_BIAS
#define TAIL(off) (tail + ((off) - OFF_BIAS))

  // ---- prologue: embedding planes [64][64] (57 real channels, zero padded) ----------
  {
    float cx = 0.f, cy = 0.f, cz = 0.f, sc = 1.f;
    if (MODE == 1) { cx = a.bb[0]; cy = a.bb[1]; cz = a.bb[2]; sc = a.bb[3]; }
    // zero the pad columns 57..63
    for (int i = tid; i < HM * 7; i += 256) {
      const int row = i / 7, c = 57 + (i - row * 7);
      const int o = e_idx(row, c >> 3) + (c & 7);
      eh[o] = (_Float16)0.f; el[o] = (_Float16)0.f;
    }
    if (MODE == 1) {
      // one thread per (point, coordinate): one load, the raw slot and all nine octaves (as mlp_fwd.hip, round 4)
      if (tid < HM * 3) {
        const int row = tid / 3, c = tid - row * 3;
        const int pt = min(p0 + row, P - 1);
        const float ctr = c == 0 ? cx : (c == 1 ? cy : cz);
        const float x = (a.in[(size_t)pt * 3 + c] - ctr) * sc;
        auto put = [&](int col, float v) {
          _Float16 h, l;
          split2(v, h, l);
          const int o = e_idx(row, col >> 3) + (col & 7);
          eh[o] = h; el[o] = l;
        };
        put(c, x);
        const float t = x * 3.14159274101257324f;
        if (sincos_cw_ok(t * 256.f)) {
#pragma unroll
          for (int k = 0; k < 9; ++k) {
            float sn, cs;
            sincos_cw(t * (float)(1 << k), sn, cs);
            put(3 + 6 * k + c, sn);
            put(6 + 6 * k + c, cs);
          }
        } else {
#pragma unroll 1
          for (int k = 0; k < 9; ++k) {
            float sn, cs;
            sincosf(t * (float)(1 << k), &sn, &cs);
            put(3 + 6 * k + c, sn);
            put(6 + 6 * k + c, cs);
          }
        }
      }
    } else
    for (int i = tid; i < HM * 30; i += 256) {
      const int row = i / 30, rem = i - row * 30;
      const int c = rem / 10, s = rem - c * 10;
      const int pt = min(p0 + row, P - 1);
      float v0, v1 = 0.f;
      int col0, col1 = -1;
      if (MODE == 0) {
        // x rows already hold gamma(x): copy channel pairs (s==0: raw, s>=1: sin/cos of freq s-1)
        const float* xr = a.in + (size_t)pt * 60;
        if (s == 0) { col0 = c; v0 = xr[c]; }
        else { col0 = 3 + 6 * (s - 1) + c; col1 = col0 + 3; v0 = xr[col0]; v1 = xr[col1]; }
      } else {
        const float ctr = c == 0 ? cx : (c == 1 ? cy : cz);
        const float x = (a.in[(size_t)pt * 3 + c] - ctr) * sc;
        if (s == 0) { col0 = c; v0 = x; }
        else {
          const float arg = (x * 3.14159274101257324f) * (float)(1 << (s - 1));
          sincosf(arg, &v0, &v1);
          col0 = 3 + 6 * (s - 1) + c; col1 = col0 + 3;
        }
      }
      _Float16 h, l;
      split2(v0, h, l);
      int o = e_idx(row, col0 >> 3) + (col0 & 7);
      eh[o] = h; el[o] = l;
      if (col1 >= 0) {
        split2(v1, h, l);
        o = e_idx(row, col1 >> 3) + (col1 & 7);
        eh[o] = h; el[o] = l;
      }
    }
  }
  __syncthreads();
  if (SAVE) {   // emb rows [64]: gamma(57) | 0 0 0 | view(3) | 0  (same layout as the exact kernel)
    float* eo = a.acts + acts_emb_off(P);
    for (int i = tid; i < HM * 64; i += 256) {
      const int row = i >> 6, c = i & 63;
      const int pt = p0 + row;
      if (pt < P) {
        float v = 0.f;
        if (c < EMB) {
          const int o = e_idx(row, c >> 3) + (c & 7);
          v = (float)eh[o] + (float)el[o] * LINV;
        } else if (c >= 60 && c < 63) {
          v = MODE == 0 ? a.in[(size_t)pt * 60 + 57 + (c - 60)]
                        : a.viewdirs[(size_t)(pt / a.S) * a.vd_stride + (c - 60)];
        }
        eo[(size_t)pt * 64 + c] = v;
      }
    }
  }

  f32x16 acc0[2][2], acc1[2][2];
  f32x16 cb[2];            // this lane's bias values of the NEXT layer, loaded one epilogue ahead
  AFrag an;
  const int nt0 = wave * 2;
#define WHBASE(L) (reinterpret_cast<const half8*>(wpk + off_wh(L)) + \
                   ((L) == L_VIEWS ? wave : nt0) * kb16(L) * 2 * 64)

#define PTS_LAYER_H(L, LNEXT, KBP)                                                               \
  {                                                                                              \
    layer_gemm_h<2, KBP, kbh16(L), false>(acc0, acc1, an, WHBASE(L), WHBASE(LNEXT), kb16(LNEXT), \
                                          eh, el, xh, xl, lane, cb);                             \
    __syncthreads();                                                                             \
    const unsigned long long bits_ =                                                             \
        layer_store_h<2, true, 2>(acc0, acc1, nt0, xh, xl, lane, cb, TAIL(off_b(LNEXT)), nt0);   \
    if (SAVE) store_relu_words<2>(a.acts, P, L, tid, bits_);                                     \
    if (SAVE) save_tile_h_wave<64, SAVE == 2>(xh, xl, a.acts + acts_slot_off(P, L), p0, P, 64 * wave, nullptr, lane); \
    __syncthreads();                                                                             \
  }

  an.t0h = WHBASE(0)[lane];
  an.t0l = WHBASE(0)[64 + lane];
  an.t1h = WHBASE(0)[(kb16(0) * 2 + 0) * 64 + lane];
  an.t1l = WHBASE(0)[(kb16(0) * 2 + 1) * 64 + lane];
  load_bias16_h<2>(cb, TAIL(off_b(0)), nt0, lane);
  PTS_LAYER_H(0, 1, 4)
  PTS_LAYER_H(1, 2, 0)
  PTS_LAYER_H(2, 3, 0)
  PTS_LAYER_H(3, 4, 0)
  PTS_LAYER_H(4, 5, 0)
  PTS_LAYER_H(5, 6, 4)

  // embedding planes are dead: view pad [64][16] halves per plane (3 real channels)
  {
    const int row = tid >> 2, c = tid & 3;
    const int pt = min(p0 + row, P - 1);
    float v = 0.f;
    if (c < 3) v = MODE == 0 ? a.in[(size_t)pt * 60 + 57 + c] : a.viewdirs[(size_t)(pt / a.S) * a.vd_stride + c];
    _Float16 h, l;
    split2(v, h, l);
    eh[row * 16 + c] = h; el[row * 16 + c] = l;
#pragma unroll
    for (int j = 1; j < 4; ++j) { eh[row * 16 + 4 * j + c] = (_Float16)0.f; el[row * 16 + 4 * j + c] = (_Float16)0.f; }
  }

  PTS_LAYER_H(6, 7, 0)
  PTS_LAYER_H(7, L_FEAT, 0)
#undef PTS_LAYER_H

  // ---- alpha head on the VALU (fp32 weights, x = h + l/2048) ------------------------
  float alpha;
  {
    const int row = tid >> 2, sub = tid & 3;
    const float* wa = TAIL(OFF_WA);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = i * 4 + sub;                       // 8-half chunk
      const half8 vh = *reinterpret_cast<const half8*>(xh + x_idx(row, c));
      const half8 vl = *reinterpret_cast<const half8*>(xl + x_idx(row, c));
#pragma unroll
      for (int j = 0; j < 8; ++j) s = fmaf((float)vh[j] + (float)vl[j] * LINV, wa[c * 8 + j], s);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    alpha = s + TAIL(OFF_BA)[0];
    if (SAVE && sub == 0 && p0 + row < P) a.acts[acts_alpha_off(P) + p0 + row] = alpha;
  }

  // ---- feature_linear ------------------------------------------------------------------
  layer_gemm_h<2, 0, 16, false>(acc0, acc1, an, WHBASE(L_FEAT), WHBASE(L_VIEWS), 0, eh, el, xh, xl, lane, cb);
  __syncthreads();
  layer_store_h<2, false, 1>(acc0, acc1, nt0, xh, xl, lane, cb, TAIL(off_b(L_VIEWS)), wave);
  if (SAVE) save_tile_h_wave<64, SAVE == 2>(xh, xl, a.acts + acts_slot_off(P, SLOT_FEAT), p0, P, 64 * wave, nullptr, lane);
  __syncthreads();

  // ---- views layer: [view pad | feature] -> 128, ReLU --------------------------------
  {
    f32x16 av0[1][2], av1[1][2];
    layer_gemm_h<1, 1, 16, true>(av0, av1, an, WHBASE(L_VIEWS), WHBASE(L_VIEWS), 0, eh, el, xh, xl, lane, cb);
    __syncthreads();
    layer_store_h<1, true, 0>(av0, av1, wave, xh, xl, lane, cb, nullptr, 0);
    if (SAVE) save_tile_h_wave<32, SAVE == 2>(xh, xl, a.acts + acts_slot_off(P, SLOT_VIEWS_H), p0, P, 32 * wave, nullptr, lane);
    __syncthreads();
  }
#undef WHBASE

  // ---- rgb head + softplus ---------------------------------------------------------------
  {
    const int row = tid >> 2, sub = tid & 3;
    const float* wr = TAIL(OFF_WR);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = i * 4 + sub;
      const half8 vh = *reinterpret_cast<const half8*>(xh + x_idx(row, c));
      const half8 vl = *reinterpret_cast<const half8*>(xl + x_idx(row, c));
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = (float)vh[j] + (float)vl[j] * LINV;
        s0 = fmaf(x, wr[c * 8 + j], s0);
        s1 = fmaf(x, wr[128 + c * 8 + j], s1);
        s2 = fmaf(x, wr[256 + c * 8 + j], s2);
      }
    }
    s0 += __shfl_xor(s0, 1, 64); s0 += __shfl_xor(s0, 2, 64);
    s1 += __shfl_xor(s1, 1, 64); s1 += __shfl_xor(s1, 2, 64);
    s2 += __shfl_xor(s2, 1, 64); s2 += __shfl_xor(s2, 2, 64);
    if (sub == 0 && p0 + row < P) {
      const float bx = alpha * 10.f;
      const float sp = bx > 20.f ? alpha : log1pf(expf(bx)) / 10.f;
      const float* br = TAIL(OFF_BR);
      f32x4 o = {s0 + br[0], s1 + br[1], s2 + br[2], sp};
      *reinterpret_cast<f32x4*>(a.out + (size_t)(p0 + row) * 4) = o;
    }
  }
#undef TAIL
}

// ---------------------------------------------------------------------------
// pack: fp32 parameters -> two fp16 planes in fragment order + fp32 tail
// ---------------------------------------------------------------------------
struct PackF16Args {
  const float* p[N_PARAM_TENSORS];
  void* packed;
};

__global__ void mlp_pack_f16_kernel(PackF16Args a) { pack_f16_row(a.p, a.packed, blockIdx.y, blockIdx.x, gridDim.x); }

}  // namespace scade

using namespace scade;

extern "C" long scade_mlp_packed_f16_bytes(void) { return PACKED_F16_BYTES; }

extern "C" int scade_mlp_pack_f16(const float* const* params, void* packed, void* stream) {
  SCADE_REQUIRE(params && packed, -1, "scade_mlp_pack_f16: null pointer");
  PackF16Args a;
  for (int i = 0; i < N_PARAM_TENSORS; ++i) {
    SCADE_REQUIRE(params[i], -1, "scade_mlp_pack_f16: params[%d] is null", i);
    a.p[i] = params[i];
  }
  a.packed = packed;
  hipLaunchKernelGGL(mlp_pack_f16_kernel, dim3(64, NLAYER_MFMA + 1), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_mlp_pack_f16");
}

template <int MODE, int SAVE>
static int launch_f16(const MlpF16Args& a, hipStream_t s) {
  static unsigned long long attr_set = 0;   // one bit per device ordinal
  auto kern = mlp_fwd_f16_kernel<MODE, SAVE>;
  if (scade_attr_needed(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, F16_LDS_BYTES);
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_fwd_f16: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(attr_set);
  }
  hipLaunchKernelGGL(kern, dim3((a.P + HM - 1) / HM), dim3(256), F16_LDS_BYTES, s, a);
  return scade_check_launch("scade_mlp_fwd_f16");
}

extern "C" int scade_mlp_fwd_f16(const void* packed_f16, int mode, const float* in,
                                 const float* viewdirs, int vd_stride, const float* bb, int P, int S,
                                 float* out, float* acts, void* stream) {
  if (P == 0) return 0;
  SCADE_REQUIRE(packed_f16 && in && out, -1, "scade_mlp_fwd_f16: null pointer");
  SCADE_REQUIRE(mode >= 0 && mode <= 3, -2, "scade_mlp_fwd_f16: mode must be 0 or 1 (+ 2: 24-bit saved rows)");
  const bool rows24 = (mode & 2) != 0;
  mode &= 1;
  SCADE_REQUIRE(!rows24 || acts, -2, "scade_mlp_fwd_f16: mode + 2 (24-bit saved rows) needs the training workspace");
  if (mode == 1) {
    SCADE_REQUIRE(viewdirs && bb && vd_stride >= 3, -1, "scade_mlp_fwd_f16: mode 1 needs viewdirs and bb");
    SCADE_REQUIRE(S > 0 && P % S == 0, -2, "scade_mlp_fwd_f16: P must be a multiple of S");
  }
  MlpF16Args a{packed_f16, in, viewdirs, bb, out, acts, P, S, vd_stride};
  hipStream_t s = (hipStream_t)stream;
  if (mode == 0) return !acts ? launch_f16<0, 0>(a, s) : rows24 ? launch_f16<0, 2>(a, s) : launch_f16<0, 1>(a, s);
  return !acts ? launch_f16<1, 0>(a, s) : rows24 ? launch_f16<1, 2>(a, s) : launch_f16<1, 1>(a, s);
}

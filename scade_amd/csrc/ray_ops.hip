// Per-ray operators of the SCADE render path for gfx950: stratified z sampling,
// alpha compositing, inverse-CDF resampling, sorted merge, space-carving loss.
// One wavefront (64 lanes) owns one ray; 4 rays per 256-thread workgroup.
//
// Reference sites (relative to the reference checkout):
//   ray_points        run_scade_scannet.py:638-657, perturb_z_vals :564-579
//   composite         compute_weights :511-522, raw2outputs :530-562
//   sample_pdf        model/run_nerf_helpers.py:337-436
//   merge_sorted      run_scade_scannet.py:713-714
//   carve             model/run_nerf_helpers.py:93-128
//   mse               model/run_nerf_helpers.py:11
//
// Numerics follow the CPU PyTorch path of the reference: fp32 element math in
// the reference's op order (this file is built with -ffp-contract=off), scans
// (cumprod / cumsum) accumulated in fp64 and rounded per prefix, as ATen's CPU
// cumsum/cumprod kernels do.
#include "common.h"
#include "ray_points_dev.h"
#include "train_loss_dev.h"
#include "mlp_layout.h"
#include "mlp_pack.h"

namespace scade {

// fp64 scans / reductions of a ray's samples: the DPP primitives of common.h (round 3; until then
// __shfl_up(double) = two ds_bpermute per step)
__device__ __forceinline__ double wave_incl_prod(double v) { return wave_incl_scan_d<DppProd>(v); }
__device__ __forceinline__ double wave_incl_sum(double v) { return wave_incl_scan_d<DppSum>(v); }
// suffix (inclusive, from the top lane down) sum
__device__ __forceinline__ double wave_incl_sum_rev(double v) { return wave_incl_sum_rev_d(v); }
__device__ __forceinline__ double wave_sum_d(double v) { return wave_sum_dpp_d(v); }

// ---------------------------------------------------------------------------
// ray_points: z_vals (+ stratified jitter) and sample positions (ray_points_dev.h)
// ---------------------------------------------------------------------------
__global__ void ray_points_kernel(RayPointsArgs a) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const float* r = a.rays + (size_t)ray * a.ray_stride;
  ray_points_ray(a, ray, lane_id(), r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
}

// ---------------------------------------------------------------------------
// composite (raw2outputs / compute_weights)
// ---------------------------------------------------------------------------
struct CompositeArgs {
  const float* raw;      // [N,S,4]
  const float* z;        // [N,S]
  const float* rays_d;   // [N, d_stride]
  const float* noise;    // [N,S] or null
  float* rgb_map;        // [N,3]
  float* disp_map;       // [N]
  float* acc_map;        // [N]
  float* weights;        // [N,S]
  float* depth_map;      // [N]
  // backward only
  const float* g_rgb; const float* g_disp; const float* g_acc; const float* g_w; const float* g_depth;
  float* g_raw;          // [N,S,4]
  int N, S, d_stride;
};

struct SampleState {
  float alpha, T, w, dist, sigpos;  // sigpos = relu(sigma+noise)
  float sr, sg, sb;                 // sigmoid(rgb)
  float z;
};

// forward pass of one ray, NC chunks of 64 samples; fills st[] and the sums
template <int NC>
__device__ __forceinline__ void composite_ray(const CompositeArgs& a, int ray, int lane,
                                              SampleState (&st)[NC], double& s_r, double& s_g,
                                              double& s_b, double& s_depth, double& s_acc) {
  const int S = a.S;
  const float* d = a.rays_d + (size_t)ray * a.d_stride;
  const float dnorm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float* zr = a.z + (size_t)ray * S;
  const f32x4* rawr = reinterpret_cast<const f32x4*>(a.raw) + (size_t)ray * S;
  double carry = 1.0;
  s_r = s_g = s_b = s_depth = s_acc = 0.0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = c * 64 + lane;
    const bool valid = i < S;
    SampleState s{};
    double xd = 1.0;
    if (valid) {
      const float zi = zr[i];
      float dist = (i + 1 < S) ? zr[i + 1] - zi : 1e10f;            // :514-515
      dist = dist * dnorm;                                           // :516
      const f32x4 rv = rawr[i];
      float sig = rv[3];
      if (a.noise) sig = sig + a.noise[(size_t)ray * S + i];
      const float sp = sig < 0.f ? 0.f : sig;   // relu that propagates NaN like F.relu (:512)
      const float alpha = 1.0f - expf(-sp * dist);                   // :512
      const float x = (1.0f - alpha) + 1e-10f;                       // :520
      xd = (double)x;
      s.alpha = alpha; s.dist = dist; s.sigpos = sig <= 0.f ? -1.f : sp; s.z = zi;      // (NaN stays NaN: threshold_backward)
      s.sr = 1.0f / (1.0f + expf(-rv[0]));                           // :543 sigmoid
      s.sg = 1.0f / (1.0f + expf(-rv[1]));
      s.sb = 1.0f / (1.0f + expf(-rv[2]));
    }
    const double incl = wave_incl_prod(xd);
    const double excl = wave_shr1_d(incl, 1.0);
    s.T = (float)(carry * excl);
    carry = carry * wave_lane_d<63>(incl);
    s.w = valid ? s.alpha * s.T : 0.f;
    if (valid) {
      s_r += (double)(s.w * s.sr);
      s_g += (double)(s.w * s.sg);
      s_b += (double)(s.w * s.sb);
      s_depth += (double)(s.w * s.z);
      s_acc += (double)s.w;
    }
    st[c] = s;
  }
  s_r = wave_sum_d(s_r); s_g = wave_sum_d(s_g); s_b = wave_sum_d(s_b);
  s_depth = wave_sum_d(s_depth); s_acc = wave_sum_d(s_acc);
}

template <int NC>
__global__ void composite_fwd_kernel(CompositeArgs a) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  SampleState st[NC];
  double sr, sg, sb, sd, sa;
  composite_ray<NC>(a, ray, lane, st, sr, sg, sb, sd, sa);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = c * 64 + lane;
    if (i < a.S) a.weights[(size_t)ray * a.S + i] = st[c].w;
  }
  if (lane == 0) {
    const float depth = (float)sd, acc = (float)sa;
    a.rgb_map[ray * 3 + 0] = (float)sr;
    a.rgb_map[ray * 3 + 1] = (float)sg;
    a.rgb_map[ray * 3 + 2] = (float)sb;
    a.depth_map[ray] = depth;
    a.acc_map[ray] = acc;
    const float q = depth / acc;                                      // :559 (NaN propagates like torch.max)
    a.disp_map[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
  }
}

// backward of one ray given its recomputed forward state; ``gw_inner`` (LDS, or null) holds an extra
// gradient w.r.t. weights[1 .. S-2] (the fine sampler's, scade_ray_tail_bwd)
template <int NC>
__device__ __forceinline__ void composite_bwd_ray(const CompositeArgs& a, int ray, int lane, const SampleState (&st)[NC],
                                                  double sd, double sa, const float* gw_inner,
                                                  bool reg_g = false, float rg_r = 0.f, float rg_g = 0.f, float rg_b = 0.f,
                                                  float* gmax_lane = nullptr) {
  const int S = a.S;
  const float depth = (float)sd, acc = (float)sa;
  // (reg_g: the colour gradient is handed over in registers - the fused tail + loss kernel below)
  const float gr = reg_g ? rg_r : a.g_rgb ? a.g_rgb[ray * 3 + 0] : 0.f;
  const float gg = reg_g ? rg_g : a.g_rgb ? a.g_rgb[ray * 3 + 1] : 0.f;
  const float gb = reg_g ? rg_b : a.g_rgb ? a.g_rgb[ray * 3 + 2] : 0.f;
  float gdepth = a.g_depth ? a.g_depth[ray] : 0.f;
  float gacc = a.g_acc ? a.g_acc[ray] : 0.f;
  if (a.g_disp) {
    // disp = 1/max(1e-10, depth/acc)
    const float q = depth / acc;
    if (q > 1e-10f) {
      const float gq = -a.g_disp[ray] / (q * q);
      gdepth += gq / acc;
      gacc += -gq * depth / (acc * acc);
    }
  }
  // G_i = dL/dw_i ; suffix sums of G_k * w_k processed from the last chunk down
  double carry = 0.0;
  f32x4* gout = reinterpret_cast<f32x4*>(a.g_raw) + (size_t)ray * S;
#pragma unroll
  for (int c = NC - 1; c >= 0; --c) {
    const int i = c * 64 + lane;
    const bool valid = i < S;
    const SampleState& s = st[c];
    float G = 0.f;
    if (valid) {
      G = gr * s.sr + gg * s.sg + gb * s.sb + gdepth * s.z + gacc;
      // the gradient of weights is summed first (autograd adds the sampler's slice gradient to the
      // caller's before compositing sees it), then joins G: same roundings as the separate operators
      float gws = a.g_w ? a.g_w[(size_t)ray * S + i] : 0.f;
      if (gw_inner && i >= 1 && i <= S - 2) gws = gws + gw_inner[i - 1];
      if (a.g_w || gw_inner) G += gws;
    }
    const double gw = valid ? (double)G * (double)s.w : 0.0;
    const double incl = wave_incl_sum_rev(gw);
    const double after = carry + (incl - gw);                          // sum over k > i
    carry += wave_lane_d<0>(incl);
    if (valid) {
      const float x = (1.0f - s.alpha) + 1e-10f;
      const float galpha = G * s.T - (float)(after / (double)x);
      // alpha = 1 - exp(-relu(sig)*dist)
      const float gsig = s.sigpos < 0.f ? 0.f : galpha * s.dist * expf(-s.sigpos * s.dist);
      f32x4 g;
      g[0] = gr * s.w * s.sr * (1.0f - s.sr);
      g[1] = gg * s.w * s.sg * (1.0f - s.sg);
      g[2] = gb * s.w * s.sb * (1.0f - s.sb);
      g[3] = gsig;
      gout[i] = g;
      if (gmax_lane) {
        // max over the finite entries of what ENTERS the MLP's backward: the colour channels as they are, the
        // density channel behind the softplus' derivative sigmoid(10 alpha_pre) = 1 - exp(-10 sigma) (the 16-bit
        // backward's loss scale is taken from it: mlp_bwd_lp.hip lp_effective_g3 - a raw d / d sigma of 1e3 beside
        // 1e-5 everywhere else is what the last sample's delta = 1e10 produces for a near-empty sample)
        const float e3 = gsig * -expm1f(-10.0f * s.sigpos);
        float m = *gmax_lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x = fabsf(q < 3 ? g[q] : e3);
          if (x < 3.0e38f) m = fmaxf(m, x);
        }
        *gmax_lane = m;
      }
    }
  }
}

template <int NC>
__global__ void composite_bwd_kernel(CompositeArgs a) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  SampleState st[NC];
  double sr, sg, sb, sd, sa;
  composite_ray<NC>(a, ray, lane, st, sr, sg, sb, sd, sa);
  composite_bwd_ray<NC>(a, ray, lane, st, sd, sa, nullptr);
}

// ---------------------------------------------------------------------------
// inverse-CDF sampler
// ---------------------------------------------------------------------------
struct SamplePdfArgs {
  const float* bins;     // mode 0: [N, bins_stride] row = M bins ; mode 1: z_vals [N, bins_stride], bins = mids
  const float* w;        // row pointer base, element j of ray n at w[n*w_stride + j], j in [0, M-1)
  const float* u;        // element s of ray n at u[n*u_stride + s] (u_stride 0 == shared row)
  const float* cdf_in;   // optional [N,M]: use this cdf instead of building it from w
  float* samples;        // [N,S]
  long long* inds;       // optional [N,S]
  float* cdf_out;        // optional [N,M]
  float* z_std;          // optional [N]
  // backward
  const float* g_samples;  // [N,S]
  float* g_w;              // [N, M-1] dense
  int N, M, S, bins_stride, w_stride, u_stride, bins_are_mids;
};

// builds cdf[M] and bins[M] in LDS for one ray; returns sum(w+1e-5) as float
// br: M bins (or M+1 z values when mids), wr: M-1 weights (or cdf_row: M cdf values); the rows may
// live in global memory or in LDS
__device__ __forceinline__ float build_cdf_rows(const float* br, int mids, const float* wr,
                                                const float* cdf_row, int M, int lane, float* cdf,
                                                float* bins, float* pdf /*optional LDS [M-1]*/) {
  for (int j = lane; j < M; j += 64)
    bins[j] = mids ? 0.5f * (br[j + 1] + br[j]) : br[j];
  float total = 0.f;
  if (cdf_row) {
    for (int j = lane; j < M; j += 64) cdf[j] = cdf_row[j];
  } else {
    double part = 0.0;
    for (int j = lane; j < M - 1; j += 64) part += (double)(wr[j] + 1e-5f);   // helpers:339
    total = (float)wave_sum_d(part);                                          // helpers:340 (sum)
    double carry = 0.0;
    if (lane == 0) cdf[0] = 0.f;                                              // helpers:343
    for (int j0 = 0; j0 < M - 1; j0 += 64) {
      const int j = j0 + lane;
      float p = 0.f;
      if (j < M - 1) p = (wr[j] + 1e-5f) / total;
      if (pdf && j < M - 1) pdf[j] = p;
      const double incl = wave_incl_sum((double)p);
      if (j < M - 1) cdf[j + 1] = (float)(carry + incl);                      // helpers:342 cumsum
      carry += wave_lane_d<63>(incl);
    }
  }
  __builtin_amdgcn_wave_barrier();
  return total;
}

__device__ __forceinline__ float build_cdf(const SamplePdfArgs& a, int ray, int lane, float* cdf,
                                           float* bins, float* pdf /*optional LDS [M-1]*/) {
  return build_cdf_rows(a.bins + (size_t)ray * a.bins_stride, a.bins_are_mids,
                        a.w ? a.w + (size_t)ray * a.w_stride : nullptr,
                        a.cdf_in ? a.cdf_in + (size_t)ray * a.M : nullptr, a.M, lane, cdf, bins, pdf);
}

// searchsorted(cdf, u, right=True): number of entries <= u
__device__ __forceinline__ int upper_bound(const float* cdf, int M, float u) {
  int lo = 0, hi = M;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ float inverse_cdf(const float* cdf, const float* bins, int M, float u, int& ind) {
  ind = upper_bound(cdf, M, u);                                               // helpers:366
  const int below = max(0, ind - 1), above = min(M - 1, ind);                 // :368-369
  const float c0 = cdf[below], c1 = cdf[above];
  float den = c1 - c0;                                                        // :378
  den = den < 1e-5f ? 1.0f : den;                                             // :379
  const float t = (u - c0) / den;                                             // :380
  const float b0 = bins[below], b1 = bins[above];
  return b0 + t * (b1 - b0);                                                  // :381
}

__global__ void sample_pdf_fwd_kernel(SamplePdfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int ray = blockIdx.x * RAYS_PER_WG + wv;
  if (ray >= a.N) return;
  const int M = a.M, S = a.S;
  float* cdf = smem + wv * 2 * M;
  float* bins = cdf + M;
  build_cdf(a, ray, lane, cdf, bins, nullptr);
  if (a.cdf_out)
    for (int j = lane; j < M; j += 64) a.cdf_out[(size_t)ray * M + j] = cdf[j];
  const float* ur = a.u + (size_t)ray * a.u_stride;
  double s1 = 0.0;
  for (int s = lane; s < S; s += 64) {
    int ind;
    const float smp = inverse_cdf(cdf, bins, M, ur[s], ind);
    a.samples[(size_t)ray * S + s] = smp;
    if (a.inds) a.inds[(size_t)ray * S + s] = ind;
    s1 += (double)smp;
  }
  if (a.z_std) {                                   // torch.std(unbiased=False), run_scade_scannet.py:744
    const double mean = wave_sum_d(s1) / (double)S;
    double s2 = 0.0;
    for (int s = lane; s < S; s += 64) {
      const double dlt = (double)a.samples[(size_t)ray * S + s] - mean;       // same lane wrote it
      s2 += dlt * dlt;
    }
    s2 = wave_sum_d(s2);
    if (lane == 0) a.z_std[ray] = (float)sqrt(s2 / (double)S);
  }
}

// d samples / d weights  (closed form; matches
// autograd through the reference op sequence)
// cdf / bins / pdf of the ray are in LDS (build_cdf_rows); dcdf [M] is LDS scratch; writes d samples / d w
// for the M-1 weights into ``gw_out`` (global row or LDS).  The cdf buffer is reused as scratch.
__device__ __forceinline__ void sample_pdf_bwd_rows(float* cdf, const float* bins, const float* pdf, float* dcdf,
                                                    float total, const float* ur, const float* gs_row, int M, int S,
                                                    int lane, float* gw_out) {
  for (int j = lane; j < M; j += 64) dcdf[j] = 0.f;
  __builtin_amdgcn_wave_barrier();
  for (int s = lane; s < S; s += 64) {
    const float u = ur[s];
    const float g = gs_row[s];
    const int ind = upper_bound(cdf, M, u);
    const int below = max(0, ind - 1), above = min(M - 1, ind);
    const float c0 = cdf[below], c1 = cdf[above];
    const float den = c1 - c0;
    const float db = bins[above] - bins[below];
    float g0, g1;
    if (den < 1e-5f) {          // t = u - c0
      g0 = -g * db;
      g1 = 0.f;
    } else {                    // t = (u - c0)/(c1 - c0)
      const float inv = 1.0f / den;
      const float t = (u - c0) * inv;
      g1 = -g * db * t * inv;                 // d/dc1 = -g*db*(u-c0)/den^2
      g0 = g * db * (t - 1.0f) * inv;         // d/dc0 =  g*db*(u-c1)/den^2
    }
    atomicAdd(&dcdf[below], g0);
    atomicAdd(&dcdf[above], g1);
  }
  __builtin_amdgcn_wave_barrier();
  // dpdf_i = sum_{j>i} dcdf_j  (cdf_j = sum_{i<j} pdf_i), i in [0, M-1)
  // dw_i = (dpdf_i - sum_k dpdf_k pdf_k) / total
  const int nchunk = (M - 1 + 63) / 64;
  double carry = 0.0, dot = 0.0;
  // pass 1 (reverse): dpdf into dcdf' (reuse cdf buffer as scratch), accumulate dot
  for (int c = nchunk - 1; c >= 0; --c) {
    const int i = c * 64 + lane;
    const double v = (i < M - 1) ? (double)dcdf[i + 1] : 0.0;
    const double incl = wave_incl_sum_rev(v);
    const double dp = carry + incl;            // sum_{j >= i+1} dcdf_j
    carry += wave_lane_d<0>(incl);
    if (i < M - 1) {
      cdf[i] = (float)dp;
      dot += dp * (double)pdf[i];
    }
  }
  dot = wave_sum_d(dot);
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < M - 1; i += 64) gw_out[i] = (float)(((double)cdf[i] - dot) / (double)total);
}

__global__ void sample_pdf_bwd_kernel(SamplePdfArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int ray = blockIdx.x * RAYS_PER_WG + wv;
  if (ray >= a.N) return;
  const int M = a.M, S = a.S;
  float* cdf = smem + wv * 4 * M;
  float* bins = cdf + M;
  float* pdf = bins + M;
  float* dcdf = pdf + M;          // [M] float accumulators (LDS atomics)
  const float total = build_cdf(a, ray, lane, cdf, bins, pdf);
  sample_pdf_bwd_rows(cdf, bins, pdf, dcdf, total, a.u + (size_t)ray * a.u_stride, a.g_samples + (size_t)ray * S, M, S,
                      lane, a.g_w + (size_t)ray * (M - 1));
}

// ---------------------------------------------------------------------------
// merge_sorted: z = sort(cat(z_a[Sa], z_b[Sb])) + points
// ---------------------------------------------------------------------------
struct MergeArgs {
  const float* za; const float* zb;
  const float* rays;     // [N, ray_stride] (o, d) or null
  float* z_out;          // [N, Sa+Sb]
  float* pts;            // [N, Sa+Sb, 3] or null
  int N, Sa, Sb, ray_stride;
};

// Bitonic sorting network over the concatenation padded to a power of two PB with the largest key.  The floats
// are sorted as order-preserving UNSIGNED KEYS (sign bit set: ~bits, else bits | 0x80000000; NaN
// canonicalised to the largest key, so it sorts last like torch.sort), which makes a compare-exchange
// two integer ops (v_min_u32 / v_max_u32) with no branches.  Equal values are interchangeable, so the
// output equals torch.sort(cat(z_a, z_b)).values bit for bit (NaN payloads and the order of -0 / +0
// aside).
__device__ __forceinline__ unsigned z_key(float x) {
  unsigned u = __float_as_uint(x);
  u = (x != x) ? 0x7fc00000u : u;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float z_unkey(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
constexpr unsigned Z_KEY_INF = 0xffffffffu;   // padding: after every data key, NaN included

// PB = 64*R <= 512 (SCADE: 192 -> 256, R = 4): the whole ray lives in REGISTERS, element e = lane +
// 64 r.  A compare-exchange at distance j < 64 is one __shfl_xor per register, at distance >= 64 it is
// lane-local; 36 stages of four independent shuffles instead of 36 dependent LDS round trips.
template <int R>
__device__ __forceinline__ void bitonic_sort_keys(unsigned (&v)[R], int lane) {
#pragma unroll
  for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {                                   // partner register, same lane
        const int dr = j >> 6;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & dr) == 0) {
            const bool asc = ((64 * r) & k) == 0;      // lane bits are below 64 <= j < k
            const unsigned lo = min(v[r], v[r | dr]), hi = max(v[r], v[r | dr]);
            v[r] = asc ? lo : hi;
            v[r | dr] = asc ? hi : lo;
          }
        }
      } else {                                         // partner lane, same register
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const unsigned x = v[r];
          const unsigned y = (unsigned)__shfl_xor((int)x, j, 64);
          const bool lower = (lane & j) == 0;
          const bool asc = ((lane + 64 * r) & k) == 0;
          v[r] = (lower == asc) ? min(x, y) : max(x, y);   // the lower element of an ascending pair keeps the min
        }
      }
    }
  }
}

// the first St sorted keys -> z_out row (+ o + d*z)
template <int R>
__device__ __forceinline__ void store_sorted(const unsigned (&v)[R], int lane, int ray, int St,
                                             const float* rays, int ray_stride, float* z_out, float* pts) {
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
  if (pts) {
    const float* rr = rays + (size_t)ray * ray_stride;
    ox = rr[0]; oy = rr[1]; oz = rr[2]; dx = rr[3]; dy = rr[4]; dz = rr[5];
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = lane + 64 * r;
    if (e < St) {
      const float x = z_unkey(v[r]);
      z_out[(size_t)ray * St + e] = x;
      if (pts) {
        float* p = pts + ((size_t)ray * St + e) * 3;
        p[0] = ox + dx * x;
        p[1] = oy + dy * x;
        p[2] = oz + dz * x;
      }
    }
  }
}

template <int R>
__global__ void merge_sorted_reg_kernel(MergeArgs a) {
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int ray = blockIdx.x * RAYS_PER_WG + wv;
  if (ray >= a.N) return;
  const int St = a.Sa + a.Sb;
  unsigned v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int e = lane + 64 * r;
    v[r] = e < a.Sa ? z_key(a.za[(size_t)ray * a.Sa + e])
                    : (e < St ? z_key(a.zb[(size_t)ray * a.Sb + (e - a.Sa)]) : Z_KEY_INF);
  }
  bitonic_sort_keys<R>(v, lane);
  store_sorted<R>(v, lane, ray, St, a.rays, a.ray_stride, a.z_out, a.pts);
}

// larger rows: the same network in LDS
__global__ void merge_sorted_kernel(MergeArgs a, int PB) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem_u[];
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int ray = blockIdx.x * RAYS_PER_WG + wv;
  if (ray >= a.N) return;
  const int St = a.Sa + a.Sb;
  unsigned* v = smem_u + wv * PB;
  for (int i = lane; i < a.Sa; i += 64) v[i] = z_key(a.za[(size_t)ray * a.Sa + i]);
  for (int i = lane; i < a.Sb; i += 64) v[a.Sa + i] = z_key(a.zb[(size_t)ray * a.Sb + i]);
  for (int i = St + lane; i < PB; i += 64) v[i] = Z_KEY_INF;
  __builtin_amdgcn_wave_barrier();
  for (int k = 2; k <= PB; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      // pair index t in [0, PB/2): i = element with bit j clear
      for (int t = lane; t < (PB >> 1); t += 64) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int ixj = i | j;
        const unsigned x = v[i], y = v[ixj];
        const bool asc = (i & k) == 0;
        const unsigned lo = min(x, y), hi = max(x, y);
        v[i] = asc ? lo : hi;
        v[ixj] = asc ? hi : lo;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  float ox = 0, oy = 0, oz = 0, dx = 0, dy = 0, dz = 0;
  if (a.pts) {
    const float* r = a.rays + (size_t)ray * a.ray_stride;
    ox = r[0]; oy = r[1]; oz = r[2]; dx = r[3]; dy = r[4]; dz = r[5];
  }
  for (int i = lane; i < St; i += 64) {
    const float x = z_unkey(v[i]);
    a.z_out[(size_t)ray * St + i] = x;
    if (a.pts) {
      float* p = a.pts + ((size_t)ray * St + i) * 3;
      p[0] = ox + dx * x;
      p[1] = oy + dy * x;
      p[2] = oz + dz * x;
    }
  }
}

// ---------------------------------------------------------------------------
// ray_tail: what follows an MLP launch on a ray, in ONE launch (one wave per ray as everywhere):
//   R > 0 (coarse stage, run_scade_scannet.py:660-714): raw2outputs -> sample_pdf(z_mid, w[1:-1], u)
//          -> sort(cat(z, samples)) -> o + d*z : the weights and samples never leave the CU
//   R = 0 (fine stage, :720-744): raw2outputs -> sample_pdf_return_u (+ std of the samples)
// Same device functions, same order of operations as the three separate kernels: same bits.
// ---------------------------------------------------------------------------
struct TailArgs {
  CompositeArgs c;       // forward fields; rays_d = rays + 3, d_stride = ray_stride
  const float* rays;     // [N, ray_stride] o(0..2) d(3..5)
  const float* u;        // element s of ray n at u[n*u_stride + s]
  float* samples;        // [N,Si] or null
  float* z_std;          // [N] or null
  float* z_out;          // [N,S+Si]  (R > 0)
  float* pts;            // [N,S+Si,3] or null
  int ray_stride, u_stride, Si;
};

template <int NC, int R>
__device__ __forceinline__ void ray_tail_body(const TailArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int ray = blockIdx.x * RAYS_PER_WG + wv;
  if (ray >= a.c.N) return;
  const int S = a.c.S, M = S - 1, Si = a.Si;
  float* w = smem + (size_t)wv * (4 * S + Si);   // w[S] | cat[S + Si] = z, samples | cdf[M] | bins[M]
  float* cat = w + S;
  float* cdf = cat + S + Si;
  float* bins = cdf + M;

  SampleState st[NC];
  double sr, sg, sb, sd, sa;
  composite_ray<NC>(a.c, ray, lane, st, sr, sg, sb, sd, sa);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = c * 64 + lane;
    if (i < S) {
      a.c.weights[(size_t)ray * S + i] = st[c].w;
      w[i] = st[c].w;
      cat[i] = st[c].z;
    }
  }
  if (lane == 0) {
    const float depth = (float)sd, acc = (float)sa;
    a.c.rgb_map[ray * 3 + 0] = (float)sr;
    a.c.rgb_map[ray * 3 + 1] = (float)sg;
    a.c.rgb_map[ray * 3 + 2] = (float)sb;
    a.c.depth_map[ray] = depth;
    a.c.acc_map[ray] = acc;
    const float q = depth / acc;                                      // :559
    a.c.disp_map[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
  }
  __builtin_amdgcn_wave_barrier();

  build_cdf_rows(cat, 1, w + 1, nullptr, M, lane, cdf, bins, nullptr);   // z_mid, weights[1:-1]
  const float* ur = a.u + (size_t)ray * a.u_stride;
  double s1 = 0.0;
  for (int s = lane; s < Si; s += 64) {
    int ind;
    const float smp = inverse_cdf(cdf, bins, M, ur[s], ind);
    if (a.samples) a.samples[(size_t)ray * Si + s] = smp;
    cat[S + s] = smp;
    s1 += (double)smp;
  }
  if (a.z_std) {                                   // torch.std(unbiased=False), :744
    const double mean = wave_sum_d(s1) / (double)Si;
    double s2 = 0.0;
    for (int s = lane; s < Si; s += 64) {
      const double dlt = (double)cat[S + s] - mean;                   // same lane wrote it
      s2 += dlt * dlt;
    }
    s2 = wave_sum_d(s2);
    if (lane == 0) a.z_std[ray] = (float)sqrt(s2 / (double)Si);
  }
  if constexpr (R > 0) {
    __builtin_amdgcn_wave_barrier();
    const int St = S + Si;
    unsigned v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int e = lane + 64 * r;
      v[r] = e < St ? z_key(cat[e]) : Z_KEY_INF;
    }
    bitonic_sort_keys<R>(v, lane);
    store_sorted<R>(v, lane, ray, St, a.rays, a.ray_stride, a.z_out, a.pts);
  }
}

template <int NC, int R>
__global__ void ray_tail_kernel(TailArgs a) { ray_tail_body<NC, R>(a); }
template <int NC>
__global__ void ray_tail_kernel0(TailArgs a) { ray_tail_body<NC, 0>(a); }   // one template argument for DISPATCH_NC

// Backward of the fine tail (raw2outputs -> sample_pdf(z_mid, weights[1:-1], u)) in ONE launch: the forward
// of the ray is recomputed (weights, cdf), the sampler's closed-form d samples / d weights goes to LDS and
// joins the compositing backward as an extra gradient of weights[1 .. S-2].  Same device functions, same
// operation order as scade_sample_pdf_bwd followed by scade_composite_bwd => same bits.
struct TailBwdArgs {
  CompositeArgs c;          // raw, z, rays_d (+stride), noise, g_rgb .. g_depth, g_raw, N, S
  const float* u;           // [N,Si] (u_stride 0: one shared row)
  const float* g_samples;   // [N,Si]
  int u_stride, Si;
};
template <int NC>
__global__ void ray_tail_bwd_kernel(TailBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wv = threadIdx.x >> 6, lane = lane_id();
  const int ray = blockIdx.x * RAYS_PER_WG + wv;
  if (ray >= a.c.N) return;
  const int S = a.c.S, M = S - 1;
  float* w = smem + (size_t)wv * (6 * S);      // w[S] | cdf[M] | bins[M] | pdf[M] | dcdf[M] | gw_inner[M]
  float* cdf = w + S;
  float* bins = cdf + S;
  float* pdf = bins + S;
  float* dcdf = pdf + S;
  float* gwi = dcdf + S;
  SampleState st[NC];
  double sr, sg, sb, sd, sa;
  composite_ray<NC>(a.c, ray, lane, st, sr, sg, sb, sd, sa);
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int i = c * 64 + lane;
    if (i < S) w[i] = st[c].w;
  }
  __builtin_amdgcn_wave_barrier();
  const float total = build_cdf_rows(a.c.z + (size_t)ray * S, 1, w + 1, nullptr, M, lane, cdf, bins, pdf);
  sample_pdf_bwd_rows(cdf, bins, pdf, dcdf, total, a.u + (size_t)ray * a.u_stride, a.g_samples + (size_t)ray * a.Si,
                      M, a.Si, lane, gwi);
  __builtin_amdgcn_wave_barrier();
  composite_bwd_ray<NC>(a.c, ray, lane, st, sd, sa, gwi);
}

// The fine tail, the three-term train loss (forward AND backward, unit gradient) and the backward of both tails
// of a TRAIN step in ONE launch: ray_tail_body's forward, the per-ray loss functions of train_loss_dev.h on the
// colour still in registers and the depth hypotheses just written to LDS, the sampler's backward, the fine
// compositing backward on the forward state still in registers (nothing is recomputed) and - c0.raw given - the
// coarse ray's compositing backward.  Same device functions and operation order as scade_ray_tail ->
// scade_train_loss_fb -> scade_ray_tail_bwd (+ scade_composite_bwd for the coarse ray) => the same bits; four
// launches of ~10 us became one in round 3.
// Round 5: ONE RAY PER WORKGROUP OF FOUR WAVES (WAVES = 4; launches that fit the chip's wave slots).  As one wave
// per ray the kernel was a single dependent chain of ~4,500 VALU instructions at 13 clocks each
// (profiles/r04_per_ray_pmc.txt: a lone wave per SIMD, 3 % of the time in s_waitcnt) - 24-30 us whatever the batch
// size.  A ray's work is not one chain: the coarse ray's compositing (forward recomputed + backward) depends on
// nothing of the fine ray, the loss VALUE (min over K per sample, summed) and the scale / shift gradient (the
// scatter into the winning hypotheses) are read by nobody in this kernel, the spread of the samples (z_std) only
// leaves, and d loss / d hypothesis is per sample.  So:
//   wave 0 (the critical chain): fine compositing -> cdf -> inverse cdf | barrier | d loss / d hypotheses ->
//           sampler backward -> fine compositing backward
//   wave 1: coarse compositing forward + backward (its colour gradient needs rgb0 and the target only)
//           | barrier | the loss value of the ray (tl_fwd_ray)
//   wave 2: | barrier | scale / shift gradient of the ray (tl_bwd_ray<TL_SS>: the argmin again, the scatter)
//   wave 3: | barrier | z_std
// Every output is produced by the same device function with the same operands in the same order as before (the
// parts of tl_bwd_ray are template switches around unchanged code), so the bits do not move; four waves per SIMD
// (1024 rays on 256 CUs) interleave where one left the issue slots empty: 24.4 -> 15.3 us at 128 rays, 25.6 -> 18.6
// at 1024, 29.6 -> 19.7 at 512 rays / K = 40 (rocprofv3, same box, tools/probe_tail_train.py).  (Halving wave 0's
// d loss / d hypotheses with wave 1 behind a second barrier measured SLOWER, 16.3 / 19.1 / 21.7 us: the wait costs more
// than 64 argmins.)  The form costs one more argmin pass per ray and parks three waves at a barrier, so launches
// beyond four waves per SIMD (2048 rays: 32.6 us, no gain; 4096 rays: 62 us against 50) keep WAVES = 1: four rays
// per workgroup, one wave each, the sequential form.
struct TailTrainArgs {
  TailArgs t;              // fine tail, merge-less form (z_out = pts = null)
  TrainLossArgs l;         // rgb / pred / g_rgb / g_pred unused (registers, LDS); g_rgb0 optional
  float* g_raw;            // [N,S,4] fine
  CompositeArgs c0;        // coarse ray: raw, z, rays_d, noise, g_raw (S0 <= 64), or raw = null
  float* gmax_ray;         // [2][N] per-ray maxima of the effective d loss / d raw (fine | coarse), or null
};
template <int NC, int WAVES>
__device__ __forceinline__ void ray_tail_train_body(TailTrainArgs a) {
  static_assert(WAVES == 1 || WAVES == 4, "one wave per ray (4 rays per workgroup) or four");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
  const int ray = WAVES == 4 ? (int)blockIdx.x : (int)blockIdx.x * RAYS_PER_WG + wv;
  if (WAVES == 1 && ray >= a.t.c.N) return;
  const int role = WAVES == 4 ? wv : 0;
  const int S = a.t.c.S, M = S - 1, Si = a.t.Si;
  // w[S] | cat[S+Si] | cdf[S] | bins[S] | pdf[S] | dcdf[S] | gwi[S] | gs[Si] | colour[4]
  float* w = smem + (WAVES == 4 ? 0 : (size_t)wv * (7 * S + 2 * Si + 4));
  float* cat = w + S;
  float* cdf = cat + S + Si;
  float* bins = cdf + S;
  float* pdf = bins + S;
  float* dcdf = pdf + S;
  float* gwi = dcdf + S;
  float* gs = gwi + S;
  float* colour = gs + Si;
  float* part = a.l.partial + 4 * (size_t)a.l.N;

  if (role == 0) {
    // ---- forward (ray_tail_body<NC, 0>) ----
    SampleState st[NC];
    double sr, sg, sb, sd, sa;
    composite_ray<NC>(a.t.c, ray, lane, st, sr, sg, sb, sd, sa);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int i = c * 64 + lane;
      if (i < S) {
        a.t.c.weights[(size_t)ray * S + i] = st[c].w;
        w[i] = st[c].w;
        cat[i] = st[c].z;
      }
    }
    if (lane == 0) {
      const float depth = (float)sd, acc = (float)sa;
      a.t.c.rgb_map[ray * 3 + 0] = colour[0] = (float)sr;
      a.t.c.rgb_map[ray * 3 + 1] = colour[1] = (float)sg;
      a.t.c.rgb_map[ray * 3 + 2] = colour[2] = (float)sb;
      a.t.c.depth_map[ray] = depth;
      a.t.c.acc_map[ray] = acc;
      const float q = depth / acc;                                      // :559
      a.t.c.disp_map[ray] = 1.0f / ((q != q) ? q : fmaxf(1e-10f, q));
    }
    __builtin_amdgcn_wave_barrier();
    const float total = build_cdf_rows(cat, 1, w + 1, nullptr, M, lane, cdf, bins, pdf);   // z_mid, weights[1:-1]
    const float* ur = a.t.u + (size_t)ray * a.t.u_stride;
    for (int s = lane; s < Si; s += 64) {
      int ind;
      const float smp = inverse_cdf(cdf, bins, M, ur[s], ind);
      if (a.t.samples) a.t.samples[(size_t)ray * Si + s] = smp;
      cat[S + s] = smp;
    }
    if (WAVES == 4) __syncthreads();               // the hypotheses and the colour are published
    else __builtin_amdgcn_wave_barrier();

    TlRayIo io;
    io.pred_row = cat + S; io.g_pred_row = gs; io.have_rgb = true;
    io.r = (float)sr; io.g = (float)sg; io.b = (float)sb;
    float gx;
    if (WAVES == 4) {
      // ---- d loss / d hypotheses, d loss / d colour (train_loss_fb_kernel's backward, unit gradient) ----
      io.store_gx = false;
      gx = tl_bwd_ray<TL_GX | TL_GP>(a.l, ray, lane, 1.0f, part, io);
      __builtin_amdgcn_wave_barrier();
    } else {
      // ---- z_std, the loss forward and backward, whole (the sequential form) ----
      if (a.t.z_std) {                             // torch.std(unbiased=False), :744
        double s1 = 0.0;
        for (int s = lane; s < Si; s += 64) s1 += (double)cat[S + s];
        const double mean = wave_sum_d(s1) / (double)Si;
        double s2 = 0.0;
        for (int s = lane; s < Si; s += 64) {
          const double dlt = (double)cat[S + s] - mean;
          s2 += dlt * dlt;
        }
        s2 = wave_sum_d(s2);
        if (lane == 0) a.t.z_std[ray] = (float)sqrt(s2 / (double)Si);
      }
      tl_fwd_ray(a.l, ray, lane, io);
      gx = tl_bwd_ray(a.l, ray, lane, 1.0f, part, io);
      __builtin_amdgcn_wave_barrier();
    }
    const float g_r = tl_bcast(gx, 0), g_g = tl_bcast(gx, 1), g_b = tl_bcast(gx, 2);

    // ---- backward of the fine tail (ray_tail_bwd_kernel; the forward state is still here) ----
    sample_pdf_bwd_rows(cdf, bins, pdf, dcdf, total, ur, gs, M, Si, lane, gwi);
    __builtin_amdgcn_wave_barrier();
    CompositeArgs cb = a.t.c;
    cb.g_rgb = nullptr; cb.g_disp = nullptr; cb.g_acc = nullptr; cb.g_w = nullptr; cb.g_depth = nullptr;
    cb.g_raw = a.g_raw;
    float gm = 0.f;
    composite_bwd_ray<NC>(cb, ray, lane, st, sd, sa, gwi, true, g_r, g_g, g_b, a.gmax_ray ? &gm : nullptr);
    if (a.gmax_ray) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o, 64));
      if (lane == 0) a.gmax_ray[ray] = gm;
    }
    if (WAVES == 1 && a.c0.raw) {
      // ---- backward of the coarse ray's compositing (composite_bwd_kernel<1>) ----
      const float g0_r = tl_bcast(gx, 3), g0_g = tl_bcast(gx, 4), g0_b = tl_bcast(gx, 5);
      SampleState s0[1];
      double r0, g0, b0, d0, a0;
      composite_ray<1>(a.c0, ray, lane, s0, r0, g0, b0, d0, a0);
      float gm0 = 0.f;
      composite_bwd_ray<1>(a.c0, ray, lane, s0, d0, a0, nullptr, true, g0_r, g0_g, g0_b, a.gmax_ray ? &gm0 : nullptr);
      if (a.gmax_ray) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gm0 = fmaxf(gm0, __shfl_xor(gm0, o, 64));
        if (lane == 0) a.gmax_ray[a.t.c.N + ray] = gm0;
      }
    }
  } else if (role == 1) {
    // ---- backward of the coarse ray's compositing (composite_bwd_kernel<1>); its colour gradient is the loss's
    //      d / d rgb0 = 2 (rgb0 - target) mask / (3 N): no part of the fine ray enters ----
    TlRayIo io;
    io.have_rgb = true;                            // (lanes 0..2 idle: the fine colour belongs to wave 0)
    const float gx = tl_bwd_ray<TL_GX>(a.l, ray, lane, 1.0f, part, io);
    if (a.c0.raw) {
      const float g0_r = tl_bcast(gx, 3), g0_g = tl_bcast(gx, 4), g0_b = tl_bcast(gx, 5);
      SampleState s0[1];
      double r0, g0, b0, d0, a0;
      composite_ray<1>(a.c0, ray, lane, s0, r0, g0, b0, d0, a0);
      float gm0 = 0.f;
      composite_bwd_ray<1>(a.c0, ray, lane, s0, d0, a0, nullptr, true, g0_r, g0_g, g0_b, a.gmax_ray ? &gm0 : nullptr);
      if (a.gmax_ray) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gm0 = fmaxf(gm0, __shfl_xor(gm0, o, 64));
        if (lane == 0) a.gmax_ray[a.t.c.N + ray] = gm0;
      }
    }
    __syncthreads();
    // ---- the ray's loss terms (train_loss_fb_kernel's forward) ----
    TlRayIo io2;
    io2.pred_row = cat + S; io2.have_rgb = true;
    io2.r = colour[0]; io2.g = colour[1]; io2.b = colour[2];
    tl_fwd_ray(a.l, ray, lane, io2);
  } else if (role == 2) {
    __syncthreads();
    // ---- the ray's scale / shift gradient ----
    TlRayIo io;
    io.pred_row = cat + S;
    tl_bwd_ray<TL_SS>(a.l, ray, lane, 1.0f, part, io);
  } else {
    __syncthreads();
    if (a.t.z_std) {                               // torch.std(unbiased=False), :744
      double s1 = 0.0;
      for (int s = lane; s < Si; s += 64) s1 += (double)cat[S + s];
      const double mean = wave_sum_d(s1) / (double)Si;
      double s2 = 0.0;
      for (int s = lane; s < Si; s += 64) {
        const double dlt = (double)cat[S + s] - mean;
        s2 += dlt * dlt;
      }
      s2 = wave_sum_d(s2);
      if (lane == 0) a.t.z_std[ray] = (float)sqrt(s2 / (double)Si);
    }
  }
}

// ---------------------------------------------------------------------------
// space-carving loss
// ---------------------------------------------------------------------------
struct CarveArgs {
  const float* pred;     // [N,P]
  const float* hyp;      // [K,N] (hypothesis-major, trailing 1 squeezed)
  const float* mask;     // [N] or null
  float* partial;        // fwd non-joint: [N] per-ray sums ; joint: [K,P] column sums (atomic)
  float* loss;           // [1]
  const float* g_loss;   // bwd: [1] upstream gradient
  float* g_pred;         // [N,P]
  float* g_hyp;          // [K,N]
  float threshold;
  int N, P, K;
};

// value of lane `src` (wave-uniform index) for every lane
__device__ __forceinline__ float lane_bcast(float v, int src) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src));
}

__device__ __forceinline__ float carve_dist(float pred, float h, float m, bool has_mask, float thr) {
  float dd = fabsf(pred - h);                     // norm over a size-1 axis == |.| for any p
  if (has_mask) dd = dd * m;
  if (thr > 0.f && dd < thr) dd = 0.f;
  return dd;
}

// non-joint: per (ray, sample) min over K, mean over samples, mean over rays
__global__ void carve_fwd_kernel(CarveArgs a) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  double acc = 0.0;
  if (a.K <= 64) {
    // the ray's K hypotheses in ONE load (lane k holds hyp[k]); v_readlane hands them out
    const float hreg = lane < a.K ? a.hyp[(size_t)lane * a.N + ray] : 0.f;
    for (int s = lane; s < a.P; s += 64) {
      const float p = a.pred[(size_t)ray * a.P + s];
      float best = INFINITY;
      for (int k = 0; k < a.K; ++k)
        best = min_nan(best, carve_dist(p, lane_bcast(hreg, k), m, hm, a.threshold));
      acc += (double)best;
    }
  } else {
    for (int s = lane; s < a.P; s += 64) {
      const float p = a.pred[(size_t)ray * a.P + s];
      float best = INFINITY;
      for (int k = 0; k < a.K; ++k)
        best = min_nan(best, carve_dist(p, a.hyp[(size_t)k * a.N + ray], m, hm, a.threshold));
      acc += (double)best;
    }
  }
  acc = wave_sum_d(acc);
  if (lane == 0) a.partial[ray] = (float)(acc / (double)a.P);       // helpers:125 mean over samples
}

// one workgroup of 1024 threads (a lone wave walking 16,384 partials was 60 us of dependent loads); the
// 16 wave sums meet in LDS and are added in a fixed order (deterministic)
__global__ __launch_bounds__(1024) void carve_reduce_kernel(const float* partial, int n, float* loss) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) acc += (double)partial[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = (float)(t / (double)n);         // helpers:126 mean over rays
  }
}

__global__ void carve_bwd_kernel(CarveArgs a) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  const float scale = a.g_loss[0] / ((float)a.N * (float)a.P);
  // per-lane accumulators for d/d hyp[k] are reduced per k with wave sums
  const bool pre = a.K <= 64;        // the ray's hypotheses fit one register: lane k holds hyp[k]
  const float hreg = (pre && lane < a.K) ? a.hyp[(size_t)lane * a.N + ray] : 0.f;
  for (int k0 = 0; k0 < a.K; k0 += 64) {
    float ghk = 0.f;    // lane l holds gradient of hypothesis k0 + l
    for (int s0 = 0; s0 < a.P; s0 += 64) {
      const int s = s0 + lane;
      float gp = 0.f; int kbest = -1; float sgn = 0.f;
      const float p = s < a.P ? a.pred[(size_t)ray * a.P + s] : 0.f;
      float best = INFINITY, hbest = 0.f;
      for (int k = 0; k < a.K; ++k) {      // all lanes walk the loop (readlane needs a uniform index)
        const float h = pre ? lane_bcast(hreg, k) : a.hyp[(size_t)k * a.N + ray];
        const float dd = carve_dist(p, h, m, hm, a.threshold);
        if (dd < best) { best = dd; kbest = k; hbest = h; }         // first index wins ties (torch.min)
      }
      if (s < a.P) {
        const float h = hbest;
        const float diff = p - h;
        float dd = fabsf(diff);
        if (hm) dd *= m;
        const bool dead = a.threshold > 0.f && dd < a.threshold;
        sgn = dead ? 0.f : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
        gp = sgn * m * scale;
        if (k0 == 0) a.g_pred[(size_t)ray * a.P + s] = gp;
      }
      // scatter -gp into the winning hypothesis: lane j (hypothesis k0 + j) walks the 64 samples of
      // this chunk in order (two v_readlane per sample instead of a 6-step cross-lane sum per hypothesis)
      for (int l = 0; l < 64; ++l) {
        const int kb = __builtin_amdgcn_readlane(kbest, l);
        const float g = lane_bcast(gp, l);
        if (k0 + lane == kb) ghk -= g;
      }
    }
    if (k0 + lane < a.K) a.g_hyp[(size_t)(k0 + lane) * a.N + ray] = ghk;
  }
}

// joint variant (is_joint=True): mean over rays -> min over K -> mean over samples
__global__ void carve_joint_colsum_kernel(CarveArgs a) {
  // grid: K blocks x 1 ; each block sums dist over rays for its hypothesis: partial[k*P + s]
  const int k = blockIdx.x;
  const bool hm = a.mask != nullptr;
  for (int s = threadIdx.x; s < a.P; s += blockDim.x) {
    double acc = 0.0;
    for (int r = 0; r < a.N; ++r)
      acc += (double)carve_dist(a.pred[(size_t)r * a.P + s], a.hyp[(size_t)k * a.N + r],
                                hm ? a.mask[r] : 1.f, hm, a.threshold);
    a.partial[(size_t)k * a.P + s] = (float)(acc / (double)a.N);
  }
}
__global__ void carve_joint_min_kernel(CarveArgs a, int* argmin_out) {
  double acc = 0.0;
  for (int s = threadIdx.x; s < a.P; s += 64) {
    float best = INFINITY; int kb = 0;
    for (int k = 0; k < a.K; ++k) {
      const float v = a.partial[(size_t)k * a.P + s];
      if (v < best) { best = v; kb = k; }
    }
    if (argmin_out) argmin_out[s] = kb;
    acc += (double)best;
  }
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) a.loss[0] = (float)(acc / (double)a.P);
}
__global__ void carve_joint_bwd_kernel(CarveArgs a, const int* argmin_in) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  const float scale = a.g_loss[0] / ((float)a.N * (float)a.P);
  for (int k = lane; k < a.K; k += 64) a.g_hyp[(size_t)k * a.N + ray] = 0.f;
  __builtin_amdgcn_wave_barrier();
  for (int s = lane; s < a.P; s += 64) {
    const int kb = argmin_in[s];
    const float p = a.pred[(size_t)ray * a.P + s];
    const float diff = p - a.hyp[(size_t)kb * a.N + ray];
    float dd = fabsf(diff);
    if (hm) dd *= m;
    const bool dead = a.threshold > 0.f && dd < a.threshold;
    const float sgn = dead ? 0.f : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
    const float gp = sgn * m * scale;
    a.g_pred[(size_t)ray * a.P + s] = gp;
    atomicAdd(&a.g_hyp[(size_t)kb * a.N + ray], -gp);
  }
}


// ---- hypotheses cached per sample: target_hypothesis [K,N,P] ("each quantile here already picked a
// hypothesis", helpers:100-102).  Same reductions as above with hyp indexed (k, ray, sample); a [K,N,P]
// tensor is streamed once, coalesced along the samples.
__global__ void carve_knp_fwd_kernel(CarveArgs a) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  const size_t kstride = (size_t)a.N * a.P;
  double acc = 0.0;
  for (int s = lane; s < a.P; s += 64) {
    const float p = a.pred[(size_t)ray * a.P + s];
    const float* h = a.hyp + (size_t)ray * a.P + s;
    float best = INFINITY;
    for (int k = 0; k < a.K; ++k) best = min_nan(best, carve_dist(p, h[k * kstride], m, hm, a.threshold));
    acc += (double)best;
  }
  acc = wave_sum_d(acc);
  if (lane == 0) a.partial[ray] = (float)(acc / (double)a.P);
}
__global__ void carve_knp_bwd_kernel(CarveArgs a) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  const float scale = a.g_loss[0] / ((float)a.N * (float)a.P);
  const size_t kstride = (size_t)a.N * a.P;
  for (int s = lane; s < a.P; s += 64) {
    const size_t o = (size_t)ray * a.P + s;
    const float p = a.pred[o];
    float best = INFINITY, hbest = 0.f;
    int kbest = 0;
    for (int k = 0; k < a.K; ++k) {
      const float h = a.hyp[k * kstride + o];
      const float dd = carve_dist(p, h, m, hm, a.threshold);
      if (dd < best) { best = dd; kbest = k; hbest = h; }           // first index wins ties (torch.min)
    }
    const float diff = p - hbest;
    float dd = fabsf(diff);
    if (hm) dd *= m;
    const bool dead = a.threshold > 0.f && dd < a.threshold;
    const float sgn = dead ? 0.f : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
    const float gp = sgn * m * scale;
    a.g_pred[o] = gp;
    for (int k = 0; k < a.K; ++k) a.g_hyp[k * kstride + o] = k == kbest ? -gp : 0.f;
  }
}
__global__ void carve_knp_colsum_kernel(CarveArgs a) {
  const int k = blockIdx.x;
  const bool hm = a.mask != nullptr;
  const float* hk = a.hyp + (size_t)k * a.N * a.P;
  for (int s = threadIdx.x; s < a.P; s += blockDim.x) {
    double acc = 0.0;
    for (int r = 0; r < a.N; ++r)
      acc += (double)carve_dist(a.pred[(size_t)r * a.P + s], hk[(size_t)r * a.P + s], hm ? a.mask[r] : 1.f, hm,
                                a.threshold);
    a.partial[(size_t)k * a.P + s] = (float)(acc / (double)a.N);
  }
}
__global__ void carve_knp_joint_bwd_kernel(CarveArgs a, const int* argmin_in) {
  const int ray = blockIdx.x * RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  const float scale = a.g_loss[0] / ((float)a.N * (float)a.P);
  const size_t kstride = (size_t)a.N * a.P;
  for (int s = lane; s < a.P; s += 64) {
    const size_t o = (size_t)ray * a.P + s;
    const int kb = argmin_in[s];
    const float diff = a.pred[o] - a.hyp[kb * kstride + o];
    float dd = fabsf(diff);
    if (hm) dd *= m;
    const bool dead = a.threshold > 0.f && dd < a.threshold;
    const float sgn = dead ? 0.f : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
    const float gp = sgn * m * scale;
    a.g_pred[o] = gp;
    for (int k = 0; k < a.K; ++k) a.g_hyp[k * kstride + o] = k == kb ? -gp : 0.f;
  }
}

// ---------------------------------------------------------------------------
// img2mse: mean((x-y)^2), optional per-row mask (run_scade_wild.py:978-986)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void mse_fwd_kernel(const float* x, const float* y, const float* mask, int n, int c,
                                                       float* loss) {
  __shared__ double red[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n * c; i += 1024) {
    const float dlt = x[i] - y[i];
    float sq = dlt * dlt;
    if (mask) sq = sq * mask[i / c];
    acc += (double)sq;
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += red[w];
    loss[0] = (float)(t / (double)(n * c));
  }
}
__global__ void mse_bwd_kernel(const float* x, const float* y, const float* mask, int n, int c,
                               const float* g_loss, float* g_x) {
  const float scale = 2.0f * g_loss[0] / (float)(n * c);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n * c; i += gridDim.x * 256) {
    float g = (x[i] - y[i]) * scale;
    if (mask) g = g * mask[i / c];
    g_x[i] = g;
  }
}

}  // namespace scade

// ===========================================================================
// C ABI
// ===========================================================================
using namespace scade;

static inline int grid_rays(int N) { return (N + RAYS_PER_WG - 1) / RAYS_PER_WG; }

extern "C" int scade_ray_points(const float* rays, int ray_stride, const float* t_vals,
                                const float* t_rand, int N, int S, int lindisp, float* z_vals,
                                float* pts, void* stream) {
  if (N <= 0) return 0;
  SCADE_REQUIRE(rays && t_vals && z_vals, -1, "scade_ray_points: null pointer");
  SCADE_REQUIRE(ray_stride >= 8 && S >= 1, -2, "scade_ray_points: ray_stride >= 8 and S >= 1 required");
  RayPointsArgs a{};
  a.rays = rays; a.t_vals = t_vals; a.t_rand = t_rand; a.z_vals = z_vals; a.pts = pts;
  a.N = N; a.S = S; a.ray_stride = ray_stride; a.lindisp = lindisp;
  hipLaunchKernelGGL(ray_points_kernel, dim3(grid_rays(N)), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_ray_points");
}

extern "C" int scade_ray_points_draw(const float* rays, int ray_stride, const float* t_vals, int N, int S,
                                     int lindisp, unsigned long long seed, unsigned long long step,
                                     const float* step_dev, int Si, float* z_vals, float* pts, float* u_a,
                                     float* u_b, void* stream) {
  if (N <= 0) return 0;
  SCADE_REQUIRE(rays && t_vals && z_vals, -1, "scade_ray_points_draw: null pointer");
  SCADE_REQUIRE(ray_stride >= 8 && S >= 1 && Si >= 0, -2, "scade_ray_points_draw: ray_stride >= 8, S >= 1, Si >= 0 required");
  SCADE_REQUIRE(Si > 0 || (!u_a && !u_b), -2, "scade_ray_points_draw: sampler draws requested with Si = 0");
  RayPointsArgs a{};
  a.rays = rays; a.t_vals = t_vals; a.z_vals = z_vals; a.pts = pts;
  a.N = N; a.S = S; a.ray_stride = ray_stride; a.lindisp = lindisp;
  a.draw = 1; a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.step = step; a.step_dev = step_dev;
  a.u_a = u_a; a.u_b = u_b; a.Si = Si;
  hipLaunchKernelGGL(ray_points_kernel, dim3(grid_rays(N)), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_ray_points_draw");
}

template <template <int> class>
struct Dummy {};

#define DISPATCH_NC(KERN, S, ...)                                                            \
  do {                                                                                       \
    const int nc_ = ((S) + 63) / 64;                                                         \
    switch (nc_) {                                                                           \
      case 1: hipLaunchKernelGGL(KERN<1>, __VA_ARGS__); break;                               \
      case 2: hipLaunchKernelGGL(KERN<2>, __VA_ARGS__); break;                               \
      case 3: hipLaunchKernelGGL(KERN<3>, __VA_ARGS__); break;                               \
      case 4: hipLaunchKernelGGL(KERN<4>, __VA_ARGS__); break;                               \
      case 5: case 6: hipLaunchKernelGGL(KERN<6>, __VA_ARGS__); break;                       \
      case 7: case 8: hipLaunchKernelGGL(KERN<8>, __VA_ARGS__); break;                       \
      default: hipLaunchKernelGGL(KERN<16>, __VA_ARGS__); break;                             \
    }                                                                                        \
  } while (0)

extern "C" int scade_composite_fwd(const float* raw, const float* z_vals, const float* rays_d,
                                   int d_stride, const float* noise, int N, int S, float* rgb_map,
                                   float* disp_map, float* acc_map, float* weights,
                                   float* depth_map, void* stream) {
  if (N <= 0) return 0;
  SCADE_REQUIRE(raw && z_vals && rays_d && rgb_map && disp_map && acc_map && weights && depth_map, -1,
                "scade_composite_fwd: null pointer");
  SCADE_REQUIRE(S >= 1 && S <= 1024, -2, "scade_composite_fwd: S=%d outside [1,1024]", S);
  CompositeArgs a{};
  a.raw = raw; a.z = z_vals; a.rays_d = rays_d; a.noise = noise; a.rgb_map = rgb_map;
  a.disp_map = disp_map; a.acc_map = acc_map; a.weights = weights; a.depth_map = depth_map;
  a.N = N; a.S = S; a.d_stride = d_stride;
  DISPATCH_NC(composite_fwd_kernel, S, dim3(grid_rays(N)), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_composite_fwd");
}

extern "C" int scade_composite_bwd(const float* raw, const float* z_vals, const float* rays_d,
                                   int d_stride, const float* noise, int N, int S,
                                   const float* g_rgb, const float* g_disp, const float* g_acc,
                                   const float* g_weights, const float* g_depth, float* g_raw,
                                   void* stream) {
  if (N <= 0) return 0;
  SCADE_REQUIRE(raw && z_vals && rays_d && g_raw, -1, "scade_composite_bwd: null pointer");
  SCADE_REQUIRE(S >= 1 && S <= 1024, -2, "scade_composite_bwd: S=%d outside [1,1024]", S);
  CompositeArgs a{};
  a.raw = raw; a.z = z_vals; a.rays_d = rays_d; a.noise = noise;
  a.g_rgb = g_rgb; a.g_disp = g_disp; a.g_acc = g_acc; a.g_w = g_weights; a.g_depth = g_depth;
  a.g_raw = g_raw; a.N = N; a.S = S; a.d_stride = d_stride;
  DISPATCH_NC(composite_bwd_kernel, S, dim3(grid_rays(N)), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_composite_bwd");
}

static int check_pdf(const char* fn, const SamplePdfArgs& a) {
  SCADE_REQUIRE(a.bins && a.u && (a.w || a.cdf_in), -1, "%s: null pointer", fn);
  SCADE_REQUIRE(a.M >= 2 && a.M <= 2048 && a.S >= 1, -2, "%s: M=%d outside [2,2048] or S<1", fn, a.M);
  return 0;
}

extern "C" int scade_sample_pdf_fwd(const float* bins, int bins_stride, int bins_are_mids,
                                    const float* weights, int w_stride, const float* cdf_in,
                                    const float* u, int u_stride, int N, int M, int S,
                                    float* samples, long long* inds, float* cdf_out, float* z_std,
                                    void* stream) {
  SamplePdfArgs a{};
  a.bins = bins; a.w = weights; a.u = u; a.cdf_in = cdf_in; a.samples = samples; a.inds = inds;
  a.cdf_out = cdf_out; a.z_std = z_std; a.N = N; a.M = M; a.S = S; a.bins_stride = bins_stride;
  a.w_stride = w_stride; a.u_stride = u_stride; a.bins_are_mids = bins_are_mids;
  if (N <= 0) return 0;
  if (int e = check_pdf("scade_sample_pdf_fwd", a)) return e;
  SCADE_REQUIRE(samples, -1, "scade_sample_pdf_fwd: samples is null");
  const size_t lds = (size_t)RAYS_PER_WG * 2 * M * sizeof(float);
  hipLaunchKernelGGL(sample_pdf_fwd_kernel, dim3(grid_rays(N)), dim3(256), lds, (hipStream_t)stream, a);
  return scade_check_launch("scade_sample_pdf_fwd");
}

extern "C" int scade_sample_pdf_bwd(const float* bins, int bins_stride, int bins_are_mids,
                                    const float* weights, int w_stride, const float* u,
                                    int u_stride, const float* g_samples, int N, int M, int S,
                                    float* g_weights, void* stream) {
  SamplePdfArgs a{};
  a.bins = bins; a.w = weights; a.u = u; a.g_samples = g_samples; a.g_w = g_weights; a.N = N;
  a.M = M; a.S = S; a.bins_stride = bins_stride; a.w_stride = w_stride; a.u_stride = u_stride;
  a.bins_are_mids = bins_are_mids;
  if (N <= 0) return 0;
  if (int e = check_pdf("scade_sample_pdf_bwd", a)) return e;
  SCADE_REQUIRE(weights && g_samples && g_weights, -1, "scade_sample_pdf_bwd: null pointer");
  const size_t lds = (size_t)RAYS_PER_WG * 4 * M * sizeof(float);
  hipLaunchKernelGGL(sample_pdf_bwd_kernel, dim3(grid_rays(N)), dim3(256), lds, (hipStream_t)stream, a);
  return scade_check_launch("scade_sample_pdf_bwd");
}

extern "C" int scade_merge_sorted(const float* z_a, int Sa, const float* z_b, int Sb,
                                  const float* rays, int ray_stride, int N, float* z_out,
                                  float* pts, void* stream) {
  if (N <= 0 || Sa + Sb == 0) return 0;
  SCADE_REQUIRE((z_a || Sa == 0) && (z_b || Sb == 0) && z_out, -1, "scade_merge_sorted: null pointer");
  SCADE_REQUIRE(!pts || (rays && ray_stride >= 6), -1, "scade_merge_sorted: pts needs rays");
  SCADE_REQUIRE(Sa >= 0 && Sb >= 0 && Sa + Sb <= 4096, -2, "scade_merge_sorted: Sa+Sb > 4096");
  MergeArgs a{z_a, z_b, rays, z_out, pts, N, Sa, Sb, ray_stride};
  int PB = 64;
  while (PB < Sa + Sb) PB <<= 1;
  const dim3 grid(grid_rays(N)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (PB) {
    case 64: hipLaunchKernelGGL(merge_sorted_reg_kernel<1>, grid, block, 0, s, a); break;
    case 128: hipLaunchKernelGGL(merge_sorted_reg_kernel<2>, grid, block, 0, s, a); break;
    case 256: hipLaunchKernelGGL(merge_sorted_reg_kernel<4>, grid, block, 0, s, a); break;
    case 512: hipLaunchKernelGGL(merge_sorted_reg_kernel<8>, grid, block, 0, s, a); break;
    default:
      hipLaunchKernelGGL(merge_sorted_kernel, grid, block, (size_t)RAYS_PER_WG * PB * sizeof(float), s, a, PB);
  }
  return scade_check_launch("scade_merge_sorted");
}

extern "C" int scade_ray_tail(const float* raw, const float* z_vals, const float* rays, int ray_stride,
                              const float* noise, int N, int S, const float* u, int u_stride, int Si,
                              float* rgb_map, float* disp_map, float* acc_map, float* weights,
                              float* depth_map, float* samples, float* z_std, float* z_out, float* pts,
                              void* stream) {
  if (N <= 0) return 0;
  SCADE_REQUIRE(raw && z_vals && rays && u && rgb_map && disp_map && acc_map && weights && depth_map, -1,
                "scade_ray_tail: null pointer");
  SCADE_REQUIRE(ray_stride >= 6, -2, "scade_ray_tail: ray rows need o and d");
  SCADE_REQUIRE(S >= 3 && S <= 512 && Si >= 1 && Si <= 1024, -2,
                "scade_ray_tail: S=%d outside [3,512] or Si=%d outside [1,1024]", S, Si);
  SCADE_REQUIRE(z_out || samples, -1, "scade_ray_tail: neither z_out nor samples requested");
  SCADE_REQUIRE(!pts || z_out, -1, "scade_ray_tail: pts needs z_out");
  TailArgs a{};
  a.c.raw = raw; a.c.z = z_vals; a.c.rays_d = rays + 3; a.c.noise = noise; a.c.rgb_map = rgb_map;
  a.c.disp_map = disp_map; a.c.acc_map = acc_map; a.c.weights = weights; a.c.depth_map = depth_map;
  a.c.N = N; a.c.S = S; a.c.d_stride = ray_stride;
  a.rays = rays; a.u = u; a.samples = samples; a.z_std = z_std; a.z_out = z_out; a.pts = pts;
  a.ray_stride = ray_stride; a.u_stride = u_stride; a.Si = Si;
  const size_t lds = (size_t)RAYS_PER_WG * (4 * S + Si) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(grid_rays(N)), block(256);
  if (!z_out) {
    DISPATCH_NC(ray_tail_kernel0, S, grid, block, lds, s, a);
  } else {
    SCADE_REQUIRE(S + Si <= 512 && S <= 256, -2, "scade_ray_tail: merged row S+Si=%d > 512 (or S > 256)", S + Si);
    int PB = 64;
    while (PB < S + Si) PB <<= 1;
    const int nc = (S + 63) / 64;
    bool launched = false;
#define TAIL_CASE(NCX, RX)                                                                     \
    if (nc == NCX && PB == 64 * RX) {                                                          \
      hipLaunchKernelGGL((ray_tail_kernel<NCX, RX>), grid, block, lds, s, a);                  \
      launched = true;                                                                         \
    }
    TAIL_CASE(1, 1) TAIL_CASE(1, 2) TAIL_CASE(1, 4) TAIL_CASE(1, 8)
    TAIL_CASE(2, 2) TAIL_CASE(2, 4) TAIL_CASE(2, 8)
    TAIL_CASE(3, 4) TAIL_CASE(3, 8)
    TAIL_CASE(4, 4) TAIL_CASE(4, 8)
#undef TAIL_CASE
    SCADE_REQUIRE(launched, -2, "scade_ray_tail: no kernel for S=%d, Si=%d", S, Si);
  }
  return scade_check_launch("scade_ray_tail");
}

// Backward of scade_ray_tail's fine form (no merge): d loss / d raw from the gradients of the five
// compositing outputs and of the drawn samples, one launch (see ray_tail_bwd_kernel).
extern "C" int scade_ray_tail_bwd(const float* raw, const float* z_vals, const float* rays, int ray_stride,
                                  const float* noise, int N, int S, const float* u, int u_stride, int Si,
                                  const float* g_rgb, const float* g_disp, const float* g_acc,
                                  const float* g_weights, const float* g_depth, const float* g_samples,
                                  float* g_raw, void* stream) {
  if (N <= 0) return 0;
  SCADE_REQUIRE(raw && z_vals && rays && u && g_samples && g_raw, -1, "scade_ray_tail_bwd: null pointer");
  SCADE_REQUIRE(ray_stride >= 6, -2, "scade_ray_tail_bwd: ray rows need o and d");
  SCADE_REQUIRE(S >= 3 && S <= 512 && Si >= 1 && Si <= 1024, -2,
                "scade_ray_tail_bwd: S=%d outside [3,512] or Si=%d outside [1,1024]", S, Si);
  TailBwdArgs a{};
  a.c.raw = raw; a.c.z = z_vals; a.c.rays_d = rays + 3; a.c.noise = noise; a.c.d_stride = ray_stride;
  a.c.g_rgb = g_rgb; a.c.g_disp = g_disp; a.c.g_acc = g_acc; a.c.g_w = g_weights; a.c.g_depth = g_depth;
  a.c.g_raw = g_raw; a.c.N = N; a.c.S = S;
  a.u = u; a.g_samples = g_samples; a.u_stride = u_stride; a.Si = Si;
  const size_t lds = (size_t)RAYS_PER_WG * 6 * S * sizeof(float);
  DISPATCH_NC(ray_tail_bwd_kernel, S, dim3(grid_rays(N)), dim3(256), lds, (hipStream_t)stream, a);
  return scade_check_launch("scade_ray_tail_bwd");
}

// scade_ray_tail (fine form) + scade_train_loss_fb + scade_ray_tail_bwd (+ scade_composite_bwd of the coarse
// ray, raw0 != NULL) in ONE launch + the loss's reduce: see ray_tail_train_kernel.  The loss differentiates the
// total with a UNIT gradient (scade_train_loss_fb's contract); gradients w.r.t. the five compositing outputs
// other than the colour are zero by construction of the train loss.
namespace scade {
template <int NC>
__global__ __launch_bounds__(256) void ray_tail_train_split(TailTrainArgs a) { ray_tail_train_body<NC, 4>(a); }
template <int NC>
__global__ __launch_bounds__(256) void ray_tail_train_seq(TailTrainArgs a) { ray_tail_train_body<NC, 1>(a); }
}  // namespace scade
static int ray_tail_train_impl(const float* raw, const float* z_vals, const float* rays, int ray_stride,
                                    const float* noise, int N, int S, const float* u, int u_stride, int Si,
                                    float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map,
                                    float* samples, float* z_std,
                                    const float* rgb0, const float* target, const float* hyp, const float* scales,
                                    const float* shifts, const long long* img_i_dev, int img_i, const float* mask,
                                    int mse_masked, int carve_on, float carve_weight, float threshold, float out_scale,
                                    int K, float* workspace, float* loss4, float* g_scales, float* g_shifts, int n_ss,
                                    float* g_raw, const float* raw0, const float* z0, const float* noise0, int S0,
                                    float* g_raw0, float* gmax_ws, float* gmax_fine, float* gmax_coarse, void* stream) {
  SCADE_REQUIRE(N > 0, -2, "scade_ray_tail_train: empty batch");
  SCADE_REQUIRE(!gmax_ws || (gmax_fine && gmax_coarse && raw0 && !noise && !noise0), -2,
                "scade_ray_tail_train_gmax: the maxima need both rays, both slot arrays and no raw noise");
  SCADE_REQUIRE(raw && z_vals && rays && u && rgb_map && disp_map && acc_map && weights && depth_map && g_raw, -1,
                "scade_ray_tail_train: null pointer");
  SCADE_REQUIRE(rgb0 && target && workspace && loss4, -1, "scade_ray_tail_train: null pointer (loss)");
  SCADE_REQUIRE(!carve_on || (hyp && scales && shifts && g_scales && g_shifts && K > 0), -1,
                "scade_ray_tail_train: the carving term needs hyp, scales, shifts and their gradient buffers");
  SCADE_REQUIRE(ray_stride >= 6, -2, "scade_ray_tail_train: ray rows need o and d");
  SCADE_REQUIRE(S >= 3 && S <= 256 && Si >= 1 && Si <= 1024, -2,
                "scade_ray_tail_train: S=%d outside [3,256] or Si=%d outside [1,1024]", S, Si);
  SCADE_REQUIRE(img_i_dev ? img_i > 0 : img_i >= 0, -2, "scade_ray_tail_train: img_i (host index, or n_images beside a device index)");
  SCADE_REQUIRE(n_ss >= 0 && (n_ss == 0 || (g_scales && g_shifts)), -2, "scade_ray_tail_train: n_ss rows need g_scales / g_shifts");
  SCADE_REQUIRE(!raw0 || (z0 && g_raw0 && S0 >= 1 && S0 <= 64), -2, "scade_ray_tail_train: the coarse ray needs z0, g_raw0 and S0 <= 64");
  TailTrainArgs a{};
  a.t.c.raw = raw; a.t.c.z = z_vals; a.t.c.rays_d = rays + 3; a.t.c.noise = noise; a.t.c.rgb_map = rgb_map;
  a.t.c.disp_map = disp_map; a.t.c.acc_map = acc_map; a.t.c.weights = weights; a.t.c.depth_map = depth_map;
  a.t.c.N = N; a.t.c.S = S; a.t.c.d_stride = ray_stride;
  a.t.rays = rays; a.t.u = u; a.t.samples = samples; a.t.z_std = z_std;
  a.t.ray_stride = ray_stride; a.t.u_stride = u_stride; a.t.Si = Si;
  a.l.rgb0 = rgb0; a.l.target = target; a.l.hyp = hyp; a.l.scales = scales; a.l.shifts = shifts;
  a.l.img_i_dev = img_i_dev; a.l.img_i = img_i; a.l.mask = mask; a.l.mse_masked = mse_masked; a.l.carve_on = carve_on;
  a.l.carve_weight = carve_weight; a.l.threshold = threshold; a.l.out_scale = out_scale; a.l.N = N; a.l.P = Si; a.l.K = K;
  a.l.partial = workspace; a.l.loss = loss4; a.l.g_scales = g_scales; a.l.g_shifts = g_shifts;
  a.g_raw = g_raw;
  a.gmax_ray = gmax_ws;
  if (raw0) {
    a.c0.raw = raw0; a.c0.z = z0; a.c0.rays_d = rays + 3; a.c0.noise = noise0; a.c0.d_stride = ray_stride;
    a.c0.g_raw = g_raw0; a.c0.N = N; a.c0.S = S0;
  }
  hipStream_t s = (hipStream_t)stream;
  // four waves per ray up to four waves per SIMD (4 SIMDs per CU), else one
  const size_t row = (size_t)(7 * S + 2 * Si + 4) * sizeof(float);
  if ((long)N * 4 <= 16L * device_cus()) {
    DISPATCH_NC(ray_tail_train_split, S, dim3(N), dim3(256), row, s, a);
  } else {
    DISPATCH_NC(ray_tail_train_seq, S, dim3(grid_rays(N)), dim3(256), RAYS_PER_WG * row, s, a);
  }
  if (int e = scade_check_launch("scade_ray_tail_train")) return e;
  return scade_launch_train_loss_fb_reduce(a.l, n_ss, s, gmax_ws, gmax_fine, gmax_coarse);
}

extern "C" int scade_ray_tail_train(const float* raw, const float* z_vals, const float* rays, int ray_stride,
                                    const float* noise, int N, int S, const float* u, int u_stride, int Si,
                                    float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map,
                                    float* samples, float* z_std,
                                    const float* rgb0, const float* target, const float* hyp, const float* scales,
                                    const float* shifts, const long long* img_i_dev, int img_i, const float* mask,
                                    int mse_masked, int carve_on, float carve_weight, float threshold, float out_scale,
                                    int K, float* workspace, float* loss4, float* g_scales, float* g_shifts, int n_ss,
                                    float* g_raw, const float* raw0, const float* z0, const float* noise0, int S0,
                                    float* g_raw0, void* stream) {
  return ray_tail_train_impl(raw, z_vals, rays, ray_stride, noise, N, S, u, u_stride, Si, rgb_map, disp_map, acc_map, weights,
                             depth_map, samples, z_std, rgb0, target, hyp, scales, shifts, img_i_dev, img_i, mask, mse_masked,
                             carve_on, carve_weight, threshold, out_scale, K, workspace, loss4, g_scales, g_shifts, n_ss, g_raw,
                             raw0, z0, noise0, S0, g_raw0, nullptr, nullptr, nullptr, stream);
}

// scade_ray_tail_train that also leaves the loss-scale maxima of the two output gradients it writes, for the 16-bit MLP
// backward of the same step (scade_mlp_bwd_lp2_deferred's gmax_pre: its own maxima launch is then skipped): gmax_ws
// [2 N] floats of workspace (per-ray maxima), gmax_fine / gmax_coarse [256] floats each = the slots the backward reads
// (slot 0 = the maximum over the finite entries of the EFFECTIVE gradient - colour channels as they are, the density
// channel times 1 - exp(-10 sigma) = sigmoid(10 alpha_pre) - the rest zero).  Needs the coarse ray and no raw noise.
extern "C" int scade_ray_tail_train_gmax(const float* raw, const float* z_vals, const float* rays, int ray_stride,
                                         int N, int S, const float* u, int u_stride, int Si,
                                         float* rgb_map, float* disp_map, float* acc_map, float* weights, float* depth_map,
                                         float* samples, float* z_std,
                                         const float* rgb0, const float* target, const float* hyp, const float* scales,
                                         const float* shifts, const long long* img_i_dev, int img_i, const float* mask,
                                         int mse_masked, int carve_on, float carve_weight, float threshold, float out_scale,
                                         int K, float* workspace, float* loss4, float* g_scales, float* g_shifts, int n_ss,
                                         float* g_raw, const float* raw0, const float* z0, int S0, float* g_raw0,
                                         float* gmax_ws, float* gmax_fine, float* gmax_coarse, void* stream) {
  SCADE_REQUIRE(gmax_ws && gmax_fine && gmax_coarse, -1, "scade_ray_tail_train_gmax: null pointer");
  return ray_tail_train_impl(raw, z_vals, rays, ray_stride, nullptr, N, S, u, u_stride, Si, rgb_map, disp_map, acc_map, weights,
                             depth_map, samples, z_std, rgb0, target, hyp, scales, shifts, img_i_dev, img_i, mask, mse_masked,
                             carve_on, carve_weight, threshold, out_scale, K, workspace, loss4, g_scales, g_shifts, n_ss, g_raw,
                             raw0, z0, nullptr, S0, g_raw0, gmax_ws, gmax_fine, gmax_coarse, stream);
}

extern "C" long scade_carve_workspace_floats(int N, int P, int K, int is_joint) {
  return is_joint ? (long)K * P + P : (long)N;
}

extern "C" int scade_carve_fwd(const float* pred, const float* hyp, const float* mask,
                               float threshold, int is_joint, int N, int P, int K,
                               float* workspace, float* loss, void* stream) {
  SCADE_REQUIRE(pred && hyp && workspace && loss, -1, "scade_carve_fwd: null pointer");
  SCADE_REQUIRE(N > 0 && P > 0 && K > 0, -2, "scade_carve_fwd: empty problem");
  CarveArgs a{};
  a.pred = pred; a.hyp = hyp; a.mask = mask; a.partial = workspace; a.loss = loss;
  a.threshold = threshold; a.N = N; a.P = P; a.K = K;
  hipStream_t s = (hipStream_t)stream;
  if (!is_joint) {
    hipLaunchKernelGGL(carve_fwd_kernel, dim3(grid_rays(N)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(carve_reduce_kernel, dim3(1), dim3(1024), 0, s, workspace, N, loss);
  } else {
    hipLaunchKernelGGL(carve_joint_colsum_kernel, dim3(K), dim3(128), 0, s, a);
    hipLaunchKernelGGL(carve_joint_min_kernel, dim3(1), dim3(64), 0, s, a,
                       reinterpret_cast<int*>(workspace + (size_t)K * P));
  }
  return scade_check_launch("scade_carve_fwd");
}

// the two phases of the joint forward as separate entries: a ray-sharded job all-reduces the [K,P]
// column means between them ("exchange-step exceptions")
extern "C" int scade_carve_joint_colmean(const float* pred, const float* hyp, const float* mask,
                                         float threshold, int N, int P, int K, float* workspace,
                                         void* stream) {
  SCADE_REQUIRE(pred && hyp && workspace, -1, "scade_carve_joint_colmean: null pointer");
  SCADE_REQUIRE(N > 0 && P > 0 && K > 0, -2, "scade_carve_joint_colmean: empty problem");
  CarveArgs a{};
  a.pred = pred; a.hyp = hyp; a.mask = mask; a.partial = workspace;
  a.threshold = threshold; a.N = N; a.P = P; a.K = K;
  hipLaunchKernelGGL(carve_joint_colsum_kernel, dim3(K), dim3(128), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_carve_joint_colmean");
}

extern "C" int scade_carve_joint_min(float* workspace, int P, int K, float* loss, void* stream) {
  SCADE_REQUIRE(workspace && loss, -1, "scade_carve_joint_min: null pointer");
  SCADE_REQUIRE(P > 0 && K > 0, -2, "scade_carve_joint_min: empty problem");
  CarveArgs a{};
  a.partial = workspace; a.loss = loss; a.P = P; a.K = K;
  hipLaunchKernelGGL(carve_joint_min_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a,
                     reinterpret_cast<int*>(workspace + (size_t)K * P));
  return scade_check_launch("scade_carve_joint_min");
}

extern "C" int scade_carve_bwd(const float* pred, const float* hyp, const float* mask,
                               float threshold, int is_joint, int N, int P, int K,
                               const float* workspace, const float* g_loss, float* g_pred,
                               float* g_hyp, void* stream) {
  SCADE_REQUIRE(pred && hyp && g_loss && g_pred && g_hyp, -1, "scade_carve_bwd: null pointer");
  SCADE_REQUIRE(N > 0 && P > 0 && K > 0, -2, "scade_carve_bwd: empty problem");
  CarveArgs a{};
  a.pred = pred; a.hyp = hyp; a.mask = mask; a.g_loss = g_loss; a.g_pred = g_pred; a.g_hyp = g_hyp;
  a.threshold = threshold; a.N = N; a.P = P; a.K = K;
  hipStream_t s = (hipStream_t)stream;
  if (!is_joint) {
    hipLaunchKernelGGL(carve_bwd_kernel, dim3(grid_rays(N)), dim3(256), 0, s, a);
  } else {
    SCADE_REQUIRE(workspace, -1, "scade_carve_bwd: joint mode needs the forward workspace");
    hipLaunchKernelGGL(carve_joint_bwd_kernel, dim3(grid_rays(N)), dim3(256), 0, s, a,
                       reinterpret_cast<const int*>(workspace + (size_t)K * P));
  }
  return scade_check_launch("scade_carve_bwd");
}

// ---- the same three entries for hypotheses cached per sample, hyp [K,N,P] (helpers:100-102) ----------
extern "C" int scade_carve_knp_fwd(const float* pred, const float* hyp, const float* mask, float threshold,
                                   int is_joint, int N, int P, int K, float* workspace, float* loss,
                                   void* stream) {
  SCADE_REQUIRE(pred && hyp && workspace && loss, -1, "scade_carve_knp_fwd: null pointer");
  SCADE_REQUIRE(N > 0 && P > 0 && K > 0, -2, "scade_carve_knp_fwd: empty problem");
  CarveArgs a{};
  a.pred = pred; a.hyp = hyp; a.mask = mask; a.partial = workspace; a.loss = loss;
  a.threshold = threshold; a.N = N; a.P = P; a.K = K;
  hipStream_t s = (hipStream_t)stream;
  if (!is_joint) {
    hipLaunchKernelGGL(carve_knp_fwd_kernel, dim3(grid_rays(N)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(carve_reduce_kernel, dim3(1), dim3(1024), 0, s, workspace, N, loss);
  } else {
    hipLaunchKernelGGL(carve_knp_colsum_kernel, dim3(K), dim3(128), 0, s, a);
    hipLaunchKernelGGL(carve_joint_min_kernel, dim3(1), dim3(64), 0, s, a,
                       reinterpret_cast<int*>(workspace + (size_t)K * P));
  }
  return scade_check_launch("scade_carve_knp_fwd");
}

extern "C" int scade_carve_knp_joint_colmean(const float* pred, const float* hyp, const float* mask,
                                             float threshold, int N, int P, int K, float* workspace,
                                             void* stream) {
  SCADE_REQUIRE(pred && hyp && workspace, -1, "scade_carve_knp_joint_colmean: null pointer");
  SCADE_REQUIRE(N > 0 && P > 0 && K > 0, -2, "scade_carve_knp_joint_colmean: empty problem");
  CarveArgs a{};
  a.pred = pred; a.hyp = hyp; a.mask = mask; a.partial = workspace;
  a.threshold = threshold; a.N = N; a.P = P; a.K = K;
  hipLaunchKernelGGL(carve_knp_colsum_kernel, dim3(K), dim3(128), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_carve_knp_joint_colmean");
}

extern "C" int scade_carve_knp_bwd(const float* pred, const float* hyp, const float* mask, float threshold,
                                   int is_joint, int N, int P, int K, const float* workspace,
                                   const float* g_loss, float* g_pred, float* g_hyp, void* stream) {
  SCADE_REQUIRE(pred && hyp && g_loss && g_pred && g_hyp, -1, "scade_carve_knp_bwd: null pointer");
  SCADE_REQUIRE(N > 0 && P > 0 && K > 0, -2, "scade_carve_knp_bwd: empty problem");
  CarveArgs a{};
  a.pred = pred; a.hyp = hyp; a.mask = mask; a.g_loss = g_loss; a.g_pred = g_pred; a.g_hyp = g_hyp;
  a.threshold = threshold; a.N = N; a.P = P; a.K = K;
  hipStream_t s = (hipStream_t)stream;
  if (!is_joint) {
    hipLaunchKernelGGL(carve_knp_bwd_kernel, dim3(grid_rays(N)), dim3(256), 0, s, a);
  } else {
    SCADE_REQUIRE(workspace, -1, "scade_carve_knp_bwd: joint mode needs the forward workspace");
    hipLaunchKernelGGL(carve_knp_joint_bwd_kernel, dim3(grid_rays(N)), dim3(256), 0, s, a,
                       reinterpret_cast<const int*>(workspace + (size_t)K * P));
  }
  return scade_check_launch("scade_carve_knp_bwd");
}

extern "C" int scade_mse_fwd(const float* x, const float* y, const float* row_mask, int n, int c,
                             float* loss, void* stream) {
  SCADE_REQUIRE(x && y && loss, -1, "scade_mse_fwd: null pointer");
  SCADE_REQUIRE(n > 0 && c > 0, -2, "scade_mse_fwd: empty input");
  hipLaunchKernelGGL(mse_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, y, row_mask, n, c, loss);
  return scade_check_launch("scade_mse_fwd");
}

extern "C" int scade_mse_bwd(const float* x, const float* y, const float* row_mask, int n, int c,
                             const float* g_loss, float* g_x, void* stream) {
  SCADE_REQUIRE(x && y && g_loss && g_x, -1, "scade_mse_bwd: null pointer");
  SCADE_REQUIRE(n > 0 && c > 0, -2, "scade_mse_bwd: empty input");
  const int grid = min(64, (n * c + 255) / 256);
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, row_mask, n, c,
                     g_loss, g_x);
  return scade_check_launch("scade_mse_bwd");
}

// ---------------------------------------------------------------------------
// standalone positional encoding (Embedder.embed, model/run_nerf_helpers.py:142-172)
// ---------------------------------------------------------------------------
namespace scade {
__global__ void embed_kernel(const float* x, int P, int D, int L, float* out) {
  const int C = D * (1 + 2 * L);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)P * D * (1 + L);
       i += (size_t)gridDim.x * 256) {
    const int s = (int)(i % (1 + L));
    const size_t pc = i / (1 + L);
    const int c = (int)(pc % D);
    const size_t p = pc / D;
    const float v = x[p * D + c];
    float* o = out + p * C;
    if (s == 0) {
      o[c] = v;
    } else {
      const float arg = (v * 3.14159274101257324f) * (float)(1 << (s - 1));
      float sn, cs;
      sincosf(arg, &sn, &cs);
      o[D + 2 * D * (s - 1) + c] = sn;
      o[D + 2 * D * (s - 1) + D + c] = cs;
    }
  }
}
}  // namespace scade

extern "C" int scade_embed(const float* x, int P, int D, int multires, float* out, void* stream) {
  if (P <= 0) return 0;
  SCADE_REQUIRE(x && out, -1, "scade_embed: null pointer");
  SCADE_REQUIRE(D >= 1 && multires >= 0 && multires <= 24, -2, "scade_embed: bad D/multires");
  const size_t items = (size_t)P * D * (1 + multires);
  const int grid = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
  hipLaunchKernelGGL(scade::embed_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, P, D, multires, out);
  return scade_check_launch("scade_embed");
}

// ---------------------------------------------------------------------------
// standalone stratified jitter of an arbitrary z tensor (perturb_z_vals :564-579)
// ---------------------------------------------------------------------------
namespace scade {
__global__ void perturb_z_kernel(const float* z, const float* t_rand, int N, int S, float* out) {
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < (size_t)N * S;
       idx += (size_t)gridDim.x * 256) {
    const int i = (int)(idx % S);
    const float zc = z[idx];
    const float lower = i > 0 ? 0.5f * (zc + z[idx - 1]) : zc;
    const float upper = i + 1 < S ? 0.5f * (z[idx + 1] + zc) : zc;
    out[idx] = lower + (upper - lower) * t_rand[idx];
  }
}
}  // namespace scade

extern "C" int scade_perturb_z(const float* z_vals, const float* t_rand, int N, int S, float* out,
                               void* stream) {
  if (N <= 0 || S <= 0) return 0;
  SCADE_REQUIRE(z_vals && t_rand && out, -1, "scade_perturb_z: null pointer");
  const size_t items = (size_t)N * S;
  const int grid = (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096);
  hipLaunchKernelGGL(scade::perturb_z_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, z_vals, t_rand, N, S, out);
  return scade_check_launch("scade_perturb_z");
}

// ---------------------------------------------------------------------------
// ray generation + training-batch gather:
// get_ray_dirs/get_rays (model/run_nerf_helpers.py:285-305), the ray-row assembly of
// render()/render_hyp() (run_scade_scannet.py:122-141) and the per-pixel gathers of
// get_ray_batch_from_one_image_hypothesis_idx (:784-821) in ONE launch: the reference
// generates all H*W rays every step and then gathers N_rand of them.
// ---------------------------------------------------------------------------
namespace scade {
struct GenRaysArgs {
  const int* coords;      // [N,2] (row j, col i) or null == every pixel, row-major
  const float* intrinsic; // fx fy cx cy
  const float* c2w;       // rows 0..2, 4 columns, row stride c2w_stride
  const float* image;     // [H,W,3] or null
  const float* hyps;      // [K,H,W] or null
  float* rays;            // [N,11] or null: o d near far viewdir
  float* rays_o;          // [N,3] or null
  float* rays_d;          // [N,3] or null
  float* target_s;        // [N,3]
  float* target_h;        // [K,N]
  float* mask;            // [N] or null
  float near, far;
  int N, H, W, K, c2w_stride, corner_px, edge_px;
};

struct CamRT {
  float fx, fy, cx, cy;
  float R[3][3], T[3];
};
__device__ __forceinline__ CamRT load_cam(const float* intrinsic, const float* c2w, int c2w_stride) {
  CamRT c;
  c.fx = intrinsic[0]; c.fy = intrinsic[1]; c.cx = intrinsic[2]; c.cy = intrinsic[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) c.R[r][k] = c2w[r * c2w_stride + k];
    c.T[r] = c2w[r * c2w_stride + 3];
  }
  return c;
}
// everything of pixel (row j, col i) that is per ray: direction, the [o d near far viewdir] row, the target
// colour and the corner / edge mask, written at batch position n
__device__ __forceinline__ void gen_ray_item(const GenRaysArgs& a, const CamRT& c, int n, int j, int i) {
  // helpers:296  dirs = [((i+.5)-cx)/fx, (H-(j+.5)-cy)/fy, -1]
  const float d0 = (((float)i + 0.5f) - c.cx) / c.fx;
  const float d1 = ((float)a.H - ((float)j + 0.5f) - c.cy) / c.fy;
  const float d2 = -1.0f;
  float d[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] = (d0 * c.R[r][0] + d1 * c.R[r][1]) + d2 * c.R[r][2];   // helpers:298
  if (a.rays_o) { a.rays_o[3 * n] = c.T[0]; a.rays_o[3 * n + 1] = c.T[1]; a.rays_o[3 * n + 2] = c.T[2]; }
  if (a.rays_d) { a.rays_d[3 * n] = d[0]; a.rays_d[3 * n + 1] = d[1]; a.rays_d[3 * n + 2] = d[2]; }
  if (a.rays) {
    float* o = a.rays + (size_t)n * 11;
    const float nrm = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);     // :128 viewdirs / norm
    o[0] = c.T[0]; o[1] = c.T[1]; o[2] = c.T[2];
    o[3] = d[0]; o[4] = d[1]; o[5] = d[2];
    o[6] = a.near; o[7] = a.far;
    o[8] = d[0] / nrm; o[9] = d[1] / nrm; o[10] = d[2] / nrm;
  }
  const size_t pix = (size_t)j * a.W + i;
  if (a.image && a.target_s) {
    a.target_s[3 * n] = a.image[pix * 3]; a.target_s[3 * n + 1] = a.image[pix * 3 + 1];
    a.target_s[3 * n + 2] = a.image[pix * 3 + 2];
  }
  if (a.mask) {
    float m = 1.f;
    const int cp = a.corner_px, e = a.edge_px;
    if (cp > 0 && (j < cp || j >= a.H - cp) && (i < cp || i >= a.W - cp)) m = 0.f;     // :810-817
    if (e > 0 && (j < e || j >= a.H - e || i < e || i >= a.W - e)) m = 0.f;            // wild :818-830
    a.mask[n] = m;
  }
}

__global__ void gen_rays_kernel(GenRaysArgs a) {
  const CamRT c = load_cam(a.intrinsic, a.c2w, a.c2w_stride);
  for (int n = blockIdx.x * 256 + threadIdx.x; n < a.N; n += gridDim.x * 256) {
    int j, i;
    if (a.coords) { j = a.coords[2 * n]; i = a.coords[2 * n + 1]; }
    else { j = n / a.W; i = n - j * a.W; }
    gen_ray_item(a, c, n, j, i);
    if (a.hyps && a.target_h) {
      const size_t pix = (size_t)j * a.W + i;
      for (int k = 0; k < a.K; ++k) a.target_h[(size_t)k * a.N + n] = a.hyps[(size_t)k * a.H * a.W + pix];
    }
  }
}

// The per-iteration batch assembly of the training loop (run_scade_scannet.py:946 image pick, :786 pixel pick,
// :784-821 gathers, :200-219 ray rows) as the ONE launch in front of a graph-captured train step: the N pixels are
// flat indices pix[n] = row * W + col (a slice of a device-resident permutation of the H*W pixels), the camera and
// the image / hypothesis planes are those of the step's training view; outputs go straight into the captured step's
// static input buffers.  The same launch stores the view's index where the captured step reads it (:951-954: which
// row of the depth scales / shifts) and advances the device-resident optimizer scalars (adam_tick), which is what
// scade_stage_inputs does for a batch assembled elsewhere.  One work item per ray row and one per (hypothesis, ray):
// the K gathers of a ray are independent loads, not a serial loop behind the row.
struct GatherBatchArgs {
  GenRaysArgs g;
  const long long* pix;
  long long* scalar_dst;
  long long scalar;
  float* tick[2];
  // scade_gather_batch_points: workgroups [points_block0, gridDim.x) are four waves = four rays of the step's first
  // per-ray kernel (ray_points: z_vals, coarse sample positions, the step's uniform draws), each from the ray row it
  // derives itself with gen_ray_item's arithmetic (nobody waits for the row another workgroup writes)
  RayPointsArgs pts;
  int points_block0;
  // ... and workgroups [pack_block0, points_block0 or gridDim.x) the step's weight packs (mlp_pack.h pack_item)
  PackItemsArgs pack;
  int pack_block0;
};
__global__ void gather_batch_kernel(GatherBatchArgs b) {
  const GenRaysArgs& a = b.g;
  if (b.pack_block0 >= 0 && (int)blockIdx.x >= b.pack_block0 &&
      (b.points_block0 < 0 || (int)blockIdx.x < b.points_block0)) {
    pack_item(b.pack, (int)blockIdx.x - b.pack_block0);
    return;
  }
  if (b.points_block0 >= 0 && (int)blockIdx.x >= b.points_block0) {
    const int n = ((int)blockIdx.x - b.points_block0) * RAYS_PER_WG + (int)(threadIdx.x >> 6);
    if (n >= a.N) return;
    const long long p = b.pix[n];
    const int j = (int)(p / a.W), i = (int)(p - (long long)j * a.W);
    const CamRT c = load_cam(a.intrinsic, a.c2w, a.c2w_stride);
    const float d0 = (((float)i + 0.5f) - c.cx) / c.fx;                     // (gen_ray_item, helpers:296-298)
    const float d1 = ((float)a.H - ((float)j + 0.5f) - c.cy) / c.fy;
    const float d2 = -1.0f;
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = (d0 * c.R[r][0] + d1 * c.R[r][1]) + d2 * c.R[r][2];
    ray_points_ray(b.pts, n, lane_id(), c.T[0], c.T[1], c.T[2], d[0], d[1], d[2], a.near, a.far);
    return;
  }
  const int nblk = b.pack_block0 >= 0 ? b.pack_block0 : (b.points_block0 >= 0 ? b.points_block0 : (int)gridDim.x);
  if (blockIdx.x == 0 && threadIdx.x == 0 && b.scalar_dst) *b.scalar_dst = b.scalar;
  if (blockIdx.x == 0 && threadIdx.x == 64 && b.tick[0]) adam_tick(b.tick[0]);
  if (blockIdx.x == 0 && threadIdx.x == 128 && b.tick[1]) adam_tick(b.tick[1]);
  const long items = (long)a.N * (1 + (a.hyps && a.target_h ? a.K : 0));
  const size_t plane = (size_t)a.H * a.W;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < items; t += (long)nblk * 256) {
    if (t < a.N) {
      const int n = (int)t;
      const long long p = b.pix[n];
      const int j = (int)(p / a.W), i = (int)(p - (long long)j * a.W);
      const CamRT c = load_cam(a.intrinsic, a.c2w, a.c2w_stride);
      gen_ray_item(a, c, n, j, i);
    } else {
      const long q = t - a.N;
      const int k = (int)(q / a.N), n = (int)(q - (long)k * a.N);
      a.target_h[(size_t)k * a.N + n] = a.hyps[(size_t)k * plane + (size_t)b.pix[n]];
    }
  }
}
}  // namespace scade

extern "C" int scade_gen_rays(const int* coords, int N, int H, int W, const float* intrinsic,
                              const float* c2w, int c2w_stride, float near, float far,
                              const float* image, const float* hyps, int K, int corner_px,
                              int edge_px, float* rays, float* rays_o, float* rays_d,
                              float* target_s, float* target_h, float* mask, void* stream) {
  if (N <= 0) return 0;
  SCADE_REQUIRE(intrinsic && c2w && c2w_stride >= 4, -1, "scade_gen_rays: intrinsic/c2w missing");
  SCADE_REQUIRE(H > 0 && W > 0, -2, "scade_gen_rays: bad image size");
  scade::GenRaysArgs a{coords, intrinsic, c2w, image, hyps, rays, rays_o, rays_d, target_s, target_h,
                       mask, near, far, N, H, W, K, c2w_stride, corner_px, edge_px};
  const int grid = (N + 255) / 256 < 2048 ? (N + 255) / 256 : 2048;
  hipLaunchKernelGGL(scade::gen_rays_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_gen_rays");
}

static int gather_batch_impl(const long long* pix, int N, int H, int W, const float* intrinsic,
                             const float* c2w, int c2w_stride, float near, float far, const float* image,
                             const float* hyps, int K, int corner_px, int edge_px, float* rays,
                             float* target_s, float* target_h, float* mask, long long* scalar_dst,
                             long long scalar, float* const* tick_states, const scade::RayPointsArgs* pts,
                             const scade::PackItemsArgs* pack, void* stream) {
  SCADE_REQUIRE(N >= 0 && H > 0 && W > 0 && K >= 0, -2, "scade_gather_batch: bad sizes");
  SCADE_REQUIRE(N == 0 || (pix && intrinsic && c2w && c2w_stride >= 4), -1, "scade_gather_batch: pix / intrinsic / c2w missing");
  SCADE_REQUIRE(!hyps == !target_h || K == 0, -1, "scade_gather_batch: hyps and target_h go together");
  scade::GatherBatchArgs b{};
  b.g = scade::GenRaysArgs{nullptr, intrinsic, c2w, image, hyps, rays, nullptr, nullptr, target_s, target_h,
                           mask, near, far, N, H, W, K, c2w_stride, corner_px, edge_px};
  b.pix = pix;
  b.scalar_dst = scalar_dst;
  b.scalar = scalar;
  if (tick_states) { b.tick[0] = tick_states[0]; b.tick[1] = tick_states[1]; }
  if (N == 0 && !scalar_dst && !b.tick[0] && !b.tick[1]) return 0;
  const long items = (long)N * (1 + (hyps && target_h ? K : 0));
  long grid = (items + 255) / 256;
  if (grid < 1) grid = 1;
  if (grid > 4096) grid = 4096;
  b.points_block0 = -1;
  b.pack_block0 = -1;
  if (pack && pack->n_nets > 0) {
    b.pack = *pack;
    b.pack_block0 = (int)grid;
    grid += scade::pack_item_count(pack->n_nets, pack->fmt);
  }
  if (pts && N > 0) {
    b.pts = *pts;
    b.points_block0 = (int)grid;
    grid += (N + scade::RAYS_PER_WG - 1) / scade::RAYS_PER_WG;
  }
  hipLaunchKernelGGL(scade::gather_batch_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, b);
  return scade_check_launch("scade_gather_batch");
}

extern "C" int scade_gather_batch(const long long* pix, int N, int H, int W, const float* intrinsic,
                                  const float* c2w, int c2w_stride, float near, float far, const float* image,
                                  const float* hyps, int K, int corner_px, int edge_px, float* rays,
                                  float* target_s, float* target_h, float* mask, long long* scalar_dst,
                                  long long scalar, float* const* tick_states, void* stream) {
  return gather_batch_impl(pix, N, H, W, intrinsic, c2w, c2w_stride, near, far, image, hyps, K, corner_px, edge_px, rays,
                           target_s, target_h, mask, scalar_dst, scalar, tick_states, nullptr, nullptr, stream);
}

// scade_gather_batch + scade_ray_points_draw of the gathered rays (host step index: this launch runs OUTSIDE the
// captured step, every iteration) as ONE launch: z_vals [N,S], pts [N,S,3], u_a / u_b [N,Si] (nullable) are the
// captured step's static coarse-sample buffers.  Same bits as the two launches.
extern "C" int scade_gather_batch_points(const long long* pix, int N, int H, int W, const float* intrinsic,
                                         const float* c2w, int c2w_stride, float near, float far, const float* image,
                                         const float* hyps, int K, int corner_px, int edge_px, float* rays,
                                         float* target_s, float* target_h, float* mask, long long* scalar_dst,
                                         long long scalar, float* const* tick_states, const float* t_vals, int S,
                                         int lindisp, unsigned long long seed, unsigned long long step, int Si,
                                         float* z_vals, float* pts, float* u_a, float* u_b, int pack_format, int n_nets,
                                         const float* const* net_params, float* const* packed_exact,
                                         void* const* packed_fwd, void* const* packed_t, void* stream) {
  scade::PackItemsArgs pk;
  SCADE_REQUIRE(scade::pack_items_fill(pk, pack_format, n_nets, net_params, packed_exact, packed_fwd, packed_t), -1,
                "scade_gather_batch_points: pack arguments (format 0..3, one or two networks, 24 parameter pointers each)");
  SCADE_REQUIRE(t_vals && z_vals && S >= 1 && Si >= 0, -2, "scade_gather_batch_points: t_vals, z_vals, S >= 1, Si >= 0 required");
  SCADE_REQUIRE(Si > 0 || (!u_a && !u_b), -2, "scade_gather_batch_points: sampler draws requested with Si = 0");
  scade::RayPointsArgs a{};
  a.t_vals = t_vals; a.z_vals = z_vals; a.pts = pts; a.N = N; a.S = S; a.lindisp = lindisp;
  a.draw = 1; a.seed_lo = (unsigned)seed; a.seed_hi = (unsigned)(seed >> 32); a.step = step;
  a.u_a = u_a; a.u_b = u_b; a.Si = Si;
  return gather_batch_impl(pix, N, H, W, intrinsic, c2w, c2w_stride, near, far, image, hyps, K, corner_px, edge_px, rays,
                           target_s, target_h, mask, scalar_dst, scalar, tick_states, &a, &pk, stream);
}

// The three-term training loss of the reference's train loop as ONE forward and ONE backward pair of
// launches (run_scade_scannet.py:954, :968-983; wild variant run_scade_wild.py:977-1008):
//
//     target_h = hyp * scale[img_i] + shift[img_i]
//     loss = mse(rgb, target) + w * space_carving(pred_hyp, target_h) + mse(rgb0, target)
//
// The operator API keeps img2mse / compute_space_carving_loss as separate entries (ray_ops.hip); a train
// step built from them costs ~25 launches for this scalar (two mse, carve + its reduce, the affine map of
// the hypotheses, three scalar adds / muls, and the mirror image of all that in the backward).  The
// Trainer uses this fused form: one wave per ray does the ray's part of all three terms, a one-workgroup
// kernel reduces the per-ray partials in a fixed order (deterministic).  Same arithmetic as the separate
// kernels: fp64 accumulation of the means, the affine map as a multiply and an add with separate
// roundings, first-index tie rule of torch.min, sign(0) = 0.
#include "common.h"

namespace scade {

constexpr int TL_RAYS_PER_WG = 4;

struct TrainLossArgs {
  const float* rgb;        // [N,3] fine colour
  const float* rgb0;       // [N,3] coarse colour
  const float* target;     // [N,3]
  const float* pred;       // [N,P] depth hypotheses of the fine sampler
  const float* hyp;        // [K,N] raw hypotheses (before scale / shift)
  const float* scales;     // [n_images] DEPTH_SCALES
  const float* shifts;     // [n_images] DEPTH_SHIFTS
  const long long* img_i_dev;   // device index of the image, or null -> img_i
  const float* mask;       // [N] or null
  float* partial;          // [N,4]: fwd {sq, sq0, carve_ray, -}; bwd {g_scale_ray, g_shift_ray, -, -}
  float* loss;             // [4] total, img_loss, carve, img_loss0
  // backward
  const float* g_loss;     // [1]
  float* g_rgb;            // [N,3]
  float* g_rgb0;           // [N,3]
  float* g_pred;           // [N,P]
  float* g_scales;         // [n_images] accumulated into (+=), or null
  float* g_shifts;
  float carve_weight, threshold, out_scale;
  int img_i, N, P, K, mse_masked, carve_on;
};

__device__ __forceinline__ double tl_wave_sum_d(double v) { return wave_sum_dpp_d(v); }
__device__ __forceinline__ float tl_bcast(float v, int src) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src));
}
__device__ __forceinline__ float tl_dist(float pred, float h, float m, bool has_mask, float thr) {
  float dd = fabsf(pred - h);                     // norm over a size-1 axis == |.| for any p (helpers:106)
  if (has_mask) dd = dd * m;                      // helpers:108-110
  if (thr > 0.f && dd < thr) dd = 0.f;            // helpers:112-113
  return dd;
}
// image of the batch: the host index, or (graph-captured steps) a device index bounded by img_i = n_images.
// A device index outside [0, n_images) returns -1: the affine map becomes NaN (the loss says so) and no
// scale / shift gradient is written - never an out-of-bounds access.
__device__ __forceinline__ int tl_image(const TrainLossArgs& a) {
  if (!a.img_i_dev) return a.img_i;
  const long long im = a.img_i_dev[0];
  return (im >= 0 && im < (long long)a.img_i) ? (int)im : -1;
}
__device__ __forceinline__ float tl_row(const float* v, int im) { return im >= 0 ? v[im] : __builtin_nanf(""); }

__device__ __forceinline__ void tl_fwd_ray(const TrainLossArgs& a, int ray, int lane) {
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  // photometric terms: lanes 0..2 = fine channels, 3..5 = coarse channels
  double sq = 0.0;
  if (lane < 6) {
    const int c = lane % 3;
    const float x = (lane < 3 ? a.rgb : a.rgb0)[ray * 3 + c];
    const float d = x - a.target[ray * 3 + c];
    float s = d * d;
    if (hm && a.mse_masked) s = s * m;            // run_scade_wild.py:980-982
    sq = (double)s;
  }
  const double sq_f = tl_wave_sum_d(lane < 3 ? sq : 0.0), sq_c = tl_wave_sum_d(lane >= 3 ? sq : 0.0);
  float carve_ray = 0.f;
  if (a.carve_on) {
    const int im = tl_image(a);
    const float sc = tl_row(a.scales, im), sh = tl_row(a.shifts, im);
    double acc = 0.0;
    if (a.K <= 64) {
      float hreg = 0.f;
      if (lane < a.K) { hreg = a.hyp[(size_t)lane * a.N + ray] * sc; hreg = hreg + sh; }     // :954
      for (int s = lane; s < a.P; s += 64) {
        const float p = a.pred[(size_t)ray * a.P + s];
        float best = INFINITY;
        for (int k = 0; k < a.K; ++k) best = fminf(best, tl_dist(p, tl_bcast(hreg, k), m, hm, a.threshold));
        acc += (double)best;
      }
    } else {
      for (int s = lane; s < a.P; s += 64) {
        const float p = a.pred[(size_t)ray * a.P + s];
        float best = INFINITY;
        for (int k = 0; k < a.K; ++k) {
          float h = a.hyp[(size_t)k * a.N + ray] * sc;
          h = h + sh;
          best = fminf(best, tl_dist(p, h, m, hm, a.threshold));
        }
        acc += (double)best;
      }
    }
    carve_ray = (float)(tl_wave_sum_d(acc) / (double)a.P);             // helpers:125 mean over samples
    if (im < 0) carve_ray = __builtin_nanf("");     // device image index out of range (fminf drops the NaN distances)
  }
  if (lane == 0) {
    f32x4 o = {(float)sq_f, (float)sq_c, carve_ray, 0.f};
    reinterpret_cast<f32x4*>(a.partial)[ray] = o;
  }
}

// one workgroup of 1024 threads; the 16 wave sums meet in LDS and are added in a fixed order
__global__ __launch_bounds__(1024) void train_loss_reduce_kernel(TrainLossArgs a) {
  __shared__ double red[3][16];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int i = threadIdx.x; i < a.N; i += 1024) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.partial)[i];
    s0 += (double)v[0]; s1 += (double)v[1]; s2 += (double)v[2];
  }
  s0 = tl_wave_sum_d(s0); s1 = tl_wave_sum_d(s1); s2 = tl_wave_sum_d(s2);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; red[2][threadIdx.x >> 6] = s2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s0 = s1 = s2 = 0.0;
    for (int w = 0; w < 16; ++w) { s0 += red[0][w]; s1 += red[1][w]; s2 += red[2][w]; }
    const float img = (float)(s0 / (double)(a.N * 3)), img0 = (float)(s1 / (double)(a.N * 3));   // helpers:11
    const float carve = (float)(s2 / (double)a.N);                                               // helpers:126
    float total = img;
    if (a.carve_on) total = total + a.carve_weight * carve;                                      // :976
    total = total + img0;                                                                         // :983
    if (a.out_scale != 1.0f) total = total * a.out_scale;     // this rank's share of a ray-sharded batch
    a.loss[0] = total; a.loss[1] = img; a.loss[2] = carve; a.loss[3] = img0;
  }
}

// ``g_in``: the gradient arriving at the total; ``part``: where the ray's scale / shift partial goes
__device__ __forceinline__ void tl_bwd_ray(const TrainLossArgs& a, int ray, int lane, float g_in, float* part) {
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  float g = g_in;
  if (a.out_scale != 1.0f) g = g * a.out_scale;
  if (lane < 6) {                                  // d mse / d x = 2 (x - y) mask / (3 N)
    const int c = lane % 3;
    const float x = (lane < 3 ? a.rgb : a.rgb0)[ray * 3 + c];
    const float scale = 2.0f * g / (float)(a.N * 3);
    float gx = (x - a.target[ray * 3 + c]) * scale;
    if (hm && a.mse_masked) gx = gx * m;
    (lane < 3 ? a.g_rgb : a.g_rgb0)[ray * 3 + c] = gx;
  }
  float gsc = 0.f, gsh = 0.f;
  if (a.carve_on) {
    const int im = tl_image(a);
    const float sc = tl_row(a.scales, im), sh = tl_row(a.shifts, im);
    const float gl = g * a.carve_weight;
    const float scale = gl / ((float)a.N * (float)a.P);
    for (int k0 = 0; k0 < a.K; k0 += 64) {
      const int kl = k0 + lane;
      const float hraw = kl < a.K ? a.hyp[(size_t)kl * a.N + ray] : 0.f;
      float hreg = hraw * sc;
      hreg = hreg + sh;
      float ghk = 0.f;                              // lane l: gradient w.r.t. target_h[k0 + l]
      for (int s0 = 0; s0 < a.P; s0 += 64) {
        const int s = s0 + lane;
        const float p = s < a.P ? a.pred[(size_t)ray * a.P + s] : 0.f;
        float best = INFINITY, hbest = 0.f;
        int kbest = -1;
        for (int k = 0; k < a.K; ++k) {             // all lanes walk the loop (uniform readlane index)
          float h;
          if (a.K <= 64) h = tl_bcast(hreg, k);
          else { h = a.hyp[(size_t)k * a.N + ray] * sc; h = h + sh; }
          const float dd = tl_dist(p, h, m, hm, a.threshold);
          if (dd < best) { best = dd; kbest = k; hbest = h; }          // first index wins ties (torch.min)
        }
        float gp = 0.f;
        if (s < a.P) {
          const float diff = p - hbest;
          float dd = fabsf(diff);
          if (hm) dd *= m;
          const bool dead = a.threshold > 0.f && dd < a.threshold;
          const float sgn = dead ? 0.f : (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
          gp = sgn * m * scale;
          if (k0 == 0) a.g_pred[(size_t)ray * a.P + s] = gp;
        } else {
          kbest = -1;
        }
        for (int l = 0; l < 64; ++l) {              // scatter -gp into the winning hypothesis
          const int kb = __builtin_amdgcn_readlane(kbest, l);
          const float gg = tl_bcast(gp, l);
          if (kl == kb) ghk -= gg;
        }
      }
      // d target_h / d scale = hyp_raw, d / d shift = 1
      gsc += (float)tl_wave_sum_d((double)(ghk * hraw));
      gsh += (float)tl_wave_sum_d((double)ghk);
    }
  } else if (a.g_pred) {
    for (int s = lane; s < a.P; s += 64) a.g_pred[(size_t)ray * a.P + s] = 0.f;
  }
  if (lane == 0) {
    f32x4 o = {gsc, gsh, 0.f, 0.f};
    reinterpret_cast<f32x4*>(part)[ray] = o;
  }
}

__global__ void train_loss_fwd_kernel(TrainLossArgs a) {
  const int ray = blockIdx.x * TL_RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  tl_fwd_ray(a, ray, lane_id());
}
__global__ void train_loss_bwd_kernel(TrainLossArgs a) {
  const int ray = blockIdx.x * TL_RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  tl_bwd_ray(a, ray, lane_id(), a.g_loss[0], a.partial);
}

// Forward AND backward of the loss in one launch (scade_train_loss_fb): the train step differentiates the total
// with a unit gradient, and nothing in the backward depends on the reduced loss value, so the ray's gradients can
// be written while its loss terms are computed.  partial: [N,4] loss terms | [N,4] scale / shift terms.
__global__ void train_loss_fb_kernel(TrainLossArgs a) {
  const int ray = blockIdx.x * TL_RAYS_PER_WG + (threadIdx.x >> 6);
  if (ray >= a.N) return;
  const int lane = lane_id();
  tl_fwd_ray(a, ray, lane);
  tl_bwd_ray(a, ray, lane, 1.0f, a.partial + 4 * (size_t)a.N);
}
// its one-workgroup reduce: the loss terms, and the scale / shift gradient rows - WRITTEN for all n_ss images
// (zero except the batch's image: the caller needs no zero fill of those rows), or accumulated when n_ss = 0
__global__ __launch_bounds__(1024) void train_loss_fb_reduce_kernel(TrainLossArgs a, int n_ss) {
  __shared__ double red[5][16];
  double s[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const f32x4* p0 = reinterpret_cast<const f32x4*>(a.partial);
  const f32x4* p1 = p0 + a.N;
  for (int i = threadIdx.x; i < a.N; i += 1024) {
    const f32x4 v = p0[i], w = p1[i];
    s[0] += (double)v[0]; s[1] += (double)v[1]; s[2] += (double)v[2]; s[3] += (double)w[0]; s[4] += (double)w[1];
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    s[j] = tl_wave_sum_d(s[j]);
    if ((threadIdx.x & 63) == 0) red[j][threadIdx.x >> 6] = s[j];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      s[j] = 0.0;
      for (int w = 0; w < 16; ++w) s[j] += red[j][w];
    }
    const float img = (float)(s[0] / (double)(a.N * 3)), img0 = (float)(s[1] / (double)(a.N * 3));   // helpers:11
    const float carve = (float)(s[2] / (double)a.N);                                                // helpers:126
    float total = img;
    if (a.carve_on) total = total + a.carve_weight * carve;                                         // :976
    total = total + img0;                                                                            // :983
    if (a.out_scale != 1.0f) total = total * a.out_scale;
    a.loss[0] = total; a.loss[1] = img; a.loss[2] = carve; a.loss[3] = img0;
    red[0][0] = s[3]; red[1][0] = s[4];
  }
  __syncthreads();
  if (a.g_scales) {
    const int im = a.carve_on ? tl_image(a) : -1;
    const float gs = (float)red[0][0], gh = (float)red[1][0];
    if (n_ss > 0) {
      for (int i = threadIdx.x; i < n_ss; i += 1024) {
        a.g_scales[i] = i == im ? gs : 0.f;
        a.g_shifts[i] = i == im ? gh : 0.f;
      }
    } else if (threadIdx.x == 0 && im >= 0) {
      a.g_scales[im] += gs;
      a.g_shifts[im] += gh;
    }
  }
}

__global__ __launch_bounds__(1024) void train_loss_ss_reduce_kernel(TrainLossArgs a) {
  __shared__ double red[2][16];
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < a.N; i += 1024) {
    const f32x4 v = reinterpret_cast<const f32x4*>(a.partial)[i];
    s0 += (double)v[0]; s1 += (double)v[1];
  }
  s0 = tl_wave_sum_d(s0); s1 = tl_wave_sum_d(s1);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s0 = s1 = 0.0;
    for (int w = 0; w < 16; ++w) { s0 += red[0][w]; s1 += red[1][w]; }
    const int im = tl_image(a);
    if (im >= 0) {
      a.g_scales[im] += (float)s0;
      a.g_shifts[im] += (float)s1;
    }
  }
}

}  // namespace scade

using namespace scade;

extern "C" int scade_train_loss_fwd(const float* rgb, const float* rgb0, const float* target, const float* pred,
                                    const float* hyp, const float* scales, const float* shifts,
                                    const long long* img_i_dev, int img_i, const float* mask, int mse_masked,
                                    int carve_on, float carve_weight, float threshold, float out_scale, int N,
                                    int P, int K, float* workspace, float* loss4, void* stream) {
  SCADE_REQUIRE(rgb && rgb0 && target && workspace && loss4, -1, "scade_train_loss_fwd: null pointer");
  SCADE_REQUIRE(!carve_on || (pred && hyp && scales && shifts), -1, "scade_train_loss_fwd: the carving term needs pred, hyp, scales, shifts");
  SCADE_REQUIRE(N > 0 && (!carve_on || (P > 0 && K > 0)), -2, "scade_train_loss_fwd: empty problem");
  SCADE_REQUIRE(img_i_dev ? img_i > 0 : img_i >= 0, -2, "scade_train_loss_fwd: img_i (host index, or n_images beside a device index)");
  TrainLossArgs a{};
  a.rgb = rgb; a.rgb0 = rgb0; a.target = target; a.pred = pred; a.hyp = hyp; a.scales = scales; a.shifts = shifts;
  a.img_i_dev = img_i_dev; a.img_i = img_i; a.mask = mask; a.mse_masked = mse_masked; a.carve_on = carve_on;
  a.carve_weight = carve_weight; a.threshold = threshold; a.out_scale = out_scale; a.N = N; a.P = P; a.K = K;
  a.partial = workspace; a.loss = loss4;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(train_loss_fwd_kernel, dim3((N + TL_RAYS_PER_WG - 1) / TL_RAYS_PER_WG), dim3(256), 0, s, a);
  hipLaunchKernelGGL(train_loss_reduce_kernel, dim3(1), dim3(1024), 0, s, a);
  return scade_check_launch("scade_train_loss_fwd");
}

extern "C" int scade_train_loss_bwd(const float* rgb, const float* rgb0, const float* target, const float* pred,
                                    const float* hyp, const float* scales, const float* shifts,
                                    const long long* img_i_dev, int img_i, const float* mask, int mse_masked,
                                    int carve_on, float carve_weight, float threshold, float out_scale, int N,
                                    int P, int K, float* workspace, const float* g_loss, float* g_rgb,
                                    float* g_rgb0, float* g_pred, float* g_scales, float* g_shifts, void* stream) {
  SCADE_REQUIRE(rgb && rgb0 && target && workspace && g_loss && g_rgb && g_rgb0, -1, "scade_train_loss_bwd: null pointer");
  SCADE_REQUIRE(!carve_on || (pred && hyp && scales && shifts && g_pred && g_scales && g_shifts), -1,
                "scade_train_loss_bwd: the carving term needs pred, hyp, scales, shifts and their gradient buffers");
  SCADE_REQUIRE(N > 0 && (!carve_on || (P > 0 && K > 0)), -2, "scade_train_loss_bwd: empty problem");
  SCADE_REQUIRE(img_i_dev ? img_i > 0 : img_i >= 0, -2, "scade_train_loss_bwd: img_i (host index, or n_images beside a device index)");
  TrainLossArgs a{};
  a.rgb = rgb; a.rgb0 = rgb0; a.target = target; a.pred = pred; a.hyp = hyp; a.scales = scales; a.shifts = shifts;
  a.img_i_dev = img_i_dev; a.img_i = img_i; a.mask = mask; a.mse_masked = mse_masked; a.carve_on = carve_on;
  a.carve_weight = carve_weight; a.threshold = threshold; a.out_scale = out_scale; a.N = N; a.P = P; a.K = K;
  a.partial = workspace; a.g_loss = g_loss; a.g_rgb = g_rgb; a.g_rgb0 = g_rgb0; a.g_pred = g_pred;
  a.g_scales = g_scales; a.g_shifts = g_shifts;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(train_loss_bwd_kernel, dim3((N + TL_RAYS_PER_WG - 1) / TL_RAYS_PER_WG), dim3(256), 0, s, a);
  if (carve_on) hipLaunchKernelGGL(train_loss_ss_reduce_kernel, dim3(1), dim3(1024), 0, s, a);
  return scade_check_launch("scade_train_loss_bwd");
}

// Forward and backward in ONE pair of launches, for callers that differentiate the total with a UNIT gradient
// (the train step does): arguments as scade_train_loss_fwd + the gradient outputs of scade_train_loss_bwd;
// workspace [8 N] floats.  n_ss > 0: g_scales / g_shifts [n_ss] are WRITTEN (zero but for the batch's image: no
// zero fill needed beforehand); n_ss = 0: the batch's image row is accumulated into as in scade_train_loss_bwd.
extern "C" int scade_train_loss_fb(const float* rgb, const float* rgb0, const float* target, const float* pred,
                                   const float* hyp, const float* scales, const float* shifts,
                                   const long long* img_i_dev, int img_i, const float* mask, int mse_masked,
                                   int carve_on, float carve_weight, float threshold, float out_scale, int N,
                                   int P, int K, float* workspace, float* loss4, float* g_rgb, float* g_rgb0,
                                   float* g_pred, float* g_scales, float* g_shifts, int n_ss, void* stream) {
  SCADE_REQUIRE(rgb && rgb0 && target && workspace && loss4 && g_rgb && g_rgb0, -1, "scade_train_loss_fb: null pointer");
  SCADE_REQUIRE(!carve_on || (pred && hyp && scales && shifts && g_pred && g_scales && g_shifts), -1,
                "scade_train_loss_fb: the carving term needs pred, hyp, scales, shifts and their gradient buffers");
  SCADE_REQUIRE(N > 0 && (!carve_on || (P > 0 && K > 0)), -2, "scade_train_loss_fb: empty problem");
  SCADE_REQUIRE(img_i_dev ? img_i > 0 : img_i >= 0, -2, "scade_train_loss_fb: img_i (host index, or n_images beside a device index)");
  SCADE_REQUIRE(n_ss >= 0 && (n_ss == 0 || (g_scales && g_shifts)), -2, "scade_train_loss_fb: n_ss rows need g_scales / g_shifts");
  TrainLossArgs a{};
  a.rgb = rgb; a.rgb0 = rgb0; a.target = target; a.pred = pred; a.hyp = hyp; a.scales = scales; a.shifts = shifts;
  a.img_i_dev = img_i_dev; a.img_i = img_i; a.mask = mask; a.mse_masked = mse_masked; a.carve_on = carve_on;
  a.carve_weight = carve_weight; a.threshold = threshold; a.out_scale = out_scale; a.N = N; a.P = P; a.K = K;
  a.partial = workspace; a.loss = loss4; a.g_rgb = g_rgb; a.g_rgb0 = g_rgb0; a.g_pred = g_pred;
  a.g_scales = g_scales; a.g_shifts = g_shifts;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(train_loss_fb_kernel, dim3((N + TL_RAYS_PER_WG - 1) / TL_RAYS_PER_WG), dim3(256), 0, s, a);
  hipLaunchKernelGGL(train_loss_fb_reduce_kernel, dim3(1), dim3(1024), 0, s, a, n_ss);
  return scade_check_launch("scade_train_loss_fb");
}

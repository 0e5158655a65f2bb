// Device side of the fused three-term train loss (train_loss.hip) - shared with the fused fine tail + loss +
// backward kernel of ray_ops.hip (scade_ray_tail_train), which runs the same per-ray functions on values it
// still holds in registers / LDS.
#pragma once
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

// Per-ray overrides for the fused tail + loss + backward kernel (ray_ops.hip: ray_tail_train_kernel), where the
// ray's fine colour is still in registers and its depth hypotheses / their gradient live in LDS rows.  The
// default (all null / false) reads and writes the global arrays of TrainLossArgs.
struct TlRayIo {
  const float* pred_row = nullptr;   // [P] hypotheses of this ray
  float* g_pred_row = nullptr;       // [P] receives d loss / d pred of this ray
  bool have_rgb = false;             // fine colour given below (wave-uniform)
  float r = 0.f, g = 0.f, b = 0.f;
  bool store_gx = true;              // write g_rgb / g_rgb0 (when given); false: the colour gradient is only returned
};
// the parts of tl_bwd_ray (all = the whole backward of the ray; the fused tail kernel gives the parts to different
// waves of the ray's workgroup - every output is computed by the same operations in the same order either way)
constexpr int TL_GX = 1;             // d / d colour (lanes 0..5)
constexpr int TL_GP = 2;             // d / d pred: the g_pred rows
constexpr int TL_SS = 4;             // d / d scale, shift: the scatter into the winning hypotheses + the ray's partial

__device__ __forceinline__ void tl_fwd_ray(const TrainLossArgs& a, int ray, int lane, const TlRayIo& io = TlRayIo()) {
  const float* pred_row = io.pred_row ? io.pred_row : a.pred + (size_t)ray * a.P;
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  // photometric terms: lanes 0..2 = fine channels, 3..5 = coarse channels
  double sq = 0.0;
  if (lane < 6) {
    const int c = lane % 3;
    const float x = (lane < 3 && io.have_rgb) ? (c == 0 ? io.r : c == 1 ? io.g : io.b)
                                              : (lane < 3 ? a.rgb : a.rgb0)[ray * 3 + c];
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
        const float p = pred_row[s];
        float best = INFINITY;
        for (int k = 0; k < a.K; ++k) best = min_nan(best, tl_dist(p, tl_bcast(hreg, k), m, hm, a.threshold));
        acc += (double)best;
      }
    } else {
      for (int s = lane; s < a.P; s += 64) {
        const float p = pred_row[s];
        float best = INFINITY;
        for (int k = 0; k < a.K; ++k) {
          float h = a.hyp[(size_t)k * a.N + ray] * sc;
          h = h + sh;
          best = min_nan(best, tl_dist(p, h, m, hm, a.threshold));
        }
        acc += (double)best;
      }
    }
    carve_ray = (float)(tl_wave_sum_d(acc) / (double)a.P);             // helpers:125 mean over samples
    if (im < 0) carve_ray = __builtin_nanf("");     // device image index out of range
  }
  if (lane == 0) {
    f32x4 o = {(float)sq_f, (float)sq_c, carve_ray, 0.f};
    reinterpret_cast<f32x4*>(a.partial)[ray] = o;
  }
}

// ``g_in``: the gradient arriving at the total; ``part``: where the ray's scale / shift partial goes
// returns, in lanes 0..2 / 3..5, the gradient w.r.t. the fine / coarse colour channel (also stored to g_rgb /
// g_rgb0 when those are given)
template <int PARTS = TL_GX | TL_GP | TL_SS>
__device__ __forceinline__ float tl_bwd_ray(const TrainLossArgs& a, int ray, int lane, float g_in, float* part,
                                            const TlRayIo& io = TlRayIo()) {
  const float* pred_row = io.pred_row ? io.pred_row : a.pred + (size_t)ray * a.P;
  float* g_pred_row = a.g_pred ? a.g_pred + (size_t)ray * a.P : nullptr;
  float gx_ret = 0.f;
  const bool hm = a.mask != nullptr;
  const float m = hm ? a.mask[ray] : 1.f;
  float g = g_in;
  if (a.out_scale != 1.0f) g = g * a.out_scale;
  if ((PARTS & TL_GX) && lane < 6) {               // d mse / d x = 2 (x - y) mask / (3 N)
    const int c = lane % 3;
    const float x = (lane < 3 && io.have_rgb) ? (c == 0 ? io.r : c == 1 ? io.g : io.b)
                                              : (lane < 3 ? a.rgb : a.rgb0)[ray * 3 + c];
    const float scale = 2.0f * g / (float)(a.N * 3);
    float gx = (x - a.target[ray * 3 + c]) * scale;
    if (hm && a.mse_masked) gx = gx * m;
    float* gdst = lane < 3 ? a.g_rgb : a.g_rgb0;
    if (gdst && io.store_gx) gdst[ray * 3 + c] = gx;
    gx_ret = gx;
  }
  if (!(PARTS & (TL_GP | TL_SS))) return gx_ret;
  float gsc = 0.f, gsh = 0.f;
  if (a.carve_on) {
    const int im = tl_image(a);
    const float sc = tl_row(a.scales, im), sh = tl_row(a.shifts, im);
    const float gl = g * a.carve_weight;
    const float scale = gl / ((float)a.N * (float)a.P);
    for (int k0 = 0; k0 < a.K; k0 += 64) {
      if (!(PARTS & TL_SS) && k0 > 0) break;        // the g_pred rows are complete after the first pass
      const int kl = k0 + lane;
      const float hraw = kl < a.K ? a.hyp[(size_t)kl * a.N + ray] : 0.f;
      float hreg = hraw * sc;
      hreg = hreg + sh;
      float ghk = 0.f;                              // lane l: gradient w.r.t. target_h[k0 + l]
      for (int s0 = 0; s0 < a.P; s0 += 64) {
        const int s = s0 + lane;
        const float p = s < a.P ? pred_row[s] : 0.f;
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
          if ((PARTS & TL_GP) && k0 == 0) {
            if (g_pred_row) g_pred_row[s] = gp;
            if (io.g_pred_row) io.g_pred_row[s] = gp;
          }
        } else {
          kbest = -1;
        }
        if (PARTS & TL_SS) {
          for (int l = 0; l < 64; ++l) {            // scatter -gp into the winning hypothesis
            const int kb = __builtin_amdgcn_readlane(kbest, l);
            const float gg = tl_bcast(gp, l);
            if (kl == kb) ghk -= gg;
          }
        }
      }
      if (PARTS & TL_SS) {
        // d target_h / d scale = hyp_raw, d / d shift = 1
        gsc += (float)tl_wave_sum_d((double)(ghk * hraw));
        gsh += (float)tl_wave_sum_d((double)ghk);
      }
    }
  } else if (PARTS & TL_GP) {
    for (int s = lane; s < a.P; s += 64) {
      if (g_pred_row) g_pred_row[s] = 0.f;
      if (io.g_pred_row) io.g_pred_row[s] = 0.f;
    }
  }
  if ((PARTS & TL_SS) && lane == 0) {
    f32x4 o = {gsc, gsh, 0.f, 0.f};
    reinterpret_cast<f32x4*>(part)[ray] = o;
  }
  return gx_ret;
}


}  // namespace scade

// launches the one-workgroup reduce of scade_train_loss_fb's partials (train_loss.hip)
// gmax_ray [2][N] (nullable): per-ray maxima of the two output gradients the caller's kernel wrote - reduced into the
// 256 loss-scale slots the 16-bit MLP backward reads (slot 0 = the maximum, the rest zero)
int scade_launch_train_loss_fb_reduce(const scade::TrainLossArgs& a, int n_ss, hipStream_t s,
                                      const float* gmax_ray = nullptr, float* gmax_a = nullptr, float* gmax_b = nullptr);

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
#include "train_loss_dev.h"

namespace scade {

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
struct TlGmaxArgs { const float* ray; float* slots[2]; };
__global__ __launch_bounds__(1024) void train_loss_fb_reduce_kernel(TrainLossArgs a, int n_ss, TlGmaxArgs gm) {
  __shared__ double red[5][16];
  if (gm.ray) {       // the loss-scale maxima of the step's two output gradients ride in this launch (no lp_gmax launch)
    __shared__ float gred[2][16];
    float m0 = 0.f, m1 = 0.f;
    for (int i = threadIdx.x; i < a.N; i += 1024) { m0 = fmaxf(m0, gm.ray[i]); m1 = fmaxf(m1, gm.ray[a.N + i]); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m0 = fmaxf(m0, __shfl_xor(m0, o, 64)); m1 = fmaxf(m1, __shfl_xor(m1, o, 64)); }
    if ((threadIdx.x & 63) == 0) { gred[0][threadIdx.x >> 6] = m0; gred[1][threadIdx.x >> 6] = m1; }
    __syncthreads();
    if (threadIdx.x < 512) {
      const int net = threadIdx.x >> 8, slot = threadIdx.x & 255;
      float m = 0.f;
      if (slot == 0)
        for (int w = 0; w < 16; ++w) m = fmaxf(m, gred[net][w]);
      gm.slots[net][slot] = m;
    }
  }
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

int scade_launch_train_loss_fb_reduce(const scade::TrainLossArgs& a, int n_ss, hipStream_t s, const float* gmax_ray,
                                      float* gmax_a, float* gmax_b) {
  const TlGmaxArgs gm{gmax_ray, {gmax_a, gmax_b}};
  hipLaunchKernelGGL(train_loss_fb_reduce_kernel, dim3(1), dim3(1024), 0, s, a, n_ss, gm);
  return scade_check_launch("scade_train_loss_fb");
}

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
  return scade_launch_train_loss_fb_reduce(a, n_ss, s);
}

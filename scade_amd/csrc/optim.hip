// Fused multi-tensor Adam over one flat fp32 buffer (both NeRFs + depth scale/shift live in
// one allocation, scade_amd/parallel.py FlatParams).  Same update as torch.optim.Adam with
// default flags (run_scade_scannet.py:469, :888): no weight decay, no amsgrad.
#include "common.h"

namespace scade {
__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                 float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                 float beta1, float beta2, float eps, float bc1, float bc2_sqrt,
                                 float grad_scale) {
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    const float mi = m[i] * beta1 + gi * (1.0f - beta1);          // exp_avg.lerp_(grad, 1-beta1)
    const float vi = v[i] * beta2 + (gi * gi) * (1.0f - beta2);   // exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}
// Device-resident optimizer state (float[16]) for graph-captured train steps: nothing about the step
// is a launch argument, so one captured graph stays valid for every iteration.
//   [0] t (steps taken)  [1] lr0  [2] decay_rate  [3] decay_step  [4] beta1  [5] beta2  [6] eps
//   [7] grad_scale  [8] lr of this step  [9] 1 - beta1^t  [10] sqrt(1 - beta2^t)
//   [11] iteration offset (resumed runs whose optimizer state was not restored)
__global__ void adam_tick_kernel(float* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float t = st[0] + 1.0f;
    st[0] = t;
    // staircase decay on the reference's loop index i, which counts from 1 (run_scade_scannet.py:899-900,
    // :988; train_utils/hyperparameter_update.py:8-13): i = t + offset
    const double it = (double)t + (double)st[11];
    const double k = st[3] > 0.f ? floor(it / (double)st[3]) : 0.0;
    st[8] = (float)((double)st[1] * pow((double)st[2], k));
    st[9] = (float)(1.0 - pow((double)st[4], (double)t));
    st[10] = (float)sqrt(1.0 - pow((double)st[5], (double)t));
  }
}
__global__ void adam_step_dev_kernel(float* __restrict__ p, const float* __restrict__ g,
                                     float* __restrict__ m, float* __restrict__ v, long n,
                                     const float* __restrict__ st) {
  const float beta1 = st[4], beta2 = st[5], eps = st[6], grad_scale = st[7];
  const float step_size = st[8] / st[9], bc2_sqrt = st[10];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i] * grad_scale;
    const float mi = m[i] * beta1 + gi * (1.0f - beta1);
    const float vi = v[i] * beta2 + (gi * gi) * (1.0f - beta2);
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}
}  // namespace scade

extern "C" int scade_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                   long n, float* state, void* stream) {
  SCADE_REQUIRE(params && grads && exp_avg && exp_avg_sq && state, -1, "scade_adam_step_dev: null pointer");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(scade::adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state);
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(scade::adam_step_dev_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params,
                     grads, exp_avg, exp_avg_sq, n, state);
  return scade_check_launch("scade_adam_step_dev");
}

extern "C" int scade_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               long n, float lr, float beta1, float beta2, float eps, int step,
                               float grad_scale, void* stream) {
  SCADE_REQUIRE(params && grads && exp_avg && exp_avg_sq, -1, "scade_adam_step: null pointer");
  SCADE_REQUIRE(step >= 1, -2, "scade_adam_step: step counts from 1");
  if (n <= 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, step);
  const double bc2 = 1.0 - pow((double)beta2, step);
  const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
  hipLaunchKernelGGL(scade::adam_step_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params,
                     grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, (float)bc1,
                     (float)sqrt(bc2), grad_scale);
  return scade_check_launch("scade_adam_step");
}

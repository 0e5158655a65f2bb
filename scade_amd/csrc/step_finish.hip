// The optimizer step of a train step with the weight gradient's last stage inside it (scade_step_finish): [sum of
// the weight gradient's partial rows ->] Adam on the networks' segment and on the depth scale / shift segment
// (optimizer.step() / optimizer_ss.step() of the reference's loop, run_scade_scannet.py:993-997).
//
// Until round 5 the reduce (9-11 us) and adam_step2 (6.5 us) were two launches over the same 1.18 M floats; the
// reduce is element-wise (a thread sums the partial rows of its four gradient elements - mlp_reduce.h, the
// stand-alone reduce kernels' own function and summation order), so it rides in the Adam launch of every step whose
// gradient is not exchanged between ranks.  Same arithmetic, same order, same bits (tests/test_gpu_train.py).
//
// Measured and NOT kept (round 6, profiles/r06_step_finish_grid_barrier.txt): the next step's weight packs as a
// second phase of the same launch behind a grid-wide barrier (2 workgroups per CU, a monotonic arrival counter in
// device memory).  The packs gather updated parameters written by other workgroups on other XCDs, whose L2s are not
// coherent with each other: every arrival is an agent-scope release (an L2 write-back) and a contended device-scope
// atomic, every departure an L2 invalidate - the launch took 120-200 us where the three separate launches take 23
// (1024-ray bf16-s8 step 0.91 -> 1.11 ms, 128-ray 0.244 -> 0.365).  A kernel boundary IS this machine's cheap
// grid barrier.  mlp_pack_step stays its own launch at the front of the next step (re-packing in place).
#include "mlp_reduce.h"

namespace scade {

struct FinishSeg {
  float* p; float* g; float* m; float* v;
  long n;
  float lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale;
  const float* st;       // device-resident scalars (scade_adam_step_dev's layout), or null
};
struct StepFinishArgs {
  FinishSeg s[2];                          // [0]: the networks (n_nets x N_PARAM_FLOATS), [1]: scales / shifts
  ReduceDesc red;                          // partial[i] != null: network i's gradient is summed here first
};

__device__ __forceinline__ void adam4(const FinishSeg& s, float lr_over_bc1, float beta1, float beta2, float eps,
                                      float bc2_sqrt, float gs, long i4, f32x4 g) {
  f32x4 m = reinterpret_cast<const f32x4*>(s.m)[i4], v = reinterpret_cast<const f32x4*>(s.v)[i4];
  f32x4 p = reinterpret_cast<const f32x4*>(s.p)[i4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {                       // (adam_step2_kernel's expression, element by element)
    const float gi = g[q] * gs;
    const float mi = m[q] * beta1 + gi * (1.0f - beta1);
    const float vi = v[q] * beta2 + (gi * gi) * (1.0f - beta2);
    m[q] = mi;
    v[q] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[q] = p[q] - lr_over_bc1 * (mi / denom);
  }
  reinterpret_cast<f32x4*>(s.m)[i4] = m;
  reinterpret_cast<f32x4*>(s.v)[i4] = v;
  reinterpret_cast<f32x4*>(s.p)[i4] = p;
}

__global__ __launch_bounds__(256) void step_finish_kernel(StepFinishArgs a) {
  {
    const FinishSeg& s = a.s[0];
    float lr = s.lr, beta1 = s.beta1, beta2 = s.beta2, eps = s.eps, bc1 = s.bc1, bc2_sqrt = s.bc2_sqrt, gs = s.grad_scale;
    if (s.st) { beta1 = s.st[4]; beta2 = s.st[5]; eps = s.st[6]; gs = s.st[7]; lr = s.st[8]; bc1 = s.st[9]; bc2_sqrt = s.st[10]; }
    const float step_size = lr / bc1;
    constexpr int N4 = N_PARAM_FLOATS / 4;
    const long n4 = s.n / 4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const int net = (int)(i / N4), x = (int)(i - (long)net * N4);
      f32x4 g;
      if (net < 2 && a.red.partial[net]) {
        g = reduce_rows4(a.red.partial[net], a.red.nseg[net], a.red.uniform[net], x);
        reinterpret_cast<f32x4*>(s.g)[i] = g;                    // the bucket keeps the step's gradient
      } else {
        g = reinterpret_cast<const f32x4*>(s.g)[i];
      }
      adam4(s, step_size, beta1, beta2, eps, bc2_sqrt, gs, i, g);
    }
  }
  if (blockIdx.x == gridDim.x - 1 && a.s[1].n > 0) {             // the scale / shift rows: a handful of floats
    const FinishSeg& s = a.s[1];
    float lr = s.lr, beta1 = s.beta1, beta2 = s.beta2, eps = s.eps, bc1 = s.bc1, bc2_sqrt = s.bc2_sqrt, gs = s.grad_scale;
    if (s.st) { beta1 = s.st[4]; beta2 = s.st[5]; eps = s.st[6]; gs = s.st[7]; lr = s.st[8]; bc1 = s.st[9]; bc2_sqrt = s.st[10]; }
    const float step_size = lr / bc1;
    for (long i = threadIdx.x; i < s.n; i += 256) {
      const float gi = s.g[i] * gs;
      const float mi = s.m[i] * beta1 + gi * (1.0f - beta1);
      const float vi = s.v[i] * beta2 + (gi * gi) * (1.0f - beta2);
      s.m[i] = mi;
      s.v[i] = vi;
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      s.p[i] = s.p[i] - step_size * (mi / denom);
    }
  }
}

}  // namespace scade

using namespace scade;

// Both optimizers of a train step in one launch, as scade_adam_step2 (arrays of two entries; segment 0 = the networks:
// n_nets x 589,700 consecutive floats in parameter order, segment 1 = depth scales / shifts, n[1] = 0: frozen;
// state[i] != NULL: that segment's scalars live on the device and were ALREADY advanced for this step - the launch in
// front of the step does that: scade_stage_inputs / scade_gather_batch) - with the sum of the weight gradient's partial
// rows in front: reduce_desc (nullable) = what scade_mlp_bwd*_deferred left; the networks' gradients are then summed
// from the partial rows here and written to grads[0] on the way.
extern "C" int scade_step_finish(float* const* params, float* const* grads, float* const* exp_avg,
                                 float* const* exp_avg_sq, const long* n, const float* lr, const float* beta1,
                                 const float* beta2, const float* eps, const int* step, const float* grad_scale,
                                 float* const* state, const void* reduce_desc, int n_nets, void* stream) {
  SCADE_REQUIRE(params && grads && exp_avg && exp_avg_sq && n, -1, "scade_step_finish: null pointer");
  SCADE_REQUIRE(n_nets == 1 || n_nets == 2, -2, "scade_step_finish: one or two networks");
  SCADE_REQUIRE(n[0] == (long)n_nets * N_PARAM_FLOATS, -2,
                "scade_step_finish: segment 0 must be the %d networks' %ld floats", n_nets, (long)n_nets * N_PARAM_FLOATS);
  StepFinishArgs a{};
  for (int i = 0; i < 2; ++i) {
    FinishSeg& s = a.s[i];
    s.n = n[i] > 0 ? n[i] : 0;
    if (s.n == 0) continue;
    SCADE_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], -1, "scade_step_finish: null pointer in segment %d", i);
    s.p = params[i]; s.g = grads[i]; s.m = exp_avg[i]; s.v = exp_avg_sq[i];
    s.st = state ? state[i] : nullptr;
    if (s.st) continue;
    SCADE_REQUIRE(lr && beta1 && beta2 && eps && step && grad_scale, -1, "scade_step_finish: host scalars missing");
    SCADE_REQUIRE(step[i] >= 1, -2, "scade_step_finish: step counts from 1");
    s.lr = lr[i]; s.beta1 = beta1[i]; s.beta2 = beta2[i]; s.eps = eps[i]; s.grad_scale = grad_scale[i];
    s.bc1 = (float)(1.0 - pow((double)beta1[i], step[i]));
    s.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2[i], step[i]));
  }
  SCADE_REQUIRE(((unsigned long long)a.s[0].p & 15) == 0 && ((unsigned long long)a.s[0].g & 15) == 0 &&
                ((unsigned long long)a.s[0].m & 15) == 0 && ((unsigned long long)a.s[0].v & 15) == 0, -2,
                "scade_step_finish: segment 0 must be 16-byte aligned");
  if (reduce_desc) a.red = *reinterpret_cast<const ReduceDesc*>(reduce_desc);
  const long n4 = a.s[0].n / 4;
  const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
  hipLaunchKernelGGL(step_finish_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_step_finish");
}

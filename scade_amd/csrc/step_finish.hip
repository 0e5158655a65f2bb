// The end of a train step as ONE launch (scade_step_finish): [sum of the weight gradient's partial rows] -> Adam on
// the networks' segment and on the depth scale / shift segment -> the MFMA weight blobs of the NEXT step's kernels
// rebuilt from the updated fp32 master parameters (optimizer.step() / optimizer_ss.step() of the reference's loop,
// run_scade_scannet.py:993-997, with the bookkeeping the fused kernels need around it).
//
// Until round 5 these were three launches per step (wgrad reduce 9-11 us, adam_step2 6.5 us, mlp_pack_step 7 us at
// the front of the next step): 23 us of a 243-us step at the 128 rays per GPU of a strongly scaled batch
// (BASELINE.json configs[3]) - each of them a pass over the same 1.18 M floats, each paying a kernel boundary.
// Here: phase 1 is element-wise (a thread sums the partial rows of its four gradient elements - mlp_reduce.h, the
// stand-alone reduce kernels' own function and summation order -, writes the reduced gradient where the bucket
// keeps it, and applies Adam to them); phase 2 (the packs: every blob element gathers ONE updated parameter) needs
// every update of phase 1, so the two are separated by a grid-wide barrier: the launch is 2 workgroups per CU,
// all resident at once (256 threads, no LDS to speak of), an arrival counter in device memory, monotonic - every
// launch adds gridDim.x to it and waits for the next multiple, so it is never reset (no memset node in a captured
// step: see lp_gmax_kernel in mlp_bwd_lp.hip for why that matters).  Same arithmetic, same order, same bits as
// the three separate launches (tests/test_gpu_train.py).
#include "mlp_pack.h"
#include "mlp_reduce.h"

namespace scade {

struct FinishSeg {
  float* p; float* g; float* m; float* v;
  long n;
  float lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale;
  const float* st;       // device-resident scalars (scade_adam_step_dev's layout), or null
};
struct StepFinishArgs {
  FinishSeg s[2];                          // [0]: both networks (n_nets x N_PARAM_FLOATS), [1]: scales / shifts
  ReduceDesc red;                          // partial[i] != null: network i's gradient is summed here first
  const float* p[2][N_PARAM_TENSORS];      // the networks' parameter tensors (views of s[0].p)
  float* exact[2];                         // fmt 0 / 3: exact forward blob
  void* fwd[2];                            // fmt 1 / 2: 16-bit forward blob; fmt 3: two-plane forward blob
  void* tr[2];                             // transposed blob of the format
  unsigned long long* sync;                // arrival counter of the grid barrier
  int n_nets, fmt, do_pack;
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

template <int FMT>
__device__ __forceinline__ void finish_pack(const StepFinishArgs& a) {
  constexpr int ROWS = FMT == 3 ? 2 * PACK_FWD_ROWS + PACK_T_ROWS : PACK_FWD_ROWS + PACK_T_ROWS;
  const int items = a.n_nets * ROWS * PACK_BLOCKS;
  for (int w = blockIdx.x; w < items; w += gridDim.x) {          // (block-uniform: the rows' own barriers stay legal)
    const int bx = w % PACK_BLOCKS, rw = w / PACK_BLOCKS;
    const int net = rw / ROWS, row = rw % ROWS;
    const float* const* p = a.p[net];
    if (FMT == 3) {
      if (row < PACK_FWD_ROWS) { if (a.exact[net]) pack_fwd_row(p, a.exact[net], row, bx, PACK_BLOCKS); }
      else if (row < 2 * PACK_FWD_ROWS) { if (a.fwd[net]) pack_f16_row(p, a.fwd[net], row - PACK_FWD_ROWS, bx, PACK_BLOCKS); }
      else if (a.tr[net]) pack_t_f16_row(p, a.tr[net], row - 2 * PACK_FWD_ROWS, bx, PACK_BLOCKS);
    } else if (row < PACK_FWD_ROWS) {
      if (FMT == 0) { if (a.exact[net]) pack_fwd_row(p, a.exact[net], row, bx, PACK_BLOCKS); }
      else if (a.fwd[net]) pack_lp_row<FMT == 1>(p, a.fwd[net], row, bx, PACK_BLOCKS);
    } else if (a.tr[net]) {
      if (FMT == 0) pack_t_row(p, reinterpret_cast<float*>(a.tr[net]), row - PACK_FWD_ROWS, bx, PACK_BLOCKS);
      else pack_t_lp_row<FMT == 1>(p, a.tr[net], row - PACK_FWD_ROWS, bx, PACK_BLOCKS);
    }
    __syncthreads();                                             // (a row's LDS census is reused by the next item)
  }
}

template <int FMT>
__global__ __launch_bounds__(256) void step_finish_kernel(StepFinishArgs a) {
  // ---- phase 1: [reduce] + Adam ------------------------------------------------------------------------------
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
  if (!a.do_pack) return;
  // ---- grid barrier: every parameter update above is visible to every workgroup below -------------------------
  __threadfence();                                               // this thread's stores have reached L2
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long G = gridDim.x;
    const unsigned long long old = __hip_atomic_fetch_add(a.sync, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (old / G + 1ull) * G;
    while (__hip_atomic_load(a.sync, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");             // no line of the old parameters survives in this CU's L1
  // ---- phase 2: the next step's weight blobs -------------------------------------------------------------------
  finish_pack<FMT>(a);
}

}  // namespace scade

using namespace scade;

// One launch for the end of a train step.  Arrays of two entries as scade_adam_step2 (segment 0 = the networks:
// n_nets x 589,700 consecutive floats in parameter order, segment 1 = depth scales / shifts, n[1] = 0: frozen);
// state[i] != NULL: that segment's scalars live on the device and were ALREADY advanced for this step (the launch in
// front of the step does: scade_stage_inputs / scade_gather_batch).  reduce_desc (nullable): what
// scade_mlp_bwd*_deferred left - the networks' gradients are then summed from the partial rows in this launch and
// written to grads[0].  pack_format: -1 = no pack (Adam [+ reduce] only), 0 = exact fp32 blobs (packed_exact +
// packed_t as scade_mlp_pack / scade_mlp_pack_t), 1 = bf16, 2 = fp16 (packed_fwd + packed_t as scade_mlp_pack_lp /
// scade_mlp_pack_t_lp), 3 = split precision (packed_exact + packed_fwd + packed_t as scade_mlp_pack_step_f16x3);
// blob entries may be NULL (skipped).  net_params: n_nets x 24 tensor pointers (views of params[0]).  sync: 8 bytes of
// device memory, zero when first used, owned by the caller and used by these launches only.
extern "C" int scade_step_finish(float* const* params, float* const* grads, float* const* exp_avg,
                                 float* const* exp_avg_sq, const long* n, const float* lr, const float* beta1,
                                 const float* beta2, const float* eps, const int* step, const float* grad_scale,
                                 float* const* state, const void* reduce_desc, int n_nets,
                                 const float* const* net_params, int pack_format, float* const* packed_exact,
                                 void* const* packed_fwd, void* const* packed_t, unsigned long long* sync,
                                 void* stream) {
  SCADE_REQUIRE(params && grads && exp_avg && exp_avg_sq && n, -1, "scade_step_finish: null pointer");
  SCADE_REQUIRE(n_nets == 1 || n_nets == 2, -2, "scade_step_finish: one or two networks");
  SCADE_REQUIRE(pack_format >= -1 && pack_format <= 3, -2, "scade_step_finish: pack_format -1 (none) .. 3");
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
  a.n_nets = n_nets;
  a.do_pack = pack_format >= 0;
  a.fmt = pack_format;
  if (a.do_pack) {
    SCADE_REQUIRE(net_params && packed_t && sync, -1, "scade_step_finish: the pack needs net_params, packed_t and sync");
    SCADE_REQUIRE(pack_format == 0 || pack_format == 3 ? packed_exact != nullptr : packed_fwd != nullptr, -1,
                  "scade_step_finish: forward blobs of the format missing");
    SCADE_REQUIRE(pack_format != 3 || packed_fwd, -1, "scade_step_finish: the split-precision pack needs packed_fwd");
    for (int k = 0; k < n_nets; ++k) {
      for (int i = 0; i < N_PARAM_TENSORS; ++i) {
        SCADE_REQUIRE(net_params[k * N_PARAM_TENSORS + i], -1, "scade_step_finish: net_params[%d][%d] is null", k, i);
        a.p[k][i] = net_params[k * N_PARAM_TENSORS + i];
      }
      a.exact[k] = packed_exact ? packed_exact[k] : nullptr;
      a.fwd[k] = packed_fwd ? packed_fwd[k] : nullptr;
      a.tr[k] = packed_t[k];
    }
    a.sync = sync;
  }
  // two workgroups per CU: all resident at once (the barrier needs that), and the same count for every launch of a
  // process on this device (the counter's arithmetic needs that)
  const int grid = 2 * device_cus();
  hipStream_t s = (hipStream_t)stream;
  switch (pack_format) {
    case 1: hipLaunchKernelGGL(step_finish_kernel<1>, dim3(grid), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(step_finish_kernel<2>, dim3(grid), dim3(256), 0, s, a); break;
    case 3: hipLaunchKernelGGL(step_finish_kernel<3>, dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(step_finish_kernel<0>, dim3(grid), dim3(256), 0, s, a); break;
  }
  return scade_check_launch("scade_step_finish");
}

// Fused multi-tensor Adam over one flat fp32 buffer (both NeRFs + depth scale/shift live in
// one allocation, scade_amd/parallel.py FlatParams).  Same update as torch.optim.Adam with
// default flags (run_scade_scannet.py:469, :888): no weight decay, no amsgrad.
#include "common.h"
#include "ray_points_dev.h"
#include "mlp_pack.h"

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
// Two optimizers in one launch (the networks' Adam and the depth scale / shift Adam of a train step, which differ
// in learning rate and step count): blocks [0, blocks0) update segment 0, the rest segment 1.  Scalars come
// from the launch arguments, or - st != null - from the segment's device-resident state (adam_tick2_kernel).
struct AdamSeg {
  float* p; const float* g; float* m; float* v;
  long n;
  float lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale;
  float* st;
};
struct Adam2Args {
  AdamSeg s[2];
  int blocks0;
};
__global__ void adam_step2_kernel(Adam2Args a) {
  const bool second = (int)blockIdx.x >= a.blocks0;
  const AdamSeg& s = second ? a.s[1] : a.s[0];
  const long b = (long)blockIdx.x - (second ? a.blocks0 : 0);
  const long nb = second ? (long)gridDim.x - a.blocks0 : a.blocks0;
  float lr = s.lr, beta1 = s.beta1, beta2 = s.beta2, eps = s.eps, bc1 = s.bc1, bc2_sqrt = s.bc2_sqrt, gs = s.grad_scale;
  if (s.st) { beta1 = s.st[4]; beta2 = s.st[5]; eps = s.st[6]; gs = s.st[7]; lr = s.st[8]; bc1 = s.st[9]; bc2_sqrt = s.st[10]; }
  const float step_size = lr / bc1;
  for (long i = b * 256 + threadIdx.x; i < s.n; i += nb * 256) {
    const float gi = s.g[i] * gs;
    const float mi = s.m[i] * beta1 + gi * (1.0f - beta1);
    const float vi = s.v[i] * beta2 + (gi * gi) * (1.0f - beta2);
    s.m[i] = mi;
    s.v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    s.p[i] = s.p[i] - step_size * (mi / denom);
  }
}
__global__ void adam_tick2_kernel(float* st0, float* st1) {
  if (threadIdx.x == 0 && st0) adam_tick(st0);
  if (threadIdx.x == 64 && st1) adam_tick(st1);
}
// ---- batch staging: up to eight device-to-device copies (+ one 8-byte scalar) in ONE launch -------------------
// A graph-captured train step reads its inputs from static buffers; refreshing them with tensor.copy_() is one
// launch per tensor (rays, target, depth hypotheses, image index: 4 x 5 us in front of a 128-ray step of 0.3 ms).
constexpr int STAGE_MAX = 8;
struct StageArgs {
  const unsigned char* src[STAGE_MAX];
  unsigned char* dst[STAGE_MAX];
  long bytes[STAGE_MAX];
  int first_block[STAGE_MAX + 1];
  int n;
  long long* scalar_dst;
  long long scalar;
  float* tick[2];        // device-resident optimizer scalars to advance (scade_adam_step_dev's layout), or null
  // scade_stage_inputs_points: workgroups [points_block0, gridDim.x) are four waves = four rays of ray_points over
  // the SOURCE ray rows (pts.rays): the step's first per-ray kernel rides in the launch that stages its inputs
  RayPointsArgs pts;
  int points_block0;
  // ... and workgroups [pack_block0, points_block0 or gridDim.x) the step's weight packs (mlp_pack.h pack_item)
  PackItemsArgs pack;
  int pack_block0;
};
__global__ void stage_inputs_kernel(StageArgs a) {
  if (a.pack_block0 >= 0 && (int)blockIdx.x >= a.pack_block0 &&
      (a.points_block0 < 0 || (int)blockIdx.x < a.points_block0)) {
    pack_item(a.pack, (int)blockIdx.x - a.pack_block0);
    return;
  }
  if (a.points_block0 >= 0 && (int)blockIdx.x >= a.points_block0) {
    const int ray = ((int)blockIdx.x - a.points_block0) * RAYS_PER_WG + (int)(threadIdx.x >> 6);
    if (ray >= a.pts.N) return;
    const float* r = a.pts.rays + (size_t)ray * a.pts.ray_stride;
    ray_points_ray(a.pts, ray, lane_id(), r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    return;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.scalar_dst) *a.scalar_dst = a.scalar;
  if (blockIdx.x == 0 && threadIdx.x == 64 && a.tick[0]) adam_tick(a.tick[0]);
  if (blockIdx.x == 0 && threadIdx.x == 128 && a.tick[1]) adam_tick(a.tick[1]);
  int g = 0;
  while (g + 1 < a.n && (int)blockIdx.x >= a.first_block[g + 1]) ++g;
  if (g >= a.n) return;
  const unsigned char* __restrict__ src = a.src[g];
  unsigned char* __restrict__ dst = a.dst[g];
  const long bytes = a.bytes[g];
  const long off = ((long)((int)blockIdx.x - a.first_block[g]) * 256 + threadIdx.x) * 16;
  if (off >= bytes) return;
  const bool vec = ((reinterpret_cast<unsigned long long>(src) | reinterpret_cast<unsigned long long>(dst)) & 15ull) == 0;
  if (vec && off + 16 <= bytes) {
    *reinterpret_cast<f32x4*>(dst + off) = *reinterpret_cast<const f32x4*>(src + off);
  } else {
    for (long o = off; o < bytes && o < off + 16; o += 4)      // (sizes and addresses are multiples of 4: checked on the host)
      *reinterpret_cast<unsigned*>(dst + o) = *reinterpret_cast<const unsigned*>(src + o);
  }
}
}  // namespace scade

static int stage_inputs_impl(const void* const* src, void* const* dst, const long* bytes, int n,
                             long long* scalar_dst, long long scalar, float* const* tick_states,
                             const scade::RayPointsArgs* pts, const scade::PackItemsArgs* pack, void* stream) {
  SCADE_REQUIRE(n >= 0 && n <= scade::STAGE_MAX, -2, "scade_stage_inputs: 0..%d copies per launch", scade::STAGE_MAX);
  SCADE_REQUIRE(n == 0 || (src && dst && bytes), -1, "scade_stage_inputs: null pointer");
  scade::StageArgs a{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    SCADE_REQUIRE(src[i] && dst[i] && bytes[i] >= 0, -1, "scade_stage_inputs: null pointer / negative size in copy %d", i);
    SCADE_REQUIRE(bytes[i] % 4 == 0 && ((unsigned long long)src[i] & 3) == 0 && ((unsigned long long)dst[i] & 3) == 0, -2,
                  "scade_stage_inputs: copy %d is not a whole number of aligned 4-byte words", i);
    a.src[i] = static_cast<const unsigned char*>(src[i]);
    a.dst[i] = static_cast<unsigned char*>(dst[i]);
    a.bytes[i] = bytes[i];
    a.first_block[i] = blocks;
    const long nb = (bytes[i] + 256 * 16 - 1) / (256 * 16);
    SCADE_REQUIRE(blocks + nb < (1L << 30), -2, "scade_stage_inputs: too large");
    blocks += (int)nb;
  }
  a.first_block[n] = blocks;
  a.n = n;
  a.scalar_dst = scalar_dst;
  a.scalar = scalar;
  if (tick_states) { a.tick[0] = tick_states[0]; a.tick[1] = tick_states[1]; }
  if (blocks == 0 && !scalar_dst && !a.tick[0] && !a.tick[1] && !pts && !(pack && pack->n_nets > 0)) return 0;
  if (blocks == 0) blocks = 1;
  a.points_block0 = -1;
  a.pack_block0 = -1;
  if (pack && pack->n_nets > 0) {
    a.pack = *pack;
    a.pack_block0 = blocks;
    blocks += scade::pack_item_count(pack->n_nets, pack->fmt);
  }
  if (pts && pts->N > 0) {
    a.pts = *pts;
    a.points_block0 = blocks;
    blocks += (pts->N + scade::RAYS_PER_WG - 1) / scade::RAYS_PER_WG;
  }
  hipLaunchKernelGGL(scade::stage_inputs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return scade_check_launch("scade_stage_inputs");
}

extern "C" int scade_stage_inputs(const void* const* src, void* const* dst, const long* bytes, int n,
                                  long long* scalar_dst, long long scalar, float* const* tick_states, void* stream) {
  return stage_inputs_impl(src, dst, bytes, n, scalar_dst, scalar, tick_states, nullptr, nullptr, stream);
}

// scade_stage_inputs + scade_ray_points_draw (host step index) of the ray rows ``rays`` [N, ray_stride] - the rows being
// staged, read at their source - as ONE launch; outputs as scade_ray_points_draw's.  Same bits as the two launches.
extern "C" int scade_stage_inputs_points(const void* const* src, void* const* dst, const long* bytes, int n,
                                         long long* scalar_dst, long long scalar, float* const* tick_states,
                                         const float* rays, int ray_stride, const float* t_vals, int N, int S,
                                         int lindisp, unsigned long long seed, unsigned long long step, int Si,
                                         float* z_vals, float* pts, float* u_a, float* u_b, int pack_format, int n_nets,
                                         const float* const* net_params, float* const* packed_exact,
                                         void* const* packed_fwd, void* const* packed_t, void* stream) {
  scade::PackItemsArgs pk;
  SCADE_REQUIRE(scade::pack_items_fill(pk, pack_format, n_nets, net_params, packed_exact, packed_fwd, packed_t), -1,
                "scade_stage_inputs_points: pack arguments (format 0..3, one or two networks, 24 parameter pointers each)");
  if (N <= 0) return stage_inputs_impl(src, dst, bytes, n, scalar_dst, scalar, tick_states, nullptr, &pk, stream);
  SCADE_REQUIRE(N <= 0 || (rays && t_vals && z_vals), -1, "scade_stage_inputs_points: null pointer");
  SCADE_REQUIRE(ray_stride >= 8 && S >= 1 && Si >= 0, -2, "scade_stage_inputs_points: ray_stride >= 8, S >= 1, Si >= 0 required");
  SCADE_REQUIRE(Si > 0 || (!u_a && !u_b), -2, "scade_stage_inputs_points: sampler draws requested with Si = 0");
  scade::RayPointsArgs p{};
  p.rays = rays; p.t_vals = t_vals; p.z_vals = z_vals; p.pts = pts; p.N = N; p.S = S; p.ray_stride = ray_stride;
  p.lindisp = lindisp; p.draw = 1; p.seed_lo = (unsigned)seed; p.seed_hi = (unsigned)(seed >> 32); p.step = step;
  p.u_a = u_a; p.u_b = u_b; p.Si = Si;
  return stage_inputs_impl(src, dst, bytes, n, scalar_dst, scalar, tick_states, &p, &pk, stream);
}

static int adam_blocks(long n) { return (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048); }

// Both optimizers of a train step in ONE launch.  Arrays of two entries; n[1] = 0 skips the second segment
// (scale / shift frozen, run_scade_scannet.py:996).  state[i] != NULL: that segment's scalars live on the device
// (scade_adam_step_dev's layout) and are advanced by one tick launch for both before the update.
extern "C" int scade_adam_step2(float* const* params, const float* const* grads, float* const* exp_avg,
                                float* const* exp_avg_sq, const long* n, const float* lr, const float* beta1,
                                const float* beta2, const float* eps, const int* step, const float* grad_scale,
                                float* const* state, int ticked, void* stream) {
  SCADE_REQUIRE(params && grads && exp_avg && exp_avg_sq && n, -1, "scade_adam_step2: null pointer");
  scade::Adam2Args a{};
  bool any_state = false;
  for (int i = 0; i < 2; ++i) {
    scade::AdamSeg& s = a.s[i];
    s.n = n[i];
    if (n[i] <= 0) { s.n = 0; continue; }
    SCADE_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i], -1, "scade_adam_step2: null pointer in segment %d", i);
    s.p = params[i]; s.g = grads[i]; s.m = exp_avg[i]; s.v = exp_avg_sq[i];
    s.st = state ? state[i] : nullptr;
    if (s.st) { any_state = true; continue; }
    SCADE_REQUIRE(lr && beta1 && beta2 && eps && step && grad_scale, -1, "scade_adam_step2: host scalars missing");
    SCADE_REQUIRE(step[i] >= 1, -2, "scade_adam_step2: step counts from 1");
    s.lr = lr[i]; s.beta1 = beta1[i]; s.beta2 = beta2[i]; s.eps = eps[i]; s.grad_scale = grad_scale[i];
    s.bc1 = (float)(1.0 - pow((double)beta1[i], step[i]));
    s.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2[i], step[i]));
  }
  if (a.s[0].n <= 0 && a.s[1].n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (any_state && !ticked)
    hipLaunchKernelGGL(scade::adam_tick2_kernel, dim3(1), dim3(128), 0, st, a.s[0].n > 0 ? a.s[0].st : nullptr,
                       a.s[1].n > 0 ? a.s[1].st : nullptr);
  a.blocks0 = a.s[0].n > 0 ? adam_blocks(a.s[0].n) : 0;
  const int blocks1 = a.s[1].n > 0 ? adam_blocks(a.s[1].n) : 0;
  hipLaunchKernelGGL(scade::adam_step2_kernel, dim3(a.blocks0 + blocks1), dim3(256), 0, st, a);
  return scade_check_launch("scade_adam_step2");
}

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

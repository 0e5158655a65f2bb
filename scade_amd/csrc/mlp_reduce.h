// The sum of a weight-gradient launch's partial rows as a device function of (descriptor, network, float4 index),
// shared by the stand-alone reduce kernels (wgrad_lp_reduce_kernel, wgrad2_reduce_pair_kernel: one launch behind the
// weight gradient) and by scade_step_finish (step_finish.hip: the same sums INSIDE the optimizer launch of a train
// step whose gradient is not exchanged between ranks - reduce, Adam and the weight packs of the next step as ONE
// launch).  Element x of a network's flat gradient = sum over the rows [0, rows(x)) of partial[row][x]: rows(x) is
// one number per network for the (chunk, job) grids of the exact and split-precision kernels and of the small 16-bit
// launches, and a per-job count for the 16-bit kernel's balanced persistent plan.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace scade {

constexpr int REDUCE_MAX_JOBS = 16;       // == MAX_WGRAD_JOBS (mlp_wgrad.h)

// what a deferred backward hands to scade_step_finish (include/scade_hip.h: scade_reduce_desc, 64 opaque bytes)
struct ReduceDesc {
  const float* partial[2];                       // [rows][N_PARAM_FLOATS] of network 0 / 1 (null: absent)
  int uniform[2];                                // > 0: every element of the network has this many rows
  unsigned char nseg[2][REDUCE_MAX_JOBS];        // else rows per job of the 16-bit balanced plan (lp_param_job)
};
static_assert(sizeof(ReduceDesc) <= 64, "scade_reduce_desc is 64 bytes");

// flat-gradient offsets of the parameter tensors as compile-time constants (param_offsets() of mlp_wgrad.h)
struct LpParamOffsets { int v[N_PARAM_TENSORS + 1]; };
constexpr LpParamOffsets lp_make_offsets() {
  LpParamOffsets o{};
  int at = 0, i = 0;
  for (int l = 0; l < 8; ++l) {
    const int k = l == 0 ? 57 : (l == 5 ? 313 : 256);
    o.v[i++] = at; at += 256 * k;
    o.v[i++] = at; at += 256;
  }
  o.v[i++] = at; at += 128 * 259;
  o.v[i++] = at; at += 128;
  o.v[i++] = at; at += 256 * 256;
  o.v[i++] = at; at += 256;
  o.v[i++] = at; at += 256;
  o.v[i++] = at; at += 1;
  o.v[i++] = at; at += 3 * 128;
  o.v[i++] = at; at += 3;
  o.v[i] = at;
  return o;
}
constexpr LpParamOffsets LP_OFF = lp_make_offsets();
static_assert(LP_OFF.v[N_PARAM_TENSORS] == N_PARAM_FLOATS, "parameter layout");
static_assert(N_PARAM_FLOATS % 4 == 0, "float4 reduce");

// job (index into build_wgrad_lp_jobs' table) that writes flat-gradient element x.  The offsets are immediates
// (an unrolled compare chain): indexed from the kernel arguments every step of the search was a dependent
// memory load per lane, which doubled the reduce kernel's time.
__host__ __device__ inline int lp_param_job(int x) {
  int t = 0;
#pragma unroll
  for (int k = 1; k < N_PARAM_TENSORS; ++k) t += x >= LP_OFF.v[k] ? 1 : 0;
  if (t < 16) {                                   // pts_linears[l]: weight (even t), bias (odd t)
    const int l = t >> 1;
    if (l == 0) return 9;                         // the embedding job of layer 0
    if (t == 10 && (x - LP_OFF.v[10]) % 313 < 57) return 10;   // skip-connection columns: embedding job
    return l - 1;
  }
  if (t == 16) return (x - LP_OFF.v[16]) % 259 < 256 ? 8 : 11;   // views_linears.0: hidden | view-direction columns
  if (t == 17) return 8;
  if (t <= 21) return 7;                          // feature_linear + the alpha head riding on its job
  return 12;                                      // rgb head
}

// elements [4 i, 4 i + 4) of one network's gradient: one float4 per row (four scalar sums where a float4 straddles
// two jobs), four accumulators - the summation order every reduce kernel of the library has used, so the bits of a
// gradient do not depend on which launch sums it
__device__ __forceinline__ f32x4 reduce_rows4(const float* __restrict__ part, const unsigned char* nseg, int uni, int i) {
  constexpr size_t ST = N_PARAM_FLOATS / 4;
  const f32x4* p = reinterpret_cast<const f32x4*>(part) + i;
  int n = uni;
  if (uni <= 0) {
    int job[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) job[q] = lp_param_job(4 * i + q);
    if (job[0] != job[3]) {                        // (jobs own contiguous runs within a tensor row: ends equal = all equal)
      f32x4 res;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* ps = part + 4 * i + q;
        const int nq = nseg[job[q]];
        float t = 0.f;
        for (int c = 0; c < nq; ++c) t += ps[(size_t)c * N_PARAM_FLOATS];
        res[q] = t;
      }
      return res;
    }
    n = nseg[job[0]];
  }
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int c = 0;
  for (; c + 4 <= n; c += 4) {
    s0 += p[(size_t)c * ST]; s1 += p[(size_t)(c + 1) * ST];
    s2 += p[(size_t)(c + 2) * ST]; s3 += p[(size_t)(c + 3) * ST];
  }
  for (; c < n; ++c) s0 += p[(size_t)c * ST];
  return (s0 + s1) + (s2 + s3);
}

}  // namespace scade

// ray_points - the coarse samples of a ray: z_vals (+ stratified jitter) and sample positions
// (run_scade_scannet.py:638-657, perturb_z_vals :564-579) - as a device function of (arguments, ray, lane, ray row),
// shared by ray_points_kernel (ray_ops.hip) and by the launches that OPEN a graph-captured train step
// (scade_stage_inputs_points in optim.hip, scade_gather_batch_points in ray_ops.hip: the step's batch assembly and
// its first per-ray kernel as one launch - round 6).
#pragma once
#include "common.h"

namespace scade {

constexpr int RAYS_PER_WG = 4;

struct RayPointsArgs {
  const float* rays;     // [N, ray_stride]: o(0..2) d(3..5) near(6) far(7)
  const float* t_vals;   // [S] torch.linspace(0,1,S)
  const float* t_rand;   // [N,S] or null
  float* z_vals;         // [N,S]
  float* pts;            // [N,S,3] or null
  int N, S, ray_stride, lindisp;
  // draw mode (scade_ray_points_draw): the step's uniform draws come from a counter-based generator inside
  // this kernel instead of a tensor - the jitter is consumed in registers, the two samplers' draws are written
  // out for the ray tails
  int draw;                  // 0: t_rand as above
  unsigned seed_lo, seed_hi; // Philox key
  unsigned long long step;   // host step index ...
  const float* step_dev;     // ... or the device-resident step count (FusedAdam.state[0], graph-captured steps)
  float* u_a;                // [N,Si] draws of the coarse importance sampler (helpers:346-361), or null
  float* u_b;                // [N,Si] draws of the depth-hypothesis sampler (helpers:395-410), or null
  int Si;
};

// Philox4x32-10 (Salmon et al., SC'11; the generator family torch's device RNG uses): a pure function
// (key, counter) -> 4 x 32 random bits, so every (step, ray, draw) has its value without any state to carry.
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0];
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (unsigned)p1; c[3] = (unsigned)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// draw d of ray `ray` at step `step`: block d / 4 of the ray's stream, element d % 4; 24 random bits -> [0,1)
// (the construction of torch.rand for float32)
__device__ __forceinline__ void draw_block(unsigned seed_lo, unsigned seed_hi, unsigned long long step, int ray,
                                           int block, float (&u)[4]) {
  unsigned c[4] = {(unsigned)block, (unsigned)ray, (unsigned)step, (unsigned)(step >> 32)};
  philox4x32_10(c, seed_lo, seed_hi);
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * 5.9604644775390625e-8f;     // 2^-24
}

// one wave: the samples of ray `ray` whose row is (o, d, near, far)
__device__ __forceinline__ void ray_points_ray(const RayPointsArgs& a, int ray, int lane, float ox, float oy, float oz,
                                               float dx, float dy, float dz, float near, float far) {
  const int S = a.S;
  auto zlin = [&](int i) {
    const float t = a.t_vals[i];
    const float om = 1.0f - t;
    if (!a.lindisp) return near * om + far * t;                      // :642
    return 1.0f / (1.0f / near * om + 1.0f / far * t);               // :645
  };
  // draw mode: the ray's stream is [jitter: S draws | sampler a: Si | sampler b: Si], every block of four
  // padded up separately so that the arrays start on a block boundary
  const unsigned long long step = a.draw ? (a.step_dev ? (unsigned long long)a.step_dev[0] : a.step) : 0ull;
  const int jb = (S + 3) >> 2, sb = (a.Si + 3) >> 2;
  if (a.draw) {
    for (int which = 0; which < 2; ++which) {
      float* dst = which ? a.u_b : a.u_a;
      if (!dst) continue;
      for (int b = lane; b < sb; b += 64) {
        float u[4];
        draw_block(a.seed_lo, a.seed_hi, step, ray, jb + which * sb + b, u);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * b + j < a.Si) dst[(size_t)ray * a.Si + 4 * b + j] = u[j];
      }
    }
  }
  for (int i = lane; i < S; i += 64) {
    float z = zlin(i);
    if (a.t_rand || a.draw) {                                        // :564-579
      float t;
      if (a.draw) {
        float u[4];
        draw_block(a.seed_lo, a.seed_hi, step, ray, i >> 2, u);
        t = u[i & 3];
      } else {
        t = a.t_rand[(size_t)ray * S + i];
      }
      const float zm = i > 0 ? zlin(i - 1) : z;
      const float zp = i + 1 < S ? zlin(i + 1) : z;
      const float lower = i > 0 ? 0.5f * (z + zm) : z;
      const float upper = i + 1 < S ? 0.5f * (zp + z) : z;
      z = lower + (upper - lower) * t;
    }
    a.z_vals[(size_t)ray * S + i] = z;
    if (a.pts) {
      float* p = a.pts + ((size_t)ray * S + i) * 3;                  // :657  o + d*z
      p[0] = ox + dx * z;
      p[1] = oy + dy * z;
      p[2] = oz + dz * z;
    }
  }
}

}  // namespace scade

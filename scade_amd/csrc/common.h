// Shared helpers for the scade_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SCADE_WAVE 64

// ---- error plumbing (capi.hip) -------------------------------------------
void scade_set_error(const char* fmt, ...);
int scade_check_launch(const char* what);

#define SCADE_REQUIRE(cond, code, ...)        \
  do {                                        \
    if (!(cond)) {                            \
      scade_set_error(__VA_ARGS__);           \
      return (code);                          \
    }                                         \
  } while (0)

// hipFuncSetAttribute applies to the CURRENT device only: the launchers keep one "done" bit per device
// ordinal, so a process that drives several GPUs sets the large-LDS attribute on each of them
static inline int scade_current_device() {
  int dv = 0;
  (void)hipGetDevice(&dv);
  return dv & 63;
}
static inline bool scade_attr_needed(unsigned long long mask) { return !((mask >> scade_current_device()) & 1ull); }
static inline void scade_attr_done(unsigned long long& mask) { mask |= 1ull << scade_current_device(); }

// wgrad + reduce launcher shared by the backward variants (mlp_bwd.hip)
int scade_launch_wgrad(const float* acts, const float* dz, const float* g_out, int P, float* partial,
                       float* grad_flat, hipStream_t s);

// ---- wave-level primitives -------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive scan over the 64 lanes of a wave with a binary op
template <typename T, typename Op>
__device__ __forceinline__ T wave_scan_incl(T v, Op op) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T n = __shfl_up(v, o, 64);
    if (l >= o) v = op(n, v);
  }
  return v;
}

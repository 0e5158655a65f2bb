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

// ---- device-resident optimizer scalars (optim.hip: float[16] per FusedAdam) -------------------------------
// One tick = the start of a step: t += 1, the staircase learning rate on the reference's loop index
// (train_utils/hyperparameter_update.py:8-13, run_scade_scannet.py:988-991) and Adam's bias corrections.
// Shared by the kernels that open a graph-captured step (scade_stage_inputs, scade_gather_batch).
namespace scade {
__device__ __forceinline__ void adam_tick(float* st) {
  st[13] = st[0];                      // steps taken BEFORE this one: the step's index for in-kernel draws
  const float t = st[0] + 1.0f;
  st[0] = t;
  const double it = (double)t + (double)st[11];
  const double k = st[3] > 0.f ? floor(it / (double)st[3]) : 0.0;
  st[8] = (float)((double)st[1] * pow((double)st[2], k));
  st[9] = (float)(1.0 - pow((double)st[4], (double)t));
  st[10] = (float)sqrt(1.0 - pow((double)st[5], (double)t));
}
}  // namespace scade

// Store-data hazard (found in round 5; tools/check_store_hazard.py scans every build for it).  A store of more than
// 64 bits reads its data registers in the issue slots BEHIND it; gfx940-class parts need two wait states before a VALU
// write of those registers.  LLVM's hazard recognizer covers flat / global stores and buffer stores without an SGPR
// soffset (GCNHazardRecognizer::createsVALUHazard) - a `buffer_store_dwordx4 ..., sN offen` followed at once by a VALU
// write of its first data register reached memory with the NEW value in lanes 12..15 of each 16 on this part.
// STORE_DATA_HOLD(v) behind the store keeps v's registers untouched over two wait states; STORE_DATA_PIN(w) behind
// the HOLD extends that to the data of the stores issued just before it.
#define STORE_DATA_HOLD(v) asm volatile("s_nop 1" : "+v"(v))
#define STORE_DATA_PIN(v) asm volatile("" : "+v"(v))

// rows a tile copy's buffer descriptor spans: from the tile's first row p0 to the end of the slot at row P, at most
// 4096 (no tile is larger; 4096 rows x 1 KiB fits the descriptor's 32-bit range with room to spare)
__device__ __forceinline__ unsigned tile_rows_left(int p0, int P) {
  const int n = P - p0;
  return (unsigned)(n < 0 ? 0 : (n > 4096 ? 4096 : n));
}

// ---- wave-level primitives -------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// minimum that PROPAGATES NaN like torch.min (v_minimum3_f32; fminf / v_min_f32 return the other operand)
__device__ __forceinline__ float min_nan(float a, float b) { return __builtin_elementwise_minimum(a, b); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- fp64 wave scans / reductions on DPP ----------------------------------------------------------------
// The per-ray kernels scan and reduce in fp64 (ATen's CPU cumsum / cumprod accumulate in double and round each
// prefix).  As __shfl_up(double) every step of a scan was two ds_bpermute_b32 through the LDS crossbar plus
// their ~100-cycle round trip in a six-step dependent chain; on the data-parallel-primitive path a step is two
// v_mov_b32 with a DPP modifier (row_shr inside the rows of 16 lanes, row_bcast:15 / :31 across them - the
// classic GCN scan) and the f64 operation, no LDS traffic and a handful of cycles of latency.
// dpp_ctrl encodings (gfx9): row_shl:n 0x100+n, row_shr:n 0x110+n, wave_shl:1 0x130, wave_shr:1 0x138,
// row_bcast:15 0x142, row_bcast:31 0x143.  With bound_ctrl = false a lane whose source lies outside its row
// (or whose row is masked off) keeps `old` - the identity of the operation.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_mov_d(double old, double src) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
// value of one lane in every lane, through two SGPRs (wave-uniform)
template <int LANE>
__device__ __forceinline__ double wave_lane_d(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), LANE);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), LANE);
  return __hiloint2double(hi, lo);
}
struct DppSum {
  static constexpr double identity() { return 0.0; }
  static __device__ __forceinline__ double op(double lower, double v) { return lower + v; }
};
struct DppProd {
  static constexpr double identity() { return 1.0; }
  static __device__ __forceinline__ double op(double lower, double v) { return lower * v; }
};
// inclusive scan, lane 0 first
template <typename OP>
__device__ __forceinline__ double wave_incl_scan_d(double v) {
  constexpr double e = OP::identity();
  v = OP::op(dpp_mov_d<0x111>(e, v), v);
  v = OP::op(dpp_mov_d<0x112>(e, v), v);
  v = OP::op(dpp_mov_d<0x114>(e, v), v);
  v = OP::op(dpp_mov_d<0x118>(e, v), v);              // every row of 16 scanned
  v = OP::op(dpp_mov_d<0x142, 0xA>(e, v), v);         // rows 1, 3 += last lane of rows 0, 2
  v = OP::op(dpp_mov_d<0x143, 0xC>(e, v), v);         // rows 2, 3 += lane 31
  return v;
}
// the wave shifted up by one lane (lane i gets lane i-1; lane 0 gets `first`)
__device__ __forceinline__ double wave_shr1_d(double v, double first) { return dpp_mov_d<0x138>(first, v); }
// inclusive SUFFIX sum (lane 63 first): rows scanned with row_shl, the row totals joined through SGPRs
__device__ __forceinline__ double wave_incl_sum_rev_d(double v) {
  v = v + dpp_mov_d<0x101>(0.0, v);
  v = v + dpp_mov_d<0x102>(0.0, v);
  v = v + dpp_mov_d<0x104>(0.0, v);
  v = v + dpp_mov_d<0x108>(0.0, v);                   // lane 16 r = sum of row r
  const double r1 = wave_lane_d<16>(v), r2 = wave_lane_d<32>(v), r3 = wave_lane_d<48>(v);
  const int row = lane_id() >> 4;
  const double above = row == 0 ? r1 + (r2 + r3) : (row == 1 ? r2 + r3 : (row == 2 ? r3 : 0.0));
  return v + above;
}
// sum over the wave, in every lane (wave-uniform)
__device__ __forceinline__ double wave_sum_dpp_d(double v) { return wave_lane_d<63>(wave_incl_scan_d<DppSum>(v)); }

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

// workgroup barrier that orders LDS only (lgkmcnt(0), never vmcnt(0)): tiles are exchanged through LDS and nothing a
// workgroup stores to HBM is read back by it, so the stores of a tile copy and the weight fragments fetched ahead
// across a layer boundary stay in flight over the barrier (__syncthreads() drains both: its release fence is
// s_waitcnt vmcnt(0))
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
#define LDS_SYNC() lds_barrier()

// sin and cos of an fp32 argument in ~35 VALU operations (the library's sincosf is ~3x that and branches): three-term
// Cody-Waite reduction by pi/2 (fused multiply-adds: the products are exact, |a| < 3e4) and the minimax polynomials
// of the classic single-precision kernels on [-pi/4, pi/4]: 8.6e-8 absolute against fp64 over the embedding's
// argument range (1.4 ulp at 1).  VALID FOR |a| < 3e4 ONLY: callers route larger or non-finite arguments to the
// library (sincos_cw_ok).
__device__ __forceinline__ bool sincos_cw_ok(float a) { return fabsf(a) < 3.0e4f; }
__device__ __forceinline__ void sincos_cw(float a, float& sn, float& cs) {
  const float qf = rintf(a * 0.636619772367581343076f);              // 2 / pi
  const int q = (int)qf;
  float r = __builtin_fmaf(qf, -3.1414794921875f * 0.5f, a);
  r = __builtin_fmaf(qf, -0.00011315941810607910156f * 0.5f, r);
  r = __builtin_fmaf(qf, -1.9841872589410058936e-09f * 0.5f, r);
  const float r2 = r * r;
  float u = -0.000195169282960705459117889f;
  u = __builtin_fmaf(u, r2, 0.00833215750753879547119141f);
  u = __builtin_fmaf(u, r2, -0.166666537523269653320312f);
  float rs = __builtin_fmaf(u * r2, r, r);
  float v = -2.71811842367242206819355e-07f;
  v = __builtin_fmaf(v, r2, 2.47990446951007470488548e-05f);
  v = __builtin_fmaf(v, r2, -0.00138888787478208541870117f);
  v = __builtin_fmaf(v, r2, 0.0416666641831398010253906f);
  v = __builtin_fmaf(v, r2, -0.5f);
  float rc = __builtin_fmaf(r2, v, 1.0f);
  if (q & 1) { const float t = rs; rs = rc; rc = t; }
  if (q & 2) rs = -rs;
  if ((q + 1) & 2) rc = -rc;
  sn = rs; cs = rc;
}

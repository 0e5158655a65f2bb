// The weight packs a train step needs, for BOTH networks, in ONE launch (scade_mlp_pack_step).
//
// After every optimizer step the MFMA kernels' weight blobs are rebuilt from the fp32 master parameters: the
// forward layout and the transposed (dgrad) layout of the coarse and of the fine NeRF - four pack launches of
// ~5 us each per step, which is 5 % of a 128-ray bf16 step.  The rows of the four stand-alone pack kernels
// (mlp_pack.h) are independent, so one grid of PACK_BLOCKS x (networks x 20 rows) runs them all.
#include "mlp_pack.h"

namespace scade {

constexpr int PACK_STEP_ROWS = PACK_FWD_ROWS + PACK_T_ROWS;     // rows per network
struct PackStepArgs {
  const float* p[2][N_PARAM_TENSORS];
  void* fwd[2];      // forward-layout blob per network (exact: floats; 16-bit: the lp blob), or null: skipped
  void* tr[2];       // transposed blob per network, or null
};

// FMT 0: exact fp32 packs; 1: bf16; 2: fp16
template <int FMT>
__global__ void mlp_pack_step_kernel(PackStepArgs a) {
  const int net = blockIdx.y / PACK_STEP_ROWS, row = blockIdx.y % PACK_STEP_ROWS;
  const float* const* p = a.p[net];
  if (row < PACK_FWD_ROWS) {
    void* out = a.fwd[net];
    if (!out) return;
    if (FMT == 0) pack_fwd_row(p, reinterpret_cast<float*>(out), row, blockIdx.x, gridDim.x);
    else pack_lp_row<FMT == 1>(p, out, row, blockIdx.x, gridDim.x);
  } else {
    void* out = a.tr[net];
    if (!out) return;
    if (FMT == 0) pack_t_row(p, reinterpret_cast<float*>(out), row - PACK_FWD_ROWS, blockIdx.x, gridDim.x);
    else pack_t_lp_row<FMT == 1>(p, out, row - PACK_FWD_ROWS, blockIdx.x, gridDim.x);
  }
}

// the split-precision ("f16x3") training packs: per network the exact forward blob (the dgrad heads read the fp32 head
// weights from it), the two-plane forward blob and the two-plane transposed blob - six stand-alone launches per step
struct PackStepF16Args {
  const float* p[2][N_PARAM_TENSORS];
  float* exact[2];
  void* fwd[2];
  void* tr[2];
};
constexpr int PACK_STEP_F16_ROWS = 2 * PACK_FWD_ROWS + PACK_T_ROWS;
__global__ void mlp_pack_step_f16_kernel(PackStepF16Args a) {
  const int net = blockIdx.y / PACK_STEP_F16_ROWS, row = blockIdx.y % PACK_STEP_F16_ROWS;
  const float* const* p = a.p[net];
  if (row < PACK_FWD_ROWS) {
    if (a.exact[net]) pack_fwd_row(p, a.exact[net], row, blockIdx.x, gridDim.x);
  } else if (row < 2 * PACK_FWD_ROWS) {
    if (a.fwd[net]) pack_f16_row(p, a.fwd[net], row - PACK_FWD_ROWS, blockIdx.x, gridDim.x);
  } else {
    if (a.tr[net]) pack_t_f16_row(p, a.tr[net], row - 2 * PACK_FWD_ROWS, blockIdx.x, gridDim.x);
  }
}

}  // namespace scade

using namespace scade;

// the same for the split-precision kernels: packed_exact (scade_mlp_pack layout), packed_f16 (scade_mlp_pack_f16),
// packed_t_f16 (scade_mlp_pack_t_f16): n_nets blobs each, entries may be NULL (skipped)
extern "C" int scade_mlp_pack_step_f16x3(int n_nets, const float* const* params, float* const* packed_exact,
                                         void* const* packed_f16, void* const* packed_t_f16, void* stream) {
  SCADE_REQUIRE(n_nets == 1 || n_nets == 2, -2, "scade_mlp_pack_step_f16x3: one or two networks");
  SCADE_REQUIRE(params && packed_exact && packed_f16 && packed_t_f16, -1, "scade_mlp_pack_step_f16x3: null pointer");
  PackStepF16Args a{};
  for (int n = 0; n < n_nets; ++n) {
    for (int i = 0; i < N_PARAM_TENSORS; ++i) {
      SCADE_REQUIRE(params[n * N_PARAM_TENSORS + i], -1, "scade_mlp_pack_step_f16x3: params[%d][%d] is null", n, i);
      a.p[n][i] = params[n * N_PARAM_TENSORS + i];
    }
    a.exact[n] = packed_exact[n];
    a.fwd[n] = packed_f16[n];
    a.tr[n] = packed_t_f16[n];
  }
  hipLaunchKernelGGL(mlp_pack_step_f16_kernel, dim3(PACK_BLOCKS, n_nets * PACK_STEP_F16_ROWS), dim3(256), 0,
                     (hipStream_t)stream, a);
  return scade_check_launch("scade_mlp_pack_step_f16x3");
}

// n_nets = 1 or 2; params: n_nets x 24 parameter pointers (the scade_mlp_pack order, network-major);
// format: 0 = exact fp32 (scade_mlp_pack + scade_mlp_pack_t layouts), 1 = bf16, 2 = fp16 (scade_mlp_pack_lp +
// scade_mlp_pack_t_lp layouts); packed_fwd / packed_t: n_nets output blobs each (entries may be NULL: skipped).
extern "C" int scade_mlp_pack_step(int n_nets, const float* const* params, int format, void* const* packed_fwd,
                                   void* const* packed_t, void* stream) {
  SCADE_REQUIRE(n_nets == 1 || n_nets == 2, -2, "scade_mlp_pack_step: one or two networks");
  SCADE_REQUIRE(format >= 0 && format <= 2, -2, "scade_mlp_pack_step: format 0 (fp32), 1 (bf16) or 2 (fp16)");
  SCADE_REQUIRE(params && packed_fwd && packed_t, -1, "scade_mlp_pack_step: null pointer");
  PackStepArgs a{};
  for (int n = 0; n < n_nets; ++n) {
    for (int i = 0; i < N_PARAM_TENSORS; ++i) {
      SCADE_REQUIRE(params[n * N_PARAM_TENSORS + i], -1, "scade_mlp_pack_step: params[%d][%d] is null", n, i);
      a.p[n][i] = params[n * N_PARAM_TENSORS + i];
    }
    a.fwd[n] = packed_fwd[n];
    a.tr[n] = packed_t[n];
  }
  const dim3 grid(PACK_BLOCKS, n_nets * PACK_STEP_ROWS);
  hipStream_t s = (hipStream_t)stream;
  if (format == 0) hipLaunchKernelGGL(mlp_pack_step_kernel<0>, grid, dim3(256), 0, s, a);
  else if (format == 1) hipLaunchKernelGGL(mlp_pack_step_kernel<1>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(mlp_pack_step_kernel<2>, grid, dim3(256), 0, s, a);
  return scade_check_launch("scade_mlp_pack_step");
}

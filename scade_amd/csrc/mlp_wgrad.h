// Job table of the weight-gradient kernels (exact fp32 and split-precision variants share it).
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace scade {

enum { WF_BIAS = 1, WF_ALPHA = 2, WF_VIEWCOLS = 4, WF_RGB = 8 };

struct WgradJob {
  long dz_off;       // float offset of the dZ matrix (row stride 256) in the dz workspace
  long in_off;       // float offset of the input matrix in the acts workspace
  int in_stride;     // 256 (activation slot) or 64 (emb)
  int kw;            // tile width in k: 256 or 64
  int n_rows;        // valid output rows (256 or 128)
  int w_off;         // flat-gradient offset of the weight tensor
  int ld;            // its row length
  int kcol0;         // first column written
  int kvalid;        // columns written
  int b_off;         // flat-gradient offset of the bias (WF_BIAS)
  int flags;
  int aux_off;       // WF_ALPHA: offset of alpha weight (bias follows at +256); WF_VIEWCOLS: unused
};

constexpr int MAX_WGRAD_JOBS = 16;
struct WgradArgs {
  WgradJob jobs[MAX_WGRAD_JOBS];
  const float* acts;
  const float* dz;
  const float* g_out;   // [P,4] (rgb head)
  float* partial;       // [nchunks][N_PARAM_FLOATS]
  int P;
  int chunk;            // points per chunk (multiple of WG_PT)
  int njobs;
};

// flat-gradient offsets of the 24 parameter tensors (PARAM order of mlp_layout.h)
inline void param_offsets(int off[N_PARAM_TENSORS + 1]) {
  int o = 0, i = 0;
  for (int l = 0; l < 8; ++l) {
    const int k = l == 0 ? 57 : (l == 5 ? 313 : 256);
    off[i++] = o; o += 256 * k;
    off[i++] = o; o += 256;
  }
  off[i++] = o; o += 128 * 259;
  off[i++] = o; o += 128;
  off[i++] = o; o += 256 * 256;
  off[i++] = o; o += 256;
  off[i++] = o; o += 256;
  off[i++] = o; o += 1;
  off[i++] = o; o += 3 * 128;
  off[i++] = o; o += 3;
  off[i] = o;
}

// Chunks of points per weight-gradient launch.  Every weight-gradient workgroup owns a whole CU
// (LDS) and the nine 256x256 jobs dominate, so the chunk count is chosen as the largest one for
// which 9 x chunks fits k "rounds" of one workgroup per CU with chunks near 2400 points (85 chunks
// for the fine pass of a 1024-ray batch, 28 for the coarse pass), never shorter than 512 points.
// Measured against the former fixed 1536-point chunks: neutral for the MFMA-bound exact kernel
// (the small jobs fill the tail either way), -4 % for the HBM-bound 16-bit kernel, whose fp32
// per-chunk partials are a fifth of its traffic - which is why that kernel asks for longer chunks still
// (target_pts = LP_CHUNK_PTS = 3500: 56 chunks instead of 85 for the fine pass, bf16 train step -2.5 %;
// the same choice costs the exact kernel 0.5 %).
inline int pick_chunks(int P, int target_pts = 2400) {
  const int ncu = device_cus();
  long k = (9L * P + (long)ncu * target_pts - 1) / ((long)ncu * target_pts);
  if (k < 1) k = 1;
  long n = ncu * k / 9;
  // a launch of ONE round: all twelve jobs of a chunk in it (21 chunks on 256 CUs), not nine heavy ones filling the
  // CUs and the three light ones behind them (28 chunks = 1.3 rounds: the coarse pass of a 1024-ray step 362 us)
  if (k == 1) n = ncu / 12;
  const long nmax = P / 512 > 1 ? P / 512 : 1;
  if (n > nmax) n = nmax;
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}

// fills w (12 jobs, chunking) and returns grid.x; chunk is rounded to a multiple of `stage_pts`
inline int build_wgrad_jobs(WgradArgs& w, const float* acts, const float* dz, const float* g_out,
                            float* partial, int P, int stage_pts) {
  int off[N_PARAM_TENSORS + 1];
  param_offsets(off);
  w.acts = acts; w.dz = dz; w.g_out = g_out; w.partial = partial; w.P = P;
  const int nchunks = pick_chunks(P);
  int chunk = (P + nchunks - 1) / nchunks;
  chunk = (chunk + stage_pts - 1) / stage_pts * stage_pts;
  w.chunk = chunk;
  int nj = 0;
  auto slot = [&](int sidx) { return acts_slot_off(P, sidx); };
  auto add = [&](long dzo, long ino, int ins, int kw, int nrows, int woff, int ld, int kcol0,
                 int kvalid, int boff, int flags, int aux) {
    WgradJob& j = w.jobs[nj++];
    j.dz_off = dzo; j.in_off = ino; j.in_stride = ins; j.kw = kw; j.n_rows = nrows; j.w_off = woff;
    j.ld = ld; j.kcol0 = kcol0; j.kvalid = kvalid; j.b_off = boff; j.flags = flags; j.aux_off = aux;
  };
  // big jobs first, small last (tail filling)
  for (int l = 1; l <= 7; ++l) {
    const int ld = l == 5 ? 313 : 256, kc0 = l == 5 ? 57 : 0;
    add(slot(l), slot(l - 1), 256, 256, 256, off[2 * l], ld, kc0, 256, off[2 * l + 1], WF_BIAS, 0);
  }
  add(slot(SLOT_FEAT), slot(7), 256, 256, 256, off[18], 256, 0, 256, off[19], WF_BIAS | WF_ALPHA, off[20]);
  add(slot(SLOT_VIEWS_H), slot(SLOT_FEAT), 256, 256, 128, off[16], 259, 0, 256, off[17],
      WF_BIAS | WF_VIEWCOLS, 0);
  add(slot(0), acts_emb_off(P), 64, 64, 256, off[0], 57, 0, 57, off[1], WF_BIAS, 0);
  add(slot(5), acts_emb_off(P), 64, 64, 256, off[10], 313, 0, 57, 0, 0, 0);
  add(0, slot(SLOT_VIEWS_H), 256, 0, 0, off[22], 128, 0, 0, off[23], WF_RGB, 0);
  w.njobs = nj;
  return (P + chunk - 1) / chunk;
}

// rgb head: dW_r[c][k] = sum_pt g[pt][c] * hv[pt][k], db_r[c] = sum_pt g[pt][c].
// A thread owns 4 columns (one 16-byte load per point) of every 16th point, four points in
// flight: this job is pure load latency and, as the LAST job of the table, its workgroups
// would otherwise set the end of the whole launch.
// R24: the views rows are 24-bit rows (mlp_tile_f16.h: h plane fp16 [P][256], l8 plane e5m2 [P][256] behind it)
template <bool R24 = false>
__device__ __forceinline__ void wgrad_rgb_job(const WgradArgs& a, const WgradJob& jb, float* lds,
                                              int c0, int c1, float* __restrict__ out) {
  const int tid = threadIdx.x;
  const int k4 = tid & 31, pl = tid >> 5;       // 32 column groups x 16 point lanes
  const float* __restrict__ hv = a.acts + jb.in_off + 4 * k4;
  const unsigned char* __restrict__ hvb = reinterpret_cast<const unsigned char*>(a.acts + jb.in_off);
  const int P = a.P;
  float s[3][4], b[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) s[c][j] = 0.f;
  for (int pt0 = c0 + pl; pt0 < c1; pt0 += 64) {
    f32x4 h[4], g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pt = min(pt0 + 16 * q, P - 1);
      if (R24) {
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
        const u32x2_ hh_ = *reinterpret_cast<const u32x2_*>(hvb + (size_t)pt * 512 + 8 * k4);
        const unsigned mm_ = *reinterpret_cast<const unsigned*>(hvb + (size_t)P * 512 + (size_t)pt * 256 + 4 * k4);
        auto val = [](unsigned h16, unsigned l8) {          // h + l * 2^-11 (l8: the upper byte of the fp16 l)
          return (float)__builtin_bit_cast(_Float16, (unsigned short)h16) +
                 (float)__builtin_bit_cast(_Float16, (unsigned short)(l8 << 8)) * (1.0f / 2048.0f);
        };
        h[q][0] = val(hh_[0] & 0xffffu, mm_ & 0xffu);
        h[q][1] = val(hh_[0] >> 16, (mm_ >> 8) & 0xffu);
        h[q][2] = val(hh_[1] & 0xffffu, (mm_ >> 16) & 0xffu);
        h[q][3] = val(hh_[1] >> 16, mm_ >> 24);
      } else {
        h[q] = *reinterpret_cast<const f32x4*>(hv + (size_t)pt * 256);
      }
      g[q] = *reinterpret_cast<const f32x4*>(a.g_out + (size_t)pt * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (pt0 + 16 * q < c1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s[0][j] = fmaf(g[q][0], h[q][j], s[0][j]);
          s[1][j] = fmaf(g[q][1], h[q][j], s[1][j]);
          s[2][j] = fmaf(g[q][2], h[q][j], s[2][j]);
        }
        b[0] += g[q][0]; b[1] += g[q][1]; b[2] += g[q][2];
      }
    }
  }
  float* red = lds;                               // [16 point lanes][3][128] + [16][4]
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(pl * 3 + c) * 128 + 4 * k4 + j] = s[c][j];
  float* redb = red + 16 * 3 * 128;
  if (k4 == 0) { redb[pl * 4 + 0] = b[0]; redb[pl * 4 + 1] = b[1]; redb[pl * 4 + 2] = b[2]; }
  __syncthreads();
  if (tid < 384) {
    float t = 0.f;
    for (int p = 0; p < 16; ++p) t += red[p * 384 + tid];
    out[jb.w_off + tid] = t;
  }
  if (tid < 3) {
    float t = 0.f;
    for (int p = 0; p < 16; ++p) t += redb[p * 4 + tid];
    out[jb.b_off + tid] = t;
  }
}

// sum of the per-chunk partials: one float4 per thread, four independent accumulators
__global__ static void wgrad_reduce4_kernel(const float* partial, int nchunks, float* grad) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N_PARAM_FLOATS / 4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(partial) + i;
  constexpr size_t ST = N_PARAM_FLOATS / 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
  int c = 0;
  for (; c + 4 <= nchunks; c += 4) {
    s0 += p[(size_t)c * ST]; s1 += p[(size_t)(c + 1) * ST];
    s2 += p[(size_t)(c + 2) * ST]; s3 += p[(size_t)(c + 3) * ST];
  }
  for (; c < nchunks; ++c) s0 += p[(size_t)c * ST];
  reinterpret_cast<f32x4*>(grad)[i] = (s0 + s1) + (s2 + s3);
}
static_assert(N_PARAM_FLOATS % 4 == 0, "float4 reduce");
constexpr int WGRAD_REDUCE_BLOCKS = (N_PARAM_FLOATS / 4 + 255) / 256;

}  // namespace scade

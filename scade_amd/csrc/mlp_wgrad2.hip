// B2 of the exact fp32 backward: dW[n][k] = sum_points dZ[point][n] * In[point][k]   (what autograd's
// addmm-backward computes for the 12 nn.Linear layers of NeRF.forward, model/run_nerf_helpers.py:223-247).
//
// The contraction runs over POINTS while both operands sit in HBM point-major, so the MFMA fragments
// ("k-index = point") are columns of the staged tiles.  Second design of this kernel (the first fed
// v_mfma_f32_32x32x2_f32 from one ds_read_b32 per operand and k-step, 512-thread workgroups, one per CU,
// all eight waves in lock step on one barrier per stage):
//
//  * workgroup = 4 waves = HALF of a layer's weight gradient (128 output features n x 256 inputs k) for
//    one chunk of points; 48 KiB of LDS => TWO independent workgroups per CU, so the staging / barrier
//    phase of one hides under the MFMAs of the other (the forward kernel's recipe);
//  * tiles are TRANSPOSED on the way into LDS - a thread loads a 4-point x 4-feature block (four
//    coalesced 16-byte row loads) and writes four ds_write_b128 "feature f: points p..p+3" - so that one
//    ds_read_b128 is the operand of FOUR MFMA k-steps (lanes 0-31 carry points 8g..8g+3, lanes 32-63
//    points 8g+4..8g+7 of k-group g; both operands use the same point <-> (k-step, lane half) map, and any
//    permutation of the contraction index is legal): 6 LDS reads per 32 MFMAs instead of 24;
//  * LDS image: 384 rows (128 dZ features | 256 input features) of 32 points = BOTH pipeline buffers of 16
//    points side by side in one 128-byte row, 16-byte chunk c stored at c ^ key(row), key = (row ^ row>>2)
//    & 7 for rows stored four per lane and (row ^ row>>1) & 7 for the dZ rows stored two per lane:
//    conflict free for the transposed stores (8-lane groups) and for the fragment reads (the
//    ds_read_b128 lane groups {0-3,12-15,20-27}, ...) - measured SQ_LDS_BANK_CONFLICT = 0;
//  * bias / alpha-head / view-column riders work on the STAGING REGISTERS (no LDS traffic) and are reduced
//    over the four waves once per workgroup;
//  * the barrier sits BETWEEN the two k-groups of a stage and orders LDS only (fences restricted to the
//    local address space: lgkmcnt(0), never vmcnt(0) - the global prefetch stays in flight); every
//    fragment is read one k-group (32 MFMAs) ahead of its use, also across stages.
//
// Measured (MI355X, 196,608 points, alone on the chip): 1.93 ms against 2.01 ms of the first design, 0.67
// against 0.80 ms at 65,536 points; knock-outs of this kernel: no global loads -7 %, no LDS stores -1 %,
// no barrier -1.5 %, no fragment reads -2 %, all four -16 % (= the bare MFMA stream with its prologue /
// epilogue, 84 % of the pipe's peak).  A ONE-ROUND variant (511 equal-cost workgroups resident from start
// to end, per-job chunk lengths, both embedding blocks as one 512 x 64 job, halves of a layer on the same
// XCD for L2 reuse: FETCH_SIZE -17 %) measured 3-7 % SLOWER at every size and was dropped.
#include "mlp_wgrad.h"

namespace scade {

constexpr int W2_PT = 16;                         // points per pipeline stage (two stages per LDS row)
constexpr int W2_ROWS = 128 + 256;
constexpr int W2_LDS_BYTES = W2_ROWS * 32 * 4;    // 49,152

struct Wgrad2Job {
  long dz_off;       // float offset of the dZ matrix (row stride 256) in the dz workspace, n_base included
  long in_off;       // float offset of the input matrix in the acts workspace
  int in_stride;     // 256 (activation slot) or 64 (emb)
  int kw;            // k width of the tile: 256 or 64; 0 = the rgb-head job
  int n_base;        // first output row of this half (0 / 128)
  int n_rows;        // output rows of the tensor (256 / 128)
  int w_off, ld, kcol0, kvalid, b_off, flags, aux_off;   // as WgradJob
};
constexpr int MAX_WGRAD2_JOBS = 24;
struct Wgrad2Args {
  Wgrad2Job jobs[MAX_WGRAD2_JOBS];
  const float* acts;
  const float* dz;
  const float* g_out;
  float* partial;
  int P, chunk, njobs;
};

// float offset of 16-byte chunk lc (0..7 = buffer*4 + point group) of LDS row `row` (swizzle key from the
// row index LOCAL to its region, which is what both the stores and the reads use)
__device__ __forceinline__ int w2_off(int row_abs, int row_local, int lc) {
  return row_abs * 32 + ((lc ^ ((row_local ^ (row_local >> 2)) & 7)) << 2);
}
// the dZ rows are stored two per lane (rows 2 lane + j), which wants a different key: (row ^ row>>1) & 7
__device__ __forceinline__ int w2_off_a(int row, int lc) { return row * 32 + ((lc ^ ((row ^ (row >> 1)) & 7)) << 2); }
// workgroup barrier that orders LDS only: release/acquire fences restricted to the local address space
// (lgkmcnt(0), no vmcnt(0) - the global prefetch stays in flight), visible to the compiler's own wait
// counting, pinned in place against the MFMA groups on either side
__device__ __forceinline__ void w2_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  __builtin_amdgcn_sched_barrier(0);
}

typedef __amdgpu_buffer_rsrc_t w2_rsrc_t;
// buffer descriptor over [base, base + bytes): loads past the end return 0 (the tail rows of a chunk need
// no per-load select) and take a 32-bit lane offset + a scalar row offset instead of 64-bit addresses
__device__ __forceinline__ w2_rsrc_t w2_make_rsrc(const float* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}

template <int KW, int FLAGS>
__device__ __forceinline__ void wgrad2_mfma_job(const Wgrad2Args& a, const Wgrad2Job& jb, float* lds, int c0,
                                                int c1, float* __restrict__ out) {
  constexpr int NKT = KW == 256 ? 4 : 1;          // k-tiles of 32 per wave
  constexpr int flags = FLAGS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int nsub = (wave >> 1) * 64;              // this wave's 64 output features inside the half
  const int k0 = (wave & 1) * (KW / 2);
  const int P = a.P;
  const int npts = c1 - c0;
  // dZ rows of this chunk (row stride 1 KiB; n_base is inside dz_off) and input rows (1 KiB or 256 B)
  const w2_rsrc_t ra = w2_make_rsrc(a.dz + jb.dz_off + (size_t)c0 * 256, (unsigned)npts * 1024u);
  const w2_rsrc_t rb = w2_make_rsrc(a.acts + jb.in_off + (size_t)c0 * (KW == 256 ? 256 : 64),
                                    (unsigned)npts * (KW == 256 ? 1024u : 256u));

  f32x16 acc[2][NKT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NKT; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

  // ---- staging: A block = 4 points x features {2 lane, 2 lane + 1}; B block = 4 points x features
  // 4 lane .. 4 lane + 3 (KW = 64: wave 0 only, 16 column groups x 4 point groups); points 4 wave + q
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 pa[4];
  f32x4 pb[4];
  float da[4] = {0.f, 0.f, 0.f, 0.f};             // d alpha_pre of the wave's 4 points (WF_ALPHA)
  float vw[4][3];                                 // view directions of the wave's 4 points (WF_VIEWCOLS)
  float bias_acc[2] = {0.f, 0.f}, alpha_acc[4] = {0.f, 0.f, 0.f, 0.f}, dal_acc = 0.f;
  float vc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
  const int b_c4 = KW == 256 ? lane : (lane & 15), b_pg = KW == 256 ? wave : (lane >> 4);
  const bool b_on = KW == 256 || wave == 0;

  const int va = lane * 8;                                                       // byte offset inside a dZ row
  const int vb = KW == 256 ? lane * 16 : (4 * b_pg) * 256 + b_c4 * 16;         // inside an input row (+ point group)
  auto issue = [&](int pt0) {
    const int rel = pt0 - c0 + 4 * wave;                                           // wave-uniform row index
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pa[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(ra, va, (rel + q) * 1024, 2));   // read once: nt
      const int pt = pt0 + 4 * wave + q;
      if (flags & WF_ALPHA) da[q] = pt < c1 ? a.dz[dz_dalpha_off(P) + pt] : 0.f;
      if (flags & WF_VIEWCOLS) {
#pragma unroll
        for (int c = 0; c < 3; ++c) vw[q][c] = pt < c1 ? a.acts[acts_emb_off(P) + (size_t)pt * 64 + 60 + c] : 0.f;
      }
    }
    if (b_on) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        pb[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rb, vb, KW == 256 ? (rel + q) * 1024 : (pt0 - c0 + q) * 256, 2));
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 2 * lane + j;
      const f32x4 v = {pa[0][j], pa[1][j], pa[2][j], pa[3][j]};
      *reinterpret_cast<f32x4*>(lds + w2_off_a(row, 4 * buf + wave)) = v;
      bias_acc[j] += (pa[0][j] + pa[1][j]) + (pa[2][j] + pa[3][j]);
    }
    if (b_on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = 4 * b_c4 + j;
        const f32x4 v = {pb[0][j], pb[1][j], pb[2][j], pb[3][j]};
        *reinterpret_cast<f32x4*>(lds + w2_off(128 + row, row, 4 * buf + b_pg)) = v;
      }
    }
    if (flags & WF_ALPHA) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) alpha_acc[j] = fmaf(da[q], pb[q][j], alpha_acc[j]);
        dal_acc += da[q];
      }
    }
    if (flags & WF_VIEWCOLS) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int c = 0; c < 3; ++c) vc[j][c] = fmaf(pa[q][j], vw[q][c], vc[j][c]);
    }
  };

  // ---- fragments: one ds_read_b128 = 4 points of this lane half = operand of 4 MFMA k-steps
  struct Frag { f32x4 a[2]; f32x4 b[NKT]; };
  int rowA[2], rowB[NKT];
#pragma unroll
  for (int t = 0; t < 2; ++t) rowA[t] = nsub + 32 * t + r;
#pragma unroll
  for (int u = 0; u < NKT; ++u) rowB[u] = k0 + 32 * u + r;
  auto read_frag = [&](Frag& f, int buf, int g) {
    const int lc = 4 * buf + 2 * g + hh;
#pragma unroll
    for (int t = 0; t < 2; ++t) f.a[t] = *reinterpret_cast<const f32x4*>(lds + w2_off_a(rowA[t], lc));
#pragma unroll
    for (int u = 0; u < NKT; ++u) f.b[u] = *reinterpret_cast<const f32x4*>(lds + w2_off(128 + rowB[u], rowB[u], lc));
  };
  auto mfma_group = [&](const Frag& f) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NKT; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[t][j], f.b[u][j], acc[t][u], 0, 0, 0);
  };

  // ---- pipeline: stage s+2 in flight to registers, stage s+1 committed to the other half of the rows
  // while stage s is multiplied; the barrier between the two k-groups of a stage publishes stage s+1
  issue(c0);
  commit(0);
  if (c0 + W2_PT < c1) issue(c0 + W2_PT);
  w2_barrier();
  Frag f0, f1;
  read_frag(f0, 0, 0);
  int buf = 0;
  for (int pt0 = c0; pt0 < c1; pt0 += W2_PT, buf ^= 1) {
    const bool has_next = pt0 + W2_PT < c1;
    if (has_next) commit(buf ^ 1);
    if (pt0 + 2 * W2_PT < c1) issue(pt0 + 2 * W2_PT);
    read_frag(f1, buf, 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(f0);
    w2_barrier();
    if (has_next) read_frag(f0, buf ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(f1);
  }

  // ---- write the partial ---------------------------------------------------------
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NKT; ++u) {
      const int k = k0 + 32 * u + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = jb.n_base + nsub + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (n < jb.n_rows && k < jb.kvalid) out[jb.w_off + (size_t)n * jb.ld + jb.kcol0 + k] = acc[t][u][i];
      }
    }
  // riders: every wave saw a quarter of the points -> sum the four waves through LDS
  if (flags & (WF_BIAS | WF_ALPHA | WF_VIEWCOLS)) {
    w2_barrier();                                  // all fragment reads done: the rows are free
    float* red = lds;                              // [4 waves][128 bias | 256 alpha | 384 view cols | 1]
    float* rw = red + wave * 772;
    rw[2 * lane] = bias_acc[0];
    rw[2 * lane + 1] = bias_acc[1];
    if (flags & WF_ALPHA) {
#pragma unroll
      for (int j = 0; j < 4; ++j) rw[128 + 4 * lane + j] = alpha_acc[j];
      if (lane == 0) rw[768] = dal_acc;
    }
    if (flags & WF_VIEWCOLS) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int c = 0; c < 3; ++c) rw[384 + (2 * lane + j) * 3 + c] = vc[j][c];
    }
    w2_barrier();
    auto sum4 = [&](int i) { return (red[i] + red[772 + i]) + (red[1544 + i] + red[2316 + i]); };
    if ((jb.flags & WF_BIAS) && tid < 128 && jb.n_base + tid < jb.n_rows) out[jb.b_off + jb.n_base + tid] = sum4(tid);
    if (flags & WF_ALPHA) {
      out[jb.aux_off + tid] = sum4(128 + tid);
      if (tid == 0) out[jb.aux_off + 256] = sum4(768);
    }
    if ((flags & WF_VIEWCOLS) && tid < 128 && jb.n_base + tid < jb.n_rows) {
#pragma unroll
      for (int c = 0; c < 3; ++c) out[jb.w_off + (size_t)(jb.n_base + tid) * jb.ld + 256 + c] = sum4(384 + tid * 3 + c);
    }
  }
}

// rgb head: dW_r[c][k] = sum_pt g[pt][c] * hv[pt][k], db_r[c] = sum_pt g[pt][c].  A thread owns 4 columns
// (one 16-byte load per point) of every 8th point, four points in flight: this job is pure load latency.
__device__ __forceinline__ void wgrad2_rgb_job(const Wgrad2Args& a, const Wgrad2Job& jb, float* lds, int c0,
                                               int c1, float* __restrict__ out) {
  const int tid = threadIdx.x;
  const int k4 = tid & 31, pl = tid >> 5;       // 32 column groups x 8 point lanes
  const float* __restrict__ hv = a.acts + jb.in_off + 4 * k4;
  const int P = a.P;
  float s[3][4], b[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) s[c][j] = 0.f;
  for (int pt0 = c0 + pl; pt0 < c1; pt0 += 32) {
    f32x4 h[4], g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pt = min(pt0 + 8 * q, P - 1);
      h[q] = *reinterpret_cast<const f32x4*>(hv + (size_t)pt * 256);
      g[q] = *reinterpret_cast<const f32x4*>(a.g_out + (size_t)pt * 4);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (pt0 + 8 * q < c1) {
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
  float* red = lds;                               // [8 point lanes][3][128] + [8][4]
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(pl * 3 + c) * 128 + 4 * k4 + j] = s[c][j];
  float* redb = red + 8 * 3 * 128;
  if (k4 == 0) { redb[pl * 4 + 0] = b[0]; redb[pl * 4 + 1] = b[1]; redb[pl * 4 + 2] = b[2]; }
  __syncthreads();
  for (int i = tid; i < 384; i += 256) {
    float t = 0.f;
    for (int p = 0; p < 8; ++p) t += red[p * 384 + i];
    out[jb.w_off + i] = t;
  }
  if (tid < 3) {
    float t = 0.f;
    for (int p = 0; p < 8; ++p) t += redb[p * 4 + tid];
    out[jb.b_off + tid] = t;
  }
}

__global__ __launch_bounds__(256, 2) void mlp_wgrad2_kernel(Wgrad2Args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const Wgrad2Job& jb = a.jobs[blockIdx.y];
  const int c0 = blockIdx.x * a.chunk;
  const int c1 = min(a.P, c0 + a.chunk);
  float* out = a.partial + (size_t)blockIdx.x * N_PARAM_FLOATS;
  if (jb.kw == 0) {
    wgrad2_rgb_job(a, jb, lds, c0, c1, out);
  } else if (jb.kw == 256) {
    if (jb.flags == WF_BIAS) wgrad2_mfma_job<256, WF_BIAS>(a, jb, lds, c0, c1, out);
    else if (jb.flags == (WF_BIAS | WF_ALPHA)) wgrad2_mfma_job<256, WF_BIAS | WF_ALPHA>(a, jb, lds, c0, c1, out);
    else wgrad2_mfma_job<256, WF_BIAS | WF_VIEWCOLS>(a, jb, lds, c0, c1, out);
  } else {
    wgrad2_mfma_job<64, WF_BIAS>(a, jb, lds, c0, c1, out);     // (the layer-5 block has no bias: b_off checked)
  }
}

// Chunks of points per launch.  Two workgroups share a CU; per chunk there are 17 half-layer workgroups of
// equal length (8 layers x 2 halves + the 128-row views layer) plus five short ones (four embedding
// columns blocks, the rgb head) worth about 1.5 more.  The chunk count fills k whole "rounds" of the
// 2 x CUs slots with chunks near `target_pts` points (SCADE_WGRAD_PTS overrides it for experiments).
static int w2_target_pts() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("SCADE_WGRAD_PTS");
    v = e ? atoi(e) : 0;
    if (v < 64) v = 2400;
  }
  return v;
}
int pick_chunks_v2(int P) {
  const double slots = 2.0 * device_cus(), per_chunk = 18.5;
  const int target = w2_target_pts();
  long k = (long)((double)P * per_chunk / (slots * target) + 0.5);
  if (k < 1) k = 1;
  long n = (long)(slots * k / per_chunk);
  const long nmax = P / 256 > 1 ? P / 256 : 1;
  if (n > nmax) n = nmax;
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}

static int build_wgrad2_jobs(Wgrad2Args& w, const float* acts, const float* dz, const float* g_out, float* partial,
                             int P) {
  int off[N_PARAM_TENSORS + 1];
  param_offsets(off);
  w.acts = acts; w.dz = dz; w.g_out = g_out; w.partial = partial; w.P = P;
  const int nchunks = pick_chunks_v2(P);
  int chunk = (P + nchunks - 1) / nchunks;
  chunk = (chunk + W2_PT - 1) / W2_PT * W2_PT;
  w.chunk = chunk;
  int nj = 0;
  auto slot = [&](int sidx) { return acts_slot_off(P, sidx); };
  auto add = [&](long dzo, long ino, int ins, int kw, int nbase, int nrows, int woff, int ld, int kcol0, int kvalid,
                 int boff, int flags, int aux) {
    Wgrad2Job& j = w.jobs[nj++];
    j.dz_off = dzo + nbase; j.in_off = ino; j.in_stride = ins; j.kw = kw; j.n_base = nbase; j.n_rows = nrows;
    j.w_off = woff; j.ld = ld; j.kcol0 = kcol0; j.kvalid = kvalid; j.b_off = boff; j.flags = flags; j.aux_off = aux;
  };
  // long jobs first, short last (tail filling)
  for (int l = 1; l <= 7; ++l) {
    const int ld = l == 5 ? 313 : 256, kc0 = l == 5 ? 57 : 0;
    for (int h = 0; h < 2; ++h)
      add(slot(l), slot(l - 1), 256, 256, 128 * h, 256, off[2 * l], ld, kc0, 256, off[2 * l + 1], WF_BIAS, 0);
  }
  for (int h = 0; h < 2; ++h)   // the alpha head rides on ONE half only (it needs the whole input row, not dZ)
    add(slot(SLOT_FEAT), slot(7), 256, 256, 128 * h, 256, off[18], 256, 0, 256, off[19],
        WF_BIAS | (h == 0 ? WF_ALPHA : 0), off[20]);
  add(slot(SLOT_VIEWS_H), slot(SLOT_FEAT), 256, 256, 0, 128, off[16], 259, 0, 256, off[17], WF_BIAS | WF_VIEWCOLS, 0);
  for (int h = 0; h < 2; ++h) {
    add(slot(0), acts_emb_off(P), 64, 64, 128 * h, 256, off[0], 57, 0, 57, off[1], WF_BIAS, 0);
    add(slot(5), acts_emb_off(P), 64, 64, 128 * h, 256, off[10], 313, 0, 57, 0, 0, 0);
  }
  add(0, slot(SLOT_VIEWS_H), 256, 0, 0, 0, off[22], 128, 0, 0, off[23], WF_RGB, 0);
  w.njobs = nj;
  return (P + chunk - 1) / chunk;
}

}  // namespace scade

using namespace scade;

// wgrad + reduce (shared by the exact backward and the split-precision backward's exact-wgrad mode)
int scade_launch_wgrad(const float* acts, const float* dz, const float* g_out, int P, float* partial,
                       float* grad_flat, hipStream_t s) {
  static unsigned long long attr_set = 0;   // one bit per device ordinal
  if (scade_attr_needed(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wgrad2_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES);
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(attr_set);
  }
  Wgrad2Args w{};
  const int grid_x = build_wgrad2_jobs(w, acts, dz, g_out, partial, P);
  hipLaunchKernelGGL(mlp_wgrad2_kernel, dim3(grid_x, w.njobs), dim3(256), W2_LDS_BYTES, s, w);
  if (int e = scade_check_launch("scade_mlp_bwd(wgrad)")) return e;
  hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(WGRAD_REDUCE_BLOCKS), dim3(256), 0, s, partial, grid_x, grad_flat);
  return scade_check_launch("scade_mlp_bwd(reduce)");
}

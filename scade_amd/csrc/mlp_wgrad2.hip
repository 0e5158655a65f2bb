// B2 of the exact fp32 backward: dW[n][k] = sum_points dZ[point][n] * In[point][k]   (what autograd's
// addmm-backward computes for the 12 nn.Linear layers of NeRF.forward, model/run_nerf_helpers.py:223-247).
//
// The contraction runs over POINTS while both operands sit in HBM point-major.  Third design of this kernel
// (first: 512-thread workgroups, one ds_read_b32 per operand and k-step; second: two 4-wave workgroups per
// CU, tiles transposed through staging registers so that one ds_read_b128 fed four k-steps):
//
//  * workgroup = 4 waves = HALF of a layer's weight gradient (128 output features n x 256 inputs k) for
//    one chunk of points, two workgroups per CU;
//  * the tiles go HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), POINT-MAJOR AS THEY LIE IN HBM: no
//    staging registers, no ds_write, no transposition, and out-of-chunk rows arrive as zeros (buffer
//    range check).  What makes the point-major image usable is a permutation of the OUTPUT rows: lane r of
//    a fragment read takes 8 (dZ) / 16 (input) consecutive bytes of point p - features 2r, 2r+1 (4r..4r+3),
//    the operands of TWO (FOUR) different 32x32 output tiles at the same k-step - so MFMA row r of tile t is
//    feature 2r + t, column r of tile u is input 4r + u, and the epilogue un-permutes (which also turns the
//    partial's stores into 16-byte ones).  8 LDS reads per 32 MFMAs, conflict free by construction (a
//    ds_read_b64 lane group covers 256 contiguous bytes, a ds_read_b128 group 16 x 16 bytes in distinct
//    banks);
//  * a ring of W2_D = 3 slots of W2_S = 8 points (36 KiB): the slot of stage s is refilled with stage s+3
//    right after the barrier that publishes stage s+1 (every wave has read all of stage s by then) and is
//    waited for with a COUNTED vmcnt two barriers later - two stages stay in flight across every barrier.
//    The barrier sits between the two halves of a stage and orders LDS only (fences restricted to the local
//    address space); every fragment is read one half stage (32 MFMAs) ahead of its use;
//  * bias / alpha-head / view-column riders read the published slot; their per-point scalars (d alpha, the
//    view direction) ride the same ring as 4-byte LDS-DMA pieces, because an ordinary load's in-order vmcnt
//    wait would drain the ring.
//
// Measured (MI355X, 196,608 points, same box, back to back): dgrad + wgrad + reduce 3.66 ms against 3.76 ms
// with the second design (wgrad 1.83 against 1.93 ms; 0.64 against 0.67 ms at 65,536 points).  Ring depth
// 2..6, 16-point stages, nt on/off and chunks of 1,600..3,600 points all measure within 1 %; knock-outs of
// this kernel: no DMA -6 %, no fragment reads -6 %, no barrier 0, all three -11.5 % (= the bare MFMA stream
// with its prologue / epilogue: 141 TFLOP/s, 90 % of the 2.4 GHz peak - the forward kernel's 88.8 % says
// that is the clock the chip sustains under fp32 MFMA load).  Putting the two halves of a layer on the same
// XCD (L2 reuse of the input rows) measured 2 % slower (padding of the chunk count to 8) and was dropped,
// as were THREE workgroups per CU (168 VGPRs, no spill: +5 % time - a third wave per SIMD only adds contention
// for a matrix pipe two already fill) and a ONE-ROUND variant of the second design (511 equal-cost resident workgroups: 3-7 % slower).
#include "mlp_wgrad.h"
#include "mlp_reduce.h"

namespace scade {

constexpr int W2_PT = 16;                         // chunk lengths are multiples of this
constexpr int W2_S = 8;                           // points per ring slot
constexpr int W2_D = 3;                           // ring slots
constexpr int W2_LDS_BYTES = W2_D * (W2_S * (128 + 256) + 64) * 4;    // 37,632

struct Wgrad2Job {
  long dz_off;       // float offset of the dZ matrix (row stride 256) in the dz workspace, n_base included
  long in_off;       // float offset of the input matrix in the acts workspace
                     // (both filled in by the kernel from the slots below and ITS network's point count)
  int dz_slot;       // workspace slot of the dZ matrix
  int in_slot;       // workspace slot of the input matrix; -1 = the embedding rows
  int in_stride;     // 256 (activation slot) or 64 (emb)
  int kw;            // k width of the tile: 256 or 64; 0 = the rgb-head job
  int n_base;        // first output row of this half (0 / 128)
  int n_rows;        // output rows of the tensor (256 / 128)
  int w_off, ld, kcol0, kvalid, b_off, flags, aux_off;   // as WgradJob
};
constexpr int MAX_WGRAD2_JOBS = 24;
// the per-network part of a launch: a launch covers one network or two (the coarse and the fine NeRF of a
// train step); chunks [0, gx0) of grid.x belong to net[0], the rest to net[1]
struct Wgrad2Net {
  const float* acts;
  const float* dz;
  const float* g_out;
  float* partial;
  int P;
};
struct Wgrad2Args {
  Wgrad2Job jobs[MAX_WGRAD2_JOBS];
  Wgrad2Net net[2];
  int chunk, njobs, gx0;
};

// workgroup barrier that orders LDS only: release/acquire fences restricted to the local address space
// (lgkmcnt(0), no vmcnt(0) - the ring stays in flight), visible to the compiler's own wait counting, pinned
// in place against the MFMA groups on either side
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
__device__ __forceinline__ void wgrad2_mfma_job(const Wgrad2Net& a, const Wgrad2Job& jb, float* lds, int c0,
                                                int c1, float* __restrict__ out) {
  constexpr int S = W2_S, D = W2_D;
  constexpr int NKT = KW == 256 ? 4 : 1;          // k-tiles of 32 per wave
  constexpr int flags = FLAGS;
  constexpr int XTRA = (FLAGS & (WF_ALPHA | WF_VIEWCOLS)) ? 64 : 0;   // rider scalars: d alpha [S] / view dirs [S][3]
  constexpr int STAGE = S * (128 + KW) + XTRA;    // floats per ring slot: [S][128] dZ | [S][KW] inputs | riders
  constexpr int NI = (KW == 256 ? (S / 8) * 3 : (S / 8) * 2) + (XTRA ? 1 : 0);   // LDS-DMA instructions per wave and stage
  constexpr int H = S / 4;                        // k-pairs (2 points) per half stage
  static_assert(D * STAGE * 4 <= W2_LDS_BYTES, "ring");
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const int nsub = (wave >> 1) * 64;              // this wave's 64 output features inside the half
  const int k0 = (wave & 1) * (KW / 2);
  const int P = a.P;
  const int npts = c1 - c0;
  // dZ rows of this chunk (row stride 1 KiB; n_base is inside dz_off), input rows (1 KiB or 256 B), and
  // the rider scalars of the chunk
  const w2_rsrc_t ra = w2_make_rsrc(a.dz + jb.dz_off + (size_t)c0 * 256, (unsigned)npts * 1024u);
  const w2_rsrc_t rb = w2_make_rsrc(a.acts + jb.in_off + (size_t)c0 * (KW == 256 ? 256 : 64),
                                    (unsigned)npts * (KW == 256 ? 1024u : 256u));
  const w2_rsrc_t rx = (FLAGS & WF_ALPHA) ? w2_make_rsrc(a.dz + dz_dalpha_off(P) + c0, (unsigned)npts * 4u)
                                          : w2_make_rsrc(a.acts + acts_emb_off(P) + (size_t)c0 * 64, (unsigned)npts * 256u);

  f32x16 acc[2][NKT];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int u = 0; u < NKT; ++u)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][u][i] = 0.f;

  // ---- LDS-DMA of stage st into slot sl: per 8 points a wave moves dZ rows 2w, 2w+1 (one instruction:
  // lanes 0-31 | 32-63) and input rows 2w, 2w+1 (KW = 256: one instruction each; KW = 64: 256-byte rows,
  // lanes 0-31 carry both).  Read once per workgroup: nt.
  const int va = hh * 1024 + r * 16;
  const int vb = lane * 16;
  auto issue = [&](int st, int sl) {
    float* slot = lds + sl * STAGE;
#pragma unroll
    for (int e = 0; e < S / 8; ++e) {
      const int row = 8 * e + 2 * wave;
      const int grow = st * S + row;              // row of the chunk (past its end: the buffer returns zeros)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(slot + row * 128), 16, va, grow * 1024, 0, 2);
      if (KW == 256) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(slot + S * 128 + (row + q) * 256), 16, vb,
                                                   (grow + q) * 1024, 0, 2);
      } else {
        if (lane < 32)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(slot + S * 128 + row * 64), 16, vb, grow * 256, 0, 2);
      }
    }
    constexpr int Q = S / 4;                      // rider points per wave
    if (FLAGS & WF_ALPHA) {
      if (lane < Q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(slot + S * (128 + KW) + Q * wave), 4, lane * 4,
                                                 (st * S + Q * wave) * 4, 0, 0);
    } else if (FLAGS & WF_VIEWCOLS) {
      if (lane < 3 * Q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(slot + S * (128 + KW) + 3 * Q * wave), 4,
                                                 (lane / 3) * 256 + (60 + lane % 3) * 4, (st * S + Q * wave) * 256, 0, 0);
    }
  };

  // ---- fragments of one half stage: k-pair j = points 2j + hh of the slot
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  struct Frag { f32x2 a[H]; f32x4 b[H]; };
  const int fa_off = hh * 128 + nsub + 2 * r;
  const int fb_off = S * 128 + hh * KW + k0 + NKT * r;
  auto read_frag = [&](Frag& f, int sl, int half) {
    const float* slot = lds + sl * STAGE;
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const int pp = 2 * (half * H + j);
      f.a[j] = *reinterpret_cast<const f32x2*>(slot + fa_off + pp * 128);
      if (NKT == 4) f.b[j] = *reinterpret_cast<const f32x4*>(slot + fb_off + pp * KW);
      else f.b[j][0] = slot[fb_off + pp * KW];
    }
  };
  auto mfma_group = [&](const Frag& f) {
#pragma unroll
    for (int j = 0; j < H; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NKT; ++u)
          acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[j][t], f.b[j][u], acc[t][u], 0, 0, 0);
  };

  // ---- riders read the published slot: thread = (feature n = tid & 127, point half tid >> 7) for the bias
  // and the view columns, thread = input column for the alpha head.  The reads are issued with the fragment
  // reads, the arithmetic runs BEHIND the MFMA group (in front of it, its lgkmcnt(0) would hold the MFMAs
  // back by an LDS round trip per half stage).
  float bias_acc = 0.f, alpha_acc = 0.f, dal_acc = 0.f, vc[3] = {0.f, 0.f, 0.f};
  const int rn = tid & 127, rph = __builtin_amdgcn_readfirstlane(tid >> 7);
  float rv[S / 2], rs[(FLAGS & WF_VIEWCOLS) ? 3 * (S / 2) : 1], ra_in[(FLAGS & WF_ALPHA) ? S : 1], ra_da[(FLAGS & WF_ALPHA) ? S : 1];
  auto riders_read = [&](int sl) {
    const float* slot = lds + sl * STAGE;
    if (flags & (WF_BIAS | WF_VIEWCOLS)) {
#pragma unroll
      for (int q = 0; q < S / 2; ++q) {
        const int pp = rph * (S / 2) + q;
        rv[q] = slot[pp * 128 + rn];
        if (flags & WF_VIEWCOLS) {
#pragma unroll
          for (int c = 0; c < 3; ++c) rs[3 * q + c] = slot[S * (128 + KW) + 3 * pp + c];
        }
      }
    }
    if (flags & WF_ALPHA) {
#pragma unroll
      for (int pp = 0; pp < S; ++pp) {
        ra_da[pp] = slot[S * (128 + KW) + pp];
        ra_in[pp] = slot[S * 128 + pp * KW + tid];
      }
    }
  };
  auto riders_add = [&]() {
    if (flags & (WF_BIAS | WF_VIEWCOLS)) {
#pragma unroll
      for (int q = 0; q < S / 2; ++q) {
        bias_acc += rv[q];
        if (flags & WF_VIEWCOLS) {
#pragma unroll
          for (int c = 0; c < 3; ++c) vc[c] = fmaf(rv[q], rs[3 * q + c], vc[c]);
        }
      }
    }
    if (flags & WF_ALPHA) {
#pragma unroll
      for (int pp = 0; pp < S; ++pp) {
        alpha_acc = fmaf(ra_da[pp], ra_in[pp], alpha_acc);
        dal_acc += ra_da[pp];
      }
    }
  };

  // ---- pipeline.  Stages past the chunk are issued too: zeros, no memory traffic, and the vmcnt stays a
  // compile-time constant.
  const int ns = (npts + S - 1) / S;
#pragma unroll
  for (int st = 0; st < D; ++st) issue(st, st);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
  w2_barrier();
  Frag f0, f1;
  read_frag(f0, 0, 0);
  int sl = 0;
  for (int st = 0; st < ns; ++st) {
    const int sl1 = sl + 1 == D ? 0 : sl + 1;
    read_frag(f1, sl, 1);
    riders_read(sl);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(f0);
    __builtin_amdgcn_sched_barrier(0);
    riders_add();
    // stage st+1: this wave's pieces have landed (D-2 younger stages stay in flight) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NI) : "memory");
    w2_barrier();                                 // ... and so have everybody's; slot sl is read out
    issue(st + D, sl);
    read_frag(f0, sl1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_group(f1);
    sl = sl1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // nothing may land after the workgroup is gone / on the scratch below

  // ---- write the partial: MFMA row m of tile t = feature 2m + t, column r of tile u = input NKT r + u
  const bool vec_ok = NKT == 4 && (jb.ld & 3) == 0 && (jb.kcol0 & 3) == 0 && (jb.w_off & 3) == 0 &&
                      (reinterpret_cast<unsigned long long>(out) & 15) == 0 && jb.kvalid == 256;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = (i & 3) + 8 * (i >> 2) + 4 * hh;
      const int n = jb.n_base + nsub + 2 * m + t;
      if (n < jb.n_rows) {
        float* dst = out + jb.w_off + (size_t)n * jb.ld + jb.kcol0 + k0 + NKT * r;
        if (NKT == 4 && vec_ok) {
          const f32x4 v = {acc[t][0][i], acc[t][NKT > 1 ? 1 : 0][i], acc[t][NKT > 2 ? 2 : 0][i], acc[t][NKT > 3 ? 3 : 0][i]};
          *reinterpret_cast<f32x4*>(dst) = v;
        } else {
#pragma unroll
          for (int u = 0; u < NKT; ++u)
            if (k0 + NKT * r + u < jb.kvalid) dst[u] = acc[t][u][i];
        }
      }
    }
  // riders: the two point halves meet in LDS
  if (flags & (WF_BIAS | WF_ALPHA | WF_VIEWCOLS)) {
    w2_barrier();                                  // all fragment reads done: the ring is free
    float* red = lds;                              // [2 point halves][128 bias | 384 view cols]
    red[rph * 512 + rn] = bias_acc;
    if (flags & WF_VIEWCOLS) {
#pragma unroll
      for (int c = 0; c < 3; ++c) red[rph * 512 + 128 + rn * 3 + c] = vc[c];
    }
    w2_barrier();
    if ((jb.flags & WF_BIAS) && tid < 128 && jb.n_base + tid < jb.n_rows)
      out[jb.b_off + jb.n_base + tid] = red[tid] + red[512 + tid];
    if (flags & WF_ALPHA) {
      out[jb.aux_off + tid] = alpha_acc;
      if (tid == 0) out[jb.aux_off + 256] = dal_acc;
    }
    if ((flags & WF_VIEWCOLS) && tid < 128 && jb.n_base + tid < jb.n_rows) {
#pragma unroll
      for (int c = 0; c < 3; ++c)
        out[jb.w_off + (size_t)(jb.n_base + tid) * jb.ld + 256 + c] = red[128 + tid * 3 + c] + red[512 + 128 + tid * 3 + c];
    }
  }
}

// rgb head: dW_r[c][k] = sum_pt g[pt][c] * hv[pt][k], db_r[c] = sum_pt g[pt][c].  A thread owns 4 columns
// (one 16-byte load per point) of every 8th point, four points in flight: this job is pure load latency.
__device__ __forceinline__ void wgrad2_rgb_job(const Wgrad2Net& a, const Wgrad2Job& jb, float* lds, int c0,
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

__global__ __launch_bounds__(256, 2) void mlp_wgrad2_kernel(Wgrad2Args aa) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const bool second = (int)blockIdx.x >= aa.gx0;                  // wave-uniform: scalar selects
  const Wgrad2Net& a = second ? aa.net[1] : aa.net[0];
  const int bx = (int)blockIdx.x - (second ? aa.gx0 : 0);
  Wgrad2Job jb = aa.jobs[blockIdx.y];
  jb.dz_off = acts_slot_off(a.P, jb.dz_slot) + jb.n_base;
  jb.in_off = jb.in_slot >= 0 ? acts_slot_off(a.P, jb.in_slot) : acts_emb_off(a.P);
  const int c0 = bx * aa.chunk;
  const int c1 = min(a.P, c0 + aa.chunk);
  float* out = a.partial + (size_t)bx * N_PARAM_FLOATS;
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

// sum of the per-chunk partials of one or two networks (blocks [0, WGRAD_REDUCE_BLOCKS) -> network 0)
struct Reduce2Args {
  const float* partial[2];
  float* grad[2];
  int nchunks[2];
};
__global__ static void wgrad2_reduce_pair_kernel(Reduce2Args r) {
  const bool second = blockIdx.x >= WGRAD_REDUCE_BLOCKS;
  const int i = (blockIdx.x - (second ? WGRAD_REDUCE_BLOCKS : 0)) * 256 + threadIdx.x;
  if (i >= N_PARAM_FLOATS / 4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(second ? r.partial[1] : r.partial[0]) + i;
  const int nchunks = second ? r.nchunks[1] : r.nchunks[0];
  constexpr size_t ST = N_PARAM_FLOATS / 4;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;     // the order of wgrad_reduce4_kernel
  int c = 0;
  for (; c + 4 <= nchunks; c += 4) {
    s0 += p[(size_t)c * ST]; s1 += p[(size_t)(c + 1) * ST];
    s2 += p[(size_t)(c + 2) * ST]; s3 += p[(size_t)(c + 3) * ST];
  }
  for (; c < nchunks; ++c) s0 += p[(size_t)c * ST];
  reinterpret_cast<f32x4*>(second ? r.grad[1] : r.grad[0])[i] = (s0 + s1) + (s2 + s3);
}

// Chunks of points per launch.  Two workgroups share a CU; per chunk there are 17 half-layer workgroups of
// equal length (8 layers x 2 halves + the 128-row views layer) plus five short ones (four embedding
// columns blocks, the rgb head) worth about 1.5 more.  The chunk count fills k whole "rounds" of the
// 2 x CUs slots with chunks near `target_pts` points.
static constexpr int W2_TARGET_PTS = 2400;   // ring depth 2..6 and chunks of 1,600..3,600 points measure within 1 %
int pick_chunks_v2(int P) {
  const double slots = 2.0 * device_cus();
  const int target = W2_TARGET_PTS;
  long k = (long)((double)P * 18.5 / (slots * target) + 0.5);
  if (k < 1) k = 1;
  // one round (the 128 rays per GPU of a strongly scaled batch): fewer, longer chunks, so that the light jobs are
  // resident beside the heavy ones instead of queueing behind them (per chunk 20: 1.002 ms per graphed 128-ray
  // step against 1.006 with 18.5 and 1.085 with 17); several rounds: the 17 heavy workgroups of a chunk alone fill
  // the slots and the light ones slip into the gaps (1024 rays: 6.92 ms against 7.00 with 18.5, 7.07 with 20)
  const double per_chunk = k == 1 ? 20.0 : 17.0;
  long n = (long)(slots * k / per_chunk);
  const long nmax = P / 256 > 1 ? P / 256 : 1;
  if (n > nmax) n = nmax;
  if (n < 1) n = 1;
  if (n > 1024) n = 1024;
  return (int)n;
}

// chunking of a joint launch over two networks: one chunk length for both, chosen so that the JOINT chunk
// count fills whole rounds of two workgroups per CU
void wgrad2_joint_chunking(const int* P, int& chunk, int& gx0, int& gx1) {
  const long Pt = (long)P[0] + P[1];
  const int nchunks = pick_chunks_v2((int)Pt);
  chunk = (int)((Pt + nchunks - 1) / nchunks);
  chunk = (chunk + W2_PT - 1) / W2_PT * W2_PT;
  gx0 = (P[0] + chunk - 1) / chunk;
  gx1 = (P[1] + chunk - 1) / chunk;
}

// fills the job table (identical for every network: offsets are slots) and returns the job count
static int build_wgrad2_jobs(Wgrad2Args& w) {
  int off[N_PARAM_TENSORS + 1];
  param_offsets(off);
  int nj = 0;
  constexpr int EMB_SLOT = -1;
  auto add = [&](int dzs, int ins, int in_stride, int kw, int nbase, int nrows, int woff, int ld, int kcol0, int kvalid,
                 int boff, int flags, int aux) {
    Wgrad2Job& j = w.jobs[nj++];
    j.dz_off = 0; j.in_off = 0; j.dz_slot = dzs; j.in_slot = ins; j.in_stride = in_stride; j.kw = kw; j.n_base = nbase;
    j.n_rows = nrows; j.w_off = woff; j.ld = ld; j.kcol0 = kcol0; j.kvalid = kvalid; j.b_off = boff; j.flags = flags;
    j.aux_off = aux;
  };
  // long jobs first, short last (tail filling)
  for (int l = 1; l <= 7; ++l) {
    const int ld = l == 5 ? 313 : 256, kc0 = l == 5 ? 57 : 0;
    for (int h = 0; h < 2; ++h)
      add(l, l - 1, 256, 256, 128 * h, 256, off[2 * l], ld, kc0, 256, off[2 * l + 1], WF_BIAS, 0);
  }
  for (int h = 0; h < 2; ++h)   // the alpha head rides on ONE half only (it needs the whole input row, not dZ)
    add(SLOT_FEAT, 7, 256, 256, 128 * h, 256, off[18], 256, 0, 256, off[19],
        WF_BIAS | (h == 0 ? WF_ALPHA : 0), off[20]);
  add(SLOT_VIEWS_H, SLOT_FEAT, 256, 256, 0, 128, off[16], 259, 0, 256, off[17], WF_BIAS | WF_VIEWCOLS, 0);
  for (int h = 0; h < 2; ++h) {
    add(0, EMB_SLOT, 64, 64, 128 * h, 256, off[0], 57, 0, 57, off[1], WF_BIAS, 0);
    add(5, EMB_SLOT, 64, 64, 128 * h, 256, off[10], 313, 0, 57, 0, 0, 0);
  }
  add(0, SLOT_VIEWS_H, 256, 0, 0, 0, off[22], 128, 0, 0, off[23], WF_RGB, 0);
  w.njobs = nj;
  return nj;
}

}  // namespace scade

using namespace scade;

static int wgrad2_set_attr() {
  static unsigned long long attr_set = 0;   // one bit per device ordinal
  if (scade_attr_needed(attr_set)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wgrad2_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS_BYTES);
    SCADE_REQUIRE(e == hipSuccess, (int)e, "scade_mlp_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    scade_attr_done(attr_set);
  }
  return 0;
}

// wgrad + reduce (shared by the exact backward and the split-precision backward's exact-wgrad mode)
int scade_launch_wgrad(const float* acts, const float* dz, const float* g_out, int P, float* partial,
                       float* grad_flat, hipStream_t s) {
  if (int e = wgrad2_set_attr()) return e;
  Wgrad2Args w{};
  build_wgrad2_jobs(w);
  const int nchunks = pick_chunks_v2(P);
  int chunk = (P + nchunks - 1) / nchunks;
  chunk = (chunk + W2_PT - 1) / W2_PT * W2_PT;
  const int grid_x = (P + chunk - 1) / chunk;
  w.net[0] = Wgrad2Net{acts, dz, g_out, partial, P};
  w.chunk = chunk; w.gx0 = grid_x;
  hipLaunchKernelGGL(mlp_wgrad2_kernel, dim3(grid_x, w.njobs), dim3(256), W2_LDS_BYTES, s, w);
  if (int e = scade_check_launch("scade_mlp_bwd(wgrad)")) return e;
  hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(WGRAD_REDUCE_BLOCKS), dim3(256), 0, s, partial, grid_x, grad_flat);
  return scade_check_launch("scade_mlp_bwd(reduce)");
}

// the same for TWO networks in one weight-gradient launch and one reduce launch (scade_mlp_bwd2)
int scade_launch_wgrad2(const float* const* acts, const float* const* dz, const float* const* g_out, const int* P,
                        float* const* partial, float* const* grad_flat, hipStream_t s, ReduceDesc* defer) {
  if (int e = wgrad2_set_attr()) return e;
  Wgrad2Args w{};
  build_wgrad2_jobs(w);
  int chunk, gx0, gx1;     // the partial buffers are sized for this by scade_mlp_bwd2_workspace_floats
  wgrad2_joint_chunking(P, chunk, gx0, gx1);
  w.net[0] = Wgrad2Net{acts[0], dz[0], g_out[0], partial[0], P[0]};
  w.net[1] = Wgrad2Net{acts[1], dz[1], g_out[1], partial[1], P[1]};
  w.chunk = chunk; w.gx0 = gx0;
  hipLaunchKernelGGL(mlp_wgrad2_kernel, dim3(gx0 + gx1, w.njobs), dim3(256), W2_LDS_BYTES, s, w);
  if (int e = scade_check_launch("scade_mlp_bwd2(wgrad)")) return e;
  if (defer) {          // the partial rows are summed by scade_step_finish, inside the optimizer's launch
    *defer = ReduceDesc{{partial[0], partial[1]}, {gx0, gx1}, {}};
    return 0;
  }
  Reduce2Args r{{partial[0], partial[1]}, {grad_flat[0], grad_flat[1]}, {gx0, gx1}};
  hipLaunchKernelGGL(wgrad2_reduce_pair_kernel, dim3(2 * WGRAD_REDUCE_BLOCKS), dim3(256), 0, s, r);
  return scade_check_launch("scade_mlp_bwd2(reduce)");
}

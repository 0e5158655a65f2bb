// Does a weight k-block that reaches a CU ONCE (staged through LDS by the LDS-DMA engine, read by all eight waves
// of one 256-point workgroup) make the 16-bit k-loop faster than today's per-wave fragment stream from L2?
// (the k-loops of mlp_fwd_lp_kernel / mlp_dgrad_lp_kernel with everything else taken away.)
//
// A synthetic trunk of NLAY 256 x 256 bf16 layers in the REAL packed fragment layout (mlp_tile_lp.h: per layer
// [n-tile 8][k-block 16][64 lanes][8 elements] = 128 KiB; NLAY distinct layers = the 1 MiB the real kernels keep
// L2-resident), the activation tile in LDS with the real swizzle, the real MFMA (v_mfma_f32_32x32x16_bf16), every
// wave = 64 output rows x 128 points (8 MFMAs per k-block, the real register blocking), one LDS-only barrier pair
// per layer where the real kernels have theirs, NO epilogue arithmetic (accumulators are pinned, not converted):
//   mode 0  today's shape: two 4-wave workgroups per CU (128 points each), A fragments per wave from L2 three
//           k-blocks ahead in four rotating register sets - the REAL layer_gemm_lp, included from the product source
//   mode 1  one 8-wave workgroup per CU (256 points), A per wave from L2 as in mode 0 (waves w and w + 4 read the
//           same fragments in lock-step: an L1 hit for the second if the line survives) - round 5's lock-step variant
//   mode 2  one 8-wave workgroup per CU (256 points), weight k-blocks through a ring of R 8-KiB LDS slots filled by
//           buffer_load_dwordx4 ... lds (wave w fetches n-tile w's 1 KiB of k-block kb + R - 1: ONE vector-memory
//           instruction per wave and k-block, no staging registers), one LDS-only barrier per k-block, A fragments by
//           ds_read_b128 from the slot (lane-linear: conflict free)
// Printed: time per tile-layer, MFMA rate against the 2.5 PFLOP/s peak.  Under rocprofv3 --pmc TCP_TCC_READ_REQ_sum the
// same binary gives the L2 -> CU request count of every mode (tools/pmc_kloop_wlds.sh).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I scade_amd/csrc tools/probe_kloop_wlds.hip -o tools/scratch/probe_kloop_wlds
#include "mlp_tile_lp.h"
#include <cstdio>
#include <cstdlib>
void scade_set_error(const char*, ...) {}
int scade_check_launch(const char*) { return 0; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace scade;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __bf16 V8 __attribute__((ext_vector_type(8)));
constexpr int NLAY = 8;
constexpr long LAYER_ELEMS = 8L * 16 * 64 * 8;          // 65536 bf16 = 128 KiB

struct Args { const __bf16* w; float* out; int tiles; unsigned long long* trace; };   // trace: [64 workgroups][8 waves][4]

__device__ __forceinline__ void pin(f32x16 (&acc)[2][4]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int p = 0; p < 4; ++p) asm volatile("" ::"v"(acc[t][p]));
}

// modes 0 / 1: NW waves per workgroup (4: 128 points, 8: 256 points), the real gemm
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_l2(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
  __bf16* x = reinterpret_cast<__bf16*>(lds16);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, pg = wave >> 2;
  for (int i = tid; i < NW * 32 * W; i += NW * 64) x[i] = (__bf16)(0.001f * (float)((i * 7) & 255));
  __syncthreads();
  const __bf16* xw = x + pg * 128 * W;
  f32x16 acc[2][4];
  AFragN<true, 4> A;
  auto wl = [&](int l) { return reinterpret_cast<const V8*>(a.w + (long)(l % NLAY) * LAYER_ELEMS) + (2 * rg) * 16 * 64; };
  A.s[0].t0 = wl(0)[lane]; A.s[0].t1 = wl(0)[16 * 64 + lane];
  A.s[1].t0 = wl(0)[64 + lane]; A.s[1].t1 = wl(0)[16 * 64 + 64 + lane];
  A.s[2].t0 = wl(0)[128 + lane]; A.s[2].t1 = wl(0)[16 * 64 + 128 + lane];
  unsigned long long tk = 0, tb = 0;        // core clocks inside the k-loops / at the two layer barriers
  for (int tile = 0; tile < a.tiles; ++tile) {
#pragma unroll 1
    for (int l = 0; l < NLAY; ++l) {
      const unsigned long long t0 = __builtin_amdgcn_s_memtime();
      layer_gemm_lp<true, 2, 0, 16, false, 0, 4, 4>(acc, A, wl(l), wl(l + 1), 16, xw, xw, lane, nullptr);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      LP_SYNC();
      pin(acc);
      LP_SYNC();
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      tk += t1 - t0; tb += t2 - t1;
    }
  }
  if (a.trace && blockIdx.x < 64 && lane == 0) {
    unsigned long long* o = a.trace + ((size_t)blockIdx.x * 8 + wave) * 4;
    o[0] = tk; o[1] = tb; o[2] = 0; o[3] = (unsigned long long)a.tiles * NLAY;
  }
  if (acc[0][0][0] == 12345.678f) a.out[0] = acc[1][3][5];
}

// mode 2: weights through an LDS ring
template <int R>
__global__ __launch_bounds__(512) void k_lds(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
  __bf16* x = reinterpret_cast<__bf16*>(lds16);
  unsigned char* ring = reinterpret_cast<unsigned char*>(lds16) + 256 * W * 2;     // R slots of 8 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, pg = wave >> 2;
  for (int i = tid; i < 256 * W; i += 512) x[i] = (__bf16)(0.001f * (float)((i * 7) & 255));
  __syncthreads();
  const __bf16* xw = x + pg * 128 * W;
  const int r = lane & 31, hh = lane >> 5;
  const unsigned long long wp = reinterpret_cast<unsigned long long>(a.w);
  const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)wp), whi = __builtin_amdgcn_readfirstlane((unsigned)(wp >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)whi << 32) | wlo), 0, (unsigned)(NLAY * LAYER_ELEMS * 2), 0x00020000);
  // global k-block index g = 16 layer + kb; this wave fetches n-tile `wave` of block g into slot g % R
  const int total = a.tiles * NLAY * 16;
  auto issue = [&](int g) {
    if (g < total) {
      const int l = (g >> 4) % NLAY, kb = g & 15;
      const int soff = __builtin_amdgcn_readfirstlane((int)((l * LAYER_ELEMS + ((long)wave * 16 + kb) * 512) * 2));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(ring + (g % R) * 8192 + wave * 1024), 16, lane * 16, soff, 0, 0);
    }
  };
#pragma unroll
  for (int g = 0; g < R - 1; ++g) issue(g);
  f32x16 acc[2][4];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int g = 0;
  for (int tile = 0; tile < a.tiles; ++tile) {
#pragma unroll 1
    for (int l = 0; l < NLAY; ++l) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[t][p] = zero16;
      V8 b0, b1, b2, b3;
#define LOAD_B(KBX, PX, B) B = *reinterpret_cast<const V8*>(xw + x_idx((PX)*32 + r, 2 * (KBX) + hh));
      LOAD_B(0, 0, b0) LOAD_B(0, 1, b1) LOAD_B(0, 2, b2) LOAD_B(0, 3, b3)
#pragma unroll 2
      for (int kb = 0; kb < 16; ++kb, ++g) {
        // block g's 1 KiB of every wave has landed and block g - 1's slot has no reader left
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 2) : "memory");
        lds_barrier();
        issue(g + R - 1);
        const unsigned char* slot = ring + (g % R) * 8192;
        const V8 a0 = *reinterpret_cast<const V8*>(slot + (2 * rg) * 1024 + lane * 16);
        const V8 a1 = *reinterpret_cast<const V8*>(slot + (2 * rg + 1) * 1024 + lane * 16);
        const int kn = kb + 1 < 16 ? kb + 1 : kb;
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        LOAD_B(kn, 0, b0)
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        LOAD_B(kn, 1, b1)
        __builtin_amdgcn_sched_barrier(0);
        acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[0][2], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[1][2], 0, 0, 0);
        LOAD_B(kn, 2, b2)
        __builtin_amdgcn_sched_barrier(0);
        acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b3, acc[0][3], 0, 0, 0);
        acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[1][3], 0, 0, 0);
        LOAD_B(kn, 3, b3)
        __builtin_amdgcn_sched_barrier(0);
      }
#undef LOAD_B
      LP_SYNC();
      pin(acc);
      LP_SYNC();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0][0][0] == 12345.678f) a.out[0] = acc[1][3][5];
}

// modes 4 / 5: the ring with the A fragments read ONE k-block ahead (the barrier of block g also covers block g + 1's
// DMA, so block g + 1's fragments are fetched from LDS under block g's MFMAs) - PAIR = false: one barrier per k-block,
// four 8-KiB slots; PAIR = true: one barrier per TWO k-blocks, two 16-KiB slots (the verdict's "two-slot ring of 16 KB
// k-blocks"), the fragments of a pair's first block read right behind the barrier
template <bool PAIR>
__global__ __launch_bounds__(512) void k_lds2(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds16[];
  __bf16* x = reinterpret_cast<__bf16*>(lds16);
  unsigned char* ring = reinterpret_cast<unsigned char*>(lds16) + 256 * W * 2;     // 32 KiB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 3, pg = wave >> 2;
  for (int i = tid; i < 256 * W; i += 512) x[i] = (__bf16)(0.001f * (float)((i * 7) & 255));
  __syncthreads();
  const __bf16* xw = x + pg * 128 * W;
  const int r = lane & 31, hh = lane >> 5;
  const unsigned long long wp = reinterpret_cast<unsigned long long>(a.w);
  const unsigned wlo = __builtin_amdgcn_readfirstlane((unsigned)wp), whi = __builtin_amdgcn_readfirstlane((unsigned)(wp >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)whi << 32) | wlo), 0, (unsigned)(NLAY * LAYER_ELEMS * 2), 0x00020000);
  const int total = a.tiles * NLAY * 16;
  auto issue = [&](int g) {                 // block g -> slot g % 4 (8 KiB each; a pair = two neighbouring slots)
    if (g < total) {
      const int l = (g >> 4) % NLAY, kb = g & 15;
      const int soff = __builtin_amdgcn_readfirstlane((int)((l * LAYER_ELEMS + ((long)wave * 16 + kb) * 512) * 2));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(ring + (g & 3) * 8192 + wave * 1024), 16, lane * 16, soff, 0, 0);
    }
  };
  auto read_a = [&](int g, V8& a0, V8& a1) {
    const unsigned char* slot = ring + (g & 3) * 8192;
    a0 = *reinterpret_cast<const V8*>(slot + (2 * rg) * 1024 + lane * 16);
    a1 = *reinterpret_cast<const V8*>(slot + (2 * rg + 1) * 1024 + lane * 16);
  };
  f32x16 acc[2][4];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  V8 a0, a1, n0, n1;
  unsigned long long tw = 0, tc = 0, tb = 0;   // core clocks: waiting for the slot (vmcnt + barrier) | issue + reads + MFMAs | layer barriers
  if (PAIR) {
    issue(0); issue(1);
  } else {
    issue(0); issue(1); issue(2);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    lds_barrier();
    read_a(0, a0, a1);
  }
  int g = 0;
  for (int tile = 0; tile < a.tiles; ++tile) {
#pragma unroll 1
    for (int l = 0; l < NLAY; ++l) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[t][p] = zero16;
      V8 b0, b1, b2, b3;
#define LOAD_B(KBX, PX, B) B = *reinterpret_cast<const V8*>(xw + x_idx((PX)*32 + r, 2 * (KBX) + hh));
      LOAD_B(0, 0, b0) LOAD_B(0, 1, b1) LOAD_B(0, 2, b2) LOAD_B(0, 3, b3)
#pragma unroll 2
      for (int kb = 0; kb < 16; ++kb, ++g) {
        if (PAIR) {
          if ((kb & 1) == 0) {              // pair g / 2: both blocks landed, the other pair slot has no reader left
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            issue(g + 2); issue(g + 3);
            read_a(g, a0, a1);
            read_a(g + 1, n0, n1);
          }
        } else {
          // blocks <= g + 1 landed everywhere; block g - 1's slot is free (its fragments were read an iteration ago)
          const unsigned long long s0 = __builtin_amdgcn_s_memtime();
          asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
          lds_barrier();
          const unsigned long long s1 = __builtin_amdgcn_s_memtime();
          tw += s1 - s0; tc -= s1;
          issue(g + 3);
          read_a(g + 1, n0, n1);
        }
        const int kn = kb + 1 < 16 ? kb + 1 : kb;
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        LOAD_B(kn, 0, b0)
        __builtin_amdgcn_sched_barrier(0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        LOAD_B(kn, 1, b1)
        __builtin_amdgcn_sched_barrier(0);
        acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[0][2], 0, 0, 0);
        acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[1][2], 0, 0, 0);
        LOAD_B(kn, 2, b2)
        __builtin_amdgcn_sched_barrier(0);
        acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b3, acc[0][3], 0, 0, 0);
        acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[1][3], 0, 0, 0);
        LOAD_B(kn, 3, b3)
        __builtin_amdgcn_sched_barrier(0);
        if (!PAIR) tc += __builtin_amdgcn_s_memtime();
        if (!PAIR || (kb & 1) == 0) { a0 = n0; a1 = n1; }
      }
#undef LOAD_B
      const unsigned long long e0 = __builtin_amdgcn_s_memtime();
      LP_SYNC();
      pin(acc);
      LP_SYNC();
      tb += __builtin_amdgcn_s_memtime() - e0;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!PAIR && a.trace && blockIdx.x < 64 && lane == 0) {
    unsigned long long* o = a.trace + ((size_t)blockIdx.x * 8 + wave) * 4;
    o[0] = tc; o[1] = tb; o[2] = tw; o[3] = (unsigned long long)a.tiles * NLAY;
  }
  if (acc[0][0][0] == 12345.678f) a.out[0] = acc[1][3][5];
}

template <typename F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  int cus = 256; CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  __bf16* w; float* o;
  CK(hipMalloc(&w, NLAY * LAYER_ELEMS * 2)); CK(hipMalloc(&o, 64)); CK(hipMemset(o, 0, 64));
  { // small finite values
    unsigned short* h = (unsigned short*)malloc(NLAY * LAYER_ELEMS * 2);
    for (long i = 0; i < NLAY * LAYER_ELEMS; ++i) h[i] = (unsigned short)(0x3c00 + (i * 2654435761u >> 26));   // bf16 ~ 0.0078 .. 0.0081
    CK(hipMemcpy(w, h, NLAY * LAYER_ELEMS * 2, hipMemcpyHostToDevice)); free(h);
  }
  unsigned long long* trace; CK(hipMalloc(&trace, 64 * 8 * 4 * 8)); CK(hipMemset(trace, 0, 64 * 8 * 4 * 8));
  auto print_trace = [&](const char* what, int waves) {
    unsigned long long h[64 * 8 * 4];
    CK(hipMemcpy(h, trace, sizeof(h), hipMemcpyDeviceToHost));
    double k = 0, b = 0, w = 0, n = 0; int cnt = 0;
    for (int g = 0; g < 64; ++g) for (int wv = 0; wv < waves; ++wv) {
      const unsigned long long* o = h + (g * 8 + wv) * 4;
      if (!o[3]) continue;
      k += (double)o[0] / o[3]; b += (double)o[1] / o[3]; w += (double)o[2] / o[3]; n += 1; ++cnt;
    }
    if (cnt) printf("   %s: core clocks per layer of one wave (mean of %d waves): in the k-loop %.0f (its own MFMAs: 4096, its SIMD partner's: 4096)"
                    ", of which waiting for the slot %.0f; at the two layer barriers %.0f\n", what, cnt, (k + w) / n, w / n, b / n);
    CK(hipMemset(trace, 0, sizeof(h)));
  };
  const int tiles = 12;                                   // 256-point tiles per CU (mode 0: per workgroup of 128 points)
  const double flop_cu = 2.0 * 256 * 256 * 256 * NLAY * tiles;   // per CU
  auto report = [&](const char* name, double ms) {
    const double tfl = flop_cu * cus / (ms * 1e-3) / 1e12;
    printf("%-78s %8.1f us  %7.1f TFLOP/s  %.3f of 2.5 PF  %6.2f us per 256-point tile-layer\n", name, ms * 1e3, tfl, tfl / 2500.0,
           ms * 1e3 / (NLAY * tiles));
  };
  Args a{w, o, tiles, trace};
  if (only < 0 || only == 0) {
    const int ldsb = 128 * W * 2 + 16384;                 // 80 KiB: two workgroups per CU, as the real kernels
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_l2<4>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    report("mode 0: 2 x 4 waves per CU, 128-point tiles, A per wave from L2 (real layer_gemm_lp)",
           time_ms([&] { hipLaunchKernelGGL(k_l2<4>, dim3(2 * cus), dim3(256), ldsb, 0, a); }, 10));
    print_trace("mode 0", 4);
  }
  if (only < 0 || only == 1) {
    const int ldsb = 256 * W * 2 + 32768;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_l2<8>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    report("mode 1: 1 x 8 waves per CU, 256-point tile, A per wave from L2, wave pairs in lock-step",
           time_ms([&] { hipLaunchKernelGGL(k_l2<8>, dim3(cus), dim3(512), ldsb, 0, a); }, 10));
    print_trace("mode 1", 8);
  }
  if (only < 0 || only == 2) {
    const int ldsb = 256 * W * 2 + 4 * 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    report("mode 2: 1 x 8 waves per CU, 256-point tile, weight k-blocks through a 4-slot LDS ring (LDS-DMA)",
           time_ms([&] { hipLaunchKernelGGL(k_lds<4>, dim3(cus), dim3(512), ldsb, 0, a); }, 10));
  }
  if (only < 0 || only == 3) {
    const int ldsb = 256 * W * 2 + 2 * 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds<2>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    report("mode 3: the same with a 2-slot ring (what fits beside the view pad in layers 6, 7, feature)",
           time_ms([&] { hipLaunchKernelGGL(k_lds<2>, dim3(cus), dim3(512), ldsb, 0, a); }, 10));
  }
  if (only < 0 || only == 4) {
    const int ldsb = 256 * W * 2 + 4 * 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds2<false>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    report("mode 4: 4-slot ring, A fragments read from LDS one k-block ahead, one barrier per k-block",
           time_ms([&] { hipLaunchKernelGGL(k_lds2<false>, dim3(cus), dim3(512), ldsb, 0, a); }, 10));
    print_trace("mode 4", 8);
  }
  if (only < 0 || only == 5) {
    const int ldsb = 256 * W * 2 + 4 * 8192;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds2<true>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
    report("mode 5: two 16-KiB pair slots, one barrier per TWO k-blocks",
           time_ms([&] { hipLaunchKernelGGL(k_lds2<true>, dim3(cus), dim3(512), ldsb, 0, a); }, 10));
  }
  CK(hipDeviceSynchronize());
  return 0;
}

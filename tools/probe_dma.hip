// What the memory system gives the 8-bit weight gradient's ACCESS PATTERN through LDS-DMA (no compute): one
// 512-thread workgroup per CU streams TWO private arrays of ROWB-byte rows in stages of 32 rows into a ring of D
// LDS slots with buffer_load_dwordx4 ... lds (1 KiB per wave instruction), one LDS-only barrier per stage - the
// skeleton of wgrad_lp8_dma_job (mlp_bwd_lp.hip).  Variants: row bytes 256 / 512, source swizzle on / off,
// nt on / off, ring depth, contiguous chunk per workgroup vs all workgroups sweeping the arrays together.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dma.hip -o tools/scratch/probe_dma && tools/scratch/probe_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args { const unsigned char* a; const unsigned char* b; long rows; long job_stride; int G; int spw; int inter; unsigned* out; };

__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
  const unsigned long long p = reinterpret_cast<unsigned long long>(base);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)p), hi = __builtin_amdgcn_readfirstlane((unsigned)(p >> 32));
  void* q = reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int ROWB, int D, bool SWZ, int AUX>
__global__ __launch_bounds__(512) void k_dma(Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int SLOT = 2 * 32 * ROWB;
  constexpr int RPI = 1024 / ROWB;                  // rows per wave instruction
  constexpr int NI = 2 * (32 / RPI) / 8;            // instructions per wave and stage
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = blockIdx.x, job = blockIdx.y;
  // this workgroup's rows: [r0, r0 + 32 spw) contiguous, or interleaved stages
  const long first = p.inter ? 0 : (long)w * p.spw * 32;
  const rsrc_t ra = make_rsrc(p.a + (long)job * p.job_stride + first * ROWB, (unsigned)((p.rows - first) * ROWB));
  const rsrc_t rb = make_rsrc(p.b + (long)job * p.job_stride + first * ROWB, (unsigned)((p.rows - first) * ROWB));
  constexpr int LPR = ROWB / 16;                    // lanes per row
  const int lrow = lane / LPR, pc = lane % LPR;
  auto issue = [&](int st, int sl) {
    unsigned char* slot = lds + sl * SLOT;
    const long srow = p.inter ? ((long)st * p.G + w) * 32 : (long)st * 32;
#pragma unroll
    for (int e = 0; e < 32 / RPI / 8; ++e) {
      const int row = (e * 8 + wave) * RPI;         // first row of this instruction inside the stage
      const int r = row + lrow;
      const int voff = lrow * ROWB + ((SWZ ? (pc ^ (2 * (r & 7))) & (LPR - 1) : pc) << 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(slot + row * ROWB), 16, voff, (int)((srow + row) * ROWB), 0, AUX);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(slot + 32 * ROWB + row * ROWB), 16, voff, (int)((srow + row) * ROWB), 0, AUX);
    }
  };
  unsigned acc = 0;
#pragma unroll
  for (int st = 0; st < D; ++st) issue(st, st);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * NI) : "memory");
  lds_barrier();
  int sl = 0;
  for (int st = 0; st < p.spw; ++st) {
    acc ^= *reinterpret_cast<const unsigned*>(lds + sl * SLOT + tid * 4);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NI) : "memory");
    lds_barrier();
    issue(st + D, sl);
    sl = sl + 1 == D ? 0 : sl + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345u) p.out[0] = acc;
}

__global__ void k_fill(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x;
  }
}
template <typename F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

template <int ROWB, int D, bool SWZ, int AUX>
static void run(const unsigned char* a, const unsigned char* b, unsigned* o, int G, int J, int inter) {
  const long rows = 262144;
  const long job_bytes = rows * ROWB;
  // G workgroups per job: together they cover the job's rows
  const int spw = (int)((rows / 32 + G - 1) / G);
  Args p{a, b, rows, job_bytes, G, spw, inter, o};
  const int ldsb = D * 2 * 32 * ROWB;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dma<ROWB, D, SWZ, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
  const double ms = time_ms([&] { hipLaunchKernelGGL((k_dma<ROWB, D, SWZ, AUX>), dim3(G, J), dim3(512), ldsb, 0, p); }, 5);
  const double bytes = 2.0 * job_bytes * J;
  printf("rows %3d B  D %d  swz %d  aux %d  G %3d x J %2d  %-11s  %7.1f us  %5.2f TB/s\n", ROWB, D, (int)SWZ, AUX, G, J,
         inter ? "interleaved" : "contiguous", ms * 1e3, bytes / ms / 1e9);
}

// MFMA burner: what the chip does right before the weight gradient in a train step (power / clock state)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_burn(float* out, int iters) {
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f - i * 0.01f); }
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  if (c0[0] + c1[0] + c2[0] + c3[0] == 12345.f) out[0] = c0[0];
}

int main() {
  const long bytes = 262144L * 512 * 10;
  unsigned char *a, *b; unsigned* o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 64)); CK(hipMemset(o, 0, 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)a, (size_t)bytes / 4, 1u);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)b, (size_t)bytes / 4, 77u);
  CK(hipDeviceSynchronize());
  // 256 workgroups in all (one per CU), split over J jobs
  for (int J : {1, 8}) {
    const int G = 256 / J;
    for (int inter = 0; inter < 2; ++inter) {
      run<256, 6, true, 2>(a, b, o, G, J, inter);
      run<256, 6, false, 2>(a, b, o, G, J, inter);
      run<256, 6, true, 0>(a, b, o, G, J, inter);
      run<256, 9, true, 2>(a, b, o, G, J, inter);
      run<256, 3, true, 2>(a, b, o, G, J, inter);
      run<512, 4, false, 2>(a, b, o, G, J, inter);
      run<512, 4, false, 0>(a, b, o, G, J, inter);
    }
  }
  // sustained / in-situ: the same launch 200 times back to back, and alternating with 0.4 ms of MFMA work
  {
    const long rows = 262144; const int G = 32, J = 8, spw = (int)((rows / 32 + G - 1) / G);
    Args p{a, b, rows, rows * 256, G, spw, 0, o};
    const int ldsb = 6 * 2 * 32 * 256;
    auto dma = [&] { hipLaunchKernelGGL((k_dma<256, 6, true, 2>), dim3(G, J), dim3(512), ldsb, 0, p); };
    const double bytes = 2.0 * rows * 256 * J;
    double ms = time_ms(dma, 200);
    printf("200 launches back to back: %7.1f us  %5.2f TB/s\n", ms * 1e3, bytes / ms / 1e9);
    float* fo; CK(hipMalloc(&fo, 64));
    auto burn = [&] { hipLaunchKernelGGL(k_burn, dim3(2048), dim3(256), 0, 0, fo, 3000); };
    const double tb = time_ms(burn, 20);
    printf("burner alone: %.1f us\n", tb * 1e3);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double tot = 0; int n = 0;
    for (int i = 0; i < 60; ++i) {
      burn(); burn();
      CK(hipEventRecord(e0)); dma(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (i >= 10) { tot += t; ++n; }
    }
    printf("behind 2 x %.0f us of MFMA work: %7.1f us  %5.2f TB/s\n", tb * 1e3, tot / n * 1e3, bytes / (tot / n) / 1e9);
  }
  return 0;
}

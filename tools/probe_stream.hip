// What the memory system gives the weight-gradient kernel's ACCESS PATTERN (no compute): one 512-thread
// workgroup per CU slot streams TWO private arrays of 512-byte rows in stages of 32 rows (16 KB + 16 KB per
// stage), D register stages in flight, one barrier per stage - as mlp_wgrad_lp_kernel does.  Variants:
//   contiguous   each workgroup owns one contiguous chunk of rows (the kernel's layout)
//   interleaved  stage s of workgroup w is global stage s * G + w (all workgroups sweep the arrays together)
//   jobs J       the workgroups are split over J different array pairs (the kernel's 10+ layer jobs)
//   hipcc --offload-arch=gfx950 -O3 tools/probe_stream.hip -o tools/scratch/probe_stream && tools/scratch/probe_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { const unsigned char* a; const unsigned char* b; long rows_per_job; long job_stride_bytes; int G; int stages_per_wg; int interleaved; u32x4* out; };

template <int D, bool BARRIER, bool NT>
__global__ __launch_bounds__(512) void k_stream(Args p) {
  __shared__ u32x4 lds[2 * 512 * 2];
  const int tid = threadIdx.x, cc = tid & 31, rr = tid >> 5;
  const int w = blockIdx.x, job = blockIdx.y;
  const unsigned char* a = p.a + (long)job * p.job_stride_bytes;
  const unsigned char* b = p.b + (long)job * p.job_stride_bytes;
  u32x4 r[D][4];
  u32x4 acc = {0, 0, 0, 0}, chk = {0, 0, 0, 0};
  auto row0 = [&](int s) -> long { return p.interleaved ? ((long)s * p.G + w) * 32 : ((long)w * p.stages_per_wg + s) * 32; };
  auto issue = [&](u32x4 (&q)[4], int s) {
    long r0 = row0(s); if (r0 + 32 > p.rows_per_job) r0 = p.rows_per_job - 32;
    const u32x4* pa = reinterpret_cast<const u32x4*>(a + (r0 + rr) * 512 + 16 * cc);
    const u32x4* pb = reinterpret_cast<const u32x4*>(b + (r0 + rr) * 512 + 16 * cc);
    if (NT) { q[0] = __builtin_nontemporal_load(pa); q[1] = __builtin_nontemporal_load(pa + 16 * 32); q[2] = __builtin_nontemporal_load(pb); q[3] = __builtin_nontemporal_load(pb + 16 * 32); }
    else { q[0] = pa[0]; q[1] = pa[16 * 32]; q[2] = pb[0]; q[3] = pb[16 * 32]; }
  };
#pragma unroll
  for (int d = 0; d < D; ++d) issue(r[d], d);
  for (int s = 0; s < p.stages_per_wg; s += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      // "commit": the stage's values go to LDS (as the kernel's commit does), then the slot is re-issued
      if (s + d < p.stages_per_wg) chk ^= r[d][0] ^ r[d][1] ^ r[d][2] ^ r[d][3];
      lds[(d & 1) * 1024 + tid] = r[d][0] ^ r[d][2]; lds[(d & 1) * 1024 + 512 + tid] = r[d][1] ^ r[d][3];
      issue(r[d], s + d + D);
      if (BARRIER) __syncthreads();
      acc ^= lds[(d & 1) * 1024 + (tid ^ 37)];
    }
  }
  if (acc[0] == 0x12345u) p.out[0] = acc;
  atomicXor(reinterpret_cast<unsigned*>(p.out) + 4, chk[0] ^ chk[1] ^ chk[2] ^ chk[3]);
}

__global__ void k_fill(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = x;
  }
}
__global__ void k_read(const u32x4* __restrict__ a, size_t n, u32x4* out) {
  u32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s ^= __builtin_nontemporal_load(a + i);
  if (s[0] == 0x12345u) out[0] = s;
}

template <typename F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) launch();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main() {
  // the kernel's fine + coarse launch: 262144 rows per slot, ~10 slots of 512-byte rows per operand
  const long rows = 262144; const int J = 10;
  const long job_bytes = rows * 512;
  unsigned char *a, *b; u32x4* o;
  CK(hipMalloc(&a, job_bytes * J)); CK(hipMalloc(&b, job_bytes * J)); CK(hipMalloc(&o, 64)); CK(hipMemset(o, 0, 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)a, (size_t)job_bytes * J / 4, 1u);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)b, (size_t)job_bytes * J / 4, 77u);
  CK(hipDeviceSynchronize());
  { double ms = time_ms([&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, 0, (const u32x4*)a, (size_t)job_bytes * J / 16, o); }, 5);
    printf("control: grid-stride nt read of one array (%.2f GB): %.1f us  %.2f TB/s\n", job_bytes * J / 1e9, ms * 1e3, job_bytes * J / ms / 1e9); }
  const double bytes = 2.0 * job_bytes * J;
  for (int G : {86, 128, 256}) {
    const int spw = (int)((rows / 32 + G - 1) / G);
    for (int inter = 0; inter < 2; ++inter) {
      Args p{a, b, rows, job_bytes, G, spw, inter, o};
      const dim3 g(G, J), t(512);
      double ms;
#define RUN(D, BAR, NT, label) ms = time_ms([&] { hipLaunchKernelGGL((k_stream<D, BAR, NT>), g, t, 0, 0, p); }, 5); \
      { unsigned h[8]; CK(hipMemcpy(h, o, 32, hipMemcpyDeviceToHost)); CK(hipMemset(o, 0, 64)); \
      printf("G %3d x %d jobs (%4d stages/wg) %-11s %-28s %7.1f us  %5.2f TB/s  chk %08x\n", G, J, spw, inter ? "interleaved" : "contiguous", label, ms * 1e3, bytes / ms / 1e9, h[4]); }
      RUN(3, true, true, "3 stages, barrier, nt")
      RUN(3, true, false, "3 stages, barrier")
      RUN(3, false, true, "3 stages, no barrier, nt")
      RUN(6, true, true, "6 stages, barrier, nt")
      RUN(6, false, true, "6 stages, no barrier, nt")
    }
  }
  return 0;
}

// HBM roofs of the box this runs on, for the access patterns the training kernels use:
// 16-byte-per-lane streaming reads, writes (plain / non-temporal) and a 1:1 copy, grid-stride over 2 GiB.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_hbm.hip -o tools/scratch/probe_hbm && tools/scratch/probe_hbm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_read(const f32x4* __restrict__ a, size_t n, f32x4* out) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
  if (s[0] == 123.456f) out[0] = s;
}
__global__ void k_read_nt(const f32x4* __restrict__ a, size_t n, f32x4* out) {
  f32x4 s = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    s += __builtin_nontemporal_load(a + i);
  if (s[0] == 123.456f) out[0] = s;
}
__global__ void k_write(f32x4* __restrict__ a, size_t n) {
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = v;
}
__global__ void k_write_nt(f32x4* __restrict__ a, size_t n) {
  const f32x4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(v, a + i);
}
__global__ void k_copy(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_copy_nt(const f32x4* __restrict__ a, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), b + i);
}
// two reads per write (the weight-gradient pattern is read-only; fwd/dgrad are write-mostly; this is the mix
// of a dgrad and a wgrad kernel sharing the chip)
__global__ void k_r2w1(const f32x4* __restrict__ a, const f32x4* __restrict__ c, f32x4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(c + i), b + i);
}

template <typename F> static double time_ms(F launch, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const size_t bytes = 2ull << 30, n = bytes / 16;
  f32x4 *a, *b, *c, *o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&o, 64));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes)); CK(hipMemset(c, 3, bytes));
  for (int wg_per_cu : {4, 8, 16}) {
    const dim3 g(256 * wg_per_cu), t(256);
    const int R = 10;
    double ms;
    printf("--- %d workgroups of 256 threads per CU, 2 GiB per stream\n", wg_per_cu);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read, g, t, 0, 0, a, n, o); }, R);       printf("read          %7.3f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_read_nt, g, t, 0, 0, a, n, o); }, R);    printf("read nt       %7.3f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write, g, t, 0, 0, b, n); }, R);         printf("write         %7.3f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_write_nt, g, t, 0, 0, b, n); }, R);      printf("write nt      %7.3f ms  %6.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy, g, t, 0, 0, a, b, n); }, R);       printf("copy          %7.3f ms  %6.2f TB/s (read + write bytes)\n", ms, 2 * bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_copy_nt, g, t, 0, 0, a, b, n); }, R);    printf("copy nt       %7.3f ms  %6.2f TB/s (read + write bytes)\n", ms, 2 * bytes / ms / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_r2w1, g, t, 0, 0, a, c, b, n); }, R);    printf("2 reads : 1 write nt %7.3f ms  %6.2f TB/s (all bytes)\n", ms, 3 * bytes / ms / 1e9);
  }
  return 0;
}

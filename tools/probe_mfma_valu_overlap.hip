// Does a wave's VALU work run UNDER another wave's MFMAs on the same SIMD?  (round 5: the split-precision forward's
// epilogue - combine, ReLU, v_cvt_pk_f16_f32, v_fma_mix - is 29 % of the kernel and is not hidden by the other
// workgroup's k-loop, whatever the two workgroups' relative phase.)
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) issue N dependent-free MFMAs, waves 4-7 (their SIMD
// partners) run a loop of ONE kind of VALU instruction for about the same time.  Reported: the MFMA waves' cycles
// alone, the VALU waves' cycles alone, and both together.  hipcc --offload-arch=gfx950 -O2 -o probe probe_mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__device__ __forceinline__ void valu_body(float& a, float& b, unsigned& c, unsigned& d) {
  if (KIND == 0) {        // plain fp32 fma
    asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %1, %1, %0, %0" : "+v"(a), "+v"(b));
  } else if (KIND == 1) { // v_cvt_pk_f16_f32
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n v_cvt_pk_f16_f32 %1, %3, %2" : "+v"(c), "+v"(d) : "v"(a), "v"(b));
  } else if (KIND == 2) { // v_fma_mixlo_f16
    asm volatile("v_fma_mixlo_f16 %0, %2, %3, %3 op_sel_hi:[1,0,0]\n v_fma_mixlo_f16 %1, %2, %3, %3 op_sel_hi:[1,0,0]"
                 : "+v"(c), "+v"(d) : "v"(c), "v"(a));
  } else if (KIND == 3) { // integer op
    asm volatile("v_lshl_or_b32 %0, %0, 1, %1\n v_lshl_or_b32 %1, %1, 1, %0" : "+v"(c), "+v"(d));
  } else {                // v_pk_mul_f16
    asm volatile("v_pk_mul_f16 %0, %0, %1\n v_pk_mul_f16 %1, %1, %0" : "+v"(c), "+v"(d));
  }
}

template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int n_mfma, int n_valu, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool mfma_wave = wave < 4;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if (mfma_wave) {
    if (mode & 1) {
      f32x16 acc[4] = {};
      half8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = a;
      t0 = __builtin_readcyclecounter();
      for (int i = 0; i < n_mfma; i += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
      }
      t1 = __builtin_readcyclecounter();
      sink[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    }
  } else {
    if (mode & 2) {
      float a = threadIdx.x * 1e-3f, b = 1.0f;
      unsigned c = threadIdx.x, d = 7;
      t0 = __builtin_readcyclecounter();
      for (int i = 0; i < n_valu; i += 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) valu_body<KIND>(a, b, c, d);
      }
      t1 = __builtin_readcyclecounter();
      sink[threadIdx.x] = a + b + (float)c + (float)d;
    }
  }
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, unsigned long long* d_out, float* d_sink) {
  const int n_mfma = 4096, n_valu = 4096 * 6;
  unsigned long long h[8 * 256];
  double res[4][2];
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, d_out, d_sink, n_mfma, n_valu, mode);
    hipDeviceSynchronize();
    hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w] / (256 * 4);
    res[mode][0] = m; res[mode][1] = v;
  }
  printf("%-18s MFMA alone %7.0f ticks | VALU alone %7.0f | together: MFMA %7.0f (x%.2f)  VALU %7.0f (x%.2f)\n", name,
         res[1][0], res[2][1], res[3][0], res[3][0] / res[1][0], res[3][1], res[3][1] / res[2][1]);
}

int main() {
  unsigned long long* d_out; float* d_sink;
  hipMalloc(&d_out, 8 * 256 * sizeof(unsigned long long)); hipMalloc(&d_sink, 512 * 4);
  printf("%d MFMAs (32x32x16 f16) per MFMA wave, %d VALU instructions per VALU wave; s_memtime ticks\n", 4096, 4096 * 6);
  run<0>("v_fma_f32", d_out, d_sink);
  run<1>("v_cvt_pk_f16_f32", d_out, d_sink);
  run<2>("v_fma_mixlo_f16", d_out, d_sink);
  run<3>("v_lshl_or_b32", d_out, d_sink);
  run<4>("v_pk_mul_f16", d_out, d_sink);
  return 0;
}

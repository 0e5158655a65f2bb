// HW probe (run on the MI355X): ds_read_tr8_b64 lane map, bf8 MFMA operand layout, cvt_scalef32_pk_bf8_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_tr8(unsigned long long* out, int pitch) {
  extern __shared__ unsigned char lds[];
  for (int i = threadIdx.x; i < 32 * pitch; i += 64) lds[i] = 0xEE;
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * 16; i += 64) { int r = i >> 4, c = i & 15; lds[r * pitch + c] = (unsigned char)(r * 16 + c); }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  typedef i32x2 __attribute__((address_space(3))) * lp;
  i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(lds + (8 * g + (i >> 1)) * pitch + (i & 1) * 8));
  out[lane] = ((unsigned long long)(unsigned)v[1] << 32) | (unsigned)v[0];
}

__global__ void k_mfma(const unsigned char* A, const unsigned char* B, float* D) {
  const int l = threadIdx.x;
  unsigned long long a = 0, b = 0;
  for (int q = 0; q < 8; ++q) {
    a |= (unsigned long long)A[(l & 31) * 16 + 8 * (l >> 5) + q] << (8 * q);
    b |= (unsigned long long)B[(8 * (l >> 5) + q) * 32 + (l & 31)] << (8 * q);
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8((long)a, (long)b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5), col = l & 31;
    D[row * 32 + col] = c[i];
  }
}

__global__ void k_cvt(const float* x, int n, float scale, unsigned* o_ref, unsigned* o_mul, unsigned* o_div, unsigned* o_sc) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i >= n) return;
  const float v = x[i];
  o_ref[i] = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(v, v, 0, false) & 0xffff;
  o_mul[i] = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(v * scale, v * scale, 0, false) & 0xffff;
  o_div[i] = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(v / scale, v / scale, 0, false) & 0xffff;
  bf16x2 s = {(__bf16)v, (__bf16)v};
  s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(s16x2{0, 0}, s, scale, false);
  o_sc[i] = (unsigned)(unsigned short)r[0];
}

static float e5m2_to_f(unsigned char b) {
  int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
  float v;
  if (e == 0) v = ldexpf((float)m / 4.f, -14);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexpf(1.f + m / 4.f, e - 15);
  return s ? -v : v;
}

int main() {
  for (int pitch : {16, 32, 256, 272}) {
    unsigned long long* d; CK(hipMalloc(&d, 64 * 8));
    hipLaunchKernelGGL(k_tr8, dim3(1), dim3(64), 32 * pitch, 0, d, pitch);
    unsigned long long h[64]; CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("tr8 pitch %d\n", pitch);
    for (int l = 0; l < 64; ++l) {
      if (l >= 18 && l < 62 && l != 32 && l != 33) continue;
      printf("  lane %2d:", l);
      for (int q = 0; q < 8; ++q) { unsigned b = (h[l] >> (8 * q)) & 0xff; printf(" (%d,%d)", b >> 4, b & 15); }
      printf("\n");
    }
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int q = 0; q < 8; ++q) {
      unsigned b = (h[l] >> (8 * q)) & 0xff;
      unsigned want = (unsigned)(((8 * (l >> 4) + q) * 16 + (l & 15)) & 0xff);
      if (b != want) ++bad;
    }
    printf("  hypothesis 'lane j <- column j, byte q <- row q': %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
    CK(hipFree(d));
  }
  {
    unsigned char A[32 * 16], B[16 * 32];
    srand(1);
    const unsigned char vals[] = {0x00, 0x3C, 0x40, 0x38, 0xBC, 0x42, 0x34, 0xC0};   // 0, 1, 2, .5, -1, 3, .25, -2
    for (auto& v : A) v = vals[rand() % 8];
    for (auto& v : B) v = vals[rand() % 8];
    unsigned char *dA, *dB; float* dD;
    CK(hipMalloc(&dA, sizeof(A))); CK(hipMalloc(&dB, sizeof(B))); CK(hipMalloc(&dD, 32 * 32 * 4));
    CK(hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    float D[32 * 32]; CK(hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      float s = 0;
      for (int k = 0; k < 16; ++k) s += e5m2_to_f(A[i * 16 + k]) * e5m2_to_f(B[k * 32 + j]);
      if (s != D[i * 32 + j]) { if (bad < 5) printf("  D[%d][%d] = %g want %g\n", i, j, D[i * 32 + j], s); ++bad; }
    }
    printf("mfma_f32_32x32x16_bf8_bf8 layout (lane l: row/col l&31, k = 8(l>>5)+byte): %s (%d mismatches)\n", bad ? "NO" : "YES", bad);
  }
  {
    const int n = 4096;
    float* x = (float*)malloc(n * 4);
    srand(2);
    for (int i = 0; i < n; ++i) {
      unsigned short hb = (unsigned short)(rand() & 0xffff);
      unsigned u = (unsigned)hb << 16; memcpy(&x[i], &u, 4);
    }
    x[0] = 0.f; x[1] = 1.f; x[2] = 57344.f; x[3] = 65536.f; x[4] = 1e30f; x[5] = INFINITY; x[6] = NAN; x[7] = 1.5258789e-05f; x[8] = 3e-6f;
    x[9] = 1.125f; x[10] = 1.375f; x[11] = 1.625f; x[12] = 1.875f; x[13] = -1.125f; x[14] = 61440.f; x[15] = 59392.f;
    float* dx; unsigned *r0, *r1, *r2, *r3;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&r0, n * 4)); CK(hipMalloc(&r1, n * 4)); CK(hipMalloc(&r2, n * 4)); CK(hipMalloc(&r3, n * 4));
    CK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
    for (float scale : {1.f, 4.f, 0.25f}) {
      hipLaunchKernelGGL(k_cvt, dim3(n / 256), dim3(256), 0, 0, dx, n, scale, r0, r1, r2, r3);
      unsigned *h0 = (unsigned*)malloc(n * 4), *h1 = (unsigned*)malloc(n * 4), *h2 = (unsigned*)malloc(n * 4), *h3 = (unsigned*)malloc(n * 4);
      CK(hipMemcpy(h0, r0, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, r1, n * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(h2, r2, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h3, r3, n * 4, hipMemcpyDeviceToHost));
      int eq_mul = 0, eq_div = 0, eq_ref = 0;
      for (int i = 0; i < n; ++i) { eq_mul += (h3[i] & 0xff) == (h1[i] & 0xff); eq_div += (h3[i] & 0xff) == (h2[i] & 0xff); eq_ref += (h3[i] & 0xff) == (h0[i] & 0xff); }
      printf("cvt_scalef32_pk_bf8_bf16 scale %g: == cvt(x*s) %d / %d, == cvt(x/s) %d, == cvt(x) %d\n", scale, eq_mul, n, eq_div, eq_ref);
      for (int i = 0; i < 16; ++i) printf("   x=%-14g cvt(x)=%02x cvt(x*s)=%02x cvt(x/s)=%02x scalecvt=%04x\n", x[i], h0[i] & 0xff, h1[i] & 0xff, h2[i] & 0xff, h3[i]);
      int shown = 0;
      for (int i = 16; i < n && shown < 8; ++i) if ((h3[i] & 0xff) != (h2[i] & 0xff) && (h3[i] & 0xff) != (h1[i] & 0xff)) { printf("   DIFF x=%g cvt=%02x mul=%02x div=%02x scalecvt=%02x\n", x[i], h0[i] & 0xff, h1[i] & 0xff, h2[i] & 0xff, h3[i] & 0xff); ++shown; }
    }
  }
  return 0;
}

// What v_cvt_pk_fp8_f32 and the e4m3 MFMA do above 240: bytes of the conversion, and an MFMA over operands of 256 / 448.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void cvt(const float* x, unsigned* out, int n) {
  int i = threadIdx.x;
  if (i < n) { int v = 0; v = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], -x[i], v, false); out[i] = (unsigned)v; }
}
__global__ void mm(float* D, int byteA, int byteB) {
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = byteA * 0x01010101; b[i] = byteB * 0x01010101; }
  v16f c; for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 127, 0, 127);
  if (threadIdx.x == 0) D[0] = c[0];
}
int main() {
  float h[] = {1.f, 200.f, 240.f, 248.f, 256.f, 267.f, 300.f, 416.f, 430.f, 448.f, 460.f, 470.f, 500.f, 1e4f};
  int n = sizeof(h) / 4; float* dx; unsigned* dout; unsigned o[32];
  hipMalloc(&dx, 128); hipMalloc(&dout, 128); hipMemcpy(dx, h, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dx, dout, n); hipMemcpy(o, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt(%g) = 0x%02x, cvt(-x) = 0x%02x\n", h[i], o[i] & 0xff, (o[i] >> 8) & 0xff);
  float* dD; hipMalloc(&dD, 4); float r;
  int cases[][2] = {{0x38, 0x38}, {0x38, 0x77}, {0x38, 0x78}, {0x38, 0x7e}, {0x78, 0x38}, {0x7e, 0x7e}, {0x38, 0x7f}};
  for (auto& cs : cases) { hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dD, cs[0], cs[1]); hipMemcpy(&r, dD, 4, hipMemcpyDeviceToHost);
    printf("mfma A=0x%02x B=0x%02x (x64): %g\n", cs[0], cs[1], r); }
  return 0;
}

// Sustained (power-limited) MFMA throughput of the two bf16 shapes, all CUs busy, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int SHAPE>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (threadIdx.x * 2654435761u) ^ (i * 40503u) ^ (seed * 97u) ^ (blockIdx.x * 7919u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    a[i] = (short)(0x3c00 + (h & 0x3ff) + ((h >> 16) & 0x8000));          // random sign / mantissa, exponent near 0
    b[i] = (short)(0x3c00 + ((h >> 10) & 0x3ff) + ((h >> 17) & 0x8000));
  }
  float s = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16_t acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  }
  if (s == 123.456f) out[0] = s;
}

int main() {
  float* d; (void)hipMalloc(&d, 4);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 20000, grid = 256 * 4;
  for (int rnd = 0; rnd < 3; ++rnd) {
    for (int shape : {16, 32}) {
      (void)hipEventRecord(e0);
      for (int rep = 0; rep < 4; ++rep) {
        if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(grid), dim3(512), 0, 0, d, iters, rep);
        else hipLaunchKernelGGL(k<32>, dim3(grid), dim3(512), 0, 0, d, iters, rep);
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      // per iteration per wave: 16 MFMAs x 16384 FLOPs (16x16x32) or 8 x 32768 (32x32x16) = 262144 FLOPs
      const double fl = 4.0 * grid * 8.0 * iters * 262144.0;
      printf("shape %dx%d: %.1f ms  %.0f TFLOP/s\n", shape, shape, ms, fl / ms / 1e9);
    }
  }
  return 0;
}

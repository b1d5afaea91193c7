// Sustained (power-limited) MFMA throughput, all CUs busy, operands in registers, random operands: the two bf16 shapes, the two int8 shapes
// (v_mfma_i32_16x16x64_i8 / 32x32x32_i8: the instruction of the int8 block Linears) and the block-scaled e4m3 form (v_mfma_scale_f32_16x16x128_f8f6f4).
// This is the ceiling a GEMM main loop can approach on this chip at the clock it holds under that load: "power-limited" as a number.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
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

typedef __attribute__((ext_vector_type(4))) int i32x4_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(16))) int i32x16_t;
// SHAPE 116: v_mfma_i32_16x16x64_i8, 132: v_mfma_i32_32x32x32_i8, 208: v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, unit scales)
template <int SHAPE>
__global__ __launch_bounds__(512) void k8(int* out, int iters, unsigned seed) {
  i32x8_t a8, b8;
  for (int i = 0; i < 8; ++i) {
    unsigned h = (threadIdx.x * 2654435761u) ^ (i * 40503u) ^ (seed * 97u) ^ (blockIdx.x * 7919u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    unsigned g = h * 0x9E3779B9u; g ^= g >> 16;
    if (SHAPE == 208) { h &= 0xBFBFBFBFu; g &= 0xBFBFBFBFu; }      // e4m3: keep the exponent's top bit clear (|x| < 2, no NaN bytes)
    a8[i] = (int)h; b8[i] = (int)g;
  }
  const i32x4_t a = {a8[0], a8[1], a8[2], a8[3]}, b = {b8[0], b8[1], b8[2], b8[3]};
  int s = 0;
  if constexpr (SHAPE == 116) {
    i32x4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = i32x4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
  } else if constexpr (SHAPE == 132) {
    i32x16_t acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
  } else {
    f32x4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i], 0, 0, 0, 127, 0, 127);
    }
    for (int i = 0; i < 16; ++i) s += (int)(acc[i][0] + acc[i][3]);
  }
  if (s == 123456789) out[0] = s;
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
  // 8-bit shapes: per iteration per wave 16 x (16x16x64) or 8 x (32x32x32) = 524288 OPs; 16 x (16x16x128) = 1048576 FLOPs
  for (int rnd = 0; rnd < 3; ++rnd) {
    for (int shape : {116, 132, 208}) {
      (void)hipEventRecord(e0);
      for (int rep = 0; rep < 4; ++rep) {
        if (shape == 116) hipLaunchKernelGGL(k8<116>, dim3(grid), dim3(512), 0, 0, (int*)d, iters, rep);
        else if (shape == 132) hipLaunchKernelGGL(k8<132>, dim3(grid), dim3(512), 0, 0, (int*)d, iters, rep);
        else hipLaunchKernelGGL(k8<208>, dim3(grid), dim3(512), 0, 0, (int*)d, iters, rep);
      }
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      const double fl = 4.0 * grid * 8.0 * iters * (shape == 208 ? 1048576.0 : 524288.0);
      printf("%s: %.1f ms  %.0f TOP/s\n", shape == 116 ? "i8 16x16x64" : shape == 132 ? "i8 32x32x32" : "e4m3 scaled 16x16x128", ms, fl / ms / 1e9);
    }
  }
  return 0;
}

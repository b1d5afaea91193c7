// Pins the operand and scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands) with exact integer data.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_fp8_probe.hip -o /tmp/mfma_fp8_probe && /tmp/mfma_fp8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void k(const uint8_t* A, const uint8_t* B, float* D, const int* sa, const int* sb, int mode) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  v8i a, b;
  for (int i = 0; i < 8; ++i) {   // lane (r, h): bytes 32h .. 32h+31 of row r (A) / column r (B)
    a[i] = *(const int*)(A + r * 64 + 32 * h + 4 * i);
    b[i] = *(const int*)(B + r * 64 + 32 * h + 4 * i);   // B stored [col][k]
  }
  v16f c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  const int va = sa[lane], vb = sb[lane];
  if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, va, 0, vb);
  if (mode == 1) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, va, 0, vb);
  if (mode == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 2, va, 3, vb);
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];   // D[row][col]
}

static uint8_t e4m3(int v) {   // exact for |v| <= 8
  if (v == 0) return 0;
  uint8_t s = v < 0 ? 0x80 : 0; int a = abs(v); int e = 0; while ((1 << (e + 1)) <= a) ++e;
  int m = (a - (1 << e)) * 8 >> e;
  return s | ((e + 7) << 3) | m;
}

int main() {
  std::vector<uint8_t> A(32 * 64), B(32 * 64);
  std::vector<int> Ai(32 * 64), Bi(32 * 64);
  srand(1);
  for (int i = 0; i < 32 * 64; ++i) { Ai[i] = rand() % 9 - 4; Bi[i] = rand() % 9 - 4; A[i] = e4m3(Ai[i]); B[i] = e4m3(Bi[i]); }
  uint8_t *dA, *dB; float* dD; int *dsa, *dsb;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  std::vector<float> D(32 * 32);
  auto run = [&](int mode, std::vector<int> sa, std::vector<int> sb) {
    hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, dsa, dsb, mode);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  };
  auto ref = [&](int i, int j, float w0, float w1) {   // sum over k<32 weighted w0, k>=32 weighted w1
    float s = 0; for (int kk = 0; kk < 64; ++kk) s += (kk < 32 ? w0 : w1) * Ai[i * 64 + kk] * Bi[j * 64 + kk]; return s; };
  auto check = [&](const char* name, auto f) {
    int bad = 0; for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (D[i * 32 + j] != f(i, j)) ++bad;
    printf("%-58s mismatches %d / 1024  (D[3][5] = %g, expected %g)\n", name, bad, D[3 * 32 + 5], f(3, 5)); };
  std::vector<int> unit(64, 127);
  run(0, unit, unit);
  check("unit scales, lane (r,h) = bytes 32h..32h+31", [&](int i, int j) { return ref(i, j, 1, 1); });
  {  // A scale x2 on lane half 0 only
    std::vector<int> sa(64, 127); for (int l = 0; l < 32; ++l) sa[l] = 128;
    run(0, sa, unit);
    check("scale_a = 2 on lanes 0-31: blocks follow lane halves?", [&](int i, int j) { return ref(i, j, 2, 1); });
  }
  {  // A scale per row: row i -> 2^(i&3)
    std::vector<int> sa(64); for (int l = 0; l < 64; ++l) sa[l] = 127 + (l & 3);
    run(0, sa, unit);
    check("scale_a = 2^(row&3), both halves", [&](int i, int j) { return ref(i, j, 1, 1) * (1 << (i & 3)); });
  }
  {  // B scale per column
    std::vector<int> sb(64); for (int l = 0; l < 64; ++l) sb[l] = 127 - (l & 3);
    run(0, unit, sb);
    check("scale_b = 2^-(col&3), both halves", [&](int i, int j) { return ref(i, j, 1, 1) / (1 << (j & 3)); });
  }
  {  // opsel: byte 1 of the scale register
    std::vector<int> sa(64); for (int l = 0; l < 64; ++l) sa[l] = 127 | ((127 + 1 + (l & 1)) << 8) | (130 << 16) | (131u << 24);
    run(1, sa, unit);
    check("opsel_a = 1 -> byte 1 (2^(1 + (row&1)))", [&](int i, int j) { return ref(i, j, 1, 1) * (2 << (i & 1)); });
    std::vector<int> sb(64); for (int l = 0; l < 64; ++l) sb[l] = 127 | (127 << 8) | (127 << 16) | (125u << 24);
    run(2, sa, sb);
    check("opsel_a = 2 (x8), opsel_b = 3 (x1/4)", [&](int i, int j) { return ref(i, j, 1, 1) * 2; });
  }
  return 0;
}

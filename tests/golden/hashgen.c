/* Test infrastructure: the counter-based integer generator of tests/full_depth_common.py::hash_normal, in C with OpenMP, for the build
 * container's fixture generator only (12 G elements per checkpoint: minutes in torch integer ops, seconds here).  The torch form stays the
 * definition -- the GPU tests regenerate the checkpoints with it on the device -- and tests/golden/make_full_depth_golden.py checks this
 * file against it on a sample before using it.
 *
 *   z = splitmix64(i + seed * 0x9E3779B97F4A7C15);  u = sum of the four 16-bit fields of z - 2 * 65535   (Irwin-Hall(4), exact)
 *   w = bf16(fp32(u) * scale [+ mean])         one fp32 multiply, one fp32 add when mean != 0, one round-to-nearest-even to bf16
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC hashgen.c -o hashgen.so
 */
#include <stdint.h>
#include <string.h>

static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      /* NaN (cannot occur here) */
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

void hash_normal_bf16(uint16_t* out, int64_t n, int64_t start, uint64_t seed, float scale, float mean, int add_mean) {
  const uint64_t off = seed * 0x9E3779B97F4A7C15ull;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    uint64_t z = (uint64_t)(start + i) + off;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const int64_t u = (int64_t)(z & 0xFFFF) + (int64_t)((z >> 16) & 0xFFFF) + (int64_t)((z >> 32) & 0xFFFF) + (int64_t)((z >> 48) & 0xFFFF) - 2 * 65535;
    float w = (float)u * scale;
    if (add_mean) w = w + mean;
    out[i] = f2bf(w);
  }
}

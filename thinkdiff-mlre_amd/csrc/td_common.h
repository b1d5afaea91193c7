// Shared device/host helpers for the ThinkDiff gfx950 kernels.
// Everything here is CDNA4-only (wave64, MFMA, LDS-DMA); there is no other backend.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bits in HBM

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

#define TD_LDS __attribute__((address_space(3)))

// ---- bf16 <-> f32 -------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) {
  return __builtin_bit_cast(float, (unsigned)v << 16);
}
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 v2;
  v2 p;
  p[0] = (__bf16)lo;
  p[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, p);
}
// Bit views of a scalar, taken BY VALUE: `__builtin_bit_cast(unsigned, vec[i])` on an ext_vector element reads element 0
// whatever `i` is (hipcc / clang 22: the element expression is not an addressable lvalue), silently.
__device__ __forceinline__ unsigned as_u32(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float as_f32(unsigned u) { return __builtin_bit_cast(float, u); }
// round an f32 through bf16 (emulates one torch bf16 op boundary of the reference pipeline)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// ---- activations (fp32 math) -------------------------------------------
// Written on v_exp_f32 / v_rcp_f32 directly: a plain `/` compiles to the ~12-instruction IEEE division sequence, which made the
// GELU epilogue of a 256x256 GEMM tile cost ~16 % of the tile (128 values per lane).  rcp / exp2 are good to 1 ulp, far inside
// the bf16 rounding that follows every activation on this path.
__device__ __forceinline__ float sigmoid_f(float z) {   // 1 / (1 + e^-z)
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
__device__ __forceinline__ float gelu_tanh_f(float x) {
  // 0.5 x (1 + tanh(u)) = x sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3)
  return x * sigmoid_f(x * (1.5957691216057308f + 0.07135481627260025f * x * x));
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
__device__ __forceinline__ float quick_gelu_f(float x) { return x * sigmoid_f(1.702f * x); }  // x * sigmoid(1.702 x)

// ---- wave reductions (wave64) ------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- error plumbing shared by the C-ABI translation units ---------------
void td_set_error(const char* fmt, ...);
#define TD_CHECK_ARG(cond, ...)          \
  do {                                   \
    if (!(cond)) {                       \
      td_set_error(__VA_ARGS__);         \
      return 2; /* TD_ERR_INVALID */     \
    }                                    \
  } while (0)
#define TD_CHECK_HIP(expr)                                                        \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      td_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return 3; /* TD_ERR_HIP */                                                  \
    }                                                                             \
  } while (0)
#define TD_CHECK_LAUNCH() TD_CHECK_HIP(hipGetLastError())

// ---- launch geometry -------------------------------------------------------
// A dispatch packet carries the grid as 32-bit WORK-ITEM counts: blocks x threads-per-block must stay below 2^32, and a larger
// product is truncated WITHOUT an error (round 2: td_fill_normal drew 3.3 G of 11.9 G weights that way).  Every launcher that
// derives its grid from an element count goes through td_grid_1d / TD_GRID_1D: the count either fits or the call is refused
// with TD_ERR_INVALID; kernels that must take more walk a grid-stride loop (td_fill_normal_kernel) and cap their grid themselves.
inline bool td_grid_1d(long long items, int block, unsigned* blocks) {
  if (items <= 0 || block <= 0) return false;
  const long long b = (items + block - 1) / block;
  if (b * block >= (1ll << 32)) return false;
  *blocks = (unsigned)b;
  return true;
}
#define TD_GRID_1D(var, items, block, what)                                                                                   \
  unsigned var = 0;                                                                                                           \
  TD_CHECK_ARG(td_grid_1d((long long)(items), (block), &var),                                                                 \
               "%s: %lld work-items in blocks of %d do not fit one launch (grid x block must stay below 2^32)", what, (long long)(items), (int)(block))
// for kernels that index their work-items with a 32-bit int
#define TD_GRID_1D_I32(var, items, block, what)                                                                               \
  TD_CHECK_ARG((long long)(items) < (1ll << 31), "%s: %lld work-items exceed the kernel's 32-bit index", what, (long long)(items)); \
  TD_GRID_1D(var, items, block, what)

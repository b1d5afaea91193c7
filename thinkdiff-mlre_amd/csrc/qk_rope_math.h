// Per-head QK-RMSNorm + interleaved-pair rotary embedding on 8 consecutive head elements: the arithmetic of
// td_qk_norm_rope_kernel (csrc/elementwise.hip), shared with the 8-bit attention's pack pass (csrc/attention_fp8.hip), which applies it
// on the way to e4m3 -- one definition, so that the two are bit-identical by construction ([ext] diffusers FluxAttnProcessor2_0: norm_q /
// norm_k = RMSNorm(128, eps 1e-6) in the bf16 graph, then embeddings.apply_rotary_emb(use_real_unbind_dim=-1) in fp32).
#pragma once
#include "td_common.h"

// sum of squares of the 8 elements, sequentially (the 16 partial sums of a head row are then added as an xor-butterfly 8, 4, 2, 1)
__device__ __forceinline__ float qk_sumsq8(const float (&x)[8]) {
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) sq = __builtin_fmaf(x[i], x[i], sq);
  return sq;
}
__device__ __forceinline__ float qk_rstd(float sumsq128, float eps) { return rsqrtf(__builtin_fmaf(sumsq128, 1.0f / 128.0f, eps)); }
// x <- bf16(bf16(x * rstd) * w): the two roundings of the bf16 RMSNorm module
__device__ __forceinline__ void qk_norm8(float (&x)[8], float rstd, const float (&w)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = rbf(rbf(x[i] * rstd) * w[i]);
}
// interleaved pairs (2i, 2i+1); cos / sin tables are repeat-interleaved, fp32
__device__ __forceinline__ void qk_rope_pairs8(const float (&x)[8], const float (&cs)[8], const float (&sn)[8], float (&y)[8]) {
  // Spelled out (the sin product rounded to fp32, the cos product fused into the sum): left to the compiler, which product of
  // `a * b - c * d` it contracts depends on the surrounding code, and the two kernels that share this would differ in the last bit.
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const float p0 = x[i + 1] * sn[i], p1 = x[i] * sn[i + 1];
    y[i] = __builtin_fmaf(x[i], cs[i], -p0);
    y[i + 1] = __builtin_fmaf(x[i + 1], cs[i + 1], p1);
  }
}

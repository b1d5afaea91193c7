// HBM-bound row kernels of the FLUX / aligner / Qwen2-VL path (SURVEY.md 2.3 K3,K5,K7,K9,K10,K16,K17,K20).
// All bf16 traffic is 16 B per lane (guide G13); statistics, RoPE and the Euler update are fp32.
// Rounding points mirror the reference's bf16 torch pipeline (each torch op rounds to bf16).
#include "td_common.h"
#include "td_kernels.h"
#include "qk_rope_math.h"

namespace {

__device__ __forceinline__ void unpack8(const u32x4_t v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf_lo(v[i]);
    f[2 * i + 1] = bf_hi(v[i]);
  }
}
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return v;
}

}  // namespace

// 8 floats * inv -> 8 OCP e4m3 bytes (v_cvt_pk_fp8_f32, round to nearest even; |x * inv| <= 448 by construction)
__device__ __forceinline__ u32x2_t pack8_fp8(const float (&x)[8], float inv) {
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[0] * inv, x[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(x[2] * inv, x[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[4] * inv, x[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(x[6] * inv, x[7] * inv, hi, true);
  return u32x2_t{(unsigned)lo, (unsigned)hi};
}
// 8 floats * inv -> 8 symmetric int8 bytes (round to nearest even; |x * inv| <= 127 by construction)
__device__ __forceinline__ u32x2_t pack8_i8(const float (&x)[8], float inv) {
  unsigned w[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const unsigned b0 = (unsigned)__float2int_rn(x[4 * h] * inv) & 0xffu, b1 = (unsigned)__float2int_rn(x[4 * h + 1] * inv) & 0xffu;
    const unsigned b2 = (unsigned)__float2int_rn(x[4 * h + 2] * inv) & 0xffu, b3 = (unsigned)__float2int_rn(x[4 * h + 3] * inv) & 0xffu;
    w[h] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
  }
  return u32x2_t{w[0], w[1]};
}
__device__ __forceinline__ u32x2_t pack8_q(const float (&x)[8], float inv, bool int8) { return int8 ? pack8_i8(x, inv) : pack8_fp8(x, inv); }

// ---------------------------------------------------------------------------------------------
// Row normalisation (+ optional affine weight, + optional adaLN modulation), one wave per row.
//   LayerNorm (no affine, eps) : n = (x - mean) * rsqrt(var + eps)          [diffusers AdaLayerNorm*]
//   RMSNorm                   : n = x * rsqrt(mean(x^2) + eps)             [T5LayerNorm / Qwen2RMSNorm]
//   y = bf16(n); if w: y = bf16(w*y); if mod: y = bf16(bf16(y * bf16(1+scale)) + shift)
// Rows < split use (shiftA, scaleA), rows >= split use (shiftB, scaleB): the joint [text||image]
// buffer is normalised in one launch although the two streams carry different modulations.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void td_norm_rows_kernel(const TdNormParams p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= p.rows) return;
  const bf16_t* xr = p.x + (size_t)row * p.ldx;
  // replicated channels (int8 smoothing): this lane's two table entries are fetched with the row, not behind the arithmetic that needs them last
  int ext_s0 = -1, ext_s1 = -1;
  {
    const int* ext0 = row >= p.split ? p.extB : p.extA;
    if (p.q && ext0 && lane * 2 < p.ext_n) { ext_s0 = ext0[lane * 2]; ext_s1 = ext0[lane * 2 + 1]; }
  }
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    unpack8(*(const u32x4_t*)(xr + c * 512 + lane * 8), v[c]);
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[c][i];
  }
  constexpr float inv_d = 1.0f / (NCH * 512);
  float mean = 0.f;
  if (!p.rms) mean = wave_sum(sum) * inv_d;
  float sq = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[c][i] -= mean;
      sq += v[c][i] * v[c][i];
    }
  const float rstd = rsqrtf(wave_sum(sq) * inv_d + p.eps);

  const bool partB = row >= p.split;
  const bf16_t* shift = partB ? p.shiftB : p.shiftA;
  const bf16_t* scale = partB ? p.scaleB : p.scaleA;
  bf16_t* yr = p.y + (size_t)row * p.ldy;
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = c * 512 + lane * 8;
    float y[8];
    const bool keep_f32 = p.rms == 2;  // fp32-weight T5LayerNorm under autocast: one rounding at the end
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = keep_f32 ? v[c][i] * rstd : rbf(v[c][i] * rstd);
    if (p.w) {
      float w[8];
      unpack8(*(const u32x4_t*)(p.w + col), w);
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = keep_f32 ? y[i] * w[i] : rbf(y[i] * w[i]);
    }
    if (scale) {
      float sc[8], sh[8];
      unpack8(*(const u32x4_t*)(scale + col), sc);
      unpack8(*(const u32x4_t*)(shift + col), sh);
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = rbf(rbf(y[i] * rbf(1.0f + sc[i])) + sh[i]);
    }
    if (p.q) {   // keep the bf16-rounded row in registers for the quantisation pass
      float sm[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      const bf16_t* smooth = partB ? p.smoothB : p.smoothA;      // int8 smoothing: 1 / s per channel (a power of two: exact on a bf16 value)
      if (smooth) unpack8(*(const u32x4_t*)(smooth + col), sm);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        v[c][i] = rbf(y[i]) * sm[i];
        amax = fmaxf(amax, fabsf(v[c][i]));
      }
    } else {
      *(u32x4_t*)(yr + col) = pack8(y);
    }
  }
  if (p.q) {
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax * (p.q_int8 ? 1.0f / 127.0f : 1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / s;
    if (lane == 0) p.q_scale[row] = s;
    uint8_t* qr = p.q + (size_t)row * p.ldq;
    const int* ext = partB ? p.extB : p.extA;
    __shared__ __attribute__((aligned(8))) uint8_t qrow[4][NCH * 512];      // the quantised row of each of the block's 4 waves (replicated channels only)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const u32x2_t b = pack8_q(v[c], inv, p.q_int8 != 0);
      *(u32x2_t*)(qr + c * 512 + lane * 8) = b;
      if (ext) *(u32x2_t*)(&qrow[threadIdx.x >> 6][c * 512 + lane * 8]) = b;
    }
    if (ext) {      // (wave-uniform; a wave reads only what it wrote itself: LDS operations of a wave complete in order)
      for (int e = lane * 2; e < p.ext_n; e += 128) {
        const int s0 = e < 128 ? ext_s0 : ext[e], s1 = e < 128 ? ext_s1 : ext[e + 1];
        const unsigned b0 = s0 >= 0 ? qrow[threadIdx.x >> 6][s0] : 0u, b1 = s1 >= 0 ? qrow[threadIdx.x >> 6][s1] : 0u;
        *(unsigned short*)(qr + NCH * 512 + e) = (unsigned short)(b0 | (b1 << 8));
      }
    }
  }
}

int td_norm_rows_launch(const TdNormParams& p, hipStream_t stream) {
  TD_CHECK_ARG(p.rows > 0 && p.D > 0, "td_norm_rows: empty problem");
  TD_CHECK_ARG(p.D % 512 == 0 && p.D <= 4096, "td_norm_rows: D=%d must be a multiple of 512, <= 4096", p.D);
  TD_CHECK_ARG(p.ldx % 8 == 0 && p.ldy % 8 == 0, "td_norm_rows: row strides must be multiples of 8");
  TD_CHECK_ARG((p.scaleA == nullptr) == (p.shiftA == nullptr), "td_norm_rows: shift and scale come together");
  if (p.q) TD_CHECK_ARG(p.q_scale && p.ldq % 8 == 0 && (uintptr_t)p.q % 8 == 0, "td_norm_rows: fp8 output needs a scale array and 8-byte aligned rows");
  if (p.extA || p.extB) TD_CHECK_ARG(p.q && p.q_int8 && p.extA && p.extB && p.ext_n > 0 && p.ext_n % 2 == 0 && p.ldq >= p.D + p.ext_n,
                                     "td_norm_rows: replicated channels need the int8 output form, both tables and rows of D + ext_n bytes");
  TD_GRID_1D(nblk, (long long)((p.rows + 3) / 4) * 256, 256, "td_norm_rows");
  const dim3 grid(nblk), block(256);
  switch (p.D / 512) {
#define TD_CASE(n) case n: hipLaunchKernelGGL(td_norm_rows_kernel<n>, grid, block, 0, stream, p); break;
    TD_CASE(1) TD_CASE(2) TD_CASE(3) TD_CASE(4) TD_CASE(5) TD_CASE(6) TD_CASE(7) TD_CASE(8)
#undef TD_CASE
  }
  TD_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Per-head q/k RMSNorm (eps, learned weight[128]) followed by rotary embedding, in place on the
// fused projection buffer.  16 lanes own one 128-wide head row (8 elements each).
//   FLUX  : interleaved pairs (2i, 2i+1), cos/sin tables [S,128] fp32 (repeat-interleaved)
//           [ext diffusers embeddings.apply_rotary_emb use_real_unbind_dim=-1]
//   Qwen2 : rotate_half (i, i+64), tables [S,128] fp32 already M-RoPE-section-merged; no q/k norm
//           (transformers modeling_qwen2_vl.py:180-222)
// Rows < split take the (wqA, wkA) norm weights (FLUX norm_added_q/k for text tokens), the rest
// (wqB, wkB) (norm_q/k).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_qk_norm_rope_kernel(const TdQkRopeParams p) {
  const int row = blockIdx.x;
  const int l16 = threadIdx.x & 15;
  const int unit0 = threadIdx.x >> 4;  // 16 head-units per pass
  const int nunits = p.Hq + p.Hk;
  bf16_t* base = p.qkv + (size_t)row * p.ld;
  const bool partB = row >= p.split;
  const bf16_t* wq = partB ? p.wqB : p.wqA;
  const bf16_t* wk = partB ? p.wkB : p.wkA;

  float cs[8], sn[8];
  {
    const float* cr = p.cos + (size_t)row * 128;
    const float* sr = p.sin + (size_t)row * 128;
    const int d0 = l16 * 8;
#pragma unroll
    for (int i = 0; i < 8; i += 4) {
      const f32x4_t c4 = *(const f32x4_t*)(cr + d0 + i);
      const f32x4_t s4 = *(const f32x4_t*)(sr + d0 + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        cs[i + j] = c4[j];
        sn[i + j] = s4[j];
      }
    }
  }

  for (int u = unit0; u < nunits; u += 16) {
    const bool is_k = u >= p.Hq;
    bf16_t* hp = base + (is_k ? p.k_col + (u - p.Hq) * 128 : p.q_col + u * 128) + l16 * 8;
    float x[8];
    unpack8(*(const u32x4_t*)hp, x);
    const bf16_t* w = is_k ? wk : wq;
    if (w) {
      float sq = qk_sumsq8(x);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 16);
      float wv[8];
      unpack8(*(const u32x4_t*)(w + l16 * 8), wv);
      qk_norm8(x, qk_rstd(sq, p.eps), wv);
    }
    float y[8];
    if (p.rotate_half) {
      // partner element i+64 (or i-64) lives 8 lanes away
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float other = __shfl_xor(x[i], 8, 16);
        const float rot = (l16 < 8) ? -other : other;
        // rotate_half == 2: every torch op of the bf16 graph rounds (q*cos, rotate_half(q)*sin, their sum)
        y[i] = p.rotate_half == 2 ? rbf(x[i] * cs[i]) + rbf(rot * sn[i]) : x[i] * cs[i] + rot * sn[i];
      }
    } else {
      qk_rope_pairs8(x, cs, sn, y);
    }
    if (!is_k && p.q_premul != 1.0f) {
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] *= p.q_premul;
    }
    *(u32x4_t*)hp = pack8(y);
  }
}

int td_qk_norm_rope_launch(const TdQkRopeParams& p, hipStream_t stream) {
  TD_CHECK_ARG(p.rows > 0 && p.Hq > 0 && p.Hk >= 0, "td_qk_norm_rope: empty problem");
  TD_CHECK_ARG(p.ld % 8 == 0 && p.q_col % 8 == 0 && p.k_col % 8 == 0, "td_qk_norm_rope: columns must be 16-byte aligned");
  TD_GRID_1D(nblk, (long long)p.rows * 256, 256, "td_qk_norm_rope");
  hipLaunchKernelGGL(td_qk_norm_rope_kernel, dim3(nblk), dim3(256), 0, stream, p);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// FluxPosEmbed: cos/sin[S,128] fp32 from ids[S,3]; per axis a with dim d_a: freq_j = theta^(-2j/d_a)
// in fp64, angle = id * freq, cos/sin -> fp32, each value repeated for the pair (2j, 2j+1).
// [ext diffusers embeddings.get_1d_rotary_pos_embed(repeat_interleave_real=True, freqs_dtype=float64)]
// ---------------------------------------------------------------------------------------------
__global__ void td_flux_rope_table_kernel(const float* ids, int S, int d0, int d1, int d2, double theta, float* cosT, float* sinT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = (d0 + d1 + d2) / 2;
  if (idx >= S * half) return;
  const int s = idx / half, j = idx % half;
  int axis, jj, dim;
  if (j < d0 / 2) { axis = 0; jj = j; dim = d0; }
  else if (j < (d0 + d1) / 2) { axis = 1; jj = j - d0 / 2; dim = d1; }
  else { axis = 2; jj = j - (d0 + d1) / 2; dim = d2; }
  const double freq = 1.0 / pow(theta, (double)(2 * jj) / (double)dim);
  const double ang = (double)ids[s * 3 + axis] * freq;
  const float c = (float)cos(ang), sn = (float)sin(ang);
  const size_t o = (size_t)s * (2 * half) + 2 * j;
  cosT[o] = c; cosT[o + 1] = c;
  sinT[o] = sn; sinT[o + 1] = sn;
}

int td_flux_rope_table_launch(const float* ids, int S, const int* axes, double theta, float* cosT, float* sinT, hipStream_t stream) {
  TD_CHECK_ARG(S > 0 && axes[0] + axes[1] + axes[2] == 128, "td_flux_rope_table: axes dims must sum to 128");
  const long long n = (long long)S * 64;
  TD_GRID_1D_I32(nblk, n, 256, "td_flux_rope_table");
  hipLaunchKernelGGL(td_flux_rope_table_kernel, dim3(nblk), dim3(256), 0, stream, ids, S, axes[0], axes[1], axes[2], theta, cosT, sinT);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Timesteps(256, flip_sin_to_cos=True, downscale_freq_shift=0): out[n] = [cos(t f_j) | sin(t f_j)],
// f_j = exp(-ln(1e4) j / 128), fp32 math, bf16 out.   [ext diffusers embeddings.get_timestep_embedding]
// ---------------------------------------------------------------------------------------------
__global__ void td_timestep_sincos_kernel(const float* t, int n, bf16_t* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * 128) return;
  const int r = idx / 128, j = idx % 128;
  const float f = expf(-9.210340371976184f * (float)j / 128.0f);
  const float a = t[r] * f;
  out[r * 256 + j] = f2bf(cosf(a));
  out[r * 256 + 128 + j] = f2bf(sinf(a));
}

int td_timestep_sincos_launch(const float* t, int n, bf16_t* out, hipStream_t stream) {
  TD_CHECK_ARG(n > 0, "td_timestep_sincos: n must be positive");
  TD_GRID_1D_I32(nblk, (long long)n * 128, 256, "td_timestep_sincos");
  hipLaunchKernelGGL(td_timestep_sincos_kernel, dim3(nblk), dim3(256), 0, stream, t, n, out);
  TD_CHECK_LAUNCH();
  return 0;
}

// temb[r] = bf16(bf16(te[r] + ge) + pe); out = silu(temb) (bf16).  te:[n,D], ge,pe:[D]
// [ext CombinedTimestepGuidanceTextProjEmbeddings.forward, then the nn.SiLU of every AdaLayerNorm*]
__global__ void td_temb_combine_silu_kernel(const bf16_t* te, const bf16_t* ge, const bf16_t* pe, int n, int D, bf16_t* temb, bf16_t* silu_out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * D) return;
  const int c = idx % D;
  float v = bf2f(te[idx]);
  if (ge) v = rbf(v + bf2f(ge[c]));
  v = rbf(v + bf2f(pe[c]));
  if (temb) temb[idx] = f2bf(v);
  silu_out[idx] = f2bf(silu_f(v));
}

int td_temb_combine_silu_launch(const bf16_t* te, const bf16_t* ge, const bf16_t* pe, int n, int D, bf16_t* temb, bf16_t* silu_out, hipStream_t stream) {
  TD_CHECK_ARG(n > 0 && D > 0, "td_temb_combine_silu: empty problem");
  TD_GRID_1D_I32(nblk, (long long)n * D, 256, "td_temb_combine_silu");
  hipLaunchKernelGGL(td_temb_combine_silu_kernel, dim3(nblk), dim3(256), 0, stream, te, ge, pe, n, D, temb, silu_out);
  TD_CHECK_LAUNCH();
  return 0;
}

// FlowMatchEulerDiscreteScheduler.step [ext scheduling_flow_match_euler_discrete.py]:
//   prev_sample = sample.float() + (sigma_next - sigma) * model_output;  prev_sample.to(model_output.dtype)
// `(sigma_next - sigma)` is a 0-dim fp32 tensor and `model_output` a bf16 tensor, so torch's type promotion makes the PRODUCT a bf16
// op: the scalar is cast to bf16, the product is rounded to bf16, and only the sum with the fp32 sample is fp32.  Three roundings:
//   x = bf16(float(x) + float(bf16(float(bf16(dt)) * float(v))))
__global__ void td_euler_step_kernel(bf16_t* x, const bf16_t* v, float dt, int n8) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  float a[8], b[8];
  unpack8(((const u32x4_t*)x)[idx], a);
  unpack8(((const u32x4_t*)v)[idx], b);
  const float dtb = rbf(dt);
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = a[i] + rbf(dtb * b[i]);   // rbf() between the product and the sum: nothing to contract into an fma
  ((u32x4_t*)x)[idx] = pack8(a);
}

int td_euler_step_launch(bf16_t* x, const bf16_t* v, float dt, long long n, hipStream_t stream) {
  TD_CHECK_ARG(n > 0 && n % 8 == 0, "td_euler_step: n=%lld must be a positive multiple of 8", n);
  TD_GRID_1D_I32(nblk, n / 8, 256, "td_euler_step");
  const int n8 = (int)(n / 8);
  hipLaunchKernelGGL(td_euler_step_kernel, dim3(nblk), dim3(256), 0, stream, x, v, dt, n8);
  TD_CHECK_LAUNCH();
  return 0;
}

// FluxPipeline._pack_latents / _unpack_latents: [C,H,W] <-> [(H/2)(W/2), C*4], token (i,j) holds
// x[c, 2i+di, 2j+dj] at column c*4 + di*2 + dj.  Unpack optionally applies z/scaling + shift (the
// pre-VAE affine of FluxPipeline.__call__).  One thread per token row element pair.
__global__ void td_flux_pack_kernel(const bf16_t* src, bf16_t* dst, int C, int H, int W, int unpack, float div, float add) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = C * H * W;
  if (idx >= total) return;
  // idx enumerates the packed layout so the packed side is the coalesced one
  const int cols = C * 4;
  const int tok = idx / cols, col = idx % cols;
  const int c = col >> 2, di = (col >> 1) & 1, dj = col & 1;
  const int wj = W / 2;
  const int i = tok / wj, j = tok % wj;
  const size_t sp = ((size_t)c * H + (2 * i + di)) * W + (2 * j + dj);
  // (z / scaling) + shift as two bf16 torch (CPU) ops: the quotient rounds to bf16 (fp32 scalar divisor), the python-scalar
  // addend is cast to the tensor dtype first, the sum rounds to bf16
  if (unpack) dst[sp] = f2bf(__fadd_rn(rbf(__fdiv_rn(bf2f(src[idx]), div)), rbf(add)));
  else dst[idx] = src[sp];
}

int td_flux_pack_launch(const bf16_t* src, bf16_t* dst, int C, int H, int W, int unpack, float div, float add, hipStream_t stream) {
  TD_CHECK_ARG(C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "td_flux_pack: bad latent shape %dx%dx%d", C, H, W);
  TD_CHECK_ARG(!unpack || div != 0.f, "td_flux_pack: scaling divisor must be non-zero");
  TD_GRID_1D_I32(nblk, (long long)C * H * W, 256, "td_flux_pack");
  hipLaunchKernelGGL(td_flux_pack_kernel, dim3(nblk), dim3(256), 0, stream, src, dst, C, H, W, unpack, div, add);
  TD_CHECK_LAUNCH();
  return 0;
}

// ThinkDiff-CLIP token pooling: tokens [1+G*G, C] -> [1+(G/2)^2, C]; CLS row copied, the G x G grid
// reduced by 2x2 mean ( == F.interpolate(bilinear, align_corners=False) at exactly 2x, fp32 math
// then bf16).  thinkdiff/models/blip_vision_t5_decoder.py:620-637
__global__ void td_cls_avgpool2_kernel(const bf16_t* x, bf16_t* y, int G, int C) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int g2 = G / 2;
  const int total = (1 + g2 * g2) * C;
  if (idx >= total) return;
  const int tok = idx / C, c = idx % C;
  if (tok == 0) { y[idx] = x[c]; return; }
  const int i = (tok - 1) / g2, j = (tok - 1) % g2;
  auto at = [&](int r, int s) { return bf2f(x[(size_t)(1 + r * G + s) * C + c]); };
  // bilinear with align_corners=False at scale 2 samples (2i+0.5, 2j+0.5): weights 0.5/0.5 per axis
  const float top = 0.5f * at(2 * i, 2 * j) + 0.5f * at(2 * i, 2 * j + 1);
  const float bot = 0.5f * at(2 * i + 1, 2 * j) + 0.5f * at(2 * i + 1, 2 * j + 1);
  y[idx] = f2bf(0.5f * top + 0.5f * bot);
}

int td_cls_avgpool2_launch(const bf16_t* x, bf16_t* y, int G, int C, hipStream_t stream) {
  TD_CHECK_ARG(G > 0 && G % 2 == 0 && C > 0, "td_cls_avgpool2: grid %d must be even", G);
  TD_GRID_1D_I32(nblk, (1 + (long long)(G / 2) * (G / 2)) * C, 256, "td_cls_avgpool2");
  hipLaunchKernelGGL(td_cls_avgpool2_kernel, dim3(nblk), dim3(256), 0, stream, x, y, G, C);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Qwen2-VL helpers (transformers modeling_qwen2_vl.py:117-222, 453-466)
// ---------------------------------------------------------------------------------------------
// token embedding gather: out[i,:] = table[ids[i],:]   (16 B per lane)
__global__ void td_embed_gather_kernel(const int* ids, const bf16_t* table, bf16_t* out, int n, int D, int vocab) {
  const int row = blockIdx.x;
  int id = ids[row];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const u32x4_t* src = (const u32x4_t*)(table + (size_t)id * D);
  u32x4_t* dst = (u32x4_t*)(out + (size_t)row * D);
  for (int c = threadIdx.x; c < D / 8; c += blockDim.x) dst[c] = src[c];
}

int td_embed_gather_launch(const int* ids, const bf16_t* table, bf16_t* out, int n, int D, int vocab, hipStream_t stream) {
  TD_CHECK_ARG(n > 0 && D % 8 == 0 && vocab > 0, "td_embed_gather: bad shape");
  TD_GRID_1D(nblk, (long long)n * 256, 256, "td_embed_gather");
  hipLaunchKernelGGL(td_embed_gather_kernel, dim3(nblk), dim3(256), 0, stream, ids, table, out, n, D, vocab);
  TD_CHECK_LAUNCH();
  return 0;
}

// SwiGLU combine: out[m, j] = bf16(bf16(silu(gate[m,j])) * up[m,j]) with gate|up the two halves of gu[m, 2I]
__global__ void td_silu_mul_kernel(const bf16_t* gu, bf16_t* out, int rows, int I) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = I / 8;
  if (idx >= (long long)rows * per_row) return;
  const int m = (int)(idx / per_row), c = (int)(idx % per_row);
  float g[8], u[8];
  unpack8(*(const u32x4_t*)(gu + (size_t)m * 2 * I + c * 8), g);
  unpack8(*(const u32x4_t*)(gu + (size_t)m * 2 * I + I + c * 8), u);
#pragma unroll
  for (int i = 0; i < 8; ++i) g[i] = rbf(silu_f(g[i])) * u[i];
  *(u32x4_t*)(out + (size_t)m * I + c * 8) = pack8(g);
}

int td_silu_mul_launch(const bf16_t* gu, bf16_t* out, int rows, int I, hipStream_t stream) {
  TD_CHECK_ARG(rows > 0 && I % 8 == 0, "td_silu_mul: bad shape");
  const long long n = (long long)rows * (I / 8);
  TD_GRID_1D(nblk, n, 256, "td_silu_mul");
  hipLaunchKernelGGL(td_silu_mul_kernel, dim3(nblk), dim3(256), 0, stream, gu, out, rows, I);
  TD_CHECK_LAUNCH();
  return 0;
}

// M-RoPE tables: pos int32 [3, n] (temporal, height, width), sections s0+s1+s2 = 64 rotary pairs.
// cos/sin[n, 128] fp32 with emb = cat(freqs, freqs); channel j (mod 64) takes its angle from the position
// stream of its section.  round_bf16: store bf16-rounded values (the tables are cast to the model dtype).
__global__ void td_mrope_table_kernel(const int* pos, int n, int s0, int s1, float theta, int round_bf16, float* cosT, float* sinT) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * 64) return;
  const int t = idx / 64, j = idx % 64;
  const int axis = j < s0 ? 0 : (j < s0 + s1 ? 1 : 2);
  const float inv_freq = 1.0f / powf(theta, (float)(2 * j) / 128.0f);
  const float ang = (float)pos[axis * n + t] * inv_freq;
  float c = cosf(ang), s = sinf(ang);
  if (round_bf16) { c = rbf(c); s = rbf(s); }
  cosT[(size_t)t * 128 + j] = c; cosT[(size_t)t * 128 + 64 + j] = c;
  sinT[(size_t)t * 128 + j] = s; sinT[(size_t)t * 128 + 64 + j] = s;
}

int td_mrope_table_launch(const int* pos, int n, const int* sections, float theta, int round_bf16, float* cosT, float* sinT, hipStream_t stream) {
  TD_CHECK_ARG(n > 0 && sections[0] + sections[1] + sections[2] == 64, "td_mrope_table: sections must sum to 64");
  TD_GRID_1D_I32(nblk, (long long)n * 64, 256, "td_mrope_table");
  hipLaunchKernelGGL(td_mrope_table_kernel, dim3(nblk), dim3(256), 0, stream, pos, n, sections[0], sections[1], theta, round_bf16, cosT, sinT);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Row normalisation for any D % 8 == 0 (CLIP 768, EVA 1408, Qwen-ViT 1280 ...): LayerNorm with affine weight and
// bias, or RMSNorm.  One wave per row, two sweeps over the (L2-resident) row.  y = bf16(n * w + b), fp32 statistics.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_norm_rows_generic_kernel(const bf16_t* x, int ldx, bf16_t* y, int ldy, int rows, int D, int rms,
                                                                   float eps, const bf16_t* w, const bf16_t* b) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * ldx;
  float s = 0.f, q = 0.f;
  for (int c = lane * 8; c < D; c += 512) {
    float v[8];
    unpack8(*(const u32x4_t*)(xr + c), v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { s += v[i]; q += v[i] * v[i]; }
  }
  s = wave_sum(s); q = wave_sum(q);
  const float mean = rms ? 0.f : s / D;
  const float var = rms ? q / D : fmaxf(q / D - mean * mean, 0.f);
  const float rstd = rsqrtf(var + eps);
  bf16_t* yr = y + (size_t)row * ldy;
  for (int c = lane * 8; c < D; c += 512) {
    float v[8], wv[8], bv[8];
    unpack8(*(const u32x4_t*)(xr + c), v);
    if (w) unpack8(*(const u32x4_t*)(w + c), wv);
    if (b) unpack8(*(const u32x4_t*)(b + c), bv);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = (v[i] - mean) * rstd;
      if (rms) { t = rbf(t); if (w) t = t * wv[i]; }      // T5LayerNorm: cast, then weight
      else { if (w) t = t * wv[i]; if (b) t = t + bv[i]; }  // nn.LayerNorm: one fp32 expression, one rounding
      v[i] = t;
    }
    *(u32x4_t*)(yr + c) = pack8(v);
  }
}

int td_norm_rows_generic_launch(const bf16_t* x, int ldx, bf16_t* y, int ldy, int rows, int D, int rms, float eps,
                                const bf16_t* w, const bf16_t* b, hipStream_t stream) {
  TD_CHECK_ARG(rows > 0 && D > 0 && D % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0, "td_layernorm: D and strides must be multiples of 8");
  TD_GRID_1D(nblk, (long long)((rows + 3) / 4) * 256, 256, "td_norm_rows_generic");
  hipLaunchKernelGGL(td_norm_rows_generic_kernel, dim3(nblk), dim3(256), 0, stream, x, ldx, y, ldy, rows, D, rms, eps, w, b);
  TD_CHECK_LAUNCH();
  return 0;
}

// out = a + b (b broadcast over rows when b_rows == 1 .. or cyclic with period b_rows), bf16, fp32 add
__global__ void td_add_rows_kernel(const bf16_t* a, const bf16_t* b, bf16_t* out, long long n8, int row8, int b_rows) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  const long long r = idx / row8;
  const long long bi = (r % b_rows) * row8 + idx % row8;
  float x[8], y[8];
  unpack8(((const u32x4_t*)a)[idx], x);
  unpack8(((const u32x4_t*)b)[bi], y);
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] += y[i];
  ((u32x4_t*)out)[idx] = pack8(x);
}

int td_add_rows_launch(const bf16_t* a, const bf16_t* b, bf16_t* out, int rows, int D, int b_rows, hipStream_t stream) {
  TD_CHECK_ARG(rows > 0 && D % 8 == 0 && b_rows > 0, "td_add_rows: bad shape");
  const long long n8 = (long long)rows * D / 8;
  TD_GRID_1D(nblk, n8, 256, "td_add_rows");
  hipLaunchKernelGGL(td_add_rows_kernel, dim3(nblk), dim3(256), 0, stream, a, b, out, n8, D / 8, b_rows);
  TD_CHECK_LAUNCH();
  return 0;
}

// gated unit: out[m, j] = bf16(bf16(act(g[m,j])) * u[m,j]), g | u = the two halves of gu[m, 2I]; act: TdAct code
__global__ void td_glu_mul_kernel(const bf16_t* gu, bf16_t* out, int rows, int I, int act) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_row = I / 8;
  if (idx >= (long long)rows * per_row) return;
  const int m = (int)(idx / per_row), c = (int)(idx % per_row);
  float g[8], u[8];
  unpack8(*(const u32x4_t*)(gu + (size_t)m * 2 * I + c * 8), g);
  unpack8(*(const u32x4_t*)(gu + (size_t)m * 2 * I + I + c * 8), u);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a = act == TD_ACT_GELU_TANH ? gelu_tanh_f(g[i]) : act == TD_ACT_GELU_ERF ? gelu_erf_f(g[i]) : silu_f(g[i]);
    g[i] = rbf(a) * u[i];
  }
  *(u32x4_t*)(out + (size_t)m * I + c * 8) = pack8(g);
}

int td_glu_mul_launch(const bf16_t* gu, bf16_t* out, int rows, int I, int act, hipStream_t stream) {
  TD_CHECK_ARG(rows > 0 && I % 8 == 0, "td_glu_mul: bad shape");
  const long long n = (long long)rows * (I / 8);
  TD_GRID_1D(nblk, n, 256, "td_glu_mul");
  hipLaunchKernelGGL(td_glu_mul_kernel, dim3(nblk), dim3(256), 0, stream, gu, out, rows, I, act);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---- vision-tower helpers -------------------------------------------------------------------------
// In-place rotate_half RoPE on heads that sit head_stride apart (zero-padded heads): for i < hd/2
//   x[i] <- x[i] cos_i - x[i+hd/2] sin_i,  x[i+hd/2] <- x[i+hd/2] cos_i + x[i] sin_i   (fp32, one rounding;
// [ext] transformers qwen2_vl apply_rotary_pos_emb_vision).  cos/sin: fp32 [S, hd/2].
__global__ void td_rope_half_kernel(bf16_t* x, int ldx, int S, int H, int head_stride, int hd, const float* cs, const float* sn) {
  const int half = hd >> 1, per_head = half >> 1;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)S * H * per_head) return;
  const int i = (int)(idx % per_head) * 2;
  const int h = (int)((idx / per_head) % H);
  const int s = (int)(idx / ((long long)per_head * H));
  bf16_t* p = x + (size_t)s * ldx + (size_t)h * head_stride + i;
  const unsigned a = *(const unsigned*)p, b = *(const unsigned*)(p + half);
  const float2 c = *(const float2*)(cs + (size_t)s * half + i), n = *(const float2*)(sn + (size_t)s * half + i);
  const float a0 = bf_lo(a), a1 = bf_hi(a), b0 = bf_lo(b), b1 = bf_hi(b);
  *(unsigned*)p = pack_bf2(a0 * c.x - b0 * n.x, a1 * c.y - b1 * n.y);
  *(unsigned*)(p + half) = pack_bf2(b0 * c.x + a0 * n.x, b1 * c.y + a1 * n.y);
}

int td_rope_half_launch(bf16_t* x, int ldx, int S, int H, int head_stride, int hd, const float* cs, const float* sn, hipStream_t stream) {
  TD_CHECK_ARG(S > 0 && H > 0 && hd % 4 == 0 && hd <= head_stride && head_stride % 2 == 0 && ldx % 2 == 0, "td_rope_half: bad shape");
  const long long n = (long long)S * H * (hd / 4);
  TD_GRID_1D(nblk, n, 256, "td_rope_half");
  hipLaunchKernelGGL(td_rope_half_kernel, dim3(nblk), dim3(256), 0, stream, x, ldx, S, H, head_stride, hd, cs, sn);
  TD_CHECK_LAUNCH();
  return 0;
}

// Conv2d(kernel = stride = p) as a GEMM operand: out[(py*gw+px), c*p*p + iy*p + ix] = pix[c, py*p+iy, px*p+ix] as bf16,
// columns [C*p*p, Kpad) zero.  pix: [C,H,W] fp32 (src_f32) or bf16.
__global__ void td_patchify_kernel(const void* pix, int src_f32, int C, int H, int W, int p, bf16_t* out, int Kpad) {
  const int gw = W / p;
  const int patch = blockIdx.x, py = patch / gw, px = patch % gw;
  const int K = C * p * p;
  for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
    float v = 0.f;
    if (k < K) {
      const int c = k / (p * p), iy = (k / p) % p, ix = k % p;
      const size_t src = ((size_t)c * H + (size_t)py * p + iy) * W + (size_t)px * p + ix;
      v = src_f32 ? ((const float*)pix)[src] : bf2f(((const bf16_t*)pix)[src]);
    }
    out[(size_t)patch * Kpad + k] = f2bf(v);
  }
}

int td_patchify_launch(const void* pix, int src_f32, int C, int H, int W, int p, bf16_t* out, int Kpad, hipStream_t stream) {
  TD_CHECK_ARG(C > 0 && p > 0 && H % p == 0 && W % p == 0 && Kpad >= C * p * p, "td_patchify: image %dx%d is not a multiple of the patch %d, or Kpad too small", H, W, p);
  TD_GRID_1D(nblk, (long long)(H / p) * (W / p) * 256, 256, "td_patchify");
  hipLaunchKernelGGL(td_patchify_kernel, dim3(nblk), dim3(256), 0, stream, pix, src_f32, C, H, W, p, out, Kpad);
  TD_CHECK_LAUNCH();
  return 0;
}

// Qwen2-VL image preprocessing after the resize, on the device ([ext] transformers Qwen2VLImageProcessor: rescale, normalize,
// patchify): img uint8 [H, W, 3] -> out bf16 [(H/p)(W/p), Kpad], row = ((bh * (gw/m) + bw) * m + mh) * m + mw for the patch at
// (m bh + mh, m bw + mw) (merge-window order), column = ((c * T + t) * p + py) * p + px (the still image repeated over the T
// temporal slots), value = lut[c][pixel] -- the 3 x 256 table holds the processor's own fp32 rescale / normalize results, so the
// arithmetic is the processor's by construction; columns [3 T p p, Kpad) zero.
__global__ void td_qwen2_patchify_u8_kernel(const unsigned char* img, int H, int W, const float* lut, int p, int m, int T, bf16_t* out, int Kpad) {
  __shared__ float s_lut[768];
  for (int i = threadIdx.x; i < 768; i += blockDim.x) s_lut[i] = lut[i];
  __syncthreads();
  const int gw = W / p, gw2 = gw / m;
  const int row = blockIdx.x;
  const int mw = row % m, mh = (row / m) % m, blk = row / (m * m);
  const int y = (blk / gw2) * m + mh, x = (blk % gw2) * m + mw;
  const int pp = p * p, K = 3 * T * pp;
  for (int k = threadIdx.x; k < Kpad; k += blockDim.x) {
    float v = 0.f;
    if (k < K) {
      const int c = k / (T * pp), r = k % pp, py = r / p, px = r % p;
      v = s_lut[c * 256 + img[((size_t)(y * p + py) * W + (size_t)x * p + px) * 3 + c]];
    }
    out[(size_t)row * Kpad + k] = f2bf(v);
  }
}

int td_qwen2_patchify_u8_launch(const unsigned char* img, int H, int W, const float* lut, int p, int m, int T, bf16_t* out, int Kpad, hipStream_t stream) {
  TD_CHECK_ARG(img && lut && out && p > 0 && m > 0 && T > 0 && H > 0 && W > 0 && H % (p * m) == 0 && W % (p * m) == 0 && Kpad >= 3 * T * p * p,
               "td_qwen2_patchify_u8: image %dx%d is not a multiple of patch x merge = %d, or Kpad too small", H, W, p * m);
  TD_GRID_1D(nblk, (long long)(H / p) * (W / p) * 256, 256, "td_qwen2_patchify_u8");
  hipLaunchKernelGGL(td_qwen2_patchify_u8_kernel, dim3(nblk), dim3(256), 0, stream, img, H, W, lut, p, m, T, out, Kpad);
  TD_CHECK_LAUNCH();
  return 0;
}

// out[r, 0:K] = bf16(src[r, 0:K]), out[r, K:Kpad] = 0   (pre-flattened patches -> GEMM operand)
__global__ void td_cast_pad_rows_kernel(const void* src, int src_f32, int rows, int K, bf16_t* out, int Kpad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)rows * Kpad) return;
  const int r = (int)(idx / Kpad), k = (int)(idx % Kpad);
  float v = 0.f;
  if (k < K) v = src_f32 ? ((const float*)src)[(size_t)r * K + k] : bf2f(((const bf16_t*)src)[(size_t)r * K + k]);
  out[idx] = f2bf(v);
}

int td_cast_pad_rows_launch(const void* src, int src_f32, int rows, int K, bf16_t* out, int Kpad, hipStream_t stream) {
  TD_CHECK_ARG(rows > 0 && K > 0 && Kpad >= K, "td_cast_pad_rows: bad shape");
  const long long n = (long long)rows * Kpad;
  TD_GRID_1D(nblk, n, 256, "td_cast_pad_rows");
  hipLaunchKernelGGL(td_cast_pad_rows_kernel, dim3(nblk), dim3(256), 0, stream, src, src_f32, rows, K, out, Kpad);
  TD_CHECK_LAUNCH();
  return 0;
}

// cos/sin [S, hd/2] fp32 of the Qwen2-VL vision rotary: angle[s, j] = pos[s, j >= hd/4] * theta^(-2 (j mod hd/4) / (hd/2))
// (first half of the columns follows the patch row, second half the patch column; [ext] VisionRotaryEmbedding(head_dim/2)).
__global__ void td_vision_rope_table_kernel(const int* pos, int S, int hd, float theta, float* cs, float* sn) {
  const int half = hd >> 1, quarter = hd >> 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= S * half) return;
  const int s = idx / half, j = idx % half;
  const int axis = j >= quarter, f = j - axis * quarter;
  const float inv = 1.0f / powf(theta, (float)(2 * f) / (float)half);
  const float a = (float)pos[2 * s + axis] * inv;
  cs[idx] = cosf(a);
  sn[idx] = sinf(a);
}

int td_vision_rope_table_launch(const int* pos, int S, int hd, float theta, float* cs, float* sn, hipStream_t stream) {
  TD_CHECK_ARG(pos && cs && sn && S > 0 && hd % 4 == 0, "td_vision_rope_table: bad arguments");
  TD_GRID_1D_I32(nblk, (long long)S * (hd / 2), 256, "td_vision_rope_table");
  hipLaunchKernelGGL(td_vision_rope_table_kernel, dim3(nblk), dim3(256), 0, stream, pos, S, hd, theta, cs, sn);
  TD_CHECK_LAUNCH();
  return 0;
}


// ---- per-row dynamic fp8 (OCP e4m3) quantisation: weights at load time (per output channel), activations per token ----
// col_mul (may be null): fp32 [K], every element is multiplied by its column's factor first -- the smoothing factors of the int8 mode (powers of
// two: s on a weight's input channels, 1 / s on the activation that meets it; td_flux_set_smoothing)
__device__ __forceinline__ void mul8_cols(float (&v)[8], const float* col_mul, int c) {
  if (col_mul) {
    const f32x4_t a = *(const f32x4_t*)(col_mul + c), b = *(const f32x4_t*)(col_mul + c + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] *= a[i]; v[4 + i] *= b[i]; }
  }
}
__global__ __launch_bounds__(256) void td_quant_rows_fp8_kernel(const bf16_t* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int K, int int8, unsigned* amax_out, const float* col_mul) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * ldx;
  float amax = 0.f;
  for (int c = lane * 8; c < K; c += 512) {
    float v[8];
    unpack8(*(const u32x4_t*)(xr + c), v);
    mul8_cols(v, col_mul, c);
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
  }
  amax = wave_max(amax);
  const float s = amax > 0.f ? amax * (int8 ? 1.0f / 127.0f : 1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / s;
  if (lane == 0) { scale[row] = s; if (amax_out) amax_out[row] = as_u32(amax); }
  uint8_t* qr = q + (size_t)row * ldq;
  for (int c = lane * 8; c < K; c += 512) {
    float v[8];
    unpack8(*(const u32x4_t*)(xr + c), v);
    mul8_cols(v, col_mul, c);
    *(u32x2_t*)(qr + c) = pack8_q(v, inv, int8 != 0);
  }
}

// the same with the row held in registers between the amax pass and the conversion (K = NCH * 512 <= 16384): one read of x
template <int NCH>
__global__ __launch_bounds__(256) void td_quant_rows_fp8_reg_kernel(const bf16_t* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int int8, unsigned* amax_out) {      // (col_mul launches take the generic kernel)
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* xr = x + (size_t)row * ldx + lane * 8;
  u32x4_t raw[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) raw[c] = *(const u32x4_t*)(xr + c * 512);
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(bf_lo(raw[c][i])), fabsf(bf_hi(raw[c][i]))));
  amax = wave_max(amax);
  const float s = amax > 0.f ? amax * (int8 ? 1.0f / 127.0f : 1.0f / 448.0f) : 1.0f;
  const float inv = 1.0f / s;
  if (lane == 0) { scale[row] = s; if (amax_out) amax_out[row] = as_u32(amax); }
  uint8_t* qr = q + (size_t)row * ldq + lane * 8;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float v[8];
    unpack8(raw[c], v);
    *(u32x2_t*)(qr + c * 512) = pack8_q(v, inv, int8 != 0);
  }
}

int td_quant_rows_fp8_launch(const bf16_t* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int K, hipStream_t stream, int int8, unsigned* amax_out, const float* col_mul) {
  TD_CHECK_ARG(x && q && scale && rows > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ldq % 8 == 0, "td_quant_rows_fp8: bad arguments");
  TD_CHECK_ARG(!col_mul || ((uintptr_t)col_mul) % 16 == 0, "td_quant_rows_fp8: the column factors must be 16-byte aligned");
  TD_GRID_1D(nblk, (long long)((rows + 3) / 4) * 256, 256, "td_quant_rows_fp8");
  const dim3 grid(nblk), block(256);
  switch (K % 512 == 0 && !col_mul ? K / 512 : 0) {
#define TD_CASE(n) case n: hipLaunchKernelGGL(td_quant_rows_fp8_reg_kernel<n>, grid, block, 0, stream, x, ldx, q, ldq, scale, rows, int8, amax_out); break;
    TD_CASE(1) TD_CASE(2) TD_CASE(4) TD_CASE(6) TD_CASE(8) TD_CASE(24) TD_CASE(30)
#undef TD_CASE
    default: hipLaunchKernelGGL(td_quant_rows_fp8_kernel, grid, block, 0, stream, x, ldx, q, ldq, scale, rows, K, int8, amax_out, col_mul);
  }
  TD_CHECK_LAUNCH();
  return 0;
}

// ---- int8 with scales fixed in advance (the engine's history-scaled mode) ----------------------------------------------------------------
// amax bits -> this step's scale = amax * margin / 127 and its inverse; the accumulators are cleared for the step being started.
// amax == 0 is "no history for this row" (the engine invalidates history whenever the tensor may not have been produced in 8-bit form last
// step, so what is left is a row that really was all zeros): such a row takes the unit step, not a vanishing one that would saturate
// every byte of the next non-zero value.
__global__ void td_q8_scales_from_amax_kernel(unsigned* amax, float* scale, float* inv, long long n, float margin) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = as_f32(amax[i]);
  const float s = a > 0.f ? a * margin * (1.0f / 127.0f) : 1.0f;
  scale[i] = s;
  inv[i] = 1.0f / s;
  amax[i] = 0u;
}

int td_q8_scales_from_amax_launch(unsigned* amax, float* scale, float* inv, long long n, float margin, hipStream_t stream) {
  TD_CHECK_ARG(amax && scale && inv && n > 0 && margin >= 1.0f, "td_q8_scales_from_amax: bad arguments");
  TD_GRID_1D(nblk, n, 256, "td_q8_scales_from_amax");
  hipLaunchKernelGGL(td_q8_scales_from_amax_kernel, dim3(nblk), dim3(256), 0, stream, amax, scale, inv, n, margin);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---- int8 smoothing (td_flux_set_smoothing): per-channel maxima and the factors made from them ---------------------------------------------------
// amax[c] = max(amax[c], max_r |x[r, c]|) over rows [0, rows) (atomic max on the float bits: all values are >= 0).  One thread owns 8 columns of a
// 64-row strip.  Used on activations during the calibration forward and on weights (columns = input channels) when the factors are made.
__global__ __launch_bounds__(256) void td_col_amax_kernel(const bf16_t* x, int ldx, int rows, int K, unsigned* amax) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
  if (c >= K) return;
  const int r0 = blockIdx.y * 64, r1 = min(rows, r0 + 64);
  float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = r0; r < r1; ++r) {
    float v[8];
    unpack8(*(const u32x4_t*)(x + (size_t)r * ldx + c), v);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], fabsf(v[i]));
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (m[i] > 0.f) atomicMax(amax + c + i, as_u32(m[i]));
}
int td_col_amax_launch(const bf16_t* x, int ldx, int rows, int K, unsigned* amax, hipStream_t stream) {
  TD_CHECK_ARG(x && amax && rows > 0 && K > 0 && K % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x) % 16 == 0, "td_col_amax: bad arguments");
  const dim3 grid((K / 8 + 255) / 256, (rows + 63) / 64);
  TD_CHECK_ARG(grid.y < 65536, "td_col_amax: %d rows exceed the grid", rows);
  hipLaunchKernelGGL(td_col_amax_kernel, grid, dim3(256), 0, stream, x, ldx, rows, K, amax);
  TD_CHECK_LAUNCH();
  return 0;
}
// SmoothQuant's balance with alpha = 1/2, rounded to a power of two: s[c] = 2^rint(log2(sqrt(amax_x[c] / amax_w[c]))) in [2^-8, 2^8] -- the
// activation channel is divided by s, the weight's input channel multiplied by it; both are exact on bf16 values, so the product is unchanged and
// only where the quantisation steps fall moves.  A channel never seen (either maximum 0) keeps s = 1.
__global__ void td_smooth_factors_kernel(const unsigned* ax, const unsigned* aw, int n, float* s, float* inv, bf16_t* inv16) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = as_f32(ax[i]), w = as_f32(aw[i]);
  float e = 0.f;
  if (a > 0.f && w > 0.f) e = fminf(fmaxf(rintf(0.5f * (log2f(a) - log2f(w))), -8.f), 8.f);
  const float sv = exp2f(e);
  s[i] = sv;
  inv[i] = 1.0f / sv;
  inv16[i] = f2bf(1.0f / sv);
}
int td_smooth_factors_launch(const unsigned* ax, const unsigned* aw, int n, float* s, float* inv, bf16_t* inv16, hipStream_t stream) {
  TD_CHECK_ARG(ax && aw && s && inv && inv16 && n > 0, "td_smooth_factors: bad arguments");
  hipLaunchKernelGGL(td_smooth_factors_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ax, aw, n, s, inv, inv16);
  TD_CHECK_LAUNCH();
  return 0;
}

// q[r, K + e] = q[r, ext[e]] (0 where ext[e] < 0): the replicated input channels of an int8 weight (TdNormParams::ext)
__global__ void td_ext_cols_kernel(uint8_t* q, int ld, int rows, int K, const int* ext, int ext_n) {
  const int r = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  uint8_t* qr = q + (size_t)r * ld;
  for (int e = lane; e < ext_n; e += 64) {
    const int src = ext[e];
    qr[K + e] = src >= 0 ? qr[src] : (uint8_t)0;
  }
}
int td_ext_cols_launch(uint8_t* q, int ld, int rows, int K, const int* ext, int ext_n, hipStream_t stream) {
  TD_CHECK_ARG(q && ext && rows > 0 && K > 0 && ext_n > 0 && ld >= K + ext_n, "td_ext_cols: bad arguments");
  hipLaunchKernelGGL(td_ext_cols_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, q, ld, rows, K, ext, ext_n);
  TD_CHECK_LAUNCH();
  return 0;
}

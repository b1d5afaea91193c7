// Row / pixel kernels of the FLUX VAE decoder: GroupNorm(+SiLU) over NHWC
// images, fp32 row softmax for the mid-block attention, weight packing, latent unpack and image finalisation.
// All HBM-bound; 16 B per lane.  Rounding points follow the bf16 torch graph of [ext] diffusers
// AutoencoderKL.decode (GroupNorm output, SiLU output each round to bf16; statistics in fp32).
#include "td_common.h"
#include "td_kernels.h"

namespace {

__device__ __forceinline__ void unpack8v(const u32x4_t v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = bf_lo(v[i]);
    f[2 * i + 1] = bf_hi(v[i]);
  }
}
__device__ __forceinline__ u32x4_t pack8v(const float (&f)[8]) {
  u32x4_t v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = pack_bf2(f[2 * i], f[2 * i + 1]);
  return v;
}

}  // namespace

// ---- GroupNorm, pass 1: per-block partial sums.  x [P, C] NHWC, G groups of C/G channels --------------
// partial[block][g][0..1] = (sum, sum of squares) over the block's pixels
__global__ __launch_bounds__(256) void td_gn_partial_kernel(const bf16_t* x, int P, int C, int G, int pix_per_block, float* partial) {
  // per-thread sub-sums, then a fixed-order sum per group: bit-reproducible (shared-memory float atomics are not)
  __shared__ float buf[256][8][2];
  const int tpp = C / 8;                       // threads per pixel
  const int ppp = 256 / tpp;                   // pixels per pass (C <= 2048)
  const int c0 = (threadIdx.x % tpp) * 8;
  const int p_in = threadIdx.x / tpp;
  const int cpg = C / G;                       // 1, 2, 4 or a multiple of 8 (checked by the launcher)
  float s[8], q[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
  const int pbeg = blockIdx.x * pix_per_block, pend = min(P, pbeg + pix_per_block);
  if (p_in < ppp)
    for (int px = pbeg + p_in; px < pend; px += ppp) {
      float v[8];
      unpack8v(*(const u32x4_t*)(x + (size_t)px * C + c0), v);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[i] += v[i]; q[i] += v[i] * v[i]; }
    }
  const int nsub = cpg >= 8 ? 1 : 8 / cpg, per = 8 / nsub;   // sub-groups of this thread's 8 channels
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    float su = 0.f, qu = 0.f;
    if (u < nsub)
      for (int i = 0; i < per; ++i) { su += s[u * per + i]; qu += q[u * per + i]; }
    buf[threadIdx.x][u][0] = su;
    buf[threadIdx.x][u][1] = qu;
  }
  __syncthreads();
  if (threadIdx.x < G) {
    const int g = threadIdx.x;
    float su = 0.f, qu = 0.f;
    const int j0 = g * cpg / 8, nj = cpg >= 8 ? cpg / 8 : 1, sub = cpg >= 8 ? 0 : g % nsub;
    for (int pp = 0; pp < ppp; ++pp)
      for (int j = j0; j < j0 + nj; ++j) {
        su += buf[pp * tpp + j][sub][0];
        qu += buf[pp * tpp + j][sub][1];
      }
    partial[((size_t)blockIdx.x * G + g) * 2] = su;
    partial[((size_t)blockIdx.x * G + g) * 2 + 1] = qu;
  }
}

// pass 2: stats[g] = (mean, rstd); one wave per group sums the block partials in double
__global__ __launch_bounds__(64) void td_gn_finalize_kernel(const float* partial, int nblocks, int G, double count, float eps, float* stats) {
  const int g = blockIdx.x, lane = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int b = lane; b < nblocks; b += 64) {
    s += partial[((size_t)b * G + g) * 2];
    q += partial[((size_t)b * G + g) * 2 + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane == 0) {
    const double mean = s / count;
    const double var = fmax(q / count - mean * mean, 0.0);
    stats[2 * g] = (float)mean;
    stats[2 * g + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// pass 3: y = bf16((x - mean) rstd gamma + beta), optionally y = bf16(silu(y))
__global__ __launch_bounds__(256) void td_gn_apply_kernel(const bf16_t* x, bf16_t* y, long long n8, int C, int G, const float* stats,
                                                          const bf16_t* gamma, const bf16_t* beta, int silu) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n8) return;
  const int c0 = (int)(idx % (C / 8)) * 8;
  const int cpg = C / G;
  float v[8], ga[8], be[8];
  unpack8v(((const u32x4_t*)x)[idx], v);
  unpack8v(*(const u32x4_t*)(gamma + c0), ga);
  unpack8v(*(const u32x4_t*)(beta + c0), be);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int g = (c0 + i) / cpg;
    float t = rbf((v[i] - stats[2 * g]) * stats[2 * g + 1] * ga[i] + be[i]);
    if (silu) t = rbf(silu_f(t));
    v[i] = t;
  }
  ((u32x4_t*)y)[idx] = pack8v(v);
}

int td_groupnorm_nhwc_launch(const bf16_t* x, bf16_t* y, int P, int C, int G, float eps, const bf16_t* gamma, const bf16_t* beta,
                             int silu, float* workspace, hipStream_t stream) {
  TD_CHECK_ARG(P > 0 && C % 8 == 0 && C <= 2048 && G > 0 && G <= 64 && C % G == 0, "td_groupnorm: bad shape P=%d C=%d G=%d", P, C, G);
  TD_CHECK_ARG((C / G) % 8 == 0 || 8 % (C / G) == 0, "td_groupnorm: channels per group (%d) must divide 8 or be a multiple of 8", C / G);
  TD_CHECK_ARG(workspace != nullptr, "td_groupnorm: workspace of td_groupnorm_workspace_floats() floats required");
  const int nblocks = min(1024, (P + 63) / 64);
  const int ppb = (P + nblocks - 1) / nblocks;
  float* partial = workspace;
  float* stats = workspace + (size_t)1024 * 64 * 2;
  hipLaunchKernelGGL(td_gn_partial_kernel, dim3(nblocks), dim3(256), 0, stream, x, P, C, G, ppb, partial);
  hipLaunchKernelGGL(td_gn_finalize_kernel, dim3(G), dim3(64), 0, stream, partial, nblocks, G, (double)P * (C / G), eps, stats);
  const long long n8 = (long long)P * C / 8;
  TD_GRID_1D(nblk_apply, n8, 256, "td_groupnorm_nhwc");
  hipLaunchKernelGGL(td_gn_apply_kernel, dim3(nblk_apply), dim3(256), 0, stream, x, y, n8, C, G, stats, gamma, beta, silu);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---- fp32 row softmax -> bf16:  p[r, :] = softmax(scale * s[r, :]),  one workgroup per row ---------------
__global__ __launch_bounds__(256) void td_softmax_rows_kernel(const float* s, bf16_t* p, int cols, float scale) {
  __shared__ float red[8];
  const float* sr = s + (size_t)blockIdx.x * cols;
  bf16_t* pr = p + (size_t)blockIdx.x * cols;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float mx = -INFINITY;
  for (int c = threadIdx.x * 4; c < cols; c += 1024) {
    const f32x4_t v = *(const f32x4_t*)(sr + c);
    mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
  }
  mx = wave_max(mx);
  if (lane == 0) red[w] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float k = scale * 1.4426950408889634f;
  float sum = 0.f;
  for (int c = threadIdx.x * 4; c < cols; c += 1024) {
    const f32x4_t v = *(const f32x4_t*)(sr + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += __builtin_amdgcn_exp2f((v[i] - mx) * k);
  }
  sum = wave_sum(sum);
  __syncthreads();
  if (lane == 0) red[4 + w] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
  for (int c = threadIdx.x * 4; c < cols; c += 1024) {
    const f32x4_t v = *(const f32x4_t*)(sr + c);
    u32x2_t o;
    o[0] = pack_bf2(__builtin_amdgcn_exp2f((v[0] - mx) * k) * inv, __builtin_amdgcn_exp2f((v[1] - mx) * k) * inv);
    o[1] = pack_bf2(__builtin_amdgcn_exp2f((v[2] - mx) * k) * inv, __builtin_amdgcn_exp2f((v[3] - mx) * k) * inv);
    *(u32x2_t*)(pr + c) = o;
  }
}

int td_softmax_rows_launch(const float* s, bf16_t* p, int rows, int cols, float scale, hipStream_t stream) {
  TD_CHECK_ARG(rows > 0 && cols > 0 && cols % 4 == 0, "td_softmax_rows: cols=%d must be a positive multiple of 4", cols);
  TD_GRID_1D(nblk, (long long)rows * 256, 256, "td_softmax_rows");
  hipLaunchKernelGGL(td_softmax_rows_kernel, dim3(nblk), dim3(256), 0, stream, s, p, cols, scale);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---- conv weight packing: [Cout, Cin, 3, 3] -> [Cout_pad, 9 * Cin_pad], k = (ky*3 + kx) * Cin_pad + c ------
__global__ void td_conv_pack_kernel(const bf16_t* w, bf16_t* out, int Cout, int Cin, int Cout_pad, int Cin_pad) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)Cout_pad * 9 * Cin_pad;
  if (idx >= total) return;
  const int c = (int)(idx % Cin_pad);
  const int tap = (int)((idx / Cin_pad) % 9);
  const int o = (int)(idx / ((long long)9 * Cin_pad));
  out[idx] = (o < Cout && c < Cin) ? w[((size_t)o * Cin + c) * 9 + tap] : (bf16_t)0;
}

int td_conv_pack_launch(const bf16_t* w, bf16_t* out, int Cout, int Cin, int Cout_pad, int Cin_pad, hipStream_t stream) {
  TD_CHECK_ARG(Cout > 0 && Cin > 0 && Cout_pad >= Cout && Cin_pad >= Cin, "td_conv3x3_pack_weight: bad shape");
  const long long total = (long long)Cout_pad * 9 * Cin_pad;
  TD_GRID_1D(nblk, total, 256, "td_conv_pack");
  hipLaunchKernelGGL(td_conv_pack_kernel, dim3(nblk), dim3(256), 0, stream, w, out, Cout, Cin, Cout_pad, Cin_pad);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---- FLUX packed latents [ (h/2)(w/2), 4C ] -> NHWC [h*w, Cpad] with z*mul + add (channels >= C zero) ---------
__global__ void td_latents_to_nhwc_kernel(const bf16_t* packed, bf16_t* out, int C, int h, int w, int Cpad, float div, float add) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= h * w * Cpad) return;
  const int c = idx % Cpad, pix = idx / Cpad;
  if (c >= C) { out[idx] = 0; return; }
  const int y = pix / w, x = pix % w;
  const int tok = (y >> 1) * (w >> 1) + (x >> 1);
  const int col = c * 4 + (y & 1) * 2 + (x & 1);
  // latents / scaling_factor + shift_factor on bf16 tensors ([ext] pipeline_flux.py): the quotient rounds to bf16, then the sum
  // (torch CPU semantics: fp32 scalar divisor, python-scalar addend cast to bf16 first)
  out[idx] = f2bf(__fadd_rn(rbf(__fdiv_rn(bf2f(packed[(size_t)tok * (4 * C) + col]), div)), rbf(add)));
}

int td_latents_to_nhwc_launch(const bf16_t* packed, bf16_t* out, int C, int h, int w, int Cpad, float div, float add, hipStream_t stream) {
  TD_CHECK_ARG(C > 0 && h % 2 == 0 && w % 2 == 0 && Cpad >= C, "td_latents_to_nhwc: bad shape");
  TD_GRID_1D_I32(nblk, (long long)h * w * Cpad, 256, "td_latents_to_nhwc");
  hipLaunchKernelGGL(td_latents_to_nhwc_kernel, dim3(nblk), dim3(256), 0, stream, packed, out, C, h, w, Cpad, div, add);
  TD_CHECK_LAUNCH();
  return 0;
}

// ---- VaeImageProcessor.postprocess: u8[H,W,3] = round(clamp(x/2 + 0.5, 0, 1) * 255) from NHWC [H*W, Cpad] --------
// also (optionally) the bf16 [3,H,W] tensor the VAE returns
__global__ void td_image_finalize_kernel(const bf16_t* x, int P, int Cpad, unsigned char* u8, bf16_t* chw) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * 3) return;
  const int pix = idx / 3, c = idx % 3;
  const bf16_t raw = x[(size_t)pix * Cpad + c];
  if (chw) chw[(size_t)c * P + pix] = raw;
  if (u8) {
    // (image / 2 + 0.5).clamp(0, 1) in bf16, then .float() * 255, round, uint8  [ext image_processor.py denormalize / pt_to_numpy / numpy_to_pil]
    const float v = fminf(fmaxf(rbf(rbf(bf2f(raw) * 0.5f) + 0.5f), 0.f), 1.f);
    u8[idx] = (unsigned char)rintf(v * 255.0f);
  }
}

int td_image_finalize_launch(const bf16_t* x, int P, int Cpad, unsigned char* u8, bf16_t* chw, hipStream_t stream) {
  TD_CHECK_ARG(P > 0 && Cpad >= 3 && (u8 || chw), "td_image_finalize: bad arguments");
  TD_GRID_1D_I32(nblk, (long long)P * 3, 256, "td_image_finalize");
  hipLaunchKernelGGL(td_image_finalize_kernel, dim3(nblk), dim3(256), 0, stream, x, P, Cpad, u8, chw);
  TD_CHECK_LAUNCH();
  return 0;
}

// FLUX VAE decoder engine (AutoencoderKL.decode): packed latents -> uint8 image, all on the HIP kernels.
//
// Replaces, inside the reference drivers' `diffusion_pipe(...)` call
// (scripts/test/test_blip_vision_t5_decoder_flux_text.py:234-242), the tail of [ext] diffusers 0.31.0
// FluxPipeline.__call__: `_unpack_latents`, `latents / scaling_factor + shift_factor`,
// `vae.decode` ([ext] autoencoder_kl.py / vae.py Decoder: conv_in, UNetMidBlock2D (ResnetBlock2D, single-head
// Attention, ResnetBlock2D), 4 UpDecoderBlock2D (3 ResnetBlock2D + Upsample2D), GroupNorm, SiLU, conv_out) and
// `image_processor.postprocess` (denormalise, uint8).
//
// Layout: images are NHWC ([pixels, channels] rows), so every 3x3 conv is an implicit GEMM on the MFMA GEMM
// kernel (no im2col buffer, zero padding and the nearest 2x upsample folded into the A-operand addressing),
// 1x1 convs and the attention projections are plain GEMMs, GroupNorm+SiLU is a 3-launch row kernel.
// The mid-block attention has ONE head of width 512: scores are produced in fp32 row chunks by the GEMM
// (fp32 output), softmaxed by a row kernel and multiplied with V^T by the GEMM again; to_v's bias is added
// after the product (softmax rows sum to 1).
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include <algorithm>
#include <cmath>

#include "td_kernels.h"
#include "../../include/thinkdiff_hip.h"

namespace {

struct VSlot { std::string name; bf16_t* ptr; int64_t count; int kind; int cout, cin, cout_pad, cin_pad; };  // kind 0 plain, 1 conv3x3

struct Resnet {
  int cin, cout;
  bf16_t *n1_w, *n1_b, *c1_w, *c1_b, *n2_w, *n2_b, *c2_w, *c2_b, *sc_w, *sc_b;
};

#define TDV_TRY(expr)         \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)

int pad64(int c) { return (c + 63) & ~63; }
int pad8(int c) { return (c + 7) & ~7; }

}  // namespace

struct td_vae {
  TdVaeConfig cfg;
  int nb = 0, cmid = 0, lat_pad = 0, out_pad = 0, max_lat_pixels = 0;
  bf16_t* arena = nullptr;
  int64_t arena_elems = 0;
  std::vector<VSlot> slots;
  std::unordered_map<std::string, int> index;
  bf16_t *cin_w, *cin_b, *nout_w, *nout_b, *cout_w, *cout_b;
  Resnet mid[2];
  bf16_t *agn_w, *agn_b, *aq_w, *aq_b, *ak_w, *ak_b, *av_w, *av_b, *ao_w, *ao_b;
  std::vector<std::vector<Resnet>> up;
  std::vector<bf16_t*> ups_w, ups_b;
  // workspace
  char* ws = nullptr;
  bf16_t *X, *T1, *T2, *T3, *Q, *K, *VT, *P;
  float *S, *gn;
  int chunk_rows = 0;
};

namespace {

struct Plan {
  int64_t off = 0;
  std::vector<std::pair<bf16_t**, int64_t>> fix;
  void take(bf16_t** p, int64_t n) { fix.emplace_back(p, off); off += (n + 127) & ~int64_t(127); }
};

void v_add(td_vae* f, const std::string& name, bf16_t* p, int64_t count, int kind = 0, int cout = 0, int cin = 0, int cout_pad = 0, int cin_pad = 0) {
  f->index[name] = (int)f->slots.size();
  f->slots.push_back({name, p, count, kind, cout, cin, cout_pad, cin_pad});
}

void plan_resnet(Plan& pl, Resnet& r, int cin, int cout) {
  r.cin = cin; r.cout = cout;
  pl.take(&r.n1_w, cin); pl.take(&r.n1_b, cin);
  pl.take(&r.c1_w, (int64_t)cout * 9 * cin); pl.take(&r.c1_b, cout);
  pl.take(&r.n2_w, cout); pl.take(&r.n2_b, cout);
  pl.take(&r.c2_w, (int64_t)cout * 9 * cout); pl.take(&r.c2_b, cout);
  r.sc_w = r.sc_b = nullptr;
  if (cin != cout) { pl.take(&r.sc_w, (int64_t)cout * cin); pl.take(&r.sc_b, cout); }
}

void name_resnet(td_vae* f, const std::string& p, const Resnet& r) {
  v_add(f, p + "norm1.weight", r.n1_w, r.cin); v_add(f, p + "norm1.bias", r.n1_b, r.cin);
  v_add(f, p + "conv1.weight", r.c1_w, (int64_t)r.cout * r.cin * 9, 1, r.cout, r.cin, r.cout, r.cin);
  v_add(f, p + "conv1.bias", r.c1_b, r.cout);
  v_add(f, p + "norm2.weight", r.n2_w, r.cout); v_add(f, p + "norm2.bias", r.n2_b, r.cout);
  v_add(f, p + "conv2.weight", r.c2_w, (int64_t)r.cout * r.cout * 9, 1, r.cout, r.cout, r.cout, r.cout);
  v_add(f, p + "conv2.bias", r.c2_b, r.cout);
  if (r.sc_w) { v_add(f, p + "conv_shortcut.weight", r.sc_w, (int64_t)r.cout * r.cin, 0, r.cout, r.cin); v_add(f, p + "conv_shortcut.bias", r.sc_b, r.cout); }
}

int conv3(hipStream_t s, const bf16_t* x, const bf16_t* w, const bf16_t* b, const bf16_t* res, bf16_t* y, int H, int W, int cin, int cout, int up) {
  TdGemmParams p;
  p.A = x; p.lda = cin; p.W = w; p.bias = b; p.C = y; p.ldc = cout; p.res = res; p.ldr = cout;
  p.M = H * W; p.N = cout; p.K = 9 * cin; p.conv_H = H; p.conv_W = W; p.conv_Cin = cin; p.conv_up = up;
  return td_gemm_launch(p, s);
}

int lin(hipStream_t s, const bf16_t* x, int ldx, const bf16_t* w, const bf16_t* b, bf16_t* y, int ldy, int M, int N, int K, const bf16_t* res = nullptr) {
  TdGemmParams p;
  p.A = x; p.lda = ldx; p.W = w; p.bias = b; p.C = y; p.ldc = ldy; p.M = M; p.N = N; p.K = K; p.res = res; p.ldr = ldy;
  return td_gemm_launch(p, s);
}

int gn(td_vae* f, hipStream_t s, const bf16_t* x, bf16_t* y, int P, int C, const bf16_t* w, const bf16_t* b, int silu) {
  return td_groupnorm_nhwc_launch(x, y, P, C, f->cfg.norm_groups, 1e-6f, w, b, silu, f->gn, s);
}

// x (in X) -> X, using T1..T3;  ResnetBlock2D: x + conv2(silu(gn2(conv1(silu(gn1(x))))))  (shortcut 1x1 when cin != cout)
int resnet(td_vae* f, hipStream_t s, const Resnet& r, int H, int W) {
  const int P = H * W;
  TDV_TRY(gn(f, s, f->X, f->T1, P, r.cin, r.n1_w, r.n1_b, 1));
  TDV_TRY(conv3(s, f->T1, r.c1_w, r.c1_b, nullptr, f->T2, H, W, r.cin, r.cout, 0));
  TDV_TRY(gn(f, s, f->T2, f->T1, P, r.cout, r.n2_w, r.n2_b, 1));
  const bf16_t* sc = f->X;
  if (r.sc_w) { TDV_TRY(lin(s, f->X, r.cin, r.sc_w, r.sc_b, f->T3, r.cout, P, r.cout, r.cin)); sc = f->T3; }
  TDV_TRY(conv3(s, f->T1, r.c2_w, r.c2_b, sc, f->X, H, W, r.cout, r.cout, 0));
  return 0;
}

}  // namespace

extern "C" {

int td_vae_create(const TdVaeConfig* cfg, int max_latent_h, int max_latent_w, td_vae** out) {
  TD_CHECK_ARG(cfg && out && max_latent_h > 0 && max_latent_w > 0, "td_vae_create: bad arguments");
  TD_CHECK_ARG(cfg->num_blocks >= 1 && cfg->num_blocks <= 4, "td_vae_create: 1..4 blocks supported");
  for (int i = 0; i < cfg->num_blocks; ++i)
    TD_CHECK_ARG(cfg->block_out_channels[i] % 64 == 0 && cfg->block_out_channels[i] % cfg->norm_groups == 0,
                 "td_vae_create: block_out_channels must be multiples of 64 and of norm_groups");
  td_vae* f = new td_vae();
  f->cfg = *cfg;
  const int nb = f->nb = cfg->num_blocks;
  const int cmid = f->cmid = cfg->block_out_channels[nb - 1];
  f->lat_pad = pad64(cfg->latent_channels);
  f->out_pad = pad8(cfg->out_channels);
  f->max_lat_pixels = max_latent_h * max_latent_w;

  Plan pl;
  pl.take(&f->cin_w, (int64_t)cmid * 9 * f->lat_pad); pl.take(&f->cin_b, cmid);
  plan_resnet(pl, f->mid[0], cmid, cmid);
  plan_resnet(pl, f->mid[1], cmid, cmid);
  pl.take(&f->agn_w, cmid); pl.take(&f->agn_b, cmid);
  pl.take(&f->aq_w, (int64_t)cmid * cmid); pl.take(&f->aq_b, cmid);
  pl.take(&f->ak_w, (int64_t)cmid * cmid); pl.take(&f->ak_b, cmid);
  pl.take(&f->av_w, (int64_t)cmid * cmid); pl.take(&f->av_b, cmid);
  pl.take(&f->ao_w, (int64_t)cmid * cmid); pl.take(&f->ao_b, cmid);
  f->up.resize(nb); f->ups_w.assign(nb, nullptr); f->ups_b.assign(nb, nullptr);
  int prev = cmid;
  for (int b = 0; b < nb; ++b) {   // reversed block_out_channels
    const int co = cfg->block_out_channels[nb - 1 - b];
    f->up[b].resize(cfg->layers_per_block + 1);
    for (int r = 0; r <= cfg->layers_per_block; ++r) plan_resnet(pl, f->up[b][r], r == 0 ? prev : co, co);
    if (b != nb - 1) { pl.take(&f->ups_w[b], (int64_t)co * 9 * co); pl.take(&f->ups_b[b], co); }
    prev = co;
  }
  const int clast = cfg->block_out_channels[0];
  pl.take(&f->nout_w, clast); pl.take(&f->nout_b, clast);
  pl.take(&f->cout_w, (int64_t)f->out_pad * 9 * clast); pl.take(&f->cout_b, f->out_pad);
  f->arena_elems = pl.off;
  hipError_t e = hipMalloc((void**)&f->arena, (size_t)pl.off * 2);
  if (e != hipSuccess) { td_set_error("td_vae_create: weight hipMalloc failed: %s", hipGetErrorString(e)); delete f; return TD_ERR_HIP; }
  (void)hipMemset(f->arena, 0, (size_t)pl.off * 2);   // padded weight rows / channels must be zero
  (void)hipDeviceSynchronize();   // the handle may be used from any stream next; a null-stream memset is not ordered with non-blocking streams
  for (auto& fx : pl.fix) *fx.first = f->arena + fx.second;

  // diffusers state-dict names
  v_add(f, "decoder.conv_in.weight", f->cin_w, (int64_t)cmid * cfg->latent_channels * 9, 1, cmid, cfg->latent_channels, cmid, f->lat_pad);
  v_add(f, "decoder.conv_in.bias", f->cin_b, cmid);
  name_resnet(f, "decoder.mid_block.resnets.0.", f->mid[0]);
  name_resnet(f, "decoder.mid_block.resnets.1.", f->mid[1]);
  const std::string a = "decoder.mid_block.attentions.0.";
  v_add(f, a + "group_norm.weight", f->agn_w, cmid); v_add(f, a + "group_norm.bias", f->agn_b, cmid);
  v_add(f, a + "to_q.weight", f->aq_w, (int64_t)cmid * cmid, 0, cmid, cmid); v_add(f, a + "to_q.bias", f->aq_b, cmid);
  v_add(f, a + "to_k.weight", f->ak_w, (int64_t)cmid * cmid, 0, cmid, cmid); v_add(f, a + "to_k.bias", f->ak_b, cmid);
  v_add(f, a + "to_v.weight", f->av_w, (int64_t)cmid * cmid, 0, cmid, cmid); v_add(f, a + "to_v.bias", f->av_b, cmid);
  v_add(f, a + "to_out.0.weight", f->ao_w, (int64_t)cmid * cmid, 0, cmid, cmid); v_add(f, a + "to_out.0.bias", f->ao_b, cmid);
  for (int b = 0; b < nb; ++b) {
    const std::string ub = "decoder.up_blocks." + std::to_string(b) + ".";
    for (int r = 0; r <= cfg->layers_per_block; ++r) name_resnet(f, ub + "resnets." + std::to_string(r) + ".", f->up[b][r]);
    if (f->ups_w[b]) {
      const int co = f->up[b][0].cout;
      v_add(f, ub + "upsamplers.0.conv.weight", f->ups_w[b], (int64_t)co * co * 9, 1, co, co, co, co);
      v_add(f, ub + "upsamplers.0.conv.bias", f->ups_b[b], co);
    }
  }
  v_add(f, "decoder.conv_norm_out.weight", f->nout_w, clast); v_add(f, "decoder.conv_norm_out.bias", f->nout_b, clast);
  v_add(f, "decoder.conv_out.weight", f->cout_w, (int64_t)cfg->out_channels * clast * 9, 1, cfg->out_channels, clast, f->out_pad, clast);
  v_add(f, "decoder.conv_out.bias", f->cout_b, cfg->out_channels);

  // workspace: largest image buffers along the decode path
  int64_t maxX = (int64_t)f->max_lat_pixels * cmid, maxT2 = maxX, px = f->max_lat_pixels;
  prev = cmid;
  for (int b = 0; b < nb; ++b) {
    const int co = cfg->block_out_channels[nb - 1 - b];
    maxX = std::max(maxX, px * std::max(prev, co));
    maxT2 = std::max(maxT2, px * co);
    if (b != nb - 1) { px *= 4; maxX = std::max(maxX, px * co); }
    prev = co;
  }
  f->chunk_rows = std::min(2048, f->max_lat_pixels);
  struct Req { void** p; int64_t bytes; };
  std::vector<Req> reqs = {
      {(void**)&f->X, maxX * 2}, {(void**)&f->T1, maxX * 2}, {(void**)&f->T2, maxT2 * 2}, {(void**)&f->T3, maxT2 * 2},
      {(void**)&f->Q, (int64_t)f->max_lat_pixels * cmid * 2}, {(void**)&f->K, (int64_t)f->max_lat_pixels * cmid * 2},
      {(void**)&f->VT, (int64_t)f->max_lat_pixels * cmid * 2},
      {(void**)&f->S, (int64_t)f->chunk_rows * f->max_lat_pixels * 4}, {(void**)&f->P, (int64_t)f->chunk_rows * f->max_lat_pixels * 2},
      {(void**)&f->gn, (int64_t)(1024 * 64 * 2 + 256) * 4},
  };
  int64_t total = 0;
  for (auto& r : reqs) total += (r.bytes + 255) & ~int64_t(255);
  e = hipMalloc((void**)&f->ws, (size_t)total);
  if (e != hipSuccess) {
    td_set_error("td_vae_create: hipMalloc of %.2f GiB workspace failed: %s", total / double(1 << 30), hipGetErrorString(e));
    (void)hipFree(f->arena); delete f; return TD_ERR_HIP;
  }
  int64_t o = 0;
  for (auto& r : reqs) { *r.p = f->ws + o; o += (r.bytes + 255) & ~int64_t(255); }
  *out = f;
  return TD_OK;
}

void td_vae_destroy(td_vae* f) {
  if (!f) return;
  (void)hipFree(f->arena);
  (void)hipFree(f->ws);
  delete f;
}

int td_vae_num_params(const td_vae* f) { return f ? (int)f->slots.size() : 0; }

int td_vae_param_info(const td_vae* f, int idx, char* name_buf, int buf_len, int64_t* count) {
  TD_CHECK_ARG(f && idx >= 0 && idx < (int)f->slots.size(), "td_vae_param_info: index %d out of range", idx);
  if (name_buf && buf_len > 0) { strncpy(name_buf, f->slots[idx].name.c_str(), buf_len - 1); name_buf[buf_len - 1] = 0; }
  if (count) *count = f->slots[idx].count;
  return TD_OK;
}

// src: device bf16 in the torch layout ([Cout,Cin,3,3] for 3x3 convs, [out,in(,1,1)] otherwise, [C] vectors)
int td_vae_load_param(td_vae* f, const char* name, const void* src, int64_t count, void* stream) {
  TD_CHECK_ARG(f && name && src, "td_vae_load_param: null argument");
  auto it = f->index.find(name);
  TD_CHECK_ARG(it != f->index.end(), "td_vae_load_param: unknown parameter '%s'", name);
  const VSlot& s = f->slots[it->second];
  TD_CHECK_ARG(s.count == count, "td_vae_load_param: '%s' expects %lld elements, got %lld", name, (long long)s.count, (long long)count);
  if (s.kind == 1) return td_conv_pack_launch((const bf16_t*)src, s.ptr, s.cout, s.cin, s.cout_pad, s.cin_pad, (hipStream_t)stream);
  TD_CHECK_HIP(hipMemcpyAsync(s.ptr, src, (size_t)count * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return TD_OK;
}

int td_vae_init_random(td_vae* f, uint64_t seed, float std, void* stream) {
  TD_CHECK_ARG(f, "td_vae_init_random: null handle");
  for (const VSlot& s : f->slots) {
    const bool norm_w = s.name.find("norm") != std::string::npos && s.name.find(".weight") != std::string::npos;
    const int64_t n = s.kind == 1 ? (int64_t)s.cout * 9 * s.cin_pad : s.count;   // padded output rows stay zero
    // std <= 0: variance-preserving weights (1 / sqrt(fan_in) for convolutions and linears, 0.02 for biases), so that a synthetic
    // decoder maps unit-scale latents to an image with contrast instead of a flat grey one
    float sd = std;
    if (std <= 0.f) sd = s.cin > 0 ? 1.0f / sqrtf((float)(s.kind == 1 ? 9 * s.cin : s.cin)) : 0.02f;
    TDV_TRY(td_fill_normal_bf16(s.ptr, n, seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(s.ptr - f->arena + 1)), norm_w ? 0.05f : sd, norm_w ? 1.0f : 0.0f, stream));
  }
  // padded input channels of conv_in must see zero weights regardless (their activations are zero anyway)
  return TD_OK;
}

// packed FLUX latents [ (h/2)(w/2), 4*latent_channels ] bf16 -> image.  h, w: latent height/width (image = 8h x 8w
// for the 4-block FLUX VAE).  image_u8: [H, W, 3] uint8 (may be NULL); image_chw: bf16 [3, H, W] = vae.decode output
// (may be NULL).  The z / scaling_factor + shift_factor step of the pipeline is applied first.
// Image size td_vae_decode writes for an h x w latent: one 2x upsampler behind every block but the last, and the packed latent's channel count.
int td_vae_output_shape(const td_vae* f, int h, int w, int* H, int* W, int* packed_channels) {
  TD_CHECK_ARG(f && h > 0 && w > 0, "td_vae_output_shape: null context or empty latent");
  if (H) *H = h << (f->nb - 1);
  if (W) *W = w << (f->nb - 1);
  if (packed_channels) *packed_channels = 4 * f->cfg.latent_channels;
  return TD_OK;
}

int td_vae_decode(td_vae* f, const void* packed_latents, int h, int w, float scaling_factor, float shift_factor,
                  void* image_u8, void* image_chw, void* stream) {
  TD_CHECK_ARG(f && packed_latents && (image_u8 || image_chw), "td_vae_decode: null argument");
  TD_CHECK_ARG((h * w) % 64 == 0, "td_vae_decode: latent pixel count %d must be a multiple of 64", h * w);
  TD_CHECK_ARG(h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0 && h * w <= f->max_lat_pixels, "td_vae_decode: latent %dx%d exceeds the %d-pixel capacity", h, w, f->max_lat_pixels);
  hipStream_t s = (hipStream_t)stream;
  const int cmid = f->cmid, nb = f->nb;
  int H = h, W = w;
  const int P0 = H * W;

  TDV_TRY(td_latents_to_nhwc_launch((const bf16_t*)packed_latents, f->T1, f->cfg.latent_channels, h, w, f->lat_pad, scaling_factor, shift_factor, s));
  TDV_TRY(conv3(s, f->T1, f->cin_w, f->cin_b, nullptr, f->X, H, W, f->lat_pad, cmid, 0));

  // ---- mid block ------------------------------------------------------------------------------------------
  TDV_TRY(resnet(f, s, f->mid[0], H, W));
  {
    TDV_TRY(gn(f, s, f->X, f->T1, P0, cmid, f->agn_w, f->agn_b, 0));
    TDV_TRY(lin(s, f->T1, cmid, f->aq_w, f->aq_b, f->Q, cmid, P0, cmid, cmid));
    TDV_TRY(lin(s, f->T1, cmid, f->ak_w, f->ak_b, f->K, cmid, P0, cmid, cmid));
    TDV_TRY(lin(s, f->av_w, cmid, f->T1, nullptr, f->VT, P0, cmid, P0, cmid));      // V^T [cmid, P0] = Wv . xn^T (bias added after PV)
    const float scale = 1.0f / sqrtf((float)cmid);
    for (int r0 = 0; r0 < P0; r0 += f->chunk_rows) {
      const int rows = std::min(f->chunk_rows, P0 - r0);
      TdGemmParams g;   // scores (fp32) = Q_chunk . K^T
      g.A = f->Q + (size_t)r0 * cmid; g.lda = cmid; g.W = f->K; g.C = (bf16_t*)f->S; g.ldc = P0; g.M = rows; g.N = P0; g.K = cmid; g.out_f32 = 1;
      g.cfg = P0 <= 64 ? 1 : (rows <= 32 ? 2 : 0);
      TDV_TRY(td_gemm_launch(g, s));
      TDV_TRY(td_softmax_rows_launch(f->S, f->P, rows, P0, scale, s));
      TDV_TRY(lin(s, f->P, P0, f->VT, f->av_b, f->T2 + (size_t)r0 * cmid, cmid, rows, cmid, P0));   // + b_v: softmax rows sum to 1
    }
    TDV_TRY(lin(s, f->T2, cmid, f->ao_w, f->ao_b, f->X, cmid, P0, cmid, cmid, f->X));   // to_out + residual
  }
  TDV_TRY(resnet(f, s, f->mid[1], H, W));

  // ---- up blocks ------------------------------------------------------------------------------------------
  for (int b = 0; b < nb; ++b) {
    for (auto& r : f->up[b]) TDV_TRY(resnet(f, s, r, H, W));
    if (f->ups_w[b]) {   // Upsample2D: nearest 2x + conv3x3, fused; output replaces X via T1
      const int co = f->up[b][0].cout;
      H *= 2; W *= 2;
      TDV_TRY(conv3(s, f->X, f->ups_w[b], f->ups_b[b], nullptr, f->T1, H, W, co, co, 1));
      std::swap(f->X, f->T1);
    }
  }
  const int clast = f->cfg.block_out_channels[0];
  TDV_TRY(gn(f, s, f->X, f->T1, H * W, clast, f->nout_w, f->nout_b, 1));
  TDV_TRY(conv3(s, f->T1, f->cout_w, f->cout_b, nullptr, f->T2, H, W, clast, f->out_pad, 0));
  TDV_TRY(td_image_finalize_launch(f->T2, H * W, f->out_pad, (unsigned char*)image_u8, (bf16_t*)image_chw, s));
  return TD_OK;
}

}  // extern "C"

// extern "C" surface of libthinkdiff_hip.so (declared in include/thinkdiff_hip.h).
#include <cstdarg>
#include <cstdio>
#include "td_kernels.h"
#include "../../include/thinkdiff_hip.h"

static thread_local char g_err[512] = "";

void td_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" {

const char* td_last_error(void) { return g_err; }
int td_abi_version(void) { return 1; }

int td_linear_bf16(const void* x, int64_t ldx, const void* w, const void* bias, void* y, int64_t ldy,
                   int M, int N, int K, int act, const void* gate, const void* res, int64_t ldr,
                   void* stream) {
  TdGemmParams p;
  p.A = (const bf16_t*)x; p.lda = (int)ldx;
  p.W = (const bf16_t*)w; p.bias = (const bf16_t*)bias;
  p.C = (bf16_t*)y; p.ldc = (int)ldy;
  p.gate = (const bf16_t*)gate; p.res = (const bf16_t*)res; p.ldr = (int)ldr;
  p.M = M; p.N = N; p.K = K; p.act = act;
  return td_gemm_launch(p, (hipStream_t)stream);
}

int td_linear_split_bf16(const void* x, int64_t ldx, const void* w, const void* bias,
                         void* y0, int64_t ldy0, int act0, void* y1, int64_t ldy1, int act1,
                         int M, int N, int K, int n_split, void* stream) {
  TdGemmParams p;
  p.A = (const bf16_t*)x; p.lda = (int)ldx;
  p.W = (const bf16_t*)w; p.bias = (const bf16_t*)bias;
  p.C = (bf16_t*)y0; p.ldc = (int)ldy0; p.act = act0;
  p.C2 = (bf16_t*)y1; p.ldc2 = (int)ldy1; p.act2 = act1; p.n_split = n_split;
  p.M = M; p.N = N; p.K = K;
  return td_gemm_launch(p, (hipStream_t)stream);
}

int td_linear_splitk_bf16(const void* x, int64_t ldx, const void* w, const void* bias, void* y0, int64_t ldy0, void* y1, int64_t ldy1, int n_split,
                          int M, int N, int K, const void* res, int64_t ldr, int tile_cfg, int split_k,
                          const void* norm_w, void* norm_out, int64_t ld_norm, float norm_eps, void* stream) {
  TdGemmParams p;
  p.sk_norm_w = (const bf16_t*)norm_w; p.sk_norm_out = (bf16_t*)norm_out; p.sk_norm_ld = (int)ld_norm; p.sk_norm_eps = norm_eps;
  p.A = (const bf16_t*)x; p.lda = (int)ldx; p.W = (const bf16_t*)w; p.bias = (const bf16_t*)bias;
  p.C = (bf16_t*)y0; p.ldc = (int)ldy0; p.C2 = (bf16_t*)y1; p.ldc2 = (int)ldy1; p.n_split = y1 ? n_split : 0;
  p.res = (const bf16_t*)res; p.ldr = (int)ldr; p.M = M; p.N = N; p.K = K; p.cfg = tile_cfg; p.split_k = split_k;
  return td_gemm_launch(p, (hipStream_t)stream);
}

int td_linear_grouped2_bf16(const void* x0, int M0, const void* w0, const void* bias0, const void* gate0,
                            const void* res0, void* y0, const void* x1, int M1, const void* w1,
                            const void* bias1, const void* gate1, const void* res1, void* y1,
                            int64_t ldx, int64_t ldy, int64_t ldr, int N, int K, int act, int tile_cfg,
                            void* stream) {
  TdGemmParams p;
  p.A = (const bf16_t*)x0; p.W = (const bf16_t*)w0; p.bias = (const bf16_t*)bias0; p.gate = (const bf16_t*)gate0;
  p.res = (const bf16_t*)res0; p.C = (bf16_t*)y0; p.M = M0;
  p.g_A = (const bf16_t*)x1; p.g_W = (const bf16_t*)w1; p.g_bias = (const bf16_t*)bias1; p.g_gate = (const bf16_t*)gate1;
  p.g_res = (const bf16_t*)res1; p.g_C = (bf16_t*)y1; p.g_M = M1;
  p.lda = (int)ldx; p.ldc = (int)ldy; p.ldr = (int)ldr; p.N = N; p.K = K; p.act = act; p.cfg = tile_cfg;
  return td_gemm_launch(p, (hipStream_t)stream);
}

int td_conv3x3_nhwc_bf16(const void* x, const void* w, const void* bias, const void* res, void* y,
                         int H, int W, int Cin, int Cout, int upsample2x, void* stream) {
  TdGemmParams p;
  p.A = (const bf16_t*)x; p.lda = Cin; p.W = (const bf16_t*)w; p.bias = (const bf16_t*)bias;
  p.C = (bf16_t*)y; p.ldc = Cout; p.res = (const bf16_t*)res; p.ldr = Cout;
  p.M = H * W; p.N = Cout; p.K = 9 * Cin;
  p.conv_H = H; p.conv_W = W; p.conv_Cin = Cin; p.conv_up = upsample2x ? 1 : 0;
  return td_gemm_launch(p, (hipStream_t)stream);
}

int td_linear_f32out_bf16(const void* x, int64_t ldx, const void* w, const void* bias, float* y, int64_t ldy,
                          int M, int N, int K, void* stream) {
  TdGemmParams p;
  p.A = (const bf16_t*)x; p.lda = (int)ldx; p.W = (const bf16_t*)w; p.bias = (const bf16_t*)bias;
  p.C = (bf16_t*)y; p.ldc = (int)ldy; p.M = M; p.N = N; p.K = K; p.out_f32 = 1; p.cfg = (N <= 64) ? 1 : (M <= 32 ? 2 : 0);
  return td_gemm_launch(p, (hipStream_t)stream);
}

static int g_attn_variant = 0;
int td_attention_set_variant(int variant) {
  const int prev = g_attn_variant;
  g_attn_variant = variant;
  return prev;
}

int td_attention_bf16(const void* q, int64_t ldq, int64_t q_bstride, const void* k, const void* v,
                      int64_t ldkv, int64_t kv_bstride, void* o, int64_t ldo, int64_t o_bstride,
                      int batch, int Sq, int Skv, int Hq, int Hkv, int head_dim, float scale,
                      int causal, void* stream) {
  TdAttnParams p;
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)o;
  p.batch = batch; p.Sq = Sq; p.Skv = Skv; p.Hq = Hq; p.Hkv = Hkv; p.head_dim = head_dim;
  p.ldq = (int)ldq; p.ldkv = (int)ldkv; p.ldo = (int)ldo;
  p.q_bstride = q_bstride; p.kv_bstride = kv_bstride; p.o_bstride = o_bstride;
  p.scale = scale; p.causal = causal; p.causal_offset = Skv - Sq; p.variant = g_attn_variant & ~0x800;
  // test hook (td_attention_set_variant bit 0x800): q already carries scale * log2(e) -- the form the FLUX engine's RoPE kernel hands over
  p.q_prescaled = (g_attn_variant & 0x800) && !causal ? 1 : 0;
  return td_attn_launch(p, (hipStream_t)stream);
}

int td_attention_joint_prescaled_bf16(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int64_t ldo, int S, int H, float score_bound, void* stream) {
  TdAttnParams p;
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)o;
  p.batch = 1; p.Sq = S; p.Skv = S; p.Hq = H; p.Hkv = H; p.head_dim = 128;
  p.ldq = (int)ldq; p.ldkv = (int)ldkv; p.ldo = (int)ldo; p.scale = 1.0f; p.causal = 0; p.causal_offset = 0; p.variant = g_attn_variant & 0xff;
  p.q_prescaled = 1; p.score_bound = score_bound;
  return td_attn_launch(p, (hipStream_t)stream);
}

size_t td_attention_fp8_workspace_bytes(int Sq, int Skv, int Hq) { return Sq > 0 && Skv > 0 && Hq > 0 ? td_attn_fp8_ws_bytes(Sq, Skv, Hq) : 0; }

int td_attention_fp8(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int64_t ldo,
                     int Sq, int Skv, int Hq, float scale, void* workspace, void* stream) {
  TdAttnParams p;
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)o;
  p.batch = 1; p.Sq = Sq; p.Skv = Skv; p.Hq = Hq; p.Hkv = Hq; p.head_dim = 128;
  p.ldq = (int)ldq; p.ldkv = (int)ldkv; p.ldo = (int)ldo; p.scale = scale; p.f8_ws = workspace; p.variant = ((g_attn_variant & 1) ? 0x1000 : 0) | ((g_attn_variant & 2) ? 0x2000 : 0) | (((g_attn_variant >> 4) & 7) << 16);      // td_attention_set_variant bit 0: the 4-wave A/B form, bit 1: exp2 probabilities, bits 4-6: timing-only probes (TD_ATTN8_PROBE builds)
  return td_attn_fp8_launch(p, (hipStream_t)stream);
}

int td_attention_fp8_qk_rope(const void* qkv, int64_t ld, int q_col, int k_col, int v_col, void* o, int64_t ldo, int S, int H,
                             const float* cos, const float* sin, int split, const void* wqA, const void* wkA, const void* wqB, const void* wkB,
                             float eps, float scale, void* workspace, void* stream) {
  TD_CHECK_ARG(qkv && cos && sin && q_col >= 0 && k_col >= 0 && v_col >= 0 && (q_col | k_col | v_col) % 8 == 0, "td_attention_fp8_qk_rope: projection buffer, both tables, 16-byte aligned column offsets");
  TD_CHECK_ARG(S > 0 && H > 0 && (long long)(q_col > k_col ? (q_col > v_col ? q_col : v_col) : (k_col > v_col ? k_col : v_col)) + (long long)H * 128 <= ld,
               "td_attention_fp8_qk_rope: the q / k / v head blocks (H=%d x 128 columns from their offsets) do not fit a row of ld=%lld", H, (long long)ld);
  TdAttnParams p;
  p.Q = (const bf16_t*)qkv + q_col; p.K = (const bf16_t*)qkv + k_col; p.V = (const bf16_t*)qkv + v_col; p.O = (bf16_t*)o;
  p.batch = 1; p.Sq = S; p.Skv = S; p.Hq = H; p.Hkv = H; p.head_dim = 128;
  p.ldq = (int)ld; p.ldkv = (int)ld; p.ldo = (int)ldo; p.scale = scale; p.f8_ws = workspace; p.variant = ((g_attn_variant & 1) ? 0x1000 : 0) | ((g_attn_variant & 2) ? 0x2000 : 0);
  p.rope_cos = cos; p.rope_sin = sin; p.rope_split = split; p.rope_eps = eps;
  p.rope_wqA = (const bf16_t*)wqA; p.rope_wkA = (const bf16_t*)wkA; p.rope_wqB = (const bf16_t*)wqB; p.rope_wkB = (const bf16_t*)wkB;
  // q is rounded to bf16 as td_qk_norm_rope_bf16 leaves it and scaled in the pack pass, as td_attention_fp8 does (the FLUX engine folds the
  // scale in front of that rounding instead: TdQkRopeParams::q_premul on both of its paths)
  return td_attn_fp8_launch(p, (hipStream_t)stream);
}

int td_attention_varlen_bf16(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int64_t ldo,
                             const int* seg_starts, int n_seg, int max_len, int Hq, int Hkv, float scale, void* stream) {
  TD_CHECK_ARG(seg_starts && n_seg > 0 && max_len > 0, "td_attention_varlen: empty segment list");
  TdAttnParams p;
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)o;
  p.batch = n_seg; p.Sq = max_len; p.Skv = max_len; p.Hq = Hq; p.Hkv = Hkv; p.head_dim = 128;
  p.ldq = (int)ldq; p.ldkv = (int)ldkv; p.ldo = (int)ldo;
  p.scale = scale; p.causal = 0; p.causal_offset = 0; p.variant = 1; p.seg_starts = seg_starts;
  return td_attn_launch(p, (hipStream_t)stream);
}

int td_norm_rows_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, int rms, float eps,
                      const void* w, int split, const void* shiftA, const void* scaleA,
                      const void* shiftB, const void* scaleB, void* stream) {
  TdNormParams p;
  p.x = (const bf16_t*)x; p.ldx = (int)ldx; p.y = (bf16_t*)y; p.ldy = (int)ldy; p.rows = rows; p.D = D;
  p.rms = rms; p.eps = eps; p.w = (const bf16_t*)w; p.split = split;
  p.shiftA = (const bf16_t*)shiftA; p.scaleA = (const bf16_t*)scaleA;
  p.shiftB = (const bf16_t*)shiftB; p.scaleB = (const bf16_t*)scaleB;
  if (p.scaleA && !p.scaleB) { p.scaleB = p.scaleA; p.shiftB = p.shiftA; }
  return td_norm_rows_launch(p, (hipStream_t)stream);
}

int td_qk_norm_rope_bf16(void* qkv, int64_t ld, int rows, int Hq, int Hk, int q_col, int k_col,
                         const float* cos, const float* sin, int split, const void* wqA, const void* wkA,
                         const void* wqB, const void* wkB, float eps, int rotate_half, void* stream) {
  TdQkRopeParams p;
  p.qkv = (bf16_t*)qkv; p.ld = (int)ld; p.rows = rows; p.Hq = Hq; p.Hk = Hk; p.q_col = q_col; p.k_col = k_col;
  p.cos = cos; p.sin = sin; p.split = split;
  p.wqA = (const bf16_t*)wqA; p.wkA = (const bf16_t*)wkA; p.wqB = (const bf16_t*)wqB; p.wkB = (const bf16_t*)wkB;
  p.eps = eps; p.rotate_half = rotate_half;
  return td_qk_norm_rope_launch(p, (hipStream_t)stream);
}

int td_flux_rope_table(const float* ids, int S, const int* axes_dims3, double theta, float* cos, float* sin, void* stream) {
  return td_flux_rope_table_launch(ids, S, axes_dims3, theta, cos, sin, (hipStream_t)stream);
}
int td_timestep_sincos(const float* t, int n, void* out, void* stream) {
  return td_timestep_sincos_launch(t, n, (bf16_t*)out, (hipStream_t)stream);
}
int td_euler_step_bf16(void* x, const void* v, float dt, int64_t n, void* stream) {
  return td_euler_step_launch((bf16_t*)x, (const bf16_t*)v, dt, n, (hipStream_t)stream);
}
int td_flux_pack_latents(const void* src, void* dst, int C, int H, int W, int unpack, float div, float add, void* stream) {
  return td_flux_pack_launch((const bf16_t*)src, (bf16_t*)dst, C, H, W, unpack, div, add, (hipStream_t)stream);
}
int td_cls_avgpool2_bf16(const void* x, void* y, int G, int C, void* stream) {
  return td_cls_avgpool2_launch((const bf16_t*)x, (bf16_t*)y, G, C, (hipStream_t)stream);
}

int td_aligner_mlp2x_bf16(const void* x, int64_t ldx, int M, int K, int hidden, const void* w0, const void* b0,
                          const void* w2, const void* b2, const void* norm_w, float eps, int fp32_norm,
                          void* workspace, void* y, int64_t ldy, void* stream) {
  TD_CHECK_ARG(x && w0 && w2 && norm_w && workspace && y, "td_aligner_mlp2x: null argument");
  bf16_t* t0 = (bf16_t*)workspace;
  bf16_t* t1 = t0 + (size_t)M * hidden;
  TdGemmParams g;
  g.A = (const bf16_t*)x; g.lda = (int)ldx; g.W = (const bf16_t*)w0; g.bias = (const bf16_t*)b0;
  g.C = t0; g.ldc = hidden; g.M = M; g.N = hidden; g.K = K; g.act = TD_ACT_GELU_ERF;
  int rc = td_gemm_launch(g, (hipStream_t)stream);
  if (rc) return rc;
  TdGemmParams g2;
  g2.A = t0; g2.lda = hidden; g2.W = (const bf16_t*)w2; g2.bias = (const bf16_t*)b2;
  g2.C = t1; g2.ldc = hidden; g2.M = M; g2.N = hidden; g2.K = hidden;
  rc = td_gemm_launch(g2, (hipStream_t)stream);
  if (rc) return rc;
  TdNormParams n;
  n.x = t1; n.ldx = hidden; n.y = (bf16_t*)y; n.ldy = (int)ldy; n.rows = M; n.D = hidden;
  n.rms = fp32_norm ? 2 : 1; n.eps = eps; n.w = (const bf16_t*)norm_w;
  return td_norm_rows_launch(n, (hipStream_t)stream);
}

int td_embed_gather_bf16(const int* ids, const void* table, void* out, int n, int D, int vocab, void* stream) {
  return td_embed_gather_launch(ids, (const bf16_t*)table, (bf16_t*)out, n, D, vocab, (hipStream_t)stream);
}
int td_silu_mul_bf16(const void* gate_up, void* out, int rows, int I, void* stream) {
  return td_silu_mul_launch((const bf16_t*)gate_up, (bf16_t*)out, rows, I, (hipStream_t)stream);
}
int td_mrope_table(const int* pos3n, int n, const int* sections3, float theta, int round_bf16, float* cos, float* sin, void* stream) {
  return td_mrope_table_launch(pos3n, n, sections3, theta, round_bf16, cos, sin, (hipStream_t)stream);
}

int td_conv3x3_pack_weight(const void* w_oihw, void* w_packed, int Cout, int Cin, int Cout_pad, int Cin_pad, void* stream) {
  return td_conv_pack_launch((const bf16_t*)w_oihw, (bf16_t*)w_packed, Cout, Cin, Cout_pad, Cin_pad, (hipStream_t)stream);
}
int td_groupnorm_nhwc_bf16(const void* x, void* y, int P, int C, int groups, float eps, const void* gamma,
                           const void* beta, int silu, float* workspace, void* stream) {
  return td_groupnorm_nhwc_launch((const bf16_t*)x, (bf16_t*)y, P, C, groups, eps, (const bf16_t*)gamma, (const bf16_t*)beta, silu, workspace, (hipStream_t)stream);
}
int td_groupnorm_workspace_floats(void) { return 1024 * 64 * 2 + 256; }
int td_softmax_rows_f32_bf16(const float* s, void* p, int rows, int cols, float scale, void* stream) {
  return td_softmax_rows_launch(s, (bf16_t*)p, rows, cols, scale, (hipStream_t)stream);
}

int td_layernorm_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, int rms, float eps,
                      const void* w, const void* b, void* stream) {
  return td_norm_rows_generic_launch((const bf16_t*)x, (int)ldx, (bf16_t*)y, (int)ldy, rows, D, rms, eps, (const bf16_t*)w, (const bf16_t*)b, (hipStream_t)stream);
}
int td_add_rows_bf16(const void* a, const void* b, void* out, int rows, int D, int b_rows, void* stream) {
  return td_add_rows_launch((const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, rows, D, b_rows, (hipStream_t)stream);
}
int td_glu_mul_bf16(const void* gate_up, void* out, int rows, int I, int act, void* stream) {
  return td_glu_mul_launch((const bf16_t*)gate_up, (bf16_t*)out, rows, I, act, (hipStream_t)stream);
}
int td_attention_bias_bf16(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int64_t ldo,
                           int Sq, int Skv, int Hq, int Hkv, float scale, int causal, const float* bias, void* stream) {
  TdAttnParams p;
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)o;
  p.batch = 1; p.Sq = Sq; p.Skv = Skv; p.Hq = Hq; p.Hkv = Hkv; p.head_dim = 128;
  p.ldq = (int)ldq; p.ldkv = (int)ldkv; p.ldo = (int)ldo;
  p.scale = scale; p.causal = causal; p.causal_offset = Skv - Sq; p.bias = bias;
  return td_attn_launch(p, (hipStream_t)stream);
}

int td_rope_half_bf16(void* x, int64_t ldx, int S, int H, int head_stride, int hd, const float* cos_t, const float* sin_t, void* stream) {
  return td_rope_half_launch((bf16_t*)x, (int)ldx, S, H, head_stride, hd, cos_t, sin_t, (hipStream_t)stream);
}
int td_vision_rope_table(const int* pos, int S, int hd, float theta, float* cos_t, float* sin_t, void* stream) {
  return td_vision_rope_table_launch(pos, S, hd, theta, cos_t, sin_t, (hipStream_t)stream);
}
int td_qwen2_patchify_u8(const void* img_hwc, int H, int W, const float* lut, int patch, int merge, int temporal, void* out, int Kpad, void* stream) {
  return td_qwen2_patchify_u8_launch((const unsigned char*)img_hwc, H, W, lut, patch, merge, temporal, (bf16_t*)out, Kpad, (hipStream_t)stream);
}

int td_patchify_bf16(const void* pix, int src_f32, int C, int H, int W, int p, void* out, int Kpad, void* stream) {
  return td_patchify_launch(pix, src_f32, C, H, W, p, (bf16_t*)out, Kpad, (hipStream_t)stream);
}
int td_cast_pad_rows_bf16(const void* src, int src_f32, int rows, int K, void* out, int Kpad, void* stream) {
  return td_cast_pad_rows_launch(src, src_f32, rows, K, (bf16_t*)out, Kpad, (hipStream_t)stream);
}

int td_quant_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int K, void* stream) {
  return td_quant_rows_fp8_launch((const bf16_t*)x, (int)ldx, (uint8_t*)q, (int)ldq, scale, rows, K, (hipStream_t)stream);
}
int td_linear_fp8(const void* xq, int64_t ldx, const float* x_scale, const void* wq, const float* w_scale, const void* bias,
                  void* y, int64_t ldy, int M, int N, int K, int act, const void* gate, const void* res, int64_t ldr,
                  int tile_cfg, void* stream) {
  TdGemmParams p;
  p.fp8 = 1; p.A = (const bf16_t*)xq; p.lda = (int)ldx; p.a_scale = x_scale;
  p.W = (const bf16_t*)wq; p.w_scale = w_scale; p.bias = (const bf16_t*)bias;
  p.C = (bf16_t*)y; p.ldc = (int)ldy;
  p.gate = (const bf16_t*)gate; p.res = (const bf16_t*)res; p.ldr = (int)ldr;
  p.M = M; p.N = N; p.K = K; p.act = act; p.cfg = tile_cfg;
  return td_gemm_launch(p, (hipStream_t)stream);
}
int td_quant_rows_int8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int K, void* stream) {
  return td_quant_rows_fp8_launch((const bf16_t*)x, (int)ldx, (uint8_t*)q, (int)ldq, scale, rows, K, (hipStream_t)stream, 1);
}
int td_linear_int8(const void* xq, int64_t ldx, const float* x_scale, const void* wq, const float* w_scale, const void* bias,
                   void* y, int64_t ldy, int M, int N, int K, int act, const void* gate, const void* res, int64_t ldr,
                   int tile_cfg, void* stream) {
  TdGemmParams p;
  p.i8 = 1; p.A = (const bf16_t*)xq; p.lda = (int)ldx; p.a_scale = x_scale;
  p.W = (const bf16_t*)wq; p.w_scale = w_scale; p.bias = (const bf16_t*)bias;
  p.C = (bf16_t*)y; p.ldc = (int)ldy;
  p.gate = (const bf16_t*)gate; p.res = (const bf16_t*)res; p.ldr = (int)ldr;
  p.M = M; p.N = N; p.K = K; p.act = act; p.cfg = tile_cfg;
  return td_gemm_launch(p, (hipStream_t)stream);
}
int td_norm_rows_quant_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* q_scale, int rows, int D, int rms, float eps,
                           const void* w, int split, const void* shiftA, const void* scaleA, const void* shiftB, const void* scaleB,
                           void* stream) {
  TdNormParams p;
  p.x = (const bf16_t*)x; p.ldx = (int)ldx; p.q = (uint8_t*)q; p.ldq = (int)ldq; p.q_scale = q_scale; p.rows = rows; p.D = D;
  p.rms = rms; p.eps = eps; p.w = (const bf16_t*)w; p.split = split;
  p.shiftA = (const bf16_t*)shiftA; p.scaleA = (const bf16_t*)scaleA;
  p.shiftB = (const bf16_t*)shiftB; p.scaleB = (const bf16_t*)scaleB;
  if (p.scaleA && !p.scaleB) { p.scaleB = p.scaleA; p.shiftB = p.shiftA; }
  TD_CHECK_ARG(q && q_scale, "td_norm_rows_quant_fp8: null output");
  return td_norm_rows_launch(p, (hipStream_t)stream);
}

int td_sample_top_p_bf16(const void* logits, int64_t ld, int rows, int vocab, float temperature, float top_p,
                         uint64_t seed, uint64_t offset, int32_t* out_ids, void* stream) {
  TD_CHECK_ARG(logits && out_ids, "td_sample_top_p_bf16: null pointer");
  return td_sample_top_p_launch((const bf16_t*)logits, (long long)ld, rows, vocab, temperature, top_p, seed, offset, out_ids, (hipStream_t)stream);
}

}  // extern "C"

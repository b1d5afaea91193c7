// Internal launch interface between the kernel translation units and the C-ABI / engine.
// Not installed; the public surface is include/thinkdiff_hip.h.
#pragma once
#include "td_common.h"

enum TdAct { TD_ACT_NONE = 0, TD_ACT_GELU_TANH = 1, TD_ACT_GELU_ERF = 2, TD_ACT_SILU = 3, TD_ACT_QUICK_GELU = 4 };

struct TdGemmParams {
  const bf16_t* A = nullptr;     // [M, lda]  activations, K-contiguous
  const bf16_t* W = nullptr;     // [N, K]    torch Linear weight layout
  const bf16_t* bias = nullptr;  // [N] or null
  bf16_t* C = nullptr;           // [M, ldc]
  const bf16_t* gate = nullptr;  // [N] or null : y *= gate[n]
  const bf16_t* res = nullptr;   // [M, ldr] or null : y += res[m,n]   (may alias C)
  bf16_t* C2 = nullptr;          // optional second output for columns >= n_split
  int M = 0, N = 0, K = 0;
  int lda = 0, ldc = 0, ldr = 0, ldc2 = 0;
  int n_split = 0;
  int act = TD_ACT_NONE, act2 = TD_ACT_NONE;
  // optional second problem of a grouped launch (same N, K, lda, ldc, ldr, act; own rows and weights):
  // the double-stream blocks run their text-token GEMM (M = 193) inside the image-token launch
  const bf16_t* g_A = nullptr; const bf16_t* g_W = nullptr; const bf16_t* g_bias = nullptr;
  const bf16_t* g_gate = nullptr; const bf16_t* g_res = nullptr; bf16_t* g_C = nullptr;
  int g_M = 0;
  int cfg = -1;                  // tile config override (-1: auto), see td_gemm_config_id
  // K split over workgroups (tile kernels, bf16 operands, plain / bias / residual / split-output epilogues): few output tiles and a long K -- the
  // KV-cached decode of 65-256 sequences, M <= 256 against N = hidden -- leave most CUs without a workgroup.  split_k = -1: the launcher decides
  // (parts so that tiles x parts covers the CUs, every part a whole number of k-tiles); > 1: that many parts; 0 / 1: none.  Part j contracts
  // K-range j of every tile into fp32 partial sums in sk_ws ([parts][M][N] floats; null: a pooled buffer of the (device, stream)), and a second launch
  // adds the parts in index order -- the result does not depend on the order the workgroups ran in -- and applies the epilogue with the tile
  // kernel's rounding points.  Forms the reduction does not cover (activation, gate, int8 / fp8, conv, grouped) are launched unsplit under -1.
  int split_k = 0;
  float* sk_ws = nullptr; long long sk_ws_bytes = 0;
  // split_k = -1 only, N % 512 == 0, N <= 4096, no second output: the reduction launch also RMS-normalises the finished row (Qwen2RMSNorm: the
  // arithmetic, summation order and rounding points of td_norm_rows_kernel's rms form on the bf16 row it has just written) into sk_norm_out
  // [M, sk_norm_ld] -- the norm launch that would follow the Linear (o_proj -> post_attention_layernorm, down_proj -> the next input_layernorm)
  // disappears.  The launch then always goes through the partial-sum buffer, with one part when a split does not pay.
  const bf16_t* sk_norm_w = nullptr; bf16_t* sk_norm_out = nullptr; int sk_norm_ld = 0; float sk_norm_eps = 1e-6f;
  int ldw = 0, k_parts = 1;      // filled by the launcher: row stride of W in elements (= the whole K), parts of a split launch (K = ONE part's extent)
  int out_f32 = 0;               // C is float* (ldc in floats): acc + bias stored unrounded, no act/gate/res
  // implicit-GEMM 3x3 convolution over an NHWC image (conv_H > 0): A = input [Hin*Win, Cin], W = [N, 9*Cin]
  // (k = tap*Cin + c, tap = ky*3+kx), M = conv_H*conv_W output pixels; conv_up = 1 fuses a nearest 2x upsample
  int conv_H = 0, conv_W = 0, conv_Cin = 0, conv_up = 0;
  // fp8 operands (fp8 = 1): A and W hold OCP e4m3 bytes (lda, K in elements = bytes); y = acc * a_scale[m] * w_scale[n]
  // gated MLP in one pass (skinny-M kernels only, glu_I > 0): W = [gate rows | up rows] (2 * glu_I x K), N = glu_I outputs,
  // C[m, n] = bf16(bf16(silu(bf16(x.gate_n))) * bf16(x.up_n)) -- Linear, SiLU and the product each round, as the separate kernels do
  int glu_I = 0;
  int fp8 = 0;
  // int8 operands (i8 = 1): A and W hold symmetric int8 (lda, K in elements = bytes), v_mfma_i32_16x16x64_i8, exact int32 accumulation;
  // y = float(acc) * a_scale[m] * w_scale[n] -- the same scale arrays and the same 2x-bf16 MFMA rate as the fp8 form
  int i8 = 0;
  // int8 OUTPUT of the activated result (int8 kernels, 256-wide tiles): q8[m, :] = clamp(rint(y * q8_inv[m]), +-127) with the caller's
  // per-row inverse scales, the maxima of |y| per row accumulated into q8_amax[m] (atomic max on float bits); with a split output
  // (C2) it applies to the second one, the first stays bf16.  g_* = the second problem of a grouped launch.  See the epilogue.
  uint8_t* q8 = nullptr; int ldq8 = 0; const float* q8_inv = nullptr; unsigned* q8_amax = nullptr;
  uint8_t* g_q8 = nullptr; const float* g_q8_inv = nullptr; unsigned* g_q8_amax = nullptr;
  // int8 OUTPUT only: per-column factors (bf16, indexed by the ABSOLUTE output column n of this launch; null = none) applied to the activated value
  // before its row maximum and its quantisation -- 1 / s of the int8 smoothing of the Linear that consumes this output
  const bf16_t* q8_smooth = nullptr; const bf16_t* g_q8_smooth = nullptr;
  const float* a_scale = nullptr; const float* w_scale = nullptr;        // [M], [N]
  const float* g_a_scale = nullptr; const float* g_w_scale = nullptr;    // second problem of a grouped launch
  int tiles_m = 0, tiles_m0 = 0, tiles_n = 0;  // filled by the launcher
  int ragged_rows = 64;                        // filled by the launcher: tiles with at most this many rows take the ragged loop (0 with TD_GEMM_NO_RAGGED: A/B)
  int tail_first_wg = 0;                       // filled by the launcher (tail-split launches): workgroups from here on take a sub-tile of the last tiles
  int no_tail = 0;                             // caller's hint: other kernels share the chip (several images in flight) -- an empty last round gets filled anyway, do not split it
  int probe = 0;                               // filled by the launcher from TD_GEMM_PROBE (timing experiments, results WRONG): 1 = no epilogue, 2 = no k-loop
};

int td_gemm_launch(const TdGemmParams& p, hipStream_t stream);
// weight-streaming form for M <= 8 (td_gemm_launch routes to it; csrc/gemv_bf16.hip)
int td_gemv_launch(const TdGemmParams& p, hipStream_t stream);
bool td_gemv_mfma_ok(const TdGemmParams& p);   // shapes the matrix-core weight stream takes for 16 < M <= 64
// 0: 256x256 (td_gemm_bf16_nt_kernel<8,4>), 1: 256x64, 2: 32x256, 3: 288x192 (<9,3>)
int td_gemm_config_id(int M, int N, int K);

enum { TD_TRACE_GEMM_MAIN = 0, TD_TRACE_GEMM_OTHER = 1, TD_TRACE_ATTN = 2, TD_TRACE_NORM = 3, TD_TRACE_QKROPE = 4, TD_TRACE_GEMM_288 = 5, TD_TRACE_NCAT = 6 };

// q/k/v are read in place from projection outputs: token row s, head h at column h*128.
struct TdAttnParams {
  const bf16_t* Q = nullptr;  // [batch][Sq, ldq]
  const bf16_t* K = nullptr;  // [batch][Skv, ldkv]
  const bf16_t* V = nullptr;  // [batch][Skv, ldkv]
  bf16_t* O = nullptr;        // [batch][Sq, ldo]
  int batch = 1, Sq = 0, Skv = 0, Hq = 0, Hkv = 0, head_dim = 128;
  int ldq = 0, ldkv = 0, ldo = 0;
  long long q_bstride = 0, kv_bstride = 0, o_bstride = 0;  // elements
  float scale = 0.f;
  int causal = 0, causal_offset = 0;  // key visible iff key <= q + causal_offset
  int q_per_kv = 1;                   // filled by the launcher
  int variant = 0;                    // 0: shipped (lean stream), 1: first lockstep kernel (A/B only)
  // optional additive score bias (T5 relative position bias): fp32 [Hq, Sq, Skv]; scores = q.k*scale + bias
  const float* bias = nullptr;
  // optional per-batch cache length (device int[batch], causal kernel): sequence b attends keys [0, kv_lens[b]); Skv = the largest
  const int* kv_lens = nullptr;
  // decode kernel only (Sq = 1, kv_lens given): the step's rotary embedding and cache write happen INSIDE the attention launch.  Q then holds the RAW
  // q rows of the projection, dec_kv_new [batch, 2 Hkv 128] the raw new k | v rows, dec_cos / dec_sin fp32 [batch, 128] the tokens' M-RoPE table rows,
  // dec_row_off int[batch] the ROW index of each sequence's new cache row (x ldkv elements from K).  Every workgroup rotates its q heads and the new key
  // itself (rotate_half, every product rounded to bf16: td_decode_rope_scatter_kernel's arithmetic), uses the new key / value as the last of its
  // kv_lens[b] keys from registers, and the first workgroup of each kv head writes them to the cache -- nobody reads that row in this launch.
  const bf16_t* dec_kv_new = nullptr; const float* dec_cos = nullptr; const float* dec_sin = nullptr; const int* dec_row_off = nullptr;
  // decode kernel only: sequence b's cache is slot dec_slots[b] (K / V + dec_slots[b] kv_bstride) instead of slot b -- the sequences of a step need not
  // sit in the first slots, so a finished sequence frees its slot without anybody's cache rows being moved (continuous batching); q / o stay row b
  const int* dec_slots = nullptr;
  // optional packed segments (device int[batch + 1], non-causal): segment b = rows [seg_starts[b], seg_starts[b+1]) of q/k/v/o; Sq = Skv = the longest
  const int* seg_starts = nullptr;
  // optional hand-off workspace of the persistent (stream-K) joint-attention kernel: td_attn_streamk_ws_bytes() bytes, zeroed
  // once by its owner, not shared by launches that may run concurrently (null: a per-(device, stream) one is created inside)
  void* sk_ws = nullptr;
  // q already carries scale * log2(e) (folded in where q was last rounded to bf16: TdQkRopeParams::q_premul), so scores arrive in
  // the exp2 domain and `scale` is not applied again; joint (non-causal, no bias / lengths / segments) attention only
  int q_prescaled = 0;
  // q_prescaled only: an upper bound of every score q'.k the caller can vouch for (FLUX: the QK-RMSNorm bounds |q'| and |k| by sqrt(128) x the
  // largest norm weight, so |q'.k| <= premul x 128 x max|w_q| x max|w_k| -- a per-block constant).  In (0, 48]: the kernels exponentiate the scores
  // as they are -- no reference point, no per-tile row maximum, no rescale (bf16 probabilities and fp32 sums do not care about magnitude:
  // exp2(+-48) and a sum of 2^13 of them are far inside the range); a score a few octaves above it is harmless.  0: the running-maximum form.
  float score_bound = 0.f;
  // int8 output instead of O (joint attention; the FLUX engine's history-scaled int8 mode): q8[b][row, head*128 + d] = clamp(rint(o * q8_inv[row]), +-127)
  // with the caller's per-row inverse scales, row maxima of |o| accumulated into q8_amax[row] (atomic max on float bits).  ldq8 in bytes.
  uint8_t* q8 = nullptr; int ldq8 = 0; const float* q8_inv = nullptr; unsigned* q8_amax = nullptr;
  // td_attn_fp8_launch only: td_attn_fp8_ws_bytes(Sq, Skv, Hq) bytes of scratch for the packed e4m3 operands
  void* f8_ws = nullptr;
  // td_attn_fp8_launch only: reference points carried from one denoise step to the next (ints, [Hq][Sq]).  ref_in (may be null): where each row's
  // softmax starts -- ceil(the row's largest score of the PREVIOUS step) - headroom, in log2 units -- instead of the first tile's maximum, so that
  // the reference hardly ever has to move (a move rescales O, the row sums and the scores: ~80 VALU instructions per wave); ref_out (may be null):
  // receives the same quantity of THIS launch (atomic max: both owners of a split item contribute); the launch's pack pass presets it.
  const int* ref_in = nullptr; int* ref_out = nullptr;
  // td_attn_fp8_launch only (rope_cos != null): Q / K are the RAW projection outputs and the pack pass applies the per-head
  // QK-RMSNorm + interleaved-pair rotary embedding of td_qk_norm_rope_kernel on its way (same arithmetic, same summation order and
  // the same bf16 rounding points: bit-identical to the two-pass form), so that pass and its HBM round trip disappear.  Needs
  // Sq == Skv (q and k rows are the same tokens); rows < rope_split take the A norm weights, the rest B (null: no norm).
  const float* rope_cos = nullptr; const float* rope_sin = nullptr;   // [Sq,128] fp32
  const bf16_t* rope_wqA = nullptr; const bf16_t* rope_wkA = nullptr; const bf16_t* rope_wqB = nullptr; const bf16_t* rope_wkB = nullptr;
  int rope_split = 0; float rope_eps = 1e-6f;
  float rope_q_premul = 1.0f;   // as TdQkRopeParams::q_premul (then q_prescaled = 1)
};
size_t td_attn_streamk_ws_bytes();
int td_attn_device_cus(int dev);
int td_attn_pooled_workspace(int dev, int ranges, hipStream_t stream, char** out);
// csrc/attention_fp8.hip: the joint attention with QK^T and P.V on the e4m3 MFMA (bf16 operands in, packed on the way)
size_t td_attn_fp8_ws_bytes(int Sq, int Skv, int H);
int td_attn_fp8_launch(const TdAttnParams& p, hipStream_t stream);

int td_attn_launch(const TdAttnParams& p, hipStream_t stream);
// Sq = 1 (KV-cached decode) form, csrc/attention_decode.hip; td_attn_launch routes to it
int td_attn_decode_launch(const TdAttnParams& p, hipStream_t stream);

struct TdNormParams {
  const bf16_t* x = nullptr; int ldx = 0;
  bf16_t* y = nullptr; int ldy = 0;
  int rows = 0, D = 0;
  int rms = 0;        // 0: LayerNorm statistics (mean/var), 1: RMSNorm
  float eps = 1e-6f;
  const bf16_t* w = nullptr;  // optional affine weight [D]
  int split = 0;              // rows < split use set A, others set B
  const bf16_t* shiftA = nullptr; const bf16_t* scaleA = nullptr;
  const bf16_t* shiftB = nullptr; const bf16_t* scaleB = nullptr;
  // fp8 output (q != null): the row is written as OCP e4m3 q[row, :] = fp8(y / s), s = max|y| / 448 -> q_scale[row]; y unused
  uint8_t* q = nullptr; int ldq = 0; float* q_scale = nullptr;
  int q_int8 = 0;   // q holds symmetric int8 instead: q = rint(y / s), s = max|y| / 127
  // quantised output only: per-channel factors (bf16 [D], 1 / s of the int8 smoothing, powers of two) applied before the row maximum is taken;
  // A for rows < split, B for the others; null = none
  const bf16_t* smoothA = nullptr; const bf16_t* smoothB = nullptr;
  // ... and replicated channels (int8 only): q rows carry ext_n further bytes behind the D quantised ones, byte D + e = the quantised byte of
  // channel ext[e] (or 0 where ext[e] < 0).  A channel that runs r x above the rest is divided by r (smooth) and present r times -- the
  // contraction sums r x (x / r) w -- so its weight column keeps its magnitude and the row's step is set by the rest (outlier channel splitting).
  const int* extA = nullptr; const int* extB = nullptr; int ext_n = 0;
};
// weights of such a Linear: q[r, K + e] = q[r, ext[e]] (0 where ext[e] < 0) for every row of an int8 matrix with row stride ld >= K + ext_n
int td_ext_cols_launch(uint8_t* q, int ld, int rows, int K, const int* ext, int ext_n, hipStream_t stream);
// per-row dynamic fp8 quantisation of a bf16 matrix: q[r,:] = e4m3(x[r,:] / s_r), s_r = max|x[r,:]| / 448 (1 for a zero row)
// int8 = 1: symmetric int8 instead (q = rint(x / s_r), s_r = max|x[r,:]| / 127)
// col_mul (fp32 [K], may be null): per-column factors applied first (int8 smoothing: s on a weight's input channels, 1 / s on activations)
int td_quant_rows_fp8_launch(const bf16_t* x, int ldx, uint8_t* q, int ldq, float* scale, int rows, int K, hipStream_t stream, int int8 = 0, unsigned* amax_out = nullptr,
                             const float* col_mul = nullptr);
// int8 smoothing (csrc/elementwise.hip, bottom): column maxima of a bf16 matrix (atomic max into float bits), and the factors made from two of them
int td_col_amax_launch(const bf16_t* x, int ldx, int rows, int K, unsigned* amax, hipStream_t stream);
int td_smooth_factors_launch(const unsigned* ax, const unsigned* aw, int n, float* s, float* inv, bf16_t* inv16, hipStream_t stream);
// history-scaled int8 (csrc/elementwise.hip, bottom): turn the accumulated maxima into the next step's scales
int td_q8_scales_from_amax_launch(unsigned* amax, float* scale, float* inv, long long n, float margin, hipStream_t stream);
int td_norm_rows_launch(const TdNormParams& p, hipStream_t stream);

struct TdQkRopeParams {
  bf16_t* qkv = nullptr; int ld = 0;   // fused projection rows; modified in place
  int rows = 0, Hq = 0, Hk = 0;
  int q_col = 0, k_col = 0;            // first column of the q / k head blocks
  const float* cos = nullptr; const float* sin = nullptr;  // [rows,128] fp32
  int split = 0;
  const bf16_t* wqA = nullptr; const bf16_t* wkA = nullptr;  // rows < split (null: no norm)
  const bf16_t* wqB = nullptr; const bf16_t* wkB = nullptr;
  float eps = 1e-6f;
  int rotate_half = 0;  // 0: interleaved pairs (FLUX), 1: half-split fp32, 2: half-split with bf16 op rounding (Qwen2)
  // q (not k) is multiplied by this in fp32 before its one rounding to bf16: the attention scale * log2(e), so that the attention
  // kernel needs no per-score multiply (TdAttnParams::q_prescaled).  1 = the plain reference values.
  float q_premul = 1.0f;
};
int td_qk_norm_rope_launch(const TdQkRopeParams& p, hipStream_t stream);

int td_flux_rope_table_launch(const float* ids, int S, const int* axes, double theta, float* cosT, float* sinT, hipStream_t stream);
int td_timestep_sincos_launch(const float* t, int n, bf16_t* out, hipStream_t stream);
int td_temb_combine_silu_launch(const bf16_t* te, const bf16_t* ge, const bf16_t* pe, int n, int D, bf16_t* temb, bf16_t* silu_out, hipStream_t stream);
int td_euler_step_launch(bf16_t* x, const bf16_t* v, float dt, long long n, hipStream_t stream);
int td_flux_pack_launch(const bf16_t* src, bf16_t* dst, int C, int H, int W, int unpack, float div, float add, hipStream_t stream);
int td_cls_avgpool2_launch(const bf16_t* x, bf16_t* y, int G, int C, hipStream_t stream);

int td_embed_gather_launch(const int* ids, const bf16_t* table, bf16_t* out, int n, int D, int vocab, hipStream_t stream);
int td_silu_mul_launch(const bf16_t* gu, bf16_t* out, int rows, int I, hipStream_t stream);
int td_mrope_table_launch(const int* pos, int n, const int* sections, float theta, int round_bf16, float* cosT, float* sinT, hipStream_t stream);

// VAE decoder kernels (vae_kernels.hip); groupnorm workspace: 1024*64*2 + 256 floats
int td_groupnorm_nhwc_launch(const bf16_t* x, bf16_t* y, int P, int C, int G, float eps, const bf16_t* gamma, const bf16_t* beta,
                             int silu, float* workspace, hipStream_t stream);
int td_softmax_rows_launch(const float* s, bf16_t* p, int rows, int cols, float scale, hipStream_t stream);
int td_conv_pack_launch(const bf16_t* w, bf16_t* out, int Cout, int Cin, int Cout_pad, int Cin_pad, hipStream_t stream);
int td_latents_to_nhwc_launch(const bf16_t* packed, bf16_t* out, int C, int h, int w, int Cpad, float div, float add, hipStream_t stream);
int td_image_finalize_launch(const bf16_t* x, int P, int Cpad, unsigned char* u8, bf16_t* chw, hipStream_t stream);
extern "C" int td_fill_normal_bf16(void* dst, int64_t n, uint64_t seed, float std, float mean, void* stream);

int td_norm_rows_generic_launch(const bf16_t* x, int ldx, bf16_t* y, int ldy, int rows, int D, int rms, float eps,
                                const bf16_t* w, const bf16_t* b, hipStream_t stream);
int td_add_rows_launch(const bf16_t* a, const bf16_t* b, bf16_t* out, int rows, int D, int b_rows, hipStream_t stream);
int td_glu_mul_launch(const bf16_t* gu, bf16_t* out, int rows, int I, int act, hipStream_t stream);
int td_rope_half_launch(bf16_t* x, int ldx, int S, int H, int head_stride, int hd, const float* cs, const float* sn, hipStream_t stream);
int td_vision_rope_table_launch(const int* pos, int S, int hd, float theta, float* cs, float* sn, hipStream_t stream);
int td_patchify_launch(const void* pix, int src_f32, int C, int H, int W, int p, bf16_t* out, int Kpad, hipStream_t stream);
int td_qwen2_patchify_u8_launch(const unsigned char* img, int H, int W, const float* lut, int p, int m, int T, bf16_t* out, int Kpad, hipStream_t stream);
int td_cast_pad_rows_launch(const void* src, int src_f32, int rows, int K, bf16_t* out, int Kpad, hipStream_t stream);

// temperature / top-p sampling over bf16 logits rows (sampler.hip): one workgroup per row, token ids to out_ids[rows]
int td_sample_top_p_launch(const bf16_t* logits, long long ld, int rows, int vocab, float temperature, float top_p,
                           unsigned long long seed, unsigned long long offset, int* out_ids, hipStream_t stream);

/* thinkdiff_hip.h — C ABI of the MI355X-native ThinkDiff hot path (libthinkdiff_hip.so).
 *
 * The reference (avi22bhattacharya/ThinkDiff-mlre) has no native boundary: its hot path runs inside
 * diffusers / transformers / vLLM Python calls.  Each entry point below names the reference call it
 * replaces (path:line relative to the reference tree, or [ext] for the pinned third-party package
 * the reference calls into).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - every `const void*` / `void*` tensor argument is a DEVICE pointer (HBM), 16-byte aligned,
 *    borrowed for the duration of the call; bf16 tensors are raw uint16 bit patterns, row-major;
 *  - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work and never synchronise (exceptions, by
 *    purpose: td_*_create / td_flux_set_precision allocate, td_flux_trace_end reads its events back);
 *  - return value: TD_OK or an error code; td_last_error() returns a thread-local message;
 *  - no entry point allocates device memory except td_flux_create / td_*_create (workspaces are
 *    sized once at creation), so every call is hipGraph-capturable.
 */
#ifndef THINKDIFF_HIP_H
#define THINKDIFF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { TD_OK = 0, TD_ERR_INVALID = 2, TD_ERR_HIP = 3 };
/* activation codes for td_linear_bf16 */
enum { TD_ACT_ID_NONE = 0, TD_ACT_ID_GELU_TANH = 1, TD_ACT_ID_GELU_ERF = 2, TD_ACT_ID_SILU = 3, TD_ACT_ID_QUICK_GELU = 4 };

const char* td_last_error(void);
int td_abi_version(void);

/* y[M,N] = act(x[M,K] . w[N,K]^T + bias) (* gate[N]) (+ res[M,N])      bf16 in/out, fp32 accumulate.
 * Replaces torch.nn.Linear (+ fused neighbours) wherever the reference's third-party stacks call it:
 * aligner mm_projector[0..2] (thinkdiff/models/blip_vision_t5_decoder.py:44-47), FLUX
 * to_q/k/v, to_out, ff, proj_mlp, proj_out [ext diffusers 0.31.0 transformer_flux.py], Qwen2-VL
 * q/k/v/o/gate/up/down [ext vLLM fork].  K % 64 == 0, N % 8 == 0; bias/gate/res may be NULL; res may
 * alias y.  Rounding points follow the reference's bf16 pipeline: Linear output, activation, gate
 * multiply and residual add each round to bf16. */
int td_linear_bf16(const void* x, int64_t ldx, const void* w, const void* bias, void* y, int64_t ldy,
                   int M, int N, int K, int act, const void* gate, const void* res, int64_t ldr,
                   void* stream);

/* Same contraction with two outputs: columns [0,n_split) -> y0 (act0), columns [n_split,N) -> y1
 * (act1).  This is FluxSingleTransformerBlock's fused [to_q|to_k|to_v|proj_mlp] projection
 * ([ext] transformer_flux.py FluxSingleTransformerBlock.forward).  n_split must be a multiple of the N tile the launcher
 * picks for the shape (64, 192 or 256 columns; 256 always qualifies -- TD_ERR_INVALID otherwise). */
int td_linear_split_bf16(const void* x, int64_t ldx, const void* w, const void* bias,
                         void* y0, int64_t ldy0, int act0, void* y1, int64_t ldy1, int act1,
                         int M, int N, int K, int n_split, void* stream);

/* y = x . w^T (+ bias) (+ res) with the contraction SPLIT over workgroups: few output tiles against a long K -- the KV-cached decode of 65..256
 * sequences (vLLM's max_num_seqs = 256, configs/qwen2_vl_embed_ccsbu.yaml:20: M <= 256 rows against N = hidden columns) -- would leave most CUs
 * idle.  split_k = -1: the launcher picks the parts (1 = an ordinary launch when a split does not pay), > 1: that many (must divide K / 64);
 * the parts' fp32 sums are added in index order by a second launch, so the result does not depend on scheduling.  y1 != NULL: columns >= n_split
 * go to y1 (n_split % 8 == 0).  M > 64 (smaller M is the weight-stream kernel's, which splits K itself).  tile_cfg as td_linear_grouped2_bf16
 * (4 = 256x128).  norm_w != NULL (split_k = -1, no y1, N % 512 == 0, N <= 4096): the reduction launch also writes norm_out[M, ld_norm] =
 * Qwen2RMSNorm(y0; norm_w, norm_eps) of each finished row, bit-identical to td_norm_rows_bf16 on y0 -- the decoder's o_proj -> post_attention_layernorm
 * and down_proj -> next input_layernorm pairs ([ext] transformers modeling_qwen2_vl.py Qwen2VLDecoderLayer.forward) as one Linear call. */
int td_linear_splitk_bf16(const void* x, int64_t ldx, const void* w, const void* bias, void* y0, int64_t ldy0, void* y1, int64_t ldy1, int n_split,
                          int M, int N, int K, const void* res, int64_t ldr, int tile_cfg, int split_k,
                          const void* norm_w, void* norm_out, int64_t ld_norm, float norm_eps, void* stream);

/* Two Linear problems in ONE launch (same N, K, strides, activation; own rows, weights, bias, gate,
 * residual): FluxTransformerBlock applies each projection to the image stream and to the text stream
 * with different weights ([ext] transformer_flux.py FluxTransformerBlock.forward: to_q/add_q_proj,
 * to_out/to_add_out, ff/ff_context); the 193-row text problem rides in the image problem's grid.
 * tile_cfg: -1 auto, 0 256x256, 1 256x64, 2 32x256, 3 288x192. */
int td_linear_grouped2_bf16(const void* x0, int M0, const void* w0, const void* bias0, const void* gate0,
                            const void* res0, void* y0, const void* x1, int M1, const void* w1,
                            const void* bias1, const void* gate1, const void* res1, void* y1,
                            int64_t ldx, int64_t ldy, int64_t ldr, int N, int K, int act, int tile_cfg,
                            void* stream);

/* 3x3 convolution, stride 1, zero padding 1, over an NHWC bf16 image as an implicit GEMM (no im2col buffer):
 *   y[H*W, Cout] = conv3x3(x[Hin*Win, Cin]) + bias (+ res[H*W, Cout]);   upsample2x != 0 fuses the nearest 2x
 * upsample that precedes the conv in Upsample2D (then Hin = H/2, Win = W/2), else Hin = H, Win = W.
 * w is [Cout, 9*Cin] with k = (ky*3 + kx)*Cin + c (td_conv3x3_pack_weight converts from torch's [Cout,Cin,3,3]).
 * Cin % 64 == 0, Cout % 8 == 0.  Replaces nn.Conv2d(3x3) of [ext] diffusers AutoencoderKL.decode
 * (ResnetBlock2D.conv1/conv2, Upsample2D.conv, Decoder.conv_in/conv_out). */
int td_conv3x3_nhwc_bf16(const void* x, const void* w, const void* bias, const void* res, void* y,
                         int H, int W, int Cin, int Cout, int upsample2x, void* stream);
/* [Cout, Cin, 3, 3] (torch) -> [Cout_pad, 9*Cin_pad] (k = tap*Cin_pad + c), zero filled padding. */
int td_conv3x3_pack_weight(const void* w_oihw, void* w_packed, int Cout, int Cin, int Cout_pad, int Cin_pad, void* stream);
/* y[M,N] (fp32) = x[M,K] . w[N,K]^T (+ bias): unrounded rows, e.g. attention scores. */
int td_linear_f32out_bf16(const void* x, int64_t ldx, const void* w, const void* bias, float* y, int64_t ldy,
                          int M, int N, int K, void* stream);

/* o[b,s,h*128+d] = softmax(q.k^T * scale (+causal mask)) . v, head_dim 128, fp32 softmax state.
 * q/k/v/o are token-major: row s of batch b at  ptr + b*bstride + s*ld  (elements), head h at
 * column h*128, so the fused QKV projection output is consumed in place.  Hq % Hkv == 0 (GQA).
 * causal != 0: key j visible to query i iff j <= i + (Skv - Sq).
 * Replaces F.scaled_dot_product_attention in [ext] diffusers 0.31.0 attention_processor.py
 * FluxAttnProcessor2_0 / FluxSingleAttnProcessor2_0 (joint [text||image] attention, no mask) and
 * the vLLM fork's Qwen2-VL attention (thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:1083). */
int td_attention_bf16(const void* q, int64_t ldq, int64_t q_bstride, const void* k, const void* v,
                      int64_t ldkv, int64_t kv_bstride, void* o, int64_t ldo, int64_t o_bstride,
                      int batch, int Sq, int Skv, int Hq, int Hkv, int head_dim, float scale,
                      int causal, void* stream);
/* Packed variable-length form (the vision towers' cu_seqlens, [ext] Qwen2VisionTransformerPretrainedModel.forward: full attention
 * inside each image / frame): q, k, v, o are [total rows, ld] with segment b = rows [seg_starts[b], seg_starts[b+1]); seg_starts is a
 * DEVICE int32[n_seg + 1]; max_len = the longest segment (sizes the grid).  One launch for all segments, no mask across them. */
int td_attention_varlen_bf16(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int64_t ldo,
                             const int* seg_starts, int n_seg, int max_len, int Hq, int Hkv, float scale, void* stream);
/* The joint attention in the form the FLUX engine calls it: q already carries scale x log2(e) (folded in where RoPE rounds q to bf16), so the
 * scores arrive in the exp2 domain; one batch entry, Hq == Hkv = H, head_dim 128.  score_bound in (0, 48]: a bound of |q'.k| (log2 units) the
 * caller vouches for -- the scores are then exponentiated as they are: no reference point, no per-tile row maximum, no rescale (a bf16 probability
 * keeps its mantissa at any magnitude and the sums are fp32: exp2(+-48) is far inside the range).  FLUX has such a bound by construction: the
 * QK-RMSNorm leaves |q'| <= premul x sqrt(128) x max|w_q| and |k| <= sqrt(128) x max|w_k|.  A score a few octaves past the bound is harmless.
 * 0: the running-maximum form (any scores).  Replaces F.scaled_dot_product_attention inside
 * [ext] diffusers FluxAttnProcessor2_0 (after norm_q / norm_k and apply_rotary_emb). */
int td_attention_joint_prescaled_bf16(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int64_t ldo, int S, int H,
                                      float score_bound, void* stream);
/* The joint (unmasked, one batch entry, Hq == Hkv) attention with BOTH products on the block-scaled e4m3 matrix instruction
 * (csrc/attention_fp8.hip): the same bf16 q / k / v / o and strides as td_attention_bf16; q, k, v are packed to e4m3 under
 * power-of-two scales (q, k per (token, head); v per (64-key tile, head)) in one pass into `workspace`
 * (td_attention_fp8_workspace_bytes(Sq, Skv, Hq) bytes, 16-byte aligned, contents undefined afterwards), the probabilities
 * are rounded to e4m3, accumulation and the softmax state stay fp32.  An option of the 8-bit modes (the reference graph has no
 * such path): its error is ~5 % of an attention output on random operands, see DESIGN.md section 4 for what it does to an image. */
size_t td_attention_fp8_workspace_bytes(int Sq, int Skv, int Hq);
int td_attention_fp8(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int64_t ldo,
                     int Sq, int Skv, int Hq, float scale, void* workspace, void* stream);
/* The same attention fed from the RAW fused projection buffer: the pack pass applies the per-head QK-RMSNorm (weights may be NULL) and the
 * interleaved-pair rotary embedding of td_qk_norm_rope_bf16(rotate_half = 0) on its way to e4m3 -- bit-identical to calling that kernel
 * and then td_attention_fp8, minus one HBM round trip over q | k; qkv is left unmodified.  Replaces, in the 8-bit attention mode, [ext] diffusers FluxAttnProcessor2_0: attn.norm_q / norm_k (+ norm_added_*) ->
 * apply_rotary_emb -> F.scaled_dot_product_attention.  qkv [S, ld] bf16 with q / k / v head blocks at columns q_col / k_col / v_col; cos,
 * sin fp32 [S,128]; rows < split take (wqA, wkA), the rest (wqB, wkB). */
int td_attention_fp8_qk_rope(const void* qkv, int64_t ld, int q_col, int k_col, int v_col, void* o, int64_t ldo, int S, int H,
                             const float* cos, const float* sin, int split, const void* wqA, const void* wkA, const void* wqB, const void* wkB,
                             float eps, float scale, void* workspace, void* stream);
/* Selects the kernel structure td_attention_bf16 launches: 0 = shipped (joint attention with more (query tile, head) items
 * than CUs runs as one round of persistent workgroups over equal KV-tile ranges; the first such call on a device allocates a
 * 35 MB hand-off workspace, so make it before capturing into a hipGraph), 1 = one workgroup per item for every shape, 2 = the
 * persistent form without the XCD-aware range order.  1 and 2 exist for in-process A/B measurements.  Returns the previous value. */
int td_attention_set_variant(int variant);
/* KV-cached decode attention (Sq = 1): q heads of one kv head handled per workgroup.  0 = automatic (the largest divisor of
 * Hq / Hkv among 7, 6, 4, 3, 2 that still gives every CU a workgroup, else 1); a divisor forces it (tests, A/B).  Returns the
 * previous value. */
int td_attention_decode_set_group(int g);

/* ---- row kernels (each is also used inside the FLUX engine) -------------------------------------- */

/* y = norm(x) [* w] [modulated]:  rms=0 LayerNorm(no affine, eps) as in [ext] diffusers
 * AdaLayerNormZero/Single/Continuous, rms=1 RMSNorm as T5LayerNorm (aligner tail,
 * thinkdiff/models/blip_vision_t5_decoder.py:50-53) and Qwen2RMSNorm.  Optional adaLN modulation
 * y = y*(1+scale)+shift with separate (shift,scale) for rows < split and rows >= split. D % 512 == 0. */
int td_norm_rows_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, int rms, float eps,
                      const void* w, int split, const void* shiftA, const void* scaleA,
                      const void* shiftB, const void* scaleB, void* stream);

/* In-place per-head RMSNorm(q), RMSNorm(k) (weights may be NULL = no norm) + rotary embedding on a
 * fused projection buffer.  rotate_half=0: FLUX interleaved pairs ([ext] diffusers apply_rotary_emb);
 * rotate_half=1: Qwen2 half-split.  cos/sin: fp32 [rows,128]. */
int td_qk_norm_rope_bf16(void* qkv, int64_t ld, int rows, int Hq, int Hk, int q_col, int k_col,
                         const float* cos, const float* sin, int split, const void* wqA, const void* wkA,
                         const void* wqB, const void* wkB, float eps, int rotate_half, void* stream);

/* FluxPosEmbed tables: ids fp32 [S,3] -> cos,sin fp32 [S,128]. */
int td_flux_rope_table(const float* ids, int S, const int* axes_dims3, double theta, float* cos, float* sin, void* stream);
/* Timesteps(256) sinusoid: t fp32 [n] (device) -> bf16 [n,256] = [cos | sin]. */
int td_timestep_sincos(const float* t, int n, void* out, void* stream);
/* FlowMatchEulerDiscreteScheduler.step on bf16 latents: x = bf16(float(x) + dt*float(v)), product and sum each rounded
 * to fp32 as torch's `sample + dt * model_output` does (no fused multiply-add). */
int td_euler_step_bf16(void* x, const void* v, float dt, int64_t n, void* stream);
/* FluxPipeline._pack_latents (unpack=0: [C,H,W] -> [(H/2)(W/2),4C]) / _unpack_latents (unpack=1, with
 * out = bf16(bf16(in / div) + add), i.e. the `latents / scaling_factor + shift_factor` on bf16 tensors that precedes
 * vae.decode in [ext] pipeline_flux.py, with torch's CPU scalar rules: fp32 divisor, addend cast to bf16, quotient and sum
 * each round to bf16; div = 1, add = 0 is a pure unpack). */
int td_flux_pack_latents(const void* src, void* dst, int C, int H, int W, int unpack, float div, float add, void* stream);
/* ThinkDiff-CLIP token pooling (blip_vision_t5_decoder.py:620-637): [1+G*G,C] -> [1+(G/2)^2,C]. */
int td_cls_avgpool2_bf16(const void* x, void* y, int G, int C, void* stream);
/* counter-based N(mean,std) fill (synthetic checkpoints for throughput runs). */
int td_fill_normal_bf16(void* dst, int64_t n, uint64_t seed, float std, float mean, void* stream);

/* ThinkDiff aligner `mm_projector` of type "mlp2x_gelu_t5_norm" (every shipped config):
 *   y = T5LayerNorm(Linear2(GELU_erf(Linear0(x))))        x:[M,K] -> y:[M,hidden]
 * Replaces build_vision_projector's nn.Sequential (thinkdiff/models/blip_vision_t5_decoder.py:31-61,
 * call sites :641 and thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:1113-1116).  State-dict
 * tensors: w0 = mm_projector.0.weight [hidden,K], b0, w2 = mm_projector.2.weight [hidden,hidden], b2,
 * norm_w = mm_projector.3.weight [hidden].  workspace: 2*M*hidden bf16.  K % 64 == 0, hidden % 512 == 0.
 * fp32_norm != 0 reproduces the LVLM path (fp32 norm params under autocast: no bf16 rounding inside
 * the norm, output rounded once). */
int td_aligner_mlp2x_bf16(const void* x, int64_t ldx, int M, int K, int hidden, const void* w0, const void* b0,
                          const void* w2, const void* b2, const void* norm_w, float eps, int fp32_norm,
                          void* workspace, void* y, int64_t ldy, void* stream);

/* ---- FLUX.1 MMDiT denoise engine ------------------------------------------------------------------
 * Replaces the `diffusion_pipe(prompt_embeds=..., pooled_prompt_embeds=..., height, width,
 * num_inference_steps, guidance_scale)` denoise loop the reference drivers call
 * (scripts/test/test_blip_vision_t5_decoder_flux_text.py:234-242, scripts/test/
 * test_mllama_t5_decoder_flux.py:182-192), i.e. [ext] diffusers 0.31.0 FluxTransformer2DModel.forward
 * + FlowMatchEulerDiscreteScheduler.step.  Parameters are addressed by their diffusers state-dict
 * names (e.g. "transformer_blocks.3.attn.to_k.weight"), so a FLUX.1-dev checkpoint loads unchanged. */
typedef struct td_flux td_flux;
typedef struct TdFluxConfig {
  int in_channels;        /* 64  */
  int num_layers;         /* 19  */
  int num_single_layers;  /* 38  */
  int num_heads;          /* 24  */
  int head_dim;           /* 128 */
  int joint_dim;          /* 4096 */
  int pooled_dim;         /* 768 */
  int guidance_embeds;    /* 1   */
  int mlp_ratio;          /* 4   */
  int axes_dims[3];       /* 16,56,56 */
  float rope_theta;       /* 10000 */
} TdFluxConfig;

int td_flux_create(const TdFluxConfig* cfg, int max_img_tokens, int max_txt_tokens, int max_steps, td_flux** out);
void td_flux_destroy(td_flux* f);
int64_t td_flux_param_elems(const td_flux* f);
int td_flux_num_params(const td_flux* f);
int td_flux_param_info(const td_flux* f, int idx, char* name_buf, int buf_len, int64_t* count);
/* copy one parameter (device bf16, `count` elements) into the engine's fused weight arena */
int td_flux_load_param(td_flux* f, const char* name, const void* src, int64_t count, void* stream);
/* A second context over the same weights (own workspace, conditioning, timestep schedule): independent images in flight on
 * separate streams fill the tails of each other's kernels (a 1024^2 step's grids are 1.6 - 3.2 rounds of the 256 CUs).
 * The parent must outlive its forks; parameters and precision are the parent's.  Destroy with td_flux_destroy. */
int td_flux_fork(td_flux* parent, td_flux** out);
/* td_flux_denoise for `count` contexts (a parent and its forks), advanced step by step, context k on streams[k]. */
int td_flux_denoise_multi(td_flux* const* fs, void* const* latents, int count, const float* sigmas, int n, void* const* streams);

/* Operand precision of the block GEMMs.  TD_PRECISION_FP8_E4M3 quantises every double-/single-stream Linear weight per
 * output channel from the parameters as loaded NOW (call after loading; call again after reloading) and runs those GEMMs on
 * the fp8 MFMA path with per-token dynamic activation scales (BASELINE config 5).  Accumulation, epilogues, attention,
 * normalisation and the residual stream stay as in the bf16 path. */
enum { TD_PRECISION_BF16 = 0, TD_PRECISION_FP8_E4M3 = 1,
       /* W8A8 symmetric int8 (round 3): the same per-output-channel weight / per-token activation scaling and the same 2x-bf16 MFMA
        * rate (v_mfma_i32_16x16x64_i8, exact int32 accumulation), with a uniform step of max/127 instead of e4m3's 3-bit mantissa */
       TD_PRECISION_INT8 = 2 };
int td_flux_set_precision(td_flux* f, int precision, void* stream);
/* TD_PRECISION_INT8 only: source of the per-token activation scales of the MLP operands.  0 (default) = measured on the spot (one
 * quantisation pass per tensor); 1 = the maxima the PREVIOUS denoise step accumulated for the same tensor and token x 1.25 (clip beyond):
 * the MLP intermediate then leaves the producing GEMM epilogue as int8.  First steps and out-of-order steps use mode 0.  Parent context. */
int td_flux_set_act_scales(td_flux* f, int mode);
/* TD_PRECISION_INT8 only: per-channel smoothing of the activations that carry outlier channels (SmoothQuant's balance at alpha = 1/2, factors
 * rounded to powers of two so that x / s and W s are exact).  mode 1: the FIRST int8 forward after the mode, the precision or a parameter changed
 * runs on the bf16 path and records per-channel maxima of the LayerNorm outputs and MLP intermediates; from then on those activations are divided by
 * s[channel] where they are quantised (LayerNorm kernel, int8 GEMM epilogue, quantisation pass) and the consuming weights' input channels are
 * multiplied by s before their own quantisation.  Per-token int8 then no longer spends its 8 bits on a few channels that run tens of times above
 * the rest (the failure mode of W8A8 on trained DiTs; tests/test_flux_full_depth_gpu.py grades it on the heavy-tailed fixture).  0 = off (default).
 * The reference has no such path (it runs bf16): this belongs to BASELINE config 5's 8-bit MFMA path.  Parent context. */
int td_flux_set_smoothing(td_flux* f, int mode);
/* Arithmetic of the joint attention in every block: TD_ATTENTION_BF16 (default: the reference graph's) or TD_ATTENTION_FP8 (QK^T and P.V
 * on the e4m3 matrix instruction, td_attention_fp8).  Independent of td_flux_set_precision; meant for the 8-bit modes.  Parent context. */
enum { TD_ATTENTION_BF16 = 0, TD_ATTENTION_FP8 = 1 };
int td_flux_set_attention(td_flux* f, int mode);
/* Which block Linears take the fp8 path while the precision is TD_PRECISION_FP8_E4M3 (default: all).  The rest run in bf16 from
 * the bf16 weights: a speed / deviation-from-bf16 trade.  Parent context only; takes effect at the next step. */
enum { TD_FP8_QKV = 1, TD_FP8_OUT = 2, TD_FP8_FF1 = 4, TD_FP8_FF2 = 8,      /* double-stream blocks: to_q|k|v (+add_*), to_out, ff.net.0, ff.net.2 */
       TD_FP8_SINGLE_IN = 16, TD_FP8_SINGLE_OUT = 32,                        /* single-stream blocks: to_q|k|v + proj_mlp, proj_out */
       TD_FP8_ALL_GEMMS = 63 };
int td_flux_set_fp8_gemms(td_flux* f, unsigned mask);
int td_flux_init_random(td_flux* f, uint64_t seed, float std, void* stream);
/* per prompt: prompt_embeds bf16 [T,joint_dim], pooled bf16 [pooled_dim], ids fp32 device [n,3]
 * (txt_ids NULL = zeros, thinkdiff/models/flux_prompt.py:119) */
int td_flux_set_condition(td_flux* f, const void* prompt_embeds, int T, const void* pooled, const float* txt_ids,
                          const float* img_ids, int S_img, void* stream);
/* per schedule: t_eff (host, n floats) / g_eff = the values fed to the sinusoids (timestep*1000,
 * guidance*1000 after the pipeline's dtype casts); precomputes temb and every adaLN modulation */
int td_flux_set_timesteps(td_flux* f, const float* t_eff, int n, float g_eff, void* stream);
/* The extents a prepared context expects: image / text tokens of the last td_flux_set_condition (0 before it), latent channels, prepared
 * timesteps.  Any out pointer may be NULL.  (The torch.ops layer checks tensor extents against it before handing pointers over.) */
int td_flux_prepared_shape(const td_flux* f, int* img_tokens, int* txt_tokens, int* in_channels, int* n_steps);
/* velocity[S_img,in_channels] = transformer(latents[S_img,in_channels]; prepared step) */
int td_flux_forward(td_flux* f, const void* latents, int step, void* velocity, void* stream);
/* Per-launch HIP-event trace of the engine's kernels (events recorded on the launch stream).
 * categories: 0 GEMM 256x256 tile (td_gemm_bf16_nt_kernel<8,4>), 1 small GEMM tiles, 2 attention,
 * 3 LayerNorm+modulate, 4 QK-RMSNorm+RoPE, 5 GEMM 288x192 tile (<9,3>).  trace_end synchronises and
 * fills 6-element arrays. */
int td_flux_trace_begin(td_flux* f, int max_launches);
int td_flux_trace_end(td_flux* f, void* stream, int64_t* counts, double* ms, double* flops);
/* n Euler steps in place; sigmas: n+1 host floats */
int td_flux_denoise(td_flux* f, void* latents, const float* sigmas, int n, void* stream);

/* ---- fp8 operand path (BASELINE config 5 "fp8 MFMA FLUX path") -------------------------------
 * Operands are OCP e4m3 bytes with one fp32 dequantisation scale per row: weights per output channel (quantised once at
 * load), activations per token (quantised by the kernel that produces them).  The contraction runs on
 * v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) at twice the bf16 MFMA rate; everything after the
 * accumulators (bias, activation, gate, residual, bf16 rounding points) is the bf16 epilogue. */
/* q[r,:] = e4m3(x[r,:] / s_r), s_r = max|x[r,:]| / 448 (1 for an all-zero row) -> scale[r].  K % 8 == 0. */
int td_quant_rows_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int K, void* stream);
/* y = epilogue((xq . wq^T) * x_scale[m] * w_scale[n]); same epilogue arguments as td_linear_bf16.  K % 128 == 0. */
int td_linear_fp8(const void* xq, int64_t ldx, const float* x_scale, const void* wq, const float* w_scale, const void* bias,
                  void* y, int64_t ldy, int M, int N, int K, int act, const void* gate, const void* res, int64_t ldr,
                  int tile_cfg, void* stream);
/* ---- int8 operand path (round 3; TD_PRECISION_INT8) -- the 8-bit form whose pixels stay inside the 1e-2 bar at full coverage -----
 * Same scaling scheme and call shape as the fp8 entries: q[r,:] = rint(x[r,:] / s_r) as int8, s_r = max|x[r,:]| / 127 (1 for an
 * all-zero row); the contraction runs on v_mfma_i32_16x16x64_i8 (exact int32 accumulation) at twice the bf16 MFMA rate. */
int td_quant_rows_int8(const void* x, int64_t ldx, void* q, int64_t ldq, float* scale, int rows, int K, void* stream);
int td_linear_int8(const void* xq, int64_t ldx, const float* x_scale, const void* wq, const float* w_scale, const void* bias,
                   void* y, int64_t ldy, int M, int N, int K, int act, const void* gate, const void* res, int64_t ldr,
                   int tile_cfg, void* stream);
/* td_norm_rows_bf16 whose output row is quantised in registers: q [rows, ldq] e4m3 + q_scale[rows] (no bf16 copy). */
int td_norm_rows_quant_fp8(const void* x, int64_t ldx, void* q, int64_t ldq, float* q_scale, int rows, int D, int rms, float eps,
                           const void* w, int split, const void* shiftA, const void* scaleA, const void* shiftB, const void* scaleB,
                           void* stream);

/* ---- building blocks of the text encoders (T5-XXL, CLIP-L) feeding encode_prompt
 * (thinkdiff/models/flux_prompt.py:88-104 -> [ext] FluxPipeline._get_t5_prompt_embeds / _get_clip_prompt_embeds) ---- */
/* y = norm(x) for any D % 8 == 0: rms = 0 nn.LayerNorm(w, b, eps), rms = 1 T5LayerNorm(w). */
int td_layernorm_bf16(const void* x, int64_t ldx, void* y, int64_t ldy, int rows, int D, int rms, float eps,
                      const void* w, const void* b, void* stream);
/* out[r,:] = a[r,:] + b[r % b_rows,:]  (token + position embeddings, bias rows). */
int td_add_rows_bf16(const void* a, const void* b, void* out, int rows, int D, int b_rows, void* stream);
/* out[m,j] = act(gu[m,j]) * gu[m,I+j]  (T5 gated-GELU / SwiGLU); act = TD_ACT_ID_*. */
int td_glu_mul_bf16(const void* gate_up, void* out, int rows, int I, int act, void* stream);
/* td_attention_bf16 with an additive fp32 score bias [Hq,Sq,Skv] (T5 relative position bias), batch 1. */
int td_attention_bias_bf16(const void* q, int64_t ldq, const void* k, const void* v, int64_t ldkv, void* o, int64_t ldo,
                           int Sq, int Skv, int Hq, int Hkv, float scale, int causal, const float* bias, void* stream);

/* ---- building blocks of the vision towers upstream of the aligner (EVA-ViT-g = Blip2VisionModel,
 * thinkdiff/models/blip_vision_t5_decoder.py:611-618; Qwen2-VL ViT inside the vLLM engine,
 * thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:1083-1089) ---- */
/* In-place rotate_half RoPE over the first hd columns of H heads spaced head_stride apart; cos/sin fp32 [S, hd/2]. */
int td_rope_half_bf16(void* x, int64_t ldx, int S, int H, int head_stride, int hd, const float* cos_t, const float* sin_t, void* stream);
/* cos/sin fp32 [S, hd/2] of the 2-D vision rotary from pos int32 [S,2] = (row, column) of each patch (device). */
int td_vision_rope_table(const int* pos, int S, int hd, float theta, float* cos_t, float* sin_t, void* stream);
/* Conv2d(kernel = stride = p) operand: pix [C,H,W] (fp32 if src_f32 else bf16) -> out [(H/p)(W/p), Kpad] bf16, zero padded. */
int td_patchify_bf16(const void* pix, int src_f32, int C, int H, int W, int p, void* out, int Kpad, void* stream);
/* Qwen2-VL image preprocessing after the resize ([ext] transformers Qwen2VLImageProcessor rescale / normalize / patchify, which
 * vLLM runs on the host for thinkdiff/models/mllama_vllm_generate_1.py:543-583): img uint8 [H,W,3] (device) -> out bf16
 * [(H/patch)(W/patch), Kpad] in the processor's merge-window row order, columns (c, t, py, px); lut fp32 [3,256] = the
 * processor's own rescale + normalize of every pixel value per channel (device). */
int td_qwen2_patchify_u8(const void* img_hwc, int H, int W, const float* lut, int patch, int merge, int temporal, void* out, int Kpad, void* stream);
/* out[r, :K] = bf16(src[r, :K]); out[r, K:Kpad] = 0. */
int td_cast_pad_rows_bf16(const void* src, int src_f32, int rows, int K, void* out, int Kpad, void* stream);

/* ---- FLUX VAE decoder (AutoencoderKL.decode) -------------------------------------------------------------
 * Replaces the tail of the drivers' `diffusion_pipe(...)` call: [ext] diffusers 0.31.0 FluxPipeline
 * `_unpack_latents` + `latents / scaling_factor + shift_factor` + `vae.decode` + `image_processor.postprocess`
 * (scripts/test/test_blip_vision_t5_decoder_flux_text.py:234-247).  Parameters use the diffusers names
 * ("decoder.up_blocks.2.resnets.0.conv1.weight", ...), conv weights in torch [Cout,Cin,3,3] layout. */
typedef struct td_vae td_vae;
typedef struct TdVaeConfig {
  int latent_channels;        /* 16 */
  int out_channels;           /* 3 */
  int num_blocks;             /* 4 */
  int block_out_channels[4];  /* 128,256,512,512 */
  int layers_per_block;       /* 2 */
  int norm_groups;            /* 32 */
} TdVaeConfig;
int td_vae_create(const TdVaeConfig* cfg, int max_latent_h, int max_latent_w, td_vae** out);
void td_vae_destroy(td_vae* f);
int td_vae_num_params(const td_vae* f);
int td_vae_param_info(const td_vae* f, int idx, char* name_buf, int buf_len, int64_t* count);
int td_vae_load_param(td_vae* f, const char* name, const void* src, int64_t count, void* stream);
/* seeded synthetic decoder; std <= 0: 1 / sqrt(fan_in) weights (images with contrast), else that std for every weight and bias */
int td_vae_init_random(td_vae* f, uint64_t seed, float std, void* stream);
/* packed latents bf16 [(h/2)(w/2), 4*latent_channels] (h, w = latent size, h*w % 64 == 0) -> image_u8 [8h,8w,3]
 * uint8 and/or image_chw bf16 [3,8h,8w] (= vae.decode output); either may be NULL. */
int td_vae_decode(td_vae* f, const void* packed_latents, int h, int w, float scaling_factor, float shift_factor,
                  void* image_u8, void* image_chw, void* stream);
/* The image size td_vae_decode writes for an h x w latent (2x per block but the last: 8h x 8w for the FLUX.1 VAE) and the channel count of
 * a packed latent row (4 x latent_channels); out pointers may be NULL.  (The torch.ops layer sizes and checks its tensors with it.) */
int td_vae_output_shape(const td_vae* f, int h, int w, int* H, int* W, int* packed_channels);
/* GroupNorm (+ optional SiLU) over an NHWC image x[P,C]; workspace: td_groupnorm_workspace_floats() floats. */
int td_groupnorm_nhwc_bf16(const void* x, void* y, int P, int C, int groups, float eps, const void* gamma,
                           const void* beta, int silu, float* workspace, void* stream);
int td_groupnorm_workspace_floats(void);
/* p[rows,cols] (bf16) = softmax(scale * s[rows,cols]) (fp32 in). */
int td_softmax_rows_f32_bf16(const float* s, void* p, int rows, int cols, float scale, void* stream);

/* ---- Qwen2-VL text decoder: hidden states at `model.norm` + KV-cached decoding -------------------------
 * Replaces the vLLM fork's model runner behind `self.mllama.generate(inputs, sampling_params)` with
 * `return_hidden_states=True` (thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:790-816,1083-1089;
 * thinkdiff/models/mllama_vllm_generate_1.py:382-413,586,614-615): `prompt_hidden_states` /
 * `outputs[0].hidden_states` are `hidden_out` of the prompt tokens / of the generated tokens.
 * Parameters use the Hugging Face names (model.embed_tokens.weight, model.layers.N.self_attn.q_proj.weight,
 * ..., model.norm.weight, lm_head.weight). */
typedef struct td_qwen2 td_qwen2;
typedef struct TdQwen2Config {
  int hidden;            /* 3584 (7B) / 1536 (2B) */
  int num_layers;        /* 28 */
  int num_heads;         /* 28 / 12 */
  int num_kv_heads;      /* 4 / 2 */
  int head_dim;          /* 128 */
  int intermediate;      /* 18944 / 8960 */
  int vocab;             /* 152064 / 151936 */
  int tie_embeddings;    /* 0 / 1 */
  int mrope_section[3];  /* 16,24,24 */
  float rms_eps;         /* 1e-6 */
  float rope_theta;      /* 1e6 */
} TdQwen2Config;

int td_qwen2_create(const TdQwen2Config* cfg, int max_tokens, td_qwen2** out);
void td_qwen2_destroy(td_qwen2* f);
int td_qwen2_num_params(const td_qwen2* f);
int td_qwen2_param_info(const td_qwen2* f, int idx, char* name_buf, int buf_len, int64_t* count);
int td_qwen2_load_param(td_qwen2* f, const char* name, const void* src, int64_t count, void* stream);
int td_qwen2_init_random(td_qwen2* f, uint64_t seed, float std, void* stream);
/* n new tokens at cache positions [pos0, pos0+n): token_ids int32[n] OR inputs_embeds bf16[n,hidden] (device),
 * position_ids int32[3,n] (M-RoPE t/h/w streams, device); hidden_out bf16[n,hidden] = model.norm output (may be
 * NULL); logits_last bf16[vocab] of the last token (may be NULL).  pos0 = 0 is a prefill; pos0 > 0 continues. */
int td_qwen2_forward(td_qwen2* f, const int* token_ids, const void* inputs_embeds, const int* position_ids, int n,
                     int pos0, void* hidden_out, void* logits_last, void* stream);
/* td_qwen2_forward on sequence `slot` of a handle partitioned by td_qwen2_set_slots (td_qwen2_forward = slot 0). */
int td_qwen2_forward_slot(td_qwen2* f, int slot, const int* token_ids, const void* inputs_embeds, const int* position_ids, int n,
                          int pos0, void* hidden_out, void* logits_last, void* stream);
/* Batched KV-cached decode (the precompute job, mllama_vllm_generate_1.py:585: vLLM decodes its whole request batch together):
 * the cache of max_tokens rows per layer is split into n_slots sequences of max_tokens / n_slots rows. */
int td_qwen2_create_slots(const TdQwen2Config* cfg, int slot_len, int n_slots, td_qwen2** out);   /* n_slots sequences of slot_len tokens */
/* ... and an activation workspace of ws_rows rows (>= slot_len): the capacity of a batched prefill, B x L <= ws_rows */
int td_qwen2_create_ex(const TdQwen2Config* cfg, int slot_len, int n_slots, int ws_rows, td_qwen2** out);
/* Prefill B right-padded sequences (row b*L + t) into slots 0..B-1 in one pass; lens[b] = real tokens (HOST ints);
 * hidden_out bf16[B*L,hidden], logits_last bf16[B,vocab] of each sequence's last real token (either may be NULL). */
int td_qwen2_prefill_batch(td_qwen2* f, int B, int L, const int* token_ids, const void* inputs_embeds, const int* position_ids,
                           const int* lens, void* hidden_out, void* logits_last, void* stream);
/* ... into slots slot0..slot0+B-1: a request batch with B x L above ws_rows is prefilled in several calls */
int td_qwen2_prefill_batch_at(td_qwen2* f, int slot0, int B, int L, const int* token_ids, const void* inputs_embeds, const int* position_ids,
                              const int* lens, void* hidden_out, void* logits_last, void* stream);
/* Packed prefill: the B prompts lie back to back -- rows [sum(lens[:b]), sum(lens[:b+1])) belong to sequence b, no padding rows -- and go to cache
 * slots slot0 .. slot0 + B - 1: what vLLM's scheduler does with the reference's request batches (max_num_batched_tokens rows per pass,
 * configs/qwen2_vl_embed_ccsbu.yaml:19; thinkdiff/models/mllama_vllm_generate_1.py:585).  total = sum(lens) <= the handle's workspace rows.
 * token_ids int32[total] or inputs_embeds bf16[total, hidden]; position_ids int32[3, total]; lens HOST int[B]; hidden_out bf16[total, hidden];
 * logits_last bf16[B, vocab] of each prompt's last token (either output may be NULL). */
int td_qwen2_prefill_packed(td_qwen2* f, int slot0, int B, const int* token_ids, const void* inputs_embeds, const int* position_ids,
                            const int* lens, void* hidden_out, void* logits_last, void* stream);
/* ... sequence b into cache slot slots[b] (HOST ints, distinct, any order): the free slots of a running batch (continuous batching). */
int td_qwen2_prefill_packed_slots(td_qwen2* f, int B, const int* slots, const int* token_ids, const void* inputs_embeds, const int* position_ids,
                                  const int* lens, void* hidden_out, void* logits_last, void* stream);
int td_qwen2_set_slots(td_qwen2* f, int n_slots);   /* re-partition the cache rows of an existing handle */
/* Decode step: the rotary embedding of the new q / k rows and the write of the new k | v rows into the cache happen inside the decode-attention
 * launch (default, one launch per layer fewer) or in a launch of their own (on = 0: A/B and the bit-identity test).  Same arithmetic either way
 * (the vLLM fork's rotary_emb + cache write, thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:1083).  Returns the previous setting. */
int td_qwen2_set_fused_rope(td_qwen2* f, int on);
int td_qwen2_slot_capacity(const td_qwen2* f);
/* copy the first `len` cache rows of sequence src to sequence dst (compaction when a sequence finishes) */
int td_qwen2_move_slot(td_qwen2* f, int src, int dst, int len, void* stream);
/* One new token for each of the sequences in slots 0..B-1 (B <= 64) in one pass over the weights: token_ids int32[B],
 * position_ids int32[3,B] (device); cache_pos[b] = tokens already cached for sequence b (HOST ints); hidden_out bf16[B,hidden],
 * logits bf16[B,vocab] (either may be NULL). */
int td_qwen2_decode_batch(td_qwen2* f, int B, const int* token_ids, const int* position_ids, const int* cache_pos,
                          void* hidden_out, void* logits, void* stream);
/* td_qwen2_decode_batch for the sequences in cache slots slots[0 .. B-1] (HOST ints, distinct; NULL = 0 .. B-1); row b of token_ids / position_ids /
 * cache_pos / hidden_out / logits belongs to slot slots[b].  A finished sequence then frees its slot without any cache rows being moved and a waiting
 * request is prefilled into it (td_qwen2_prefill_packed_slots): vLLM's continuous batching of the reference's request batches
 * (thinkdiff/models/mllama_vllm_generate_1.py:585, `max_num_seqs: 256`), with whole-sequence slots in place of paged blocks. */
int td_qwen2_decode_batch_slots(td_qwen2* f, int B, const int* slots, const int* token_ids, const int* position_ids, const int* cache_pos,
                                void* hidden_out, void* logits, void* stream);
/* out bf16[n,hidden] = embed_tokens[token_ids] (device int32[n]): the host splices vision tokens into this to form inputs_embeds. */
int td_qwen2_embed_tokens(td_qwen2* f, const int* token_ids, void* out, int n, void* stream);

/* Qwen2 building blocks */
int td_embed_gather_bf16(const int* ids, const void* table, void* out, int n, int D, int vocab, void* stream);
int td_silu_mul_bf16(const void* gate_up, void* out, int rows, int I, void* stream);
int td_mrope_table(const int* pos3n, int n, const int* sections3, float theta, int round_bf16, float* cos, float* sin, void* stream);
/* Temperature / top-p (nucleus) sampling, one token per row of bf16 logits [rows, ld], in ONE launch and without a sort:
 *   p = softmax(logits / temperature); keep the most likely tokens while the mass in front of a token is < top_p; renormalise;
 *   draw one token per row -> out_ids[rows] (device int32).  temperature <= 0: greedy (first index of the row maximum).
 * The draw of row r is a pure function of (seed, offset, r): callers advance `offset` once per generated token.
 * vocab % 8 == 0, ld % 8 == 0, rows <= 65535.  Replaces the sampler of vllm.SamplingParams(temperature, top_p) inside
 * LLM.generate (thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:817-823, thinkdiff/models/mllama_vllm_generate_1.py:398-405). */
int td_sample_top_p_bf16(const void* logits, int64_t ld, int rows, int vocab, float temperature, float top_p,
                         uint64_t seed, uint64_t offset, int32_t* out_ids, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* THINKDIFF_HIP_H */

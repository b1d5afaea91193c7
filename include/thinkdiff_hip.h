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
 *  - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work, never synchronise;
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
enum { TD_ACT_ID_NONE = 0, TD_ACT_ID_GELU_TANH = 1, TD_ACT_ID_GELU_ERF = 2, TD_ACT_ID_SILU = 3 };

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
 * ([ext] transformer_flux.py FluxSingleTransformerBlock.forward). n_split % 256 == 0. */
int td_linear_split_bf16(const void* x, int64_t ldx, const void* w, const void* bias,
                         void* y0, int64_t ldy0, int act0, void* y1, int64_t ldy1, int act1,
                         int M, int N, int K, int n_split, void* stream);

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

#ifdef __cplusplus
}
#endif
#endif /* THINKDIFF_HIP_H */

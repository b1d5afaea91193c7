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

int td_attention_bf16(const void* q, int64_t ldq, int64_t q_bstride, const void* k, const void* v,
                      int64_t ldkv, int64_t kv_bstride, void* o, int64_t ldo, int64_t o_bstride,
                      int batch, int Sq, int Skv, int Hq, int Hkv, int head_dim, float scale,
                      int causal, void* stream) {
  TdAttnParams p;
  p.Q = (const bf16_t*)q; p.K = (const bf16_t*)k; p.V = (const bf16_t*)v; p.O = (bf16_t*)o;
  p.batch = batch; p.Sq = Sq; p.Skv = Skv; p.Hq = Hq; p.Hkv = Hkv; p.head_dim = head_dim;
  p.ldq = (int)ldq; p.ldkv = (int)ldkv; p.ldo = (int)ldo;
  p.q_bstride = q_bstride; p.kv_bstride = kv_bstride; p.o_bstride = o_bstride;
  p.scale = scale; p.causal = causal; p.causal_offset = Skv - Sq;
  return td_attn_launch(p, (hipStream_t)stream);
}

}  // extern "C"

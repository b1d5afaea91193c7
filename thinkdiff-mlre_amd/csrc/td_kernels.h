// Internal launch interface between the kernel translation units and the C-ABI / engine.
// Not installed; the public surface is include/thinkdiff_hip.h.
#pragma once
#include "td_common.h"

enum TdAct { TD_ACT_NONE = 0, TD_ACT_GELU_TANH = 1, TD_ACT_GELU_ERF = 2, TD_ACT_SILU = 3 };

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
  int tiles_m = 0, tiles_n = 0;  // filled by the launcher
};

int td_gemm_launch(const TdGemmParams& p, hipStream_t stream);

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
};

int td_attn_launch(const TdAttnParams& p, hipStream_t stream);

// Qwen2-VL text-decoder engine: hidden-state extraction at `model.norm` (the embedding ThinkDiff-LVLM feeds
// to its aligner) and KV-cached decoding, on the same GEMM / attention / row kernels as the FLUX engine.
//
// Replaces the vLLM fork's Qwen2-VL model runner behind `self.mllama.generate(..., return_hidden_states)`
// (reference thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:790-816,1083-1089 and
// thinkdiff/models/mllama_vllm_generate_1.py:382-413,586,614-615).  Per-layer math follows transformers
// `Qwen2VLDecoderLayer` (modeling_qwen2_vl.py:453-625), which SURVEY.md 8a row A7 identifies as identical:
//   h += o_proj(Attn(RMSNorm(h)));  h += down(SiLU(gate(x)) * up(x)), x = RMSNorm(h)
//   Attn: q/k/v Linear with bias, M-RoPE (rotate_half, 3 position streams merged by mrope_section),
//   causal GQA softmax(q k^T / sqrt(128)) v, o_proj without bias.
//
// Layout: one fused [q | k | v] projection per layer whose k|v columns are written STRAIGHT into the
// layer's KV cache rows (dual-output GEMM epilogue), so prefill and decode share one code path:
// `td_qwen2_forward(tokens at positions [pos0, pos0+n))` attends over cache rows [0, pos0+n).
// Parameters are addressed by their Hugging Face names (model.layers.N.self_attn.q_proj.weight, ...).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "td_kernels.h"
#include "../../include/thinkdiff_hip.h"

namespace {

struct QSlot { std::string name; bf16_t* ptr; int64_t count; };

struct QLayer {
  bf16_t *qkv_w, *qkv_b;   // [(Hq + 2 Hkv) * 128, D]
  bf16_t* o_w;             // [D, Hq * 128]
  bf16_t* gu_w;            // [2 I, D] = gate_proj | up_proj
  bf16_t* down_w;          // [D, I]
  bf16_t *ln1_w, *ln2_w;   // [D]
  bf16_t* kv;              // cache [max_tokens, 2 * Hkv * 128]
};

}  // namespace

struct td_qwen2 {
  TdQwen2Config cfg;
  int D = 0, I = 0, Hq = 0, Hkv = 0, max_tokens = 0;
  bf16_t* arena = nullptr;
  int64_t arena_elems = 0;
  std::vector<QSlot> slots;
  std::unordered_map<std::string, int> index;
  bf16_t *embed_w, *norm_w, *lm_w;
  std::vector<QLayer> layers;
  char* ws = nullptr;
  bf16_t *h, *xn, *q, *attn, *gu, *act;
  float *cosT, *sinT;
  // batched decode: the cache rows of every layer are split into n_slots sequences of slot_len rows
  int n_slots = 1, slot_len = 0, ws_rows = 0;
  bf16_t* kvtmp = nullptr;   // [ws_rows, 2 Hkv 128] k|v rows of a batched step before they are scattered to their sequences
  bf16_t* lastrows = nullptr; // [MAX_BATCH, D] last-token rows of a batched prefill (lm_head input)
  int* ibuf = nullptr;       // device ints: kv lengths, scatter offsets
  // Decode-step graphs.  A decode step is ~8 small launches per layer (230 for 28 layers) whose grids and arguments depend on the batch size
  // only -- sequence lengths and cache rows are device ints (ibuf), tokens / positions / outputs go through the fixed buffers below -- so the
  // step is captured once per (batch size, logits wanted) on the engine's own stream and replayed as ONE hipGraphLaunch on the caller's stream:
  // the GPU then runs the step's kernels back to back instead of waiting ~6 us for the host between them (the step was launch-bound: 1.5 ms
  // at one sequence of the 2B shape against 0.4 ms of weight streaming).  TD_QWEN2_NO_GRAPH: A/B and bisecting.
  std::unordered_map<int, hipGraphExec_t> step_graphs;
  std::unordered_map<int, int> step_calls;      // calls seen per key: the first runs eagerly (one-time function attributes are set outside a capture)
  hipStream_t capture_stream = nullptr;
  bool graphs_ok = true;
  bool fused_rope = true;                       // decode: rotary embedding + cache write inside the attention launch (td_qwen2_set_fused_rope)
  int *tok_buf = nullptr, *pos_buf = nullptr;   // [MAX_BATCH], [3, MAX_BATCH]: the step's token and position ids
  bf16_t* logits_buf = nullptr;                 // [MAX_BATCH, vocab]
  int* row_map = nullptr;                       // [ws_rows] packed prefill: cache row of every packed prompt row
  std::vector<int> row_map_host;                // ... its host image (kept alive across the asynchronous upload)
  float* sk_ws = nullptr;                       // partial sums of the decode step's split-K Linears (65-256 sequences): the engine's own buffer, so a captured step allocates nothing
};

namespace {

void q_add(td_qwen2* f, const std::string& name, bf16_t* p, int64_t n) {
  f->index[name] = (int)f->slots.size();
  f->slots.push_back({name, p, n});
}

// Sequences per decode step (vLLM's max_num_seqs of the precompute job is 256: configs/qwen2_vl_embed_ccsbu.yaml:20).  Up to 64 the Linears of a step
// are weight streams (td_gemv_mfma_kernel: one pass over the weights for up to 64 rows); above, the step is a small-M GEMM problem and takes the tile
// kernels the prefill uses, with K split over workgroups where the output is narrow (td_gemm_launch's split_k form).
constexpr int MAX_BATCH = 256;
constexpr int STREAM_BATCH = 64;
constexpr int64_t SK_WS_BYTES = 32ll << 20;

struct IntPack { int v[3 * MAX_BATCH]; };      // (3 KB of kernel arguments: lengths | cache rows | cache slots of a decode step)
__global__ void td_set_ints_kernel(int* dst, IntPack vals, int n) {
  if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}
// dst_base[off[b] + c] = src[b, c]: the k|v rows of a decode step go to their sequences' cache rows
__global__ void td_scatter_rows_kernel(const bf16_t* src, bf16_t* dst_base, const int* off, int W) {
  const int b = blockIdx.y;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (c < W) *(u32x4_t*)(dst_base + (size_t)off[b] + c) = *(const u32x4_t*)(src + (size_t)b * W + c);
}

// row b*L + t of src -> cache row (b * slot_len + t): the k|v rows of a batched prefill go to their sequences' slots
__global__ void td_kv_rows_to_slots_kernel(const bf16_t* src, bf16_t* dst_base, int L, int slot_len, int W) {
  const int row = blockIdx.y;
  const int b = row / L, t = row - b * L;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (c < W) *(u32x4_t*)(dst_base + ((size_t)b * slot_len + t) * W + c) = *(const u32x4_t*)(src + (size_t)row * W + c);
}

// packed prefill: row r of src (prompts back to back) -> cache row dst_row[r] (its sequence's slot and position)
__global__ void td_kv_rows_to_rows_kernel(const bf16_t* src, bf16_t* dst_base, const int* dst_row, int W) {
  const int row = blockIdx.y;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (c < W) *(u32x4_t*)(dst_base + (size_t)dst_row[row] * W + c) = *(const u32x4_t*)(src + (size_t)row * W + c);
}
// dst[b, :] = src[seg_starts[b + 1] - 1, :]: the last row of every packed segment (lm_head input)
__global__ void td_gather_last_rows_kernel(const bf16_t* src, bf16_t* dst, const int* seg_starts, int D) {
  const int b = blockIdx.y;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (c < D) *(u32x4_t*)(dst + (size_t)b * D + c) = *(const u32x4_t*)(src + (size_t)(seg_starts[b + 1] - 1) * D + c);
}

// Decode step, one launch for: M-RoPE on the new q rows (in place), M-RoPE on the new k rows, k|v rows -> their sequences' cache
// rows.  Rotation = rotate_half with every op rounding to bf16 (td_qk_norm_rope_kernel rotate_half == 2: q*cos, rot*sin, sum).
__global__ __launch_bounds__(256) void td_decode_rope_scatter_kernel(bf16_t* q, const bf16_t* kv, bf16_t* cache, const int* row_off,
                                                                     const float* cosT, const float* sinT, int Hq, int Hkv) {
  const int b = blockIdx.x;
  const int l16 = threadIdx.x & 15, unit0 = threadIdx.x >> 4;
  const int QW = Hq * 128, KVW = 2 * Hkv * 128;
  float cs[8], sn[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { cs[i] = cosT[(size_t)b * 128 + l16 * 8 + i]; sn[i] = sinT[(size_t)b * 128 + l16 * 8 + i]; }
  bf16_t* dst = cache + (size_t)row_off[b] * KVW;      // (row index of the sequence's new cache row)
  for (int u = unit0; u < Hq + 2 * Hkv; u += 16) {
    const bool is_q = u < Hq, is_v = u >= Hq + Hkv;
    const bf16_t* src = is_q ? q + (size_t)b * QW + u * 128 + l16 * 8 : kv + (size_t)b * KVW + (u - Hq) * 128 + l16 * 8;
    u32x4_t raw = *(const u32x4_t*)src;
    if (!is_v) {
      float x[8], y[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { x[2 * i] = bf_lo(raw[i]); x[2 * i + 1] = bf_hi(raw[i]); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float other = __shfl_xor(x[i], 8, 16);
        const float rot = (l16 < 8) ? -other : other;
        y[i] = rbf(x[i] * cs[i]) + rbf(rot * sn[i]);
      }
      raw = u32x4_t{pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]), pack_bf2(y[4], y[5]), pack_bf2(y[6], y[7])};
    }
    bf16_t* out = is_q ? q + (size_t)b * QW + u * 128 + l16 * 8 : dst + (u - Hq) * 128 + l16 * 8;
    *(u32x4_t*)out = raw;
  }
}

#define TDQ_TRY(expr)         \
  do {                        \
    int _rc = (expr);         \
    if (_rc != 0) return _rc; \
  } while (0)

}  // namespace

extern "C" {

// KV cache of n_slots sequences x slot_len rows per layer; activation workspace for slot_len rows (the longest prefill)
int td_qwen2_create_ex(const TdQwen2Config* cfg, int slot_len, int n_slots, int ws_rows, td_qwen2** out) {
  TD_CHECK_ARG(cfg && out && slot_len > 0 && n_slots > 0 && (long long)slot_len * n_slots < (1ll << 30), "td_qwen2_create: bad arguments");
  if (ws_rows < slot_len) ws_rows = slot_len;
  const int max_tokens = slot_len * n_slots;
  TD_CHECK_ARG(cfg->head_dim == 128, "td_qwen2_create: head_dim must be 128");
  TD_CHECK_ARG((long long)max_tokens * 2 * cfg->num_kv_heads * 128 < (1ll << 31), "td_qwen2_create: %d cache rows of %d elements exceed the 32-bit row offsets of the decode step", max_tokens, 2 * cfg->num_kv_heads * 128);
  TD_CHECK_ARG(cfg->hidden % 512 == 0 && cfg->intermediate % 64 == 0, "td_qwen2_create: hidden %% 512 and intermediate %% 64 must be 0");
  TD_CHECK_ARG(cfg->num_heads % cfg->num_kv_heads == 0, "td_qwen2_create: heads must be a multiple of kv heads");
  TD_CHECK_ARG(cfg->mrope_section[0] + cfg->mrope_section[1] + cfg->mrope_section[2] == 64, "td_qwen2_create: mrope sections must sum to 64");
  td_qwen2* f = new td_qwen2();
  f->cfg = *cfg;
  const int D = f->D = cfg->hidden, I = f->I = cfg->intermediate;
  const int Hq = f->Hq = cfg->num_heads, Hkv = f->Hkv = cfg->num_kv_heads;
  const int NQKV = (Hq + 2 * Hkv) * 128;
  f->max_tokens = max_tokens;
  f->slot_len = slot_len;
  f->n_slots = n_slots;
  f->ws_rows = ws_rows;
  f->layers.resize(cfg->num_layers);

  int64_t off = 0;
  std::vector<std::pair<bf16_t**, int64_t>> fix;
  auto take = [&](bf16_t** p, int64_t n) { fix.emplace_back(p, off); off += (n + 127) & ~int64_t(127); };
  take(&f->embed_w, (int64_t)cfg->vocab * D);
  take(&f->norm_w, D);
  if (!cfg->tie_embeddings) take(&f->lm_w, (int64_t)cfg->vocab * D);
  for (auto& l : f->layers) {
    take(&l.qkv_w, (int64_t)NQKV * D); take(&l.qkv_b, NQKV);
    take(&l.o_w, (int64_t)D * Hq * 128);
    take(&l.gu_w, (int64_t)2 * I * D);
    take(&l.down_w, (int64_t)D * I);
    take(&l.ln1_w, D); take(&l.ln2_w, D);
    take(&l.kv, (int64_t)max_tokens * 2 * Hkv * 128);
  }
  f->arena_elems = off;
  hipError_t e = hipMalloc((void**)&f->arena, (size_t)off * 2);
  if (e != hipSuccess) {
    td_set_error("td_qwen2_create: hipMalloc of %.2f GiB failed: %s", off * 2.0 / (1 << 30), hipGetErrorString(e));
    delete f;
    return TD_ERR_HIP;
  }
  for (auto& fx : fix) *fx.first = f->arena + fx.second;
  if (cfg->tie_embeddings) f->lm_w = f->embed_w;

  q_add(f, "model.embed_tokens.weight", f->embed_w, (int64_t)cfg->vocab * D);
  q_add(f, "model.norm.weight", f->norm_w, D);
  if (!cfg->tie_embeddings) q_add(f, "lm_head.weight", f->lm_w, (int64_t)cfg->vocab * D);
  for (int i = 0; i < cfg->num_layers; ++i) {
    QLayer& l = f->layers[i];
    const std::string p = "model.layers." + std::to_string(i) + ".";
    q_add(f, p + "self_attn.q_proj.weight", l.qkv_w, (int64_t)Hq * 128 * D);
    q_add(f, p + "self_attn.q_proj.bias", l.qkv_b, Hq * 128);
    q_add(f, p + "self_attn.k_proj.weight", l.qkv_w + (int64_t)Hq * 128 * D, (int64_t)Hkv * 128 * D);
    q_add(f, p + "self_attn.k_proj.bias", l.qkv_b + Hq * 128, Hkv * 128);
    q_add(f, p + "self_attn.v_proj.weight", l.qkv_w + (int64_t)(Hq + Hkv) * 128 * D, (int64_t)Hkv * 128 * D);
    q_add(f, p + "self_attn.v_proj.bias", l.qkv_b + (Hq + Hkv) * 128, Hkv * 128);
    q_add(f, p + "self_attn.o_proj.weight", l.o_w, (int64_t)D * Hq * 128);
    q_add(f, p + "mlp.gate_proj.weight", l.gu_w, (int64_t)I * D);
    q_add(f, p + "mlp.up_proj.weight", l.gu_w + (int64_t)I * D, (int64_t)I * D);
    q_add(f, p + "mlp.down_proj.weight", l.down_w, (int64_t)D * I);
    q_add(f, p + "input_layernorm.weight", l.ln1_w, D);
    q_add(f, p + "post_attention_layernorm.weight", l.ln2_w, D);
  }

  const int64_t n = ws_rows;
  struct Req { void** p; int64_t bytes; };
  std::vector<Req> reqs = {
      {(void**)&f->h, n * D * 2}, {(void**)&f->xn, n * D * 2}, {(void**)&f->q, n * Hq * 128 * 2},
      {(void**)&f->attn, n * Hq * 128 * 2}, {(void**)&f->gu, n * 2 * I * 2}, {(void**)&f->act, n * I * 2},
      {(void**)&f->cosT, n * 128 * 4}, {(void**)&f->sinT, n * 128 * 4},
      {(void**)&f->kvtmp, n * 2 * Hkv * 128 * 2}, {(void**)&f->ibuf, (5 * MAX_BATCH + 16) * 4}, {(void**)&f->lastrows, (int64_t)MAX_BATCH * D * 2},
      {(void**)&f->tok_buf, MAX_BATCH * 4}, {(void**)&f->pos_buf, 3 * MAX_BATCH * 4}, {(void**)&f->logits_buf, (int64_t)MAX_BATCH * cfg->vocab * 2},
      {(void**)&f->sk_ws, SK_WS_BYTES}, {(void**)&f->row_map, n * 4},
  };
  int64_t total = 0;
  for (auto& r : reqs) total += (r.bytes + 255) & ~int64_t(255);
  e = hipMalloc((void**)&f->ws, (size_t)total);
  if (e != hipSuccess) {
    td_set_error("td_qwen2_create: hipMalloc of %.2f GiB workspace failed: %s", total / double(1 << 30), hipGetErrorString(e));
    (void)hipFree(f->arena);
    delete f;
    return TD_ERR_HIP;
  }
  (void)hipMemset(f->ws, 0, (size_t)total);
  (void)hipDeviceSynchronize();   // the handle may be used from any stream next; a null-stream memset is not ordered with non-blocking streams
  int64_t o = 0;
  for (auto& r : reqs) { *r.p = f->ws + o; o += (r.bytes + 255) & ~int64_t(255); }
  *out = f;
  return TD_OK;
}

static void drop_step_graphs(td_qwen2* f) {
  for (auto& kv : f->step_graphs) (void)hipGraphExecDestroy(kv.second);
  f->step_graphs.clear();
  f->step_calls.clear();
}

void td_qwen2_destroy(td_qwen2* f) {
  if (!f) return;
  drop_step_graphs(f);
  if (f->capture_stream) (void)hipStreamDestroy(f->capture_stream);
  (void)hipFree(f->arena);
  (void)hipFree(f->ws);
  delete f;
}

int td_qwen2_num_params(const td_qwen2* f) { return f ? (int)f->slots.size() : 0; }

int td_qwen2_param_info(const td_qwen2* f, int idx, char* name_buf, int buf_len, int64_t* count) {
  TD_CHECK_ARG(f && idx >= 0 && idx < (int)f->slots.size(), "td_qwen2_param_info: index %d out of range", idx);
  if (name_buf && buf_len > 0) {
    strncpy(name_buf, f->slots[idx].name.c_str(), buf_len - 1);
    name_buf[buf_len - 1] = 0;
  }
  if (count) *count = f->slots[idx].count;
  return TD_OK;
}

int td_qwen2_load_param(td_qwen2* f, const char* name, const void* src, int64_t count, void* stream) {
  TD_CHECK_ARG(f && name && src, "td_qwen2_load_param: null argument");
  auto it = f->index.find(name);
  TD_CHECK_ARG(it != f->index.end(), "td_qwen2_load_param: unknown parameter '%s'", name);
  const QSlot& s = f->slots[it->second];
  TD_CHECK_ARG(s.count == count, "td_qwen2_load_param: '%s' expects %lld elements, got %lld", name, (long long)s.count, (long long)count);
  TD_CHECK_HIP(hipMemcpyAsync(s.ptr, src, (size_t)count * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return TD_OK;
}

int td_qwen2_init_random(td_qwen2* f, uint64_t seed, float std, void* stream) {
  TD_CHECK_ARG(f, "td_qwen2_init_random: null handle");
  TDQ_TRY(td_fill_normal_bf16(f->arena, f->arena_elems, seed, std, 0.f, stream));
  for (const QSlot& s : f->slots)
    if (s.name.find("layernorm.weight") != std::string::npos || s.name == "model.norm.weight")
      TDQ_TRY(td_fill_normal_bf16(s.ptr, s.count, seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)(s.ptr - f->arena + 1)), 0.05f, 1.0f, stream));
  return TD_OK;
}

// Runs n new tokens at cache positions [pos0, pos0 + n) through the decoder.
// out[i,:] = embed_tokens[ids[i],:] -- the host builds `inputs_embeds` from this, replacing the image-placeholder
// rows with the vision tower's merged tokens ([ext] Qwen2VLModel.forward masked_scatter).
int td_qwen2_embed_tokens(td_qwen2* f, const int* token_ids, void* out, int n, void* stream) {
  TD_CHECK_ARG(f && token_ids && out && n > 0, "td_qwen2_embed_tokens: null argument");
  return td_embed_gather_launch(token_ids, f->embed_w, (bf16_t*)out, n, f->D, f->cfg.vocab, (hipStream_t)stream);
}

//   token_ids  : device int32 [n], or NULL when inputs_embeds is given
//   inputs_embeds : device bf16 [n, hidden] (token embeddings with the image-token rows replaced by the
//                vision tower's output), or NULL
//   position_ids : device int32 [3, n] M-RoPE streams (temporal, height, width); text tokens repeat one value
//   hidden_out : device bf16 [n, hidden] = model.norm(h)  -- the embedding the reference captures
//                ("embedding_layer_name: model.norm"); may be NULL
//   logits_last: device bf16 [vocab] for the LAST of the n tokens (lm_head), or NULL
int td_qwen2_forward_slot(td_qwen2* f, int slot, const int* token_ids, const void* inputs_embeds, const int* position_ids, int n,
                          int pos0, void* hidden_out, void* logits_last, void* stream) {
  TD_CHECK_ARG(f && position_ids && (token_ids || inputs_embeds), "td_qwen2_forward: null argument");
  TD_CHECK_ARG(slot >= 0 && slot < f->n_slots, "td_qwen2_forward: slot %d outside the %d configured sequences", slot, f->n_slots);
  TD_CHECK_ARG(n > 0 && pos0 >= 0 && pos0 + n <= f->slot_len, "td_qwen2_forward: positions [%d, %d) exceed the cache capacity %d", pos0, pos0 + n, f->slot_len);
  hipStream_t s = (hipStream_t)stream;
  const int D = f->D, I = f->I, Hq = f->Hq, Hkv = f->Hkv;
  const int QW = Hq * 128, KVW = 2 * Hkv * 128;

  if (inputs_embeds) TD_CHECK_HIP(hipMemcpyAsync(f->h, inputs_embeds, (size_t)n * D * 2, hipMemcpyDeviceToDevice, s));
  else TDQ_TRY(td_embed_gather_launch(token_ids, f->embed_w, f->h, n, D, f->cfg.vocab, s));
  // transformers casts the fp32 cos/sin tables to the model dtype before use
  TDQ_TRY(td_mrope_table_launch(position_ids, n, f->cfg.mrope_section, f->cfg.rope_theta, 1, f->cosT, f->sinT, s));

  TdNormParams np;
  np.x = f->h; np.ldx = D; np.y = f->xn; np.ldy = D; np.rows = n; np.D = D; np.rms = 1; np.eps = f->cfg.rms_eps;
  TdQkRopeParams rq;   // q heads in the q buffer
  rq.qkv = f->q; rq.ld = QW; rq.rows = n; rq.Hq = Hq; rq.Hk = 0; rq.q_col = 0; rq.k_col = 0;
  rq.cos = f->cosT; rq.sin = f->sinT; rq.rotate_half = 2;
  TdQkRopeParams rk = rq;  // k heads in the cache rows just written
  rk.ld = KVW; rk.Hq = Hkv;

  for (int i = 0; i < f->cfg.num_layers; ++i) {
    const QLayer& l = f->layers[i];
    bf16_t* kv_seq = l.kv + (size_t)slot * f->slot_len * KVW;   // this sequence's cache rows
    bf16_t* kv_new = kv_seq + (size_t)pos0 * KVW;
    np.w = l.ln1_w;
    TDQ_TRY(td_norm_rows_launch(np, s));
    {  // fused q | k | v projection: q -> scratch, k | v -> this layer's cache rows
      TdGemmParams g;
      g.A = f->xn; g.lda = D; g.W = l.qkv_w; g.bias = l.qkv_b; g.M = n; g.N = QW + KVW; g.K = D;
      g.C = f->q; g.ldc = QW; g.C2 = kv_new; g.ldc2 = KVW; g.n_split = QW;
      if (QW % 256 == 0) {
        // the column split needs a tile width that divides 256: every config but the 288x192 one
        g.cfg = n <= 32 ? -1 : (td_gemm_config_id(n, QW + KVW, D) == 1 ? 1 : 0);
        TDQ_TRY(td_gemm_launch(g, s));
      } else {
        TdGemmParams a = g; a.C2 = nullptr; a.N = QW;
        TDQ_TRY(td_gemm_launch(a, s));
        TdGemmParams b = g; b.C2 = nullptr; b.W = l.qkv_w + (size_t)QW * D; b.bias = l.qkv_b + QW; b.N = KVW; b.C = kv_new; b.ldc = KVW;
        TDQ_TRY(td_gemm_launch(b, s));
      }
    }
    rq.qkv = f->q; TDQ_TRY(td_qk_norm_rope_launch(rq, s));
    rk.qkv = kv_new; TDQ_TRY(td_qk_norm_rope_launch(rk, s));
    TdAttnParams ap;
    ap.Q = f->q; ap.ldq = QW; ap.K = kv_seq; ap.V = kv_seq + Hkv * 128; ap.ldkv = KVW; ap.O = f->attn; ap.ldo = QW;
    ap.batch = 1; ap.Sq = n; ap.Skv = pos0 + n; ap.Hq = Hq; ap.Hkv = Hkv; ap.scale = 0.08838834764831845f;
    ap.causal = 1; ap.causal_offset = pos0;
    TDQ_TRY(td_attn_launch(ap, s));
    {  // h += o_proj(attn)
      TdGemmParams g;
      g.A = f->attn; g.lda = QW; g.W = l.o_w; g.C = f->h; g.ldc = D; g.res = f->h; g.ldr = D; g.M = n; g.N = D; g.K = QW;
      TDQ_TRY(td_gemm_launch(g, s));
    }
    np.w = l.ln2_w;
    TDQ_TRY(td_norm_rows_launch(np, s));
    {  // gate | up, SwiGLU, down (+ residual)
      TdGemmParams g;
      g.A = f->xn; g.lda = D; g.W = l.gu_w; g.C = f->gu; g.ldc = 2 * I; g.M = n; g.N = 2 * I; g.K = D;
      TDQ_TRY(td_gemm_launch(g, s));
      TDQ_TRY(td_silu_mul_launch(f->gu, f->act, n, I, s));
      TdGemmParams d;
      d.A = f->act; d.lda = I; d.W = l.down_w; d.C = f->h; d.ldc = D; d.res = f->h; d.ldr = D; d.M = n; d.N = D; d.K = I;
      TDQ_TRY(td_gemm_launch(d, s));
    }
  }
  // model.norm -> captured embedding
  np.w = f->norm_w; np.y = f->xn;
  TDQ_TRY(td_norm_rows_launch(np, s));
  if (hidden_out) TD_CHECK_HIP(hipMemcpyAsync(hidden_out, f->xn, (size_t)n * D * 2, hipMemcpyDeviceToDevice, s));
  if (logits_last) {
    TdGemmParams g;
    g.A = f->xn + (size_t)(n - 1) * D; g.lda = D; g.W = f->lm_w; g.C = (bf16_t*)logits_last; g.ldc = f->cfg.vocab;
    g.M = 1; g.N = f->cfg.vocab; g.K = D;
    TDQ_TRY(td_gemm_launch(g, s));
  }
  return TD_OK;
}

// ---- batched KV-cached decode (the precompute job: many short sequences against one pass over the weights) --------------
int td_qwen2_forward(td_qwen2* f, const int* token_ids, const void* inputs_embeds, const int* position_ids, int n,
                     int pos0, void* hidden_out, void* logits_last, void* stream) {
  return td_qwen2_forward_slot(f, 0, token_ids, inputs_embeds, position_ids, n, pos0, hidden_out, logits_last, stream);
}

int td_qwen2_create_slots(const TdQwen2Config* cfg, int slot_len, int n_slots, td_qwen2** out) { return td_qwen2_create_ex(cfg, slot_len, n_slots, slot_len, out); }
int td_qwen2_create(const TdQwen2Config* cfg, int max_tokens, td_qwen2** out) { return td_qwen2_create_ex(cfg, max_tokens, 1, max_tokens, out); }

// re-partition the cache rows; a sequence cannot be longer than the activation workspace the handle was created with
int td_qwen2_set_slots(td_qwen2* f, int n_slots) {
  TD_CHECK_ARG(f && n_slots >= 1 && n_slots <= f->max_tokens, "td_qwen2_set_slots: bad slot count %d", n_slots);
  f->n_slots = n_slots;
  f->slot_len = f->max_tokens / n_slots < f->ws_rows ? f->max_tokens / n_slots : f->ws_rows;
  drop_step_graphs(f);      // the captured steps carry the old slot stride
  return TD_OK;
}

int td_qwen2_set_fused_rope(td_qwen2* f, int on) {
  TD_CHECK_ARG(f, "td_qwen2_set_fused_rope: null handle");
  const int prev = f->fused_rope ? 1 : 0;
  f->fused_rope = on != 0;
  drop_step_graphs(f);      // the captured steps carry the old launch list
  return prev;
}

int td_qwen2_slot_capacity(const td_qwen2* f) { return f ? f->slot_len : 0; }

int td_qwen2_move_slot(td_qwen2* f, int src, int dst, int len, void* stream) {
  TD_CHECK_ARG(f && src >= 0 && src < f->n_slots && dst >= 0 && dst < f->n_slots && len >= 0 && len <= f->slot_len, "td_qwen2_move_slot: bad arguments");
  if (src == dst || len == 0) return TD_OK;
  const size_t KVW = (size_t)2 * f->Hkv * 128;
  for (const QLayer& l : f->layers)
    TD_CHECK_HIP(hipMemcpyAsync(l.kv + (size_t)dst * f->slot_len * KVW, l.kv + (size_t)src * f->slot_len * KVW, (size_t)len * KVW * 2,
                                hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return TD_OK;
}

}  // extern "C"

namespace {
// The launches of one decode step for the sequences in slots 0 .. B-1: ids from tok_buf / pos_buf, lengths and cache rows from ibuf, the final
// hidden states left in xn and the logits in logits_buf.  Captured into a graph by td_qwen2_decode_batch (nothing here may depend on a host value
// other than B and the engine's own pointers; max_len only sizes the generic attention entry's bookkeeping, the decode kernel reads kv_lens).
int decode_step(td_qwen2* f, int B, int max_len, bool want_logits, hipStream_t s) {
  const int D = f->D, I = f->I, Hq = f->Hq, Hkv = f->Hkv;
  const int QW = Hq * 128, KVW = 2 * Hkv * 128;
  const int* kv_lens = f->ibuf;
  const int* row_off = f->ibuf + MAX_BATCH;
  const int* slot_ids = f->ibuf + 2 * MAX_BATCH;      // cache slot of each sequence of the step
  // Cross-over between the weight-stream kernels and the tile kernels, measured on both decoder shapes (profiles/r4z_decode_crossover.log): the 2B
  // shape (hidden 1536) stays ahead on the stream through 64 sequences (2.68 vs 2.79 ms per step), the 7B shape (hidden 3584) is level at 24 and
  // 18 % ahead on the tiles at 64 (5.44 vs 6.63 ms).  TD_QWEN2_STREAM_BATCH (>= 16): A/B.
  static const int env_stream_batch = getenv("TD_QWEN2_STREAM_BATCH") ? atoi(getenv("TD_QWEN2_STREAM_BATCH")) : 0;
  const int stream_batch = env_stream_batch > 0 ? (env_stream_batch < 16 ? 16 : env_stream_batch) : (D >= 3072 ? 32 : STREAM_BATCH);
  const bool wide = B > stream_batch;

  TDQ_TRY(td_embed_gather_launch(f->tok_buf, f->embed_w, f->h, B, D, f->cfg.vocab, s));
  TDQ_TRY(td_mrope_table_launch(f->pos_buf, B, f->cfg.mrope_section, f->cfg.rope_theta, 1, f->cosT, f->sinT, s));
  TdNormParams np;
  np.x = f->h; np.ldx = D; np.y = f->xn; np.ldy = D; np.rows = B; np.D = D; np.rms = 1; np.eps = f->cfg.rms_eps;
  for (int i = 0; i < f->cfg.num_layers; ++i) {
    const QLayer& l = f->layers[i];
    // (wide steps: the reduction launch of the Linear in front normalises the rows it finishes -- TdGemmParams::sk_norm_w -- so only layer 0 has a norm launch)
    if (!wide || i == 0) {
      np.w = l.ln1_w;
      TDQ_TRY(td_norm_rows_launch(np, s));
    }
    {
      TdGemmParams g;
      g.A = f->xn; g.lda = D; g.W = l.qkv_w; g.bias = l.qkv_b; g.M = B; g.N = QW + KVW; g.K = D;
      g.C = f->q; g.ldc = QW; g.C2 = f->kvtmp; g.ldc2 = KVW; g.n_split = QW;
      if (wide) { g.split_k = -1; g.sk_ws = f->sk_ws; g.sk_ws_bytes = SK_WS_BYTES; }      // (tile width and parts by the wide planner: n_split = Hq 128 fits its 64- / 128-column tiles)
      TDQ_TRY(td_gemm_launch(g, s));
    }
    // rotary embedding of the new q / k rows and the cache write ride inside the attention launch (TdAttnParams::dec_kv_new);
    // td_qwen2_set_fused_rope(f, 0) / TD_QWEN2_NO_FUSED_ROPE: the separate launch (A/B and the bit-identity test)
    static const bool env_unfused = getenv("TD_QWEN2_NO_FUSED_ROPE") != nullptr;
    const bool fused_rope = f->fused_rope && !env_unfused;
    if (!fused_rope) hipLaunchKernelGGL(td_decode_rope_scatter_kernel, dim3(B), dim3(256), 0, s, f->q, f->kvtmp, l.kv, row_off, f->cosT, f->sinT, Hq, Hkv);
    TdAttnParams ap;
    if (fused_rope) { ap.dec_kv_new = f->kvtmp; ap.dec_cos = f->cosT; ap.dec_sin = f->sinT; ap.dec_row_off = row_off; }
    ap.Q = f->q; ap.ldq = QW; ap.q_bstride = QW; ap.K = l.kv; ap.V = l.kv + Hkv * 128; ap.ldkv = KVW;
    ap.kv_bstride = (long long)f->slot_len * KVW; ap.O = f->attn; ap.ldo = QW; ap.o_bstride = QW;
    ap.batch = B; ap.Sq = 1; ap.Skv = max_len; ap.Hq = Hq; ap.Hkv = Hkv; ap.scale = 0.08838834764831845f;
    ap.causal = 1; ap.causal_offset = max_len - 1; ap.kv_lens = kv_lens; ap.dec_slots = slot_ids;
    TDQ_TRY(td_attn_launch(ap, s));
    {
      TdGemmParams g;
      g.A = f->attn; g.lda = QW; g.W = l.o_w; g.C = f->h; g.ldc = D; g.res = f->h; g.ldr = D; g.M = B; g.N = D; g.K = QW;
      if (wide) {
        g.split_k = -1; g.sk_ws = f->sk_ws; g.sk_ws_bytes = SK_WS_BYTES;
        g.sk_norm_w = l.ln2_w; g.sk_norm_out = f->xn; g.sk_norm_ld = D; g.sk_norm_eps = f->cfg.rms_eps;
      }
      TDQ_TRY(td_gemm_launch(g, s));
    }
    if (!wide) {
      np.w = l.ln2_w;
      TDQ_TRY(td_norm_rows_launch(np, s));
    }
    {
      TdGemmParams g;
      if (!wide) {
        g.A = f->xn; g.lda = D; g.W = l.gu_w; g.C = f->act; g.ldc = I; g.M = B; g.N = I; g.K = D; g.glu_I = I;   // gate | up, SiLU and product in one pass
        TDQ_TRY(td_gemm_launch(g, s));
      } else {      // the prefill's form: gate | up as one Linear, SiLU and product in a pass of their own (the same rounding points)
        g.A = f->xn; g.lda = D; g.W = l.gu_w; g.C = f->gu; g.ldc = 2 * I; g.M = B; g.N = 2 * I; g.K = D;
        g.split_k = -1; g.sk_ws = f->sk_ws; g.sk_ws_bytes = SK_WS_BYTES;
        TDQ_TRY(td_gemm_launch(g, s));
        TDQ_TRY(td_silu_mul_launch(f->gu, f->act, B, I, s));
      }
      TdGemmParams d;
      d.A = f->act; d.lda = I; d.W = l.down_w; d.C = f->h; d.ldc = D; d.res = f->h; d.ldr = D; d.M = B; d.N = D; d.K = I;
      if (wide) {
        d.split_k = -1; d.sk_ws = f->sk_ws; d.sk_ws_bytes = SK_WS_BYTES;
        d.sk_norm_w = i + 1 < f->cfg.num_layers ? f->layers[i + 1].ln1_w : f->norm_w;      // the next layer's input norm, or model.norm
        d.sk_norm_out = f->xn; d.sk_norm_ld = D; d.sk_norm_eps = f->cfg.rms_eps;
      }
      TDQ_TRY(td_gemm_launch(d, s));
    }
  }
  if (!wide) {
    np.w = f->norm_w; np.y = f->xn;
    TDQ_TRY(td_norm_rows_launch(np, s));
  }
  if (want_logits) {
    TdGemmParams g;
    g.A = f->xn; g.lda = D; g.W = f->lm_w; g.C = f->logits_buf; g.ldc = f->cfg.vocab; g.M = B; g.N = f->cfg.vocab; g.K = D;
    TDQ_TRY(td_gemm_launch(g, s));
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}
}  // namespace

extern "C" {

// One new token for each of the sequences in slots 0 .. B-1 (B <= 256): token_ids int32[B], position_ids int32[3,B] (device),
// cache_pos[b] = tokens already in slot b (HOST ints).  hidden_out bf16[B,hidden], logits bf16[B,vocab] (either may be NULL).
int td_qwen2_decode_batch(td_qwen2* f, int B, const int* token_ids, const int* position_ids, const int* cache_pos,
                          void* hidden_out, void* logits, void* stream) {
  return td_qwen2_decode_batch_slots(f, B, nullptr, token_ids, position_ids, cache_pos, hidden_out, logits, stream);
}

// ... for the sequences in cache slots slots[0 .. B-1] (HOST ints, distinct; NULL = 0 .. B-1): row b of the inputs and outputs belongs to slot slots[b].
// A finished sequence frees its slot with no cache rows moved, a new one is prefilled into any free slot (td_qwen2_prefill_packed_slots): the
// bookkeeping of continuous batching ([ext] vLLM's block tables, reduced to whole-sequence slots: 256 x 8192 rows fit the HBM outright).
int td_qwen2_decode_batch_slots(td_qwen2* f, int B, const int* slots, const int* token_ids, const int* position_ids, const int* cache_pos,
                                void* hidden_out, void* logits, void* stream) {
  TD_CHECK_ARG(f && token_ids && position_ids && cache_pos, "td_qwen2_decode_batch: null argument");
  TD_CHECK_ARG(B >= 1 && B <= MAX_BATCH && B <= f->n_slots && B <= f->ws_rows, "td_qwen2_decode_batch: batch %d exceeds min(%d, %d slots, %d workspace rows)", B, MAX_BATCH, f->n_slots, f->ws_rows);
  hipStream_t s = (hipStream_t)stream;
  const int D = f->D;
  IntPack ip;
  int max_len = 0;
  if (slots) {      // two rows on one slot would write the same cache row: refused, not left to corrupt a sequence
    int sorted[MAX_BATCH];
    std::copy(slots, slots + B, sorted);
    std::sort(sorted, sorted + B);
    TD_CHECK_ARG(std::adjacent_find(sorted, sorted + B) == sorted + B, "td_qwen2_decode_batch: cache slot %d is named by more than one row", *std::adjacent_find(sorted, sorted + B));
  }
  for (int b = 0; b < B; ++b) {
    const int slot = slots ? slots[b] : b;
    TD_CHECK_ARG(slot >= 0 && slot < f->n_slots, "td_qwen2_decode_batch: sequence %d names cache slot %d of %d", b, slot, f->n_slots);
    TD_CHECK_ARG(cache_pos[b] >= 0 && cache_pos[b] < f->slot_len, "td_qwen2_decode_batch: sequence %d is full (%d of %d)", b, cache_pos[b], f->slot_len);
    ip.v[b] = cache_pos[b] + 1;                                         // keys visible to the new token
    ip.v[MAX_BATCH + b] = slot * f->slot_len + cache_pos[b];            // its cache row (index; the kernels scale it by the row width)
    ip.v[2 * MAX_BATCH + b] = slot;
    max_len = cache_pos[b] + 1 > max_len ? cache_pos[b] + 1 : max_len;
  }
  hipLaunchKernelGGL(td_set_ints_kernel, dim3(1), dim3(3 * MAX_BATCH), 0, s, f->ibuf, ip, 3 * MAX_BATCH);
  TD_CHECK_LAUNCH();
  // the step reads its ids from fixed buffers and leaves its outputs in fixed buffers (the captured form needs stable addresses)
  TD_CHECK_HIP(hipMemcpyAsync(f->tok_buf, token_ids, (size_t)B * 4, hipMemcpyDeviceToDevice, s));
  TD_CHECK_HIP(hipMemcpyAsync(f->pos_buf, position_ids, (size_t)3 * B * 4, hipMemcpyDeviceToDevice, s));
  const bool want_logits = logits != nullptr;
  const int key = 2 * B + (want_logits ? 1 : 0);
  static const bool no_graph = getenv("TD_QWEN2_NO_GRAPH") != nullptr;
  auto it = f->step_graphs.find(key);
  if (it != f->step_graphs.end()) {
    TD_CHECK_HIP(hipGraphLaunch(it->second, s));
  } else if (no_graph || !f->graphs_ok || f->step_calls[key]++ == 0) {
    TDQ_TRY(decode_step(f, B, max_len, want_logits, s));
  } else {
    // capture on the engine's own stream (the caller's may be the legacy default stream, which cannot capture), after everything the caller
    // queued so far: nothing runs during a capture, but the eager fallback below must see the ids copied above
    if (!f->capture_stream) TD_CHECK_HIP(hipStreamCreateWithFlags(&f->capture_stream, hipStreamNonBlocking));
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(f->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const int rc = decode_step(f, B, max_len, want_logits, f->capture_stream);
      ok = hipStreamEndCapture(f->capture_stream, &graph) == hipSuccess && rc == TD_OK && graph != nullptr;
    }
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (graph) (void)hipGraphDestroy(graph);
    if (getenv("TD_QWEN2_GRAPH_DEBUG")) fprintf(stderr, "td_qwen2_decode_batch: step graph for B=%d logits=%d %s\n", B, (int)want_logits, ok ? "captured" : "NOT captured (eager from now on)");
    if (ok) {
      f->step_graphs[key] = exec;
      TD_CHECK_HIP(hipGraphLaunch(exec, s));
    } else {
      (void)hipGetLastError();
      f->graphs_ok = false;      // this runtime cannot capture the step: stay on the eager path for good
      TDQ_TRY(decode_step(f, B, max_len, want_logits, s));
    }
  }
  if (hidden_out) TD_CHECK_HIP(hipMemcpyAsync(hidden_out, f->xn, (size_t)B * D * 2, hipMemcpyDeviceToDevice, s));
  if (logits) TD_CHECK_HIP(hipMemcpyAsync(logits, f->logits_buf, (size_t)B * f->cfg.vocab * 2, hipMemcpyDeviceToDevice, s));
  return TD_OK;
}

// Prefill of B sequences in one pass: every sequence is right-padded to L tokens (row b * L + t; causal attention keeps the
// padding from influencing real tokens), sequence b goes to cache slot b.  inputs_embeds bf16[B*L, hidden] or token_ids
// int32[B*L]; position_ids int32[3, B*L]; lens[b] = real tokens of sequence b (HOST ints).  hidden_out bf16[B*L, hidden]
// (rows past lens[b] are meaningless), logits_last bf16[B, vocab] of each sequence's last real token (either may be NULL).
int td_qwen2_prefill_batch(td_qwen2* f, int B, int L, const int* token_ids, const void* inputs_embeds, const int* position_ids,
                           const int* lens, void* hidden_out, void* logits_last, void* stream) {
  return td_qwen2_prefill_batch_at(f, 0, B, L, token_ids, inputs_embeds, position_ids, lens, hidden_out, logits_last, stream);
}

// ... into cache slots slot0 .. slot0 + B - 1 (a request batch larger than the activation workspace is prefilled in several calls)
int td_qwen2_prefill_batch_at(td_qwen2* f, int slot0, int B, int L, const int* token_ids, const void* inputs_embeds, const int* position_ids,
                              const int* lens, void* hidden_out, void* logits_last, void* stream) {
  TD_CHECK_ARG(f && position_ids && lens && (token_ids || inputs_embeds), "td_qwen2_prefill_batch: null argument");
  TD_CHECK_ARG(slot0 >= 0 && B >= 1 && B <= MAX_BATCH && slot0 + B <= f->n_slots && L >= 1 && L <= f->slot_len && (long long)B * L <= f->ws_rows,
               "td_qwen2_prefill_batch: slots [%d, %d) x L=%d exceed the handle (slots %d x %d tokens, workspace %d rows)", slot0, slot0 + B, L, f->n_slots, f->slot_len, f->ws_rows);
  for (int b = 0; b < B; ++b) TD_CHECK_ARG(lens[b] >= 1 && lens[b] <= L, "td_qwen2_prefill_batch: sequence %d has %d of %d tokens", b, lens[b], L);
  hipStream_t s = (hipStream_t)stream;
  const int D = f->D, I = f->I, Hq = f->Hq, Hkv = f->Hkv;
  const int QW = Hq * 128, KVW = 2 * Hkv * 128, n = B * L;
  if (inputs_embeds) TD_CHECK_HIP(hipMemcpyAsync(f->h, inputs_embeds, (size_t)n * D * 2, hipMemcpyDeviceToDevice, s));
  else TDQ_TRY(td_embed_gather_launch(token_ids, f->embed_w, f->h, n, D, f->cfg.vocab, s));
  TDQ_TRY(td_mrope_table_launch(position_ids, n, f->cfg.mrope_section, f->cfg.rope_theta, 1, f->cosT, f->sinT, s));
  TdNormParams np;
  np.x = f->h; np.ldx = D; np.y = f->xn; np.ldy = D; np.rows = n; np.D = D; np.rms = 1; np.eps = f->cfg.rms_eps;
  TdQkRopeParams rq;
  rq.qkv = f->q; rq.ld = QW; rq.rows = n; rq.Hq = Hq; rq.Hk = 0; rq.q_col = 0; rq.k_col = 0;
  rq.cos = f->cosT; rq.sin = f->sinT; rq.rotate_half = 2;
  TdQkRopeParams rk = rq;
  rk.qkv = f->kvtmp; rk.ld = KVW; rk.Hq = Hkv;
  for (int i = 0; i < f->cfg.num_layers; ++i) {
    const QLayer& l = f->layers[i];
    np.w = l.ln1_w;
    TDQ_TRY(td_norm_rows_launch(np, s));
    {
      TdGemmParams g;
      g.A = f->xn; g.lda = D; g.W = l.qkv_w; g.bias = l.qkv_b; g.M = n; g.N = QW + KVW; g.K = D;
      g.C = f->q; g.ldc = QW; g.C2 = f->kvtmp; g.ldc2 = KVW; g.n_split = QW;
      if (QW % 256 == 0) {
        g.cfg = n <= 32 ? -1 : (td_gemm_config_id(n, QW + KVW, D) == 1 ? 1 : 0);
        TDQ_TRY(td_gemm_launch(g, s));
      } else {
        TdGemmParams a = g; a.C2 = nullptr; a.N = QW;
        TDQ_TRY(td_gemm_launch(a, s));
        TdGemmParams b2 = g; b2.C2 = nullptr; b2.W = l.qkv_w + (size_t)QW * D; b2.bias = l.qkv_b + QW; b2.N = KVW; b2.C = f->kvtmp; b2.ldc = KVW;
        TDQ_TRY(td_gemm_launch(b2, s));
      }
    }
    TDQ_TRY(td_qk_norm_rope_launch(rq, s));
    TDQ_TRY(td_qk_norm_rope_launch(rk, s));
    bf16_t* kv0 = l.kv + (size_t)slot0 * f->slot_len * KVW;      // first slot of this call
    hipLaunchKernelGGL(td_kv_rows_to_slots_kernel, dim3((KVW / 8 + 255) / 256, n), dim3(256), 0, s, f->kvtmp, kv0, L, f->slot_len, KVW);
    TdAttnParams ap;
    ap.Q = f->q; ap.ldq = QW; ap.q_bstride = (long long)L * QW; ap.K = kv0; ap.V = kv0 + Hkv * 128; ap.ldkv = KVW;
    ap.kv_bstride = (long long)f->slot_len * KVW; ap.O = f->attn; ap.ldo = QW; ap.o_bstride = (long long)L * QW;
    ap.batch = B; ap.Sq = L; ap.Skv = L; ap.Hq = Hq; ap.Hkv = Hkv; ap.scale = 0.08838834764831845f; ap.causal = 1; ap.causal_offset = 0;
    TDQ_TRY(td_attn_launch(ap, s));
    {
      TdGemmParams g;
      g.A = f->attn; g.lda = QW; g.W = l.o_w; g.C = f->h; g.ldc = D; g.res = f->h; g.ldr = D; g.M = n; g.N = D; g.K = QW;
      TDQ_TRY(td_gemm_launch(g, s));
    }
    np.w = l.ln2_w;
    TDQ_TRY(td_norm_rows_launch(np, s));
    {
      TdGemmParams g;
      g.A = f->xn; g.lda = D; g.W = l.gu_w; g.C = f->gu; g.ldc = 2 * I; g.M = n; g.N = 2 * I; g.K = D;
      TDQ_TRY(td_gemm_launch(g, s));
      TDQ_TRY(td_silu_mul_launch(f->gu, f->act, n, I, s));
      TdGemmParams d;
      d.A = f->act; d.lda = I; d.W = l.down_w; d.C = f->h; d.ldc = D; d.res = f->h; d.ldr = D; d.M = n; d.N = D; d.K = I;
      TDQ_TRY(td_gemm_launch(d, s));
    }
  }
  np.w = f->norm_w; np.y = f->xn;
  TDQ_TRY(td_norm_rows_launch(np, s));
  if (hidden_out) TD_CHECK_HIP(hipMemcpyAsync(hidden_out, f->xn, (size_t)n * D * 2, hipMemcpyDeviceToDevice, s));
  if (logits_last) {
    for (int b = 0; b < B; ++b)
      TD_CHECK_HIP(hipMemcpyAsync(f->lastrows + (size_t)b * D, f->xn + ((size_t)b * L + lens[b] - 1) * D, (size_t)D * 2, hipMemcpyDeviceToDevice, s));
    TdGemmParams g;
    g.A = f->lastrows; g.lda = D; g.W = f->lm_w; g.C = (bf16_t*)logits_last; g.ldc = f->cfg.vocab; g.M = B; g.N = f->cfg.vocab; g.K = D;
    TDQ_TRY(td_gemm_launch(g, s));
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}


// Packed prefill: the B prompts lie back to back (row offsets = the running sum of lens; no padding rows), sequence b goes to cache slot slot0 + b.
// What vLLM's scheduler does with the reference's request batches (max_num_batched_tokens rows per pass); against the padded form it saves the
// rows between each prompt and the longest.  total = sum(lens) rows: inputs_embeds bf16[total, hidden] or token_ids int32[total]; position_ids
// int32[3, total]; hidden_out bf16[total, hidden]; logits_last bf16[B, vocab] of each prompt's last token (either output may be NULL).
int td_qwen2_prefill_packed(td_qwen2* f, int slot0, int B, const int* token_ids, const void* inputs_embeds, const int* position_ids,
                            const int* lens, void* hidden_out, void* logits_last, void* stream) {
  TD_CHECK_ARG(slot0 >= 0 && B >= 1 && B <= MAX_BATCH, "td_qwen2_prefill_packed: bad slot range");
  int slots[MAX_BATCH];
  for (int b = 0; b < B; ++b) slots[b] = slot0 + b;
  return td_qwen2_prefill_packed_slots(f, B, slots, token_ids, inputs_embeds, position_ids, lens, hidden_out, logits_last, stream);
}

// ... sequence b into cache slot slots[b] (HOST ints, distinct, any order): the free slots of a running batch
int td_qwen2_prefill_packed_slots(td_qwen2* f, int B, const int* slots, const int* token_ids, const void* inputs_embeds, const int* position_ids,
                                  const int* lens, void* hidden_out, void* logits_last, void* stream) {
  TD_CHECK_ARG(f && slots && position_ids && lens && (token_ids || inputs_embeds), "td_qwen2_prefill_packed: null argument");
  TD_CHECK_ARG(B >= 1 && B <= MAX_BATCH, "td_qwen2_prefill_packed: %d sequences (1 .. %d)", B, MAX_BATCH);
  for (int b = 0; b < B; ++b) TD_CHECK_ARG(slots[b] >= 0 && slots[b] < f->n_slots, "td_qwen2_prefill_packed: sequence %d names cache slot %d of %d", b, slots[b], f->n_slots);
  {
    int sorted[MAX_BATCH];
    std::copy(slots, slots + B, sorted);
    std::sort(sorted, sorted + B);
    TD_CHECK_ARG(std::adjacent_find(sorted, sorted + B) == sorted + B, "td_qwen2_prefill_packed: cache slot %d is named by more than one sequence", *std::adjacent_find(sorted, sorted + B));
  }
  long long total = 0;
  int L = 0;
  for (int b = 0; b < B; ++b) {
    TD_CHECK_ARG(lens[b] >= 1 && lens[b] <= f->slot_len, "td_qwen2_prefill_packed: sequence %d has %d tokens (slot capacity %d)", b, lens[b], f->slot_len);
    total += lens[b];
    L = lens[b] > L ? lens[b] : L;
  }
  TD_CHECK_ARG(total <= f->ws_rows, "td_qwen2_prefill_packed: %lld packed rows exceed the %d workspace rows", total, f->ws_rows);
  hipStream_t s = (hipStream_t)stream;
  const int D = f->D, I = f->I, Hq = f->Hq, Hkv = f->Hkv;
  const int QW = Hq * 128, KVW = 2 * Hkv * 128, n = (int)total;
  // segment starts (device, for the attention and the last-row gather) and the cache row of every packed row
  IntPack ip;
  int* seg_starts = f->ibuf + 3 * MAX_BATCH;      // [B + 1] behind the decode step's lengths, rows and slots
  static_assert(MAX_BATCH + 1 <= 3 * MAX_BATCH, "seg_starts fit one IntPack");
  f->row_map_host.resize((size_t)n);
  int r = 0;
  for (int b = 0; b < B; ++b) {
    ip.v[b] = r;
    for (int t = 0; t < lens[b]; ++t) f->row_map_host[(size_t)r + t] = slots[b] * f->slot_len + t;
    r += lens[b];
  }
  ip.v[B] = r;
  hipLaunchKernelGGL(td_set_ints_kernel, dim3(1), dim3(3 * MAX_BATCH), 0, s, seg_starts, ip, B + 1);
  TD_CHECK_LAUNCH();
  TD_CHECK_HIP(hipMemcpyAsync(f->row_map, f->row_map_host.data(), (size_t)n * 4, hipMemcpyHostToDevice, s));
  if (inputs_embeds) TD_CHECK_HIP(hipMemcpyAsync(f->h, inputs_embeds, (size_t)n * D * 2, hipMemcpyDeviceToDevice, s));
  else TDQ_TRY(td_embed_gather_launch(token_ids, f->embed_w, f->h, n, D, f->cfg.vocab, s));
  TDQ_TRY(td_mrope_table_launch(position_ids, n, f->cfg.mrope_section, f->cfg.rope_theta, 1, f->cosT, f->sinT, s));
  TdNormParams np;
  np.x = f->h; np.ldx = D; np.y = f->xn; np.ldy = D; np.rows = n; np.D = D; np.rms = 1; np.eps = f->cfg.rms_eps;
  TdQkRopeParams rq;
  rq.qkv = f->q; rq.ld = QW; rq.rows = n; rq.Hq = Hq; rq.Hk = 0; rq.q_col = 0; rq.k_col = 0;
  rq.cos = f->cosT; rq.sin = f->sinT; rq.rotate_half = 2;
  TdQkRopeParams rk = rq;
  rk.qkv = f->kvtmp; rk.ld = KVW; rk.Hq = Hkv;
  for (int i = 0; i < f->cfg.num_layers; ++i) {
    const QLayer& l = f->layers[i];
    np.w = l.ln1_w;
    TDQ_TRY(td_norm_rows_launch(np, s));
    {
      TdGemmParams g;
      g.A = f->xn; g.lda = D; g.W = l.qkv_w; g.bias = l.qkv_b; g.M = n; g.N = QW + KVW; g.K = D;
      g.C = f->q; g.ldc = QW; g.C2 = f->kvtmp; g.ldc2 = KVW; g.n_split = QW;
      if (QW % 256 == 0) {
        g.cfg = n <= 32 ? -1 : (td_gemm_config_id(n, QW + KVW, D) == 1 ? 1 : 0);
        TDQ_TRY(td_gemm_launch(g, s));
      } else {
        TdGemmParams a = g; a.C2 = nullptr; a.N = QW;
        TDQ_TRY(td_gemm_launch(a, s));
        TdGemmParams b2 = g; b2.C2 = nullptr; b2.W = l.qkv_w + (size_t)QW * D; b2.bias = l.qkv_b + QW; b2.N = KVW; b2.C = f->kvtmp; b2.ldc = KVW;
        TDQ_TRY(td_gemm_launch(b2, s));
      }
    }
    TDQ_TRY(td_qk_norm_rope_launch(rq, s));
    TDQ_TRY(td_qk_norm_rope_launch(rk, s));
    hipLaunchKernelGGL(td_kv_rows_to_rows_kernel, dim3((KVW / 8 + 255) / 256, n), dim3(256), 0, s, f->kvtmp, l.kv, f->row_map, KVW);
    TdAttnParams ap;      // causal attention inside each packed prompt, straight from the projection rows (a prefill starts at position 0)
    ap.Q = f->q; ap.ldq = QW; ap.K = f->kvtmp; ap.V = f->kvtmp + Hkv * 128; ap.ldkv = KVW; ap.O = f->attn; ap.ldo = QW;
    ap.batch = B; ap.Sq = L; ap.Skv = L; ap.Hq = Hq; ap.Hkv = Hkv; ap.scale = 0.08838834764831845f; ap.causal = 1; ap.causal_offset = 0;
    ap.seg_starts = seg_starts;
    TDQ_TRY(td_attn_launch(ap, s));
    {
      TdGemmParams g;
      g.A = f->attn; g.lda = QW; g.W = l.o_w; g.C = f->h; g.ldc = D; g.res = f->h; g.ldr = D; g.M = n; g.N = D; g.K = QW;
      TDQ_TRY(td_gemm_launch(g, s));
    }
    np.w = l.ln2_w;
    TDQ_TRY(td_norm_rows_launch(np, s));
    {
      TdGemmParams g;
      g.A = f->xn; g.lda = D; g.W = l.gu_w; g.C = f->gu; g.ldc = 2 * I; g.M = n; g.N = 2 * I; g.K = D;
      TDQ_TRY(td_gemm_launch(g, s));
      TDQ_TRY(td_silu_mul_launch(f->gu, f->act, n, I, s));
      TdGemmParams d;
      d.A = f->act; d.lda = I; d.W = l.down_w; d.C = f->h; d.ldc = D; d.res = f->h; d.ldr = D; d.M = n; d.N = D; d.K = I;
      TDQ_TRY(td_gemm_launch(d, s));
    }
  }
  np.w = f->norm_w; np.y = f->xn;
  TDQ_TRY(td_norm_rows_launch(np, s));
  if (hidden_out) TD_CHECK_HIP(hipMemcpyAsync(hidden_out, f->xn, (size_t)n * D * 2, hipMemcpyDeviceToDevice, s));
  if (logits_last) {
    hipLaunchKernelGGL(td_gather_last_rows_kernel, dim3((D / 8 + 255) / 256, B), dim3(256), 0, s, f->xn, f->lastrows, seg_starts, D);
    TdGemmParams g;
    g.A = f->lastrows; g.lda = D; g.W = f->lm_w; g.C = (bf16_t*)logits_last; g.ldc = f->cfg.vocab; g.M = B; g.N = f->cfg.vocab; g.K = D;
    TDQ_TRY(td_gemm_launch(g, s));
  }
  TD_CHECK_LAUNCH();
  return TD_OK;
}

}  // extern "C"

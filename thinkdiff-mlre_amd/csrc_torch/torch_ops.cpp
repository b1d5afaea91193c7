// torch.ops.thinkdiff_hip.* -- the custom-op layer of the MI355X hot path (last row), registered from a
// shared library (lib/libthinkdiff_torch_ops.so, loaded by thinkdiff/ops.py with torch.ops.load_library).
//
// Each op is a schema plus ONE kernel, registered for the "CUDA" (= HIP on ROCm) dispatch key, that hands raw device pointers
// and sizes to the C ABI of libthinkdiff_hip.so (include/thinkdiff_hip.h).  Conventions: tensors are borrowed (caller owns,
// device-resident, innermost stride 1), outputs come from the PyTorch caching allocator on the current HIP stream, nothing
// synchronises, a rejected argument is a TORCH_CHECK failure (Python RuntimeError) carrying td_last_error().  There is no CPU,
// Meta or composite kernel: a call with only host tensors fails in the dispatcher.  The dispatcher picks this kernel as soon as
// ANY argument is on the GPU, so every tensor argument -- the optional ones too -- is checked here for device (all on x's), dtype,
// contiguity and extent before its pointer goes to the C ABI; each op also makes x's device current for its duration (the
// launchers read hipGetDevice() for per-device attributes and their stream-K / split-K workspaces).
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/thinkdiff_hip.h"

namespace {

void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }
const void* P(const c10::optional<at::Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }

void check_rows(const at::Tensor& t, const char* name, at::ScalarType ty = at::kBFloat16) {
  TORCH_CHECK(t.is_cuda(), "thinkdiff_hip: ", name, " must live on the GPU");
  TORCH_CHECK(t.scalar_type() == ty, "thinkdiff_hip: ", name, " has dtype ", t.scalar_type(), ", expected ", ty);
  TORCH_CHECK(t.dim() >= 1 && t.stride(-1) == 1, "thinkdiff_hip: ", name, " needs innermost stride 1");
}
// a per-column / per-row operand: on `like`'s device, dtype `ty`, contiguous, exactly `numel` elements
void check_vec(const at::Tensor& t, const char* name, const at::Tensor& like, int64_t numel, at::ScalarType ty = at::kBFloat16) {
  check_rows(t, name, ty);
  TORCH_CHECK(t.device() == like.device(), "thinkdiff_hip: ", name, " lives on ", t.device(), ", expected ", like.device());
  TORCH_CHECK(t.is_contiguous() && t.numel() == numel, "thinkdiff_hip: ", name, " needs ", numel, " contiguous elements, got ", t.numel());
}
void check_vec(const c10::optional<at::Tensor>& t, const char* name, const at::Tensor& like, int64_t numel, at::ScalarType ty = at::kBFloat16) {
  if (t.has_value() && t->defined()) check_vec(*t, name, like, numel, ty);
}
void same_device(const at::Tensor& t, const char* name, const at::Tensor& like) {
  TORCH_CHECK(t.device() == like.device(), "thinkdiff_hip: ", name, " lives on ", t.device(), ", expected ", like.device());
}
using DeviceGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;
void ok(int rc) { TORCH_CHECK(rc == TD_OK, "libthinkdiff_hip error ", rc, ": ", td_last_error()); }

// y = act(x . w^T + bias) * gate + res      (nn.Linear and its fused neighbours)
at::Tensor linear(const at::Tensor& x, const at::Tensor& w, const c10::optional<at::Tensor>& bias, int64_t act,
                  const c10::optional<at::Tensor>& gate, const c10::optional<at::Tensor>& res) {
  check_rows(x, "x"); check_rows(w, "w"); same_device(w, "w", x);
  TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && w.is_contiguous() && w.size(1) == x.size(1), "thinkdiff_hip::linear: x [M,K], w [N,K] contiguous");
  check_vec(bias, "bias", x, w.size(0)); check_vec(gate, "gate", x, w.size(0));
  if (res.has_value() && res->defined()) {
    check_rows(*res, "res"); same_device(*res, "res", x);
    TORCH_CHECK(res->dim() == 2 && res->size(0) == x.size(0) && res->size(1) == w.size(0), "thinkdiff_hip::linear: res must be [M,N]");
  }
  DeviceGuard guard(x.device());
  at::Tensor y = at::empty({x.size(0), w.size(0)}, x.options());
  const int64_t ldr = res.has_value() && res->defined() ? res->stride(0) : 0;
  ok(td_linear_bf16(x.data_ptr(), x.stride(0), w.data_ptr(), P(bias), y.data_ptr(), y.stride(0), (int)x.size(0), (int)w.size(0),
                    (int)x.size(1), (int)act, P(gate), P(res), ldr, stream_of(x)));
  return y;
}

// ThinkDiff aligner mm_projector "mlp2x_gelu_t5_norm": T5LayerNorm(Linear2(GELU_erf(Linear0(x))))
at::Tensor aligner_mlp2x(const at::Tensor& x, const at::Tensor& w0, const at::Tensor& b0, const at::Tensor& w2, const at::Tensor& b2,
                         const at::Tensor& norm_w, double eps, bool fp32_norm) {
  check_rows(x, "x"); check_rows(w0, "w0"); check_rows(w2, "w2");
  TORCH_CHECK(x.dim() == 2 && w0.dim() == 2 && w2.dim() == 2, "thinkdiff_hip::aligner_mlp2x: x [M,K], w0 [H,K], w2 [H,H]");
  const int64_t M = x.size(0), K = x.size(1), H = w0.size(0);
  check_vec(w0, "w0", x, H * K); check_vec(w2, "w2", x, H * H);
  TORCH_CHECK(w0.size(1) == K && w2.size(0) == H && w2.size(1) == H, "thinkdiff_hip::aligner_mlp2x: w0 must be [H,K] and w2 [H,H]");
  check_vec(b0, "b0", x, H); check_vec(b2, "b2", x, H); check_vec(norm_w, "norm_w", x, H);
  DeviceGuard guard(x.device());
  at::Tensor ws = at::empty({2 * M * H}, x.options());
  at::Tensor y = at::empty({M, H}, x.options());
  ok(td_aligner_mlp2x_bf16(x.data_ptr(), x.stride(0), (int)M, (int)K, (int)H, w0.data_ptr(), b0.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                           norm_w.data_ptr(), (float)eps, fp32_norm ? 1 : 0, ws.data_ptr(), y.data_ptr(), H, stream_of(x)));
  return y;
}

// softmax(q.k^T * scale [+ causal mask]) . v on token-major fused projections: q [B,Sq,>=Hq*128], k/v [B,Skv,>=Hkv*128]
at::Tensor attention(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, int64_t Hq, int64_t Hkv, double scale, bool causal) {
  check_rows(q, "q"); check_rows(k, "k"); check_rows(v, "v");
  TORCH_CHECK(q.dim() == 3 && k.dim() == 3 && v.dim() == 3 && k.strides() == v.strides(), "thinkdiff_hip::attention: q/k/v [B,S,cols], k and v with equal strides");
  same_device(k, "k", q); same_device(v, "v", q);
  TORCH_CHECK(Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "thinkdiff_hip::attention: Hq must be a positive multiple of Hkv");
  TORCH_CHECK(q.size(2) >= Hq * 128 && k.size(2) >= Hkv * 128 && v.size(2) >= Hkv * 128,
              "thinkdiff_hip::attention: q needs >= Hq*128 columns and k/v >= Hkv*128 (got ", q.size(2), ", ", k.size(2), ", ", v.size(2), ")");
  TORCH_CHECK(k.size(0) == q.size(0) && v.size(0) == q.size(0) && v.size(1) == k.size(1), "thinkdiff_hip::attention: q/k/v batch sizes and k/v lengths must agree");
  DeviceGuard guard(q.device());
  at::Tensor o = at::empty({q.size(0), q.size(1), Hq * 128}, q.options());
  ok(td_attention_bf16(q.data_ptr(), q.stride(1), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(1), k.stride(0), o.data_ptr(),
                       o.stride(1), o.stride(0), (int)q.size(0), (int)q.size(1), (int)k.size(1), (int)Hq, (int)Hkv, 128, (float)scale,
                       causal ? 1 : 0, stream_of(q)));
  return o;
}

// LayerNorm (no affine) / RMSNorm rows with optional adaLN modulation y*(1+scale)+shift (set A for rows < split, set B after)
at::Tensor norm_rows(const at::Tensor& x, bool rms, double eps, const c10::optional<at::Tensor>& w, int64_t split,
                     const c10::optional<at::Tensor>& shiftA, const c10::optional<at::Tensor>& scaleA,
                     const c10::optional<at::Tensor>& shiftB, const c10::optional<at::Tensor>& scaleB) {
  check_rows(x, "x");
  TORCH_CHECK(x.dim() == 2, "thinkdiff_hip::norm_rows: x [rows,D]");
  const int64_t Dn = x.size(1);
  check_vec(w, "w", x, Dn); check_vec(shiftA, "shiftA", x, Dn); check_vec(scaleA, "scaleA", x, Dn);
  check_vec(shiftB, "shiftB", x, Dn); check_vec(scaleB, "scaleB", x, Dn);
  TORCH_CHECK(split >= 0 && split <= x.size(0), "thinkdiff_hip::norm_rows: split outside [0, rows]");
  DeviceGuard guard(x.device());
  at::Tensor y = at::empty_like(x);
  ok(td_norm_rows_bf16(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), (int)x.size(0), (int)x.size(1), rms ? 1 : 0, (float)eps, P(w),
                       (int)split, P(shiftA), P(scaleA), P(shiftB), P(scaleB), stream_of(x)));
  return y;
}

// in-place per-head RMSNorm(q), RMSNorm(k) + rotary embedding on a fused projection buffer
at::Tensor& qk_norm_rope_(at::Tensor& qkv, int64_t Hq, int64_t Hk, int64_t q_col, int64_t k_col, const at::Tensor& cos, const at::Tensor& sin,
                          int64_t split, const c10::optional<at::Tensor>& wqA, const c10::optional<at::Tensor>& wkA,
                          const c10::optional<at::Tensor>& wqB, const c10::optional<at::Tensor>& wkB, double eps, bool rotate_half) {
  check_rows(qkv, "qkv"); check_rows(cos, "cos", at::kFloat); check_rows(sin, "sin", at::kFloat);
  TORCH_CHECK(qkv.dim() == 2 && cos.is_contiguous() && sin.is_contiguous() && cos.size(0) == qkv.size(0) && cos.size(1) == 128,
              "thinkdiff_hip::qk_norm_rope_: qkv [rows,cols], cos/sin fp32 [rows,128]");
  check_vec(cos, "cos", qkv, qkv.size(0) * 128, at::kFloat); check_vec(sin, "sin", qkv, qkv.size(0) * 128, at::kFloat);
  check_vec(wqA, "wqA", qkv, 128); check_vec(wkA, "wkA", qkv, 128); check_vec(wqB, "wqB", qkv, 128); check_vec(wkB, "wkB", qkv, 128);
  TORCH_CHECK(Hq >= 0 && Hk >= 0 && q_col >= 0 && k_col >= 0 && q_col + Hq * 128 <= qkv.size(1) && k_col + Hk * 128 <= qkv.size(1),
              "thinkdiff_hip::qk_norm_rope_: the q / k head blocks must lie inside a row of qkv");
  TORCH_CHECK(split >= 0 && split <= qkv.size(0), "thinkdiff_hip::qk_norm_rope_: split outside [0, rows]");
  DeviceGuard guard(qkv.device());
  ok(td_qk_norm_rope_bf16(qkv.data_ptr(), qkv.stride(0), (int)qkv.size(0), (int)Hq, (int)Hk, (int)q_col, (int)k_col, (const float*)cos.data_ptr(),
                          (const float*)sin.data_ptr(), (int)split, P(wqA), P(wkA), wqB.has_value() && wqB->defined() ? P(wqB) : P(wqA),
                          wkB.has_value() && wkB->defined() ? P(wkB) : P(wkA), (float)eps, rotate_half ? 1 : 0, stream_of(qkv)));
  return qkv;
}

// FlowMatchEulerDiscreteScheduler.step in place: x = bf16(float(x) + dt * float(v))
at::Tensor& euler_step_(at::Tensor& x, const at::Tensor& v, double dt) {
  check_rows(x, "x"); check_rows(v, "v");
  TORCH_CHECK(x.is_contiguous() && v.is_contiguous() && x.numel() == v.numel(), "thinkdiff_hip::euler_step_: contiguous x, v of equal size");
  same_device(v, "v", x);
  DeviceGuard guard(x.device());
  ok(td_euler_step_bf16(x.data_ptr(), v.data_ptr(), (float)dt, x.numel(), stream_of(x)));
  return x;
}

at::Tensor flux_pack_latents(const at::Tensor& latents) {            // [C,H,W] -> [(H/2)(W/2), 4C]
  check_rows(latents, "latents");
  TORCH_CHECK(latents.dim() == 3 && latents.is_contiguous(), "thinkdiff_hip::flux_pack_latents: contiguous [C,H,W]");
  const int64_t C = latents.size(0), H = latents.size(1), W = latents.size(2);
  DeviceGuard guard(latents.device());
  at::Tensor out = at::empty({(H / 2) * (W / 2), C * 4}, latents.options());
  ok(td_flux_pack_latents(latents.data_ptr(), out.data_ptr(), (int)C, (int)H, (int)W, 0, 1.0f, 0.0f, stream_of(latents)));
  return out;
}

at::Tensor flux_unpack_latents(const at::Tensor& packed, int64_t C, int64_t H, int64_t W, double div, double add) {   // bf16(bf16(x / div) + add)
  check_rows(packed, "packed");
  TORCH_CHECK(packed.is_contiguous() && packed.numel() == C * H * W, "thinkdiff_hip::flux_unpack_latents: contiguous [(H/2)(W/2), 4C]");
  DeviceGuard guard(packed.device());
  at::Tensor out = at::empty({C, H, W}, packed.options());
  ok(td_flux_pack_latents(packed.data_ptr(), out.data_ptr(), (int)C, (int)H, (int)W, 1, (float)div, (float)add, stream_of(packed)));
  return out;
}

at::Tensor cls_avgpool2(const at::Tensor& tokens) {                  // [1+G*G, C] -> [1+(G/2)^2, C]
  check_rows(tokens, "tokens");
  TORCH_CHECK(tokens.dim() == 2 && tokens.is_contiguous(), "thinkdiff_hip::cls_avgpool2: contiguous [1+G*G, C]");
  int64_t G = 0;
  while ((G + 1) * (G + 1) <= tokens.size(0) - 1) ++G;
  TORCH_CHECK(G * G == tokens.size(0) - 1, "thinkdiff_hip::cls_avgpool2: token count must be 1 + G*G");
  DeviceGuard guard(tokens.device());
  at::Tensor out = at::empty({1 + (G / 2) * (G / 2), tokens.size(1)}, tokens.options());
  ok(td_cls_avgpool2_bf16(tokens.data_ptr(), out.data_ptr(), (int)G, (int)tokens.size(1), stream_of(tokens)));
  return out;
}

// one token per row of bf16 logits: temperature / top-p, drawn from (seed, offset, row); temperature <= 0: greedy
at::Tensor sample_top_p(const at::Tensor& logits, double temperature, double top_p, int64_t seed, int64_t offset) {
  check_rows(logits, "logits");
  at::Tensor x = logits.dim() == 1 ? logits.unsqueeze(0) : logits;
  TORCH_CHECK(x.dim() == 2, "thinkdiff_hip::sample_top_p: logits [rows, vocab]");
  DeviceGuard guard(x.device());
  at::Tensor out = at::empty({x.size(0)}, x.options().dtype(at::kInt));
  ok(td_sample_top_p_bf16(x.data_ptr(), x.stride(0), (int)x.size(0), (int)x.size(1), (float)temperature, (float)top_p, (uint64_t)seed,
                          (uint64_t)offset, (int32_t*)out.data_ptr(), stream_of(x)));
  return out;
}

// ---- the engines: the denoise loop itself goes through the dispatcher ----------------------------------------------------------------
// An engine is a stateful C++ object behind the C ABI (weights, workspaces, prepared conditioning); the ops take its handle as an int (the
// td_flux* / td_vae* value, as thinkdiff.models.* hold it) and check every tensor against what the prepared context expects.
td_flux* flux_of(int64_t h) {
  TORCH_CHECK(h != 0, "thinkdiff_hip: null FLUX engine handle");
  return (td_flux*)(uintptr_t)h;
}
void check_latents(td_flux* f, const at::Tensor& x, const char* name) {
  int si = 0, c = 0, n = 0;
  ok(td_flux_prepared_shape(f, &si, nullptr, &c, &n));
  check_rows(x, name);
  TORCH_CHECK(si > 0 && n > 0, "thinkdiff_hip: the FLUX context has no condition / timesteps prepared");
  TORCH_CHECK(x.dim() == 2 && x.is_contiguous() && x.size(0) == si && x.size(1) == c, "thinkdiff_hip: ", name, " must be contiguous [", si, ", ", c, "] bf16, got ", x.sizes());
}
// velocity = FluxTransformer2DModel.forward(latents) at prepared step `step`
at::Tensor& flux_forward_(int64_t engine, const at::Tensor& latents, int64_t step, at::Tensor& velocity) {
  td_flux* f = flux_of(engine);
  check_latents(f, latents, "latents"); check_latents(f, velocity, "velocity"); same_device(velocity, "velocity", latents);
  DeviceGuard guard(latents.device());
  ok(td_flux_forward(f, latents.data_ptr(), (int)step, velocity.data_ptr(), stream_of(latents)));
  return velocity;
}
// the whole Euler flow-matching loop in place: FluxPipeline.__call__'s denoising loop (transformer + scheduler.step per sigma)
at::Tensor& flux_denoise_(int64_t engine, at::Tensor& latents, at::ArrayRef<double> sigmas) {
  td_flux* f = flux_of(engine);
  check_latents(f, latents, "latents");
  TORCH_CHECK(sigmas.size() >= 2, "thinkdiff_hip::flux_denoise_: sigmas needs n + 1 >= 2 entries");
  std::vector<float> sg(sigmas.begin(), sigmas.end());
  DeviceGuard guard(latents.device());
  ok(td_flux_denoise(f, latents.data_ptr(), sg.data(), (int)sg.size() - 1, stream_of(latents)));
  return latents;
}
// the same for several prepared contexts (a parent and its forks) at once, context k on streams[k] (hipStream_t values)
void flux_denoise_multi_(at::ArrayRef<int64_t> engines, at::TensorList latents, at::ArrayRef<double> sigmas, at::ArrayRef<int64_t> streams) {
  TORCH_CHECK(!engines.empty() && engines.size() == latents.size() && engines.size() == streams.size(), "thinkdiff_hip::flux_denoise_multi_: one latent tensor and one stream per engine");
  TORCH_CHECK(sigmas.size() >= 2, "thinkdiff_hip::flux_denoise_multi_: sigmas needs n + 1 >= 2 entries");
  std::vector<td_flux*> fs; std::vector<void*> ls, ss;
  for (size_t k = 0; k < engines.size(); ++k) {
    fs.push_back(flux_of(engines[k]));
    check_latents(fs.back(), latents[k], "latents[k]"); same_device(latents[k], "latents[k]", latents[0]);
    ls.push_back(latents[k].data_ptr()); ss.push_back((void*)(uintptr_t)streams[k]);
  }
  std::vector<float> sg(sigmas.begin(), sigmas.end());
  DeviceGuard guard(latents[0].device());
  ok(td_flux_denoise_multi(fs.data(), ls.data(), (int)fs.size(), sg.data(), (int)sg.size() - 1, ss.data()));
}
// AutoencoderKL.decode + VaeImageProcessor.postprocess: packed latents [(h/2)(w/2), 4C] -> uint8 [H, W, 3] (8h x 8w for the FLUX.1 VAE)
at::Tensor vae_decode_u8(int64_t engine, const at::Tensor& packed, int64_t h, int64_t w, double scaling_factor, double shift_factor) {
  TORCH_CHECK(engine != 0, "thinkdiff_hip: null VAE engine handle");
  check_rows(packed, "packed");
  int H = 0, W = 0, pc = 0;
  TORCH_CHECK(h > 0 && w > 0 && h % 2 == 0 && w % 2 == 0, "thinkdiff_hip::vae_decode_u8: even latent height and width");
  ok(td_vae_output_shape((const td_vae*)(uintptr_t)engine, (int)h, (int)w, &H, &W, &pc));
  TORCH_CHECK(packed.dim() == 2 && packed.is_contiguous() && packed.size(0) == (h / 2) * (w / 2) && packed.size(1) == pc,
              "thinkdiff_hip::vae_decode_u8: packed must be contiguous [(h/2)(w/2) = ", (h / 2) * (w / 2), ", ", pc, "] bf16, got ", packed.sizes());
  DeviceGuard guard(packed.device());
  at::Tensor img = at::empty({H, W, 3}, packed.options().dtype(at::kByte));
  ok(td_vae_decode((td_vae*)(uintptr_t)engine, packed.data_ptr(), (int)h, (int)w, (float)scaling_factor, (float)shift_factor, img.data_ptr(), nullptr, stream_of(packed)));
  return img;
}
// the joint attention with QK^T and P.V on the e4m3 MFMA (td_attention_fp8): q, k, v [S, >= H*128] bf16 views
at::Tensor attention_fp8(const at::Tensor& q, const at::Tensor& k, const at::Tensor& v, int64_t H, double scale) {
  check_rows(q, "q"); check_rows(k, "k"); check_rows(v, "v"); same_device(k, "k", q); same_device(v, "v", q);
  TORCH_CHECK(q.dim() == 2 && k.dim() == 2 && v.dim() == 2 && H > 0 && q.size(1) >= H * 128 && k.size(1) >= H * 128 && v.size(1) >= H * 128 && k.size(0) == v.size(0) &&
              k.stride(0) == v.stride(0), "thinkdiff_hip::attention_fp8: q [Sq, >= H*128], k / v [Skv, >= H*128] with equal row strides");
  DeviceGuard guard(q.device());
  at::Tensor out = at::empty({q.size(0), H * 128}, q.options());
  at::Tensor ws = at::empty({(int64_t)td_attention_fp8_workspace_bytes((int)q.size(0), (int)k.size(0), (int)H)}, q.options().dtype(at::kByte));
  ok(td_attention_fp8(q.data_ptr(), q.stride(0), k.data_ptr(), v.data_ptr(), k.stride(0), out.data_ptr(), out.stride(0), (int)q.size(0), (int)k.size(0), (int)H,
                      (float)scale, ws.data_ptr(), stream_of(q)));
  return out;
}

}  // namespace

TORCH_LIBRARY(thinkdiff_hip, m) {
  m.def("linear(Tensor x, Tensor w, Tensor? bias, int act, Tensor? gate, Tensor? res) -> Tensor");
  m.def("aligner_mlp2x(Tensor x, Tensor w0, Tensor b0, Tensor w2, Tensor b2, Tensor norm_w, float eps, bool fp32_norm) -> Tensor");
  m.def("attention(Tensor q, Tensor k, Tensor v, int Hq, int Hkv, float scale, bool causal) -> Tensor");
  m.def("norm_rows(Tensor x, bool rms, float eps, Tensor? w, int split, Tensor? shiftA, Tensor? scaleA, Tensor? shiftB, Tensor? scaleB) -> Tensor");
  m.def("qk_norm_rope_(Tensor(a!) qkv, int Hq, int Hk, int q_col, int k_col, Tensor cos, Tensor sin, int split, Tensor? wqA, Tensor? wkA, Tensor? wqB, Tensor? wkB, float eps, bool rotate_half) -> Tensor(a!)");
  m.def("euler_step_(Tensor(a!) x, Tensor v, float dt) -> Tensor(a!)");
  m.def("flux_pack_latents(Tensor latents) -> Tensor");
  m.def("flux_unpack_latents(Tensor packed, int C, int H, int W, float div, float add) -> Tensor");
  m.def("cls_avgpool2(Tensor tokens) -> Tensor");
  m.def("sample_top_p(Tensor logits, float temperature, float top_p, int seed, int offset) -> Tensor");
  m.def("flux_forward_(int engine, Tensor latents, int step, Tensor(a!) velocity) -> Tensor(a!)");
  m.def("flux_denoise_(int engine, Tensor(a!) latents, float[] sigmas) -> Tensor(a!)");
  m.def("flux_denoise_multi_(int[] engines, Tensor(a!)[] latents, float[] sigmas, int[] streams) -> ()");
  m.def("vae_decode_u8(int engine, Tensor packed, int h, int w, float scaling_factor, float shift_factor) -> Tensor");
  m.def("attention_fp8(Tensor q, Tensor k, Tensor v, int H, float scale) -> Tensor");
}

TORCH_LIBRARY_IMPL(thinkdiff_hip, CUDA, m) {
  m.impl("linear", &linear);
  m.impl("aligner_mlp2x", &aligner_mlp2x);
  m.impl("attention", &attention);
  m.impl("norm_rows", &norm_rows);
  m.impl("qk_norm_rope_", &qk_norm_rope_);
  m.impl("euler_step_", &euler_step_);
  m.impl("flux_pack_latents", &flux_pack_latents);
  m.impl("flux_unpack_latents", &flux_unpack_latents);
  m.impl("cls_avgpool2", &cls_avgpool2);
  m.impl("sample_top_p", &sample_top_p);
  m.impl("flux_forward_", &flux_forward_);
  m.impl("flux_denoise_", &flux_denoise_);
  m.impl("flux_denoise_multi_", &flux_denoise_multi_);
  m.impl("vae_decode_u8", &vae_decode_u8);
  m.impl("attention_fp8", &attention_fp8);
}

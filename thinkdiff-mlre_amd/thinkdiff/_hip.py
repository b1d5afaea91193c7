"""ctypes binding of libthinkdiff_hip.so (the C ABI in include/thinkdiff_hip.h).

torch is used here only as the owner of device memory and streams: every call passes raw device
pointers + sizes to the C ABI.  There is deliberately NO fallback: if the library is missing the
import of any op raises, so a GPU run can never silently take an eager/CPU path.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TD_HIP_LIB: another build of the same library (A/B timing of two builds on one box); never a different implementation
LIB_PATH = os.environ.get("TD_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libthinkdiff_hip.so")

ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_SILU, ACT_QUICK_GELU = 0, 1, 2, 3, 4

_lib = None


class ThinkDiffHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise loudly when the .so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ThinkDiffHipError(
                f"{LIB_PATH} not found: build it with `make -C thinkdiff-mlre_amd` "
                "(or __graft_entry__.build()); there is no fallback path")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.td_last_error.restype = ctypes.c_char_p
        _declare(_lib)
    return _lib


def _declare(L):
    vp, i64, i32, f32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float
    sig = {
        "td_abi_version": [],
        "td_linear_bf16": [vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, i64, vp],
        "td_linear_split_bf16": [vp, i64, vp, vp, vp, i64, i32, vp, i64, i32, i32, i32, i32, i32, vp],
        "td_linear_splitk_bf16": [vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp, i64, i32, i32, vp, vp, i64, f32, vp],
        "td_linear_grouped2_bf16": [vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i64, i64, i64, i32, i32, i32, i32, vp],
        "td_norm_rows_bf16": [vp, i64, vp, i64, i32, i32, i32, f32, vp, i32, vp, vp, vp, vp, vp],
        "td_qk_norm_rope_bf16": [vp, i64, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, f32, i32, vp],
        "td_flux_rope_table": [vp, i32, vp, ctypes.c_double, vp, vp, vp],
        "td_timestep_sincos": [vp, i32, vp, vp],
        "td_euler_step_bf16": [vp, vp, f32, i64, vp],
        "td_flux_pack_latents": [vp, vp, i32, i32, i32, i32, f32, f32, vp],
        "td_cls_avgpool2_bf16": [vp, vp, i32, i32, vp],
        "td_fill_normal_bf16": [vp, i64, ctypes.c_uint64, f32, f32, vp],
        "td_aligner_mlp2x_bf16": [vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, f32, i32, vp, vp, i64, vp],
        "td_flux_create": [vp, i32, i32, i32, vp],
        "td_flux_num_params": [vp],
        "td_flux_param_info": [vp, i32, ctypes.c_char_p, i32, vp],
        "td_flux_load_param": [vp, ctypes.c_char_p, vp, i64, vp],
        "td_flux_init_random": [vp, ctypes.c_uint64, f32, vp],
        "td_flux_set_precision": [vp, i32, vp],
        "td_flux_set_fp8_gemms": [vp, ctypes.c_uint],
        "td_flux_set_act_scales": [vp, i32],
        "td_flux_set_smoothing": [vp, i32],
        "td_flux_set_attention": [vp, i32],
        "td_flux_prepared_shape": [vp, vp, vp, vp, vp],
        "td_vae_output_shape": [vp, i32, i32, vp, vp, vp],
        "td_flux_fork": [vp, vp],
        "td_flux_denoise_multi": [vp, vp, i32, vp, i32, vp],
        "td_flux_set_condition": [vp, vp, i32, vp, vp, vp, i32, vp],
        "td_flux_set_timesteps": [vp, vp, i32, f32, vp],
        "td_flux_forward": [vp, vp, i32, vp, vp],
        "td_flux_denoise": [vp, vp, vp, i32, vp],
        "td_flux_trace_begin": [vp, i32],
        "td_flux_trace_end": [vp, vp, vp, vp, vp],
        "td_attention_set_variant": [i32],
        "td_attention_decode_set_group": [i32],
        "td_quant_rows_fp8": [vp, i64, vp, i64, vp, i32, i32, vp],
        "td_linear_fp8": [vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, i64, i32, vp],
        "td_quant_rows_int8": [vp, i64, vp, i64, vp, i32, i32, vp],
        "td_linear_int8": [vp, i64, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp, vp, i64, i32, vp],
        "td_norm_rows_quant_fp8": [vp, i64, vp, i64, vp, i32, i32, i32, f32, vp, i32, vp, vp, vp, vp, vp],
        "td_layernorm_bf16": [vp, i64, vp, i64, i32, i32, i32, f32, vp, vp, vp],
        "td_add_rows_bf16": [vp, vp, vp, i32, i32, i32, vp],
        "td_glu_mul_bf16": [vp, vp, i32, i32, i32, vp],
        "td_attention_bias_bf16": [vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, f32, i32, vp, vp],
        "td_attention_varlen_bf16": [vp, i64, vp, vp, i64, vp, i64, vp, i32, i32, i32, i32, f32, vp],
        "td_rope_half_bf16": [vp, i64, i32, i32, i32, i32, vp, vp, vp],
        "td_vision_rope_table": [vp, i32, i32, f32, vp, vp, vp],
        "td_patchify_bf16": [vp, i32, i32, i32, i32, i32, vp, i32, vp],
        "td_qwen2_patchify_u8": [vp, i32, i32, vp, i32, i32, i32, vp, i32, vp],
        "td_cast_pad_rows_bf16": [vp, i32, i32, i32, vp, i32, vp],
        "td_vae_create": [vp, i32, i32, vp],
        "td_vae_num_params": [vp],
        "td_vae_param_info": [vp, i32, ctypes.c_char_p, i32, vp],
        "td_vae_load_param": [vp, ctypes.c_char_p, vp, i64, vp],
        "td_vae_init_random": [vp, ctypes.c_uint64, f32, vp],
        "td_vae_decode": [vp, vp, i32, i32, f32, f32, vp, vp, vp],
        "td_conv3x3_nhwc_bf16": [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
        "td_conv3x3_pack_weight": [vp, vp, i32, i32, i32, i32, vp],
        "td_linear_f32out_bf16": [vp, i64, vp, vp, vp, i64, i32, i32, i32, vp],
        "td_groupnorm_nhwc_bf16": [vp, vp, i32, i32, i32, f32, vp, vp, i32, vp, vp],
        "td_groupnorm_workspace_floats": [],
        "td_softmax_rows_f32_bf16": [vp, vp, i32, i32, f32, vp],
        "td_qwen2_create": [vp, i32, vp],
        "td_qwen2_num_params": [vp],
        "td_qwen2_param_info": [vp, i32, ctypes.c_char_p, i32, vp],
        "td_qwen2_load_param": [vp, ctypes.c_char_p, vp, i64, vp],
        "td_qwen2_init_random": [vp, ctypes.c_uint64, f32, vp],
        "td_qwen2_forward": [vp, vp, vp, vp, i32, i32, vp, vp, vp],
        "td_qwen2_embed_tokens": [vp, vp, vp, i32, vp],
        "td_qwen2_forward_slot": [vp, i32, vp, vp, vp, i32, i32, vp, vp, vp],
        "td_qwen2_set_slots": [vp, i32],
        "td_qwen2_set_fused_rope": [vp, i32],
        "td_qwen2_create_slots": [vp, i32, i32, vp],
        "td_qwen2_create_ex": [vp, i32, i32, i32, vp],
        "td_qwen2_prefill_batch": [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp],
        "td_qwen2_prefill_batch_at": [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
        "td_qwen2_prefill_packed": [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp],
        "td_qwen2_prefill_packed_slots": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "td_qwen2_slot_capacity": [vp],
        "td_qwen2_move_slot": [vp, i32, i32, i32, vp],
        "td_qwen2_decode_batch": [vp, i32, vp, vp, vp, vp, vp, vp],
        "td_qwen2_decode_batch_slots": [vp, i32, vp, vp, vp, vp, vp, vp, vp],
        "td_embed_gather_bf16": [vp, vp, vp, i32, i32, i32, vp],
        "td_silu_mul_bf16": [vp, vp, i32, i32, vp],
        "td_mrope_table": [vp, i32, vp, f32, i32, vp, vp, vp],
        "td_attention_bf16": [vp, i64, i64, vp, vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, i32, f32, i32, vp],
        "td_attention_fp8": [vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, f32, vp, vp],
        "td_attention_joint_prescaled_bf16": [vp, i64, vp, vp, i64, vp, i64, i32, i32, f32, vp],
        "td_attention_fp8_qk_rope": [vp, i64, i32, i32, i32, vp, i64, i32, i32, vp, vp, i32, vp, vp, vp, vp, f32, f32, vp, vp],
        "td_sample_top_p_bf16": [vp, i64, i32, i32, f32, f32, ctypes.c_uint64, ctypes.c_uint64, vp, vp],
    }
    for name, args in sig.items():
        if os.environ.get("TD_HIP_LIB") and not hasattr(L, name):
            continue          # an older build under A/B timing may predate an entry point; the shipped library must export all
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int
    L.td_flux_destroy.argtypes = [vp]
    L.td_flux_destroy.restype = None
    L.td_vae_destroy.argtypes = [vp]
    L.td_vae_destroy.restype = None
    L.td_qwen2_destroy.argtypes = [vp]
    L.td_qwen2_destroy.restype = None
    L.td_flux_param_elems.argtypes = [vp]
    L.td_flux_param_elems.restype = ctypes.c_int64
    return sig


def check(status):
    if status != 0:
        raise ThinkDiffHipError(f"libthinkdiff_hip error {status}: {lib().td_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda, "thinkdiff_hip ops take device tensors only"
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rows(t):
    assert t.dim() == 2 and t.stride(1) == 1 and t.dtype == torch.bfloat16, (t.shape, t.stride(), t.dtype)
    return t.stride(0)


def linear(x, w, bias=None, act=ACT_NONE, gate=None, res=None, out=None):
    """out = act(x @ w.T + bias) * gate + res   (2-D bf16 tensors, row stride may exceed width)."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous()
    if out is None:
        out = torch.empty((M, N), dtype=torch.bfloat16, device=x.device)
    check(lib().td_linear_bf16(ptr(x), _rows(x), ptr(w), ptr(bias), ptr(out), _rows(out), M, N, K, act,
                               ptr(gate), ptr(res), _rows(res) if res is not None else 0, stream_ptr()))
    return out


def linear_splitk(x, w, bias=None, res=None, out=None, out1=None, n_split=0, split_k=-1, tile_cfg=-1, norm_w=None, norm_out=None, norm_eps=1e-6):
    """out = x @ w.T + bias + res with the contraction split over workgroups (td_linear_splitk_bf16); columns >= n_split go to out1 when given;
    norm_w / norm_out: the reduction also writes RMSNorm(out; norm_w)."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.is_contiguous()
    if out is None:
        out = torch.empty((M, n_split if out1 is not None else N), dtype=torch.bfloat16, device=x.device)
    check(lib().td_linear_splitk_bf16(ptr(x), _rows(x), ptr(w), ptr(bias), ptr(out), _rows(out), ptr(out1), _rows(out1) if out1 is not None else 0, n_split,
                                      M, N, K, ptr(res), _rows(res) if res is not None else 0, tile_cfg, split_k,
                                      ptr(norm_w), ptr(norm_out), _rows(norm_out) if norm_out is not None else 0, norm_eps, stream_ptr()))
    return out


def linear_split(x, w, bias, out0, act0, out1, act1, n_split):
    M, K = x.shape
    N = w.shape[0]
    check(lib().td_linear_split_bf16(ptr(x), _rows(x), ptr(w), ptr(bias), ptr(out0), _rows(out0), act0,
                                     ptr(out1), _rows(out1), act1, M, N, K, n_split, stream_ptr()))
    return out0, out1


def attention(q, k, v, out, Hq, Hkv, scale=None, causal=False):
    """q:[B,Sq,>=Hq*128] k,v:[B,Skv,>=Hkv*128] (views into projection outputs) -> out:[B,Sq,>=Hq*128]."""
    assert q.dim() == 3 and k.dim() == 3 and v.dim() == 3 and out.dim() == 3
    B, Sq, _ = q.shape
    Skv = k.shape[1]
    for t in (q, k, v, out):
        assert t.dtype == torch.bfloat16 and t.stride(2) == 1
    assert k.stride() == v.stride()
    if scale is None:
        scale = 128 ** -0.5
    check(lib().td_attention_bf16(ptr(q), q.stride(1), q.stride(0), ptr(k), ptr(v), k.stride(1), k.stride(0),
                                  ptr(out), out.stride(1), out.stride(0), B, Sq, Skv, Hq, Hkv, 128,
                                  float(scale), int(causal), stream_ptr()))
    return out


def attention_joint_prescaled(q, k, v, out, H, score_bound=0.0):
    """The FLUX engine's form of the joint attention: q [S, >= H*128] already multiplied by scale * log2(e); `score_bound` > 0 = a fixed
    reference point of the softmax (td_attention_joint_prescaled_bf16)."""
    for t in (q, k, v, out):
        assert t.dim() == 2 and t.dtype == torch.bfloat16 and t.stride(1) == 1
    assert k.stride() == v.stride() and q.shape[0] == k.shape[0]
    check(lib().td_attention_joint_prescaled_bf16(ptr(q), q.stride(0), ptr(k), ptr(v), k.stride(0), ptr(out), out.stride(0), q.shape[0], H, float(score_bound), stream_ptr()))
    return out


def attention_fp8(q, k, v, out, H, scale=None, workspace=None):
    """Joint attention on the e4m3 MFMA: q, k, v, out [S, >= H*128] bf16 views (one batch entry); returns out.
    `workspace`: optional uint8 tensor of td_attention_fp8_workspace_bytes (tests read the packed operands back from it)."""
    assert q.dim() == 2 and k.dim() == 2 and v.dim() == 2 and out.dim() == 2
    for t in (q, k, v, out):
        assert t.dtype == torch.bfloat16 and t.stride(1) == 1
    assert k.stride() == v.stride()
    Sq, Skv = q.shape[0], k.shape[0]
    if scale is None:
        scale = 128 ** -0.5
    L = lib()
    L.td_attention_fp8_workspace_bytes.restype = ctypes.c_size_t
    nbytes = int(L.td_attention_fp8_workspace_bytes(Sq, Skv, H))
    ws = workspace if workspace is not None else torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    assert ws.dtype == torch.uint8 and ws.numel() >= nbytes and ws.is_contiguous()
    check(L.td_attention_fp8(ptr(q), q.stride(0), ptr(k), ptr(v), k.stride(0), ptr(out), out.stride(0), Sq, Skv, H,
                             float(scale), ptr(ws), stream_ptr()))
    return out


def attention_fp8_qk_rope(qkv, out, H, cos, sin, split=0, wqA=None, wkA=None, wqB=None, wkB=None, eps=1e-6, scale=None, workspace=None):
    """td_attention_fp8 fed from the RAW fused projection qkv [S, 3*H*128] (q | k | v): QK-RMSNorm + RoPE happen inside the pack
    pass (td_attention_fp8_qk_rope); qkv is not modified.  Returns out."""
    assert qkv.dim() == 2 and qkv.dtype == torch.bfloat16 and qkv.stride(1) == 1 and out.dtype == torch.bfloat16 and out.stride(1) == 1
    S, D = qkv.shape[0], H * 128
    assert qkv.shape[1] >= 3 * D and cos.dtype == torch.float32 and cos.shape == (S, 128) and sin.shape == (S, 128)
    if scale is None:
        scale = 128 ** -0.5
    L = lib()
    L.td_attention_fp8_workspace_bytes.restype = ctypes.c_size_t
    nbytes = int(L.td_attention_fp8_workspace_bytes(S, S, H))
    ws = workspace if workspace is not None else torch.empty(nbytes, dtype=torch.uint8, device=qkv.device)
    assert ws.dtype == torch.uint8 and ws.numel() >= nbytes and ws.is_contiguous()
    check(L.td_attention_fp8_qk_rope(ptr(qkv), qkv.stride(0), 0, D, 2 * D, ptr(out), out.stride(0), S, H, ptr(cos), ptr(sin), split,
                                     ptr(wqA), ptr(wkA), ptr(wqB), ptr(wkB), float(eps), float(scale), ptr(ws), stream_ptr()))
    return out


class TdFluxConfig(ctypes.Structure):
    """Mirror of `struct TdFluxConfig` (include/thinkdiff_hip.h)."""
    _fields_ = [("in_channels", ctypes.c_int), ("num_layers", ctypes.c_int), ("num_single_layers", ctypes.c_int),
                ("num_heads", ctypes.c_int), ("head_dim", ctypes.c_int), ("joint_dim", ctypes.c_int),
                ("pooled_dim", ctypes.c_int), ("guidance_embeds", ctypes.c_int), ("mlp_ratio", ctypes.c_int),
                ("axes_dims", ctypes.c_int * 3), ("rope_theta", ctypes.c_float)]


def norm_rows(x, out=None, rms=False, eps=1e-6, w=None, split=0, shiftA=None, scaleA=None, shiftB=None, scaleB=None):
    rows, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib().td_norm_rows_bf16(ptr(x), _rows(x), ptr(out), _rows(out), rows, D, int(rms), float(eps), ptr(w), split,
                                  ptr(shiftA), ptr(scaleA), ptr(shiftB), ptr(scaleB), stream_ptr()))
    return out


def qk_norm_rope(qkv, Hq, Hk, q_col, k_col, cos, sin, split=0, wqA=None, wkA=None, wqB=None, wkB=None, eps=1e-6,
                 rotate_half=False):
    assert cos.dtype == torch.float32 and cos.shape == (qkv.shape[0], 128) and cos.is_contiguous() and sin.is_contiguous()
    check(lib().td_qk_norm_rope_bf16(ptr(qkv), _rows(qkv), qkv.shape[0], Hq, Hk, q_col, k_col, ptr(cos), ptr(sin), split,
                                     ptr(wqA), ptr(wkA), ptr(wqB if wqB is not None else wqA),
                                     ptr(wkB if wkB is not None else wkA), float(eps), int(rotate_half), stream_ptr()))
    return qkv


def flux_rope_table(ids, axes_dims=(16, 56, 56), theta=10000.0):
    assert ids.dtype == torch.float32 and ids.is_contiguous() and ids.shape[1] == 3
    S = ids.shape[0]
    cos = torch.empty(S, 128, dtype=torch.float32, device=ids.device)
    sin = torch.empty_like(cos)
    axes = (ctypes.c_int * 3)(*axes_dims)
    check(lib().td_flux_rope_table(ptr(ids), S, ctypes.cast(axes, ctypes.c_void_p), float(theta), ptr(cos), ptr(sin), stream_ptr()))
    return cos, sin


def timestep_sincos(t):
    assert t.dtype == torch.float32 and t.is_contiguous()
    out = torch.empty(t.numel(), 256, dtype=torch.bfloat16, device=t.device)
    check(lib().td_timestep_sincos(ptr(t), t.numel(), ptr(out), stream_ptr()))
    return out


def euler_step(x, v, dt):
    assert x.is_contiguous() and v.is_contiguous() and x.dtype == v.dtype == torch.bfloat16
    check(lib().td_euler_step_bf16(ptr(x), ptr(v), float(dt), x.numel(), stream_ptr()))
    return x


def flux_pack_latents(lat):
    """[C,H,W] bf16 -> [(H/2)(W/2), 4C]"""
    C, H, W = lat.shape
    out = torch.empty((H // 2) * (W // 2), C * 4, dtype=torch.bfloat16, device=lat.device)
    check(lib().td_flux_pack_latents(ptr(lat.contiguous()), ptr(out), C, H, W, 0, 1.0, 0.0, stream_ptr()))
    return out


def flux_unpack_latents(x, C, H, W, div=1.0, add=0.0):
    """[(H/2)(W/2), 4C] bf16 -> bf16(bf16([C,H,W] / div) + add)"""
    out = torch.empty(C, H, W, dtype=torch.bfloat16, device=x.device)
    check(lib().td_flux_pack_latents(ptr(x.contiguous()), ptr(out), C, H, W, 1, float(div), float(add), stream_ptr()))
    return out


def cls_avgpool2(x):
    n, C = x.shape
    G = int(round((n - 1) ** 0.5))
    assert G * G == n - 1 and x.is_contiguous()
    out = torch.empty(1 + (G // 2) ** 2, C, dtype=torch.bfloat16, device=x.device)
    check(lib().td_cls_avgpool2_bf16(ptr(x), ptr(out), G, C, stream_ptr()))
    return out


def fill_normal(t, seed, std=1.0, mean=0.0):
    assert t.is_contiguous() and t.dtype == torch.bfloat16
    check(lib().td_fill_normal_bf16(ptr(t), t.numel(), seed, float(std), float(mean), stream_ptr()))
    return t


def aligner_mlp2x(x, w0, b0, w2, b2, norm_w, eps=1e-6, fp32_norm=False):
    """y = T5LayerNorm(Linear2(GELU(Linear0(x))))  x:[M,K] bf16 -> [M,hidden] bf16"""
    M, K = x.shape
    hidden = w0.shape[0]
    ws = torch.empty(2 * M * hidden, dtype=torch.bfloat16, device=x.device)
    y = torch.empty(M, hidden, dtype=torch.bfloat16, device=x.device)
    check(lib().td_aligner_mlp2x_bf16(ptr(x), _rows(x), M, K, hidden, ptr(w0), ptr(b0), ptr(w2), ptr(b2), ptr(norm_w),
                                      float(eps), int(fp32_norm), ptr(ws), ptr(y), hidden, stream_ptr()))
    return y


def linear_grouped2(x0, w0, b0, y0, x1, w1, b1, y1, act=ACT_NONE, gate0=None, res0=None, gate1=None, res1=None, tile_cfg=-1):
    """Two Linear problems (same N, K, row strides) in one launch; x1 may be None (M1 = 0)."""
    M0, K = x0.shape
    N = w0.shape[0]
    M1 = 0 if x1 is None else x1.shape[0]
    ldr = _rows(res0) if res0 is not None else 0
    check(lib().td_linear_grouped2_bf16(ptr(x0), M0, ptr(w0), ptr(b0), ptr(gate0), ptr(res0), ptr(y0),
                                        ptr(x1), M1, ptr(w1), ptr(b1), ptr(gate1), ptr(res1), ptr(y1),
                                        _rows(x0), _rows(y0), ldr, N, K, act, tile_cfg, stream_ptr()))
    return y0, y1


class TdQwen2Config(ctypes.Structure):
    """Mirror of `struct TdQwen2Config` (include/thinkdiff_hip.h)."""
    _fields_ = [("hidden", ctypes.c_int), ("num_layers", ctypes.c_int), ("num_heads", ctypes.c_int),
                ("num_kv_heads", ctypes.c_int), ("head_dim", ctypes.c_int), ("intermediate", ctypes.c_int),
                ("vocab", ctypes.c_int), ("tie_embeddings", ctypes.c_int), ("mrope_section", ctypes.c_int * 3),
                ("rms_eps", ctypes.c_float), ("rope_theta", ctypes.c_float)]


class TdVaeConfig(ctypes.Structure):
    """Mirror of `struct TdVaeConfig` (include/thinkdiff_hip.h)."""
    _fields_ = [("latent_channels", ctypes.c_int), ("out_channels", ctypes.c_int), ("num_blocks", ctypes.c_int),
                ("block_out_channels", ctypes.c_int * 4), ("layers_per_block", ctypes.c_int), ("norm_groups", ctypes.c_int)]


def conv3x3_nhwc(x, w_packed, bias, H, W, Cout, res=None, upsample2x=False):
    """x [Hin*Win, Cin] bf16 NHWC, w_packed [Cout, 9*Cin] -> [H*W, Cout]"""
    Cin = x.shape[1]
    y = torch.empty(H * W, Cout, dtype=torch.bfloat16, device=x.device)
    check(lib().td_conv3x3_nhwc_bf16(ptr(x), ptr(w_packed), ptr(bias), ptr(res), ptr(y), H, W, Cin, Cout, int(upsample2x), stream_ptr()))
    return y


def conv3x3_pack_weight(w_oihw, Cout_pad=None, Cin_pad=None):
    Cout, Cin = w_oihw.shape[:2]
    Cout_pad, Cin_pad = Cout_pad or Cout, Cin_pad or Cin
    out = torch.empty(Cout_pad, 9 * Cin_pad, dtype=torch.bfloat16, device=w_oihw.device)
    check(lib().td_conv3x3_pack_weight(ptr(w_oihw.contiguous()), ptr(out), Cout, Cin, Cout_pad, Cin_pad, stream_ptr()))
    return out


def groupnorm_nhwc(x, gamma, beta, groups=32, eps=1e-6, silu=False):
    P, C = x.shape
    ws = torch.empty(lib().td_groupnorm_workspace_floats(), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    check(lib().td_groupnorm_nhwc_bf16(ptr(x), ptr(y), P, C, groups, float(eps), ptr(gamma), ptr(beta), int(silu), ptr(ws), stream_ptr()))
    return y


def layernorm(x, w=None, b=None, eps=1e-5, rms=False, out=None):
    rows, D = x.shape
    if out is None:
        out = torch.empty(rows, D, dtype=torch.bfloat16, device=x.device)
    check(lib().td_layernorm_bf16(ptr(x), _rows(x), ptr(out), _rows(out), rows, D, int(rms), float(eps), ptr(w), ptr(b), stream_ptr()))
    return out


def add_rows(a, b):
    rows, D = a.shape
    out = torch.empty_like(a)
    check(lib().td_add_rows_bf16(ptr(a), ptr(b.contiguous()), ptr(out), rows, D, b.shape[0], stream_ptr()))
    return out


def glu_mul(gate_up, act):
    rows, two_i = gate_up.shape
    out = torch.empty(rows, two_i // 2, dtype=torch.bfloat16, device=gate_up.device)
    check(lib().td_glu_mul_bf16(ptr(gate_up), ptr(out), rows, two_i // 2, act, stream_ptr()))
    return out


def attention_padded(qkv, H, scale, causal=False, bias=None, out=None):
    """qkv [S, 3*H*128] with every head zero-padded to 128 columns -> [S, H*128].  bias: fp32 [H,S,S] or None."""
    S = qkv.shape[0]
    W = H * 128
    if out is None:
        out = torch.empty(S, W, dtype=torch.bfloat16, device=qkv.device)
    assert out.shape == (S, W) and out.is_contiguous()
    check(lib().td_attention_bias_bf16(ptr(qkv), _rows(qkv), ptr(qkv[:, W:]), ptr(qkv[:, 2 * W:]), _rows(qkv), ptr(out), W,
                                       S, S, H, H, float(scale), int(causal), ptr(bias), stream_ptr()))
    return out


def attention_padded_varlen(qkv, H, scale, seg_starts, max_len, out=None):
    """attention_padded over packed segments in ONE launch: qkv [S, 3*H*128], seg_starts device int32 [n_seg + 1] (row offsets,
    seg_starts[-1] == S), full attention inside each segment.  max_len = the longest segment."""
    S = qkv.shape[0]
    W = H * 128
    if out is None:
        out = torch.empty(S, W, dtype=torch.bfloat16, device=qkv.device)
    assert out.shape == (S, W) and out.is_contiguous()
    assert seg_starts.dtype == torch.int32 and seg_starts.is_cuda and seg_starts.is_contiguous() and seg_starts.numel() >= 2
    check(lib().td_attention_varlen_bf16(ptr(qkv), _rows(qkv), ptr(qkv[:, W:]), ptr(qkv[:, 2 * W:]), _rows(qkv), ptr(out), W,
                                         ptr(seg_starts), seg_starts.numel() - 1, int(max_len), H, H, float(scale), stream_ptr()))
    return out


def rope_half(x, H, hd, cos, sin, head_stride=128):
    """In place on x [S, >= H*head_stride]; cos/sin fp32 [S, hd/2]."""
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (x.shape[0], hd // 2)
    check(lib().td_rope_half_bf16(ptr(x), _rows(x), x.shape[0], H, head_stride, hd, ptr(cos), ptr(sin), stream_ptr()))
    return x


def qwen2_patchify_u8(img, lut, patch, merge, temporal, Kpad, out=None):
    """img uint8 [H,W,3] (cuda), lut fp32 [3,256] (cuda) -> bf16 [(H/patch)(W/patch), Kpad]: the Qwen2-VL processor's rescale,
    normalize and patchify (merge-window row order) in one launch."""
    H, W, C = img.shape
    assert C == 3 and img.dtype == torch.uint8 and img.is_contiguous() and lut.dtype == torch.float32 and lut.shape == (3, 256) and lut.is_contiguous()
    S = (H // patch) * (W // patch)
    if out is None:
        out = torch.empty(S, Kpad, dtype=torch.bfloat16, device=img.device)
    assert out.shape == (S, Kpad) and out.is_contiguous() and out.dtype == torch.bfloat16
    check(lib().td_qwen2_patchify_u8(ptr(img), H, W, ptr(lut), int(patch), int(merge), int(temporal), ptr(out), int(Kpad), stream_ptr()))
    return out


def patchify(pix, p, Kpad):
    """pix [C,H,W] fp32|bf16 -> [(H/p)(W/p), Kpad] bf16."""
    C, H, W = pix.shape
    assert pix.is_contiguous() and pix.dtype in (torch.float32, torch.bfloat16)
    out = torch.empty((H // p) * (W // p), Kpad, dtype=torch.bfloat16, device=pix.device)
    check(lib().td_patchify_bf16(ptr(pix), int(pix.dtype == torch.float32), C, H, W, p, ptr(out), Kpad, stream_ptr()))
    return out


def cast_pad_rows(src, Kpad):
    rows, K = src.shape
    assert src.is_contiguous() and src.dtype in (torch.float32, torch.bfloat16)
    out = torch.empty(rows, Kpad, dtype=torch.bfloat16, device=src.device)
    check(lib().td_cast_pad_rows_bf16(ptr(src), int(src.dtype == torch.float32), rows, K, ptr(out), Kpad, stream_ptr()))
    return out


def vision_rope_table(pos, hd, theta=10000.0):
    """pos int32 [S,2] on the device -> (cos, sin) fp32 [S, hd/2]."""
    S = pos.shape[0]
    assert pos.dtype == torch.int32 and pos.is_contiguous() and pos.shape[1] == 2
    cos = torch.empty(S, hd // 2, dtype=torch.float32, device=pos.device)
    sin = torch.empty_like(cos)
    check(lib().td_vision_rope_table(ptr(pos), S, hd, float(theta), ptr(cos), ptr(sin), stream_ptr()))
    return cos, sin


# ---- fp8 operand path --------------------------------------------------------------------------------------------
def quant_rows_fp8(x):
    """bf16 [R,K] -> (uint8 e4m3 [R,K], fp32 scale [R])."""
    R, K = x.shape
    q = torch.empty(R, K, dtype=torch.uint8, device=x.device)
    s = torch.empty(R, dtype=torch.float32, device=x.device)
    check(lib().td_quant_rows_fp8(ptr(x), _rows(x), ptr(q), K, ptr(s), R, K, stream_ptr()))
    return q, s


def linear_fp8(xq, xs, wq, ws, bias=None, act=ACT_NONE, gate=None, res=None, out=None, tile_cfg=-1):
    M, K = xq.shape
    N = wq.shape[0]
    assert xq.dtype == torch.uint8 and wq.dtype == torch.uint8 and wq.shape[1] == K and wq.is_contiguous()
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=xq.device)
    check(lib().td_linear_fp8(ptr(xq), xq.stride(0), ptr(xs), ptr(wq), ptr(ws), ptr(bias), ptr(out), _rows(out), M, N, K, act,
                              ptr(gate), ptr(res), _rows(res) if res is not None else 0, tile_cfg, stream_ptr()))
    return out


def quant_rows_int8(x):
    """bf16 [R,K] -> (int8 [R,K], fp32 scale [R]): q = rint(x / s), s = max|x| / 127 per row."""
    R, K = x.shape
    q = torch.empty(R, K, dtype=torch.int8, device=x.device)
    s = torch.empty(R, dtype=torch.float32, device=x.device)
    check(lib().td_quant_rows_int8(ptr(x), _rows(x), ptr(q), K, ptr(s), R, K, stream_ptr()))
    return q, s


def linear_int8(xq, xs, wq, ws, bias=None, act=ACT_NONE, gate=None, res=None, out=None, tile_cfg=-1):
    M, K = xq.shape
    N = wq.shape[0]
    assert xq.dtype == torch.int8 and wq.dtype == torch.int8 and wq.shape[1] == K and wq.is_contiguous()
    if out is None:
        out = torch.empty(M, N, dtype=torch.bfloat16, device=xq.device)
    check(lib().td_linear_int8(ptr(xq), xq.stride(0), ptr(xs), ptr(wq), ptr(ws), ptr(bias), ptr(out), _rows(out), M, N, K, act,
                               ptr(gate), ptr(res), _rows(res) if res is not None else 0, tile_cfg, stream_ptr()))
    return out


def norm_rows_quant_fp8(x, rms=False, eps=1e-6, w=None, split=0, shiftA=None, scaleA=None, shiftB=None, scaleB=None):
    R, D = x.shape
    q = torch.empty(R, D, dtype=torch.uint8, device=x.device)
    s = torch.empty(R, dtype=torch.float32, device=x.device)
    check(lib().td_norm_rows_quant_fp8(ptr(x), _rows(x), ptr(q), D, ptr(s), R, D, int(rms), float(eps), ptr(w), split,
                                       ptr(shiftA), ptr(scaleA), ptr(shiftB), ptr(scaleB), stream_ptr()))
    return q, s


def sample_top_p(logits, temperature, top_p, seed, offset, out=None):
    """logits bf16 [rows, vocab] (or [vocab]) -> int32 [rows] token ids on the device; no host synchronisation."""
    x = logits if logits.dim() == 2 else logits[None]
    assert x.dtype == torch.bfloat16 and x.stride(1) == 1
    if out is None:
        out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    check(lib().td_sample_top_p_bf16(ptr(x), x.stride(0), x.shape[0], x.shape[1], float(temperature), float(top_p),
                                     int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset) & 0xFFFFFFFFFFFFFFFF, ptr(out), stream_ptr()))
    return out


def kernel_source_digest() -> str:
    """sha256 over the kernel sources of this tree (csrc/*.hip, csrc/*.h, include/thinkdiff_hip.h; names and contents, sorted): parity records
    written on the GPU box carry it, and bench.py quotes a record only when it was measured on the sources it is running."""
    import glob
    import hashlib
    pkg = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")))
    files.append(os.path.join(os.path.dirname(pkg), "include", "thinkdiff_hip.h"))
    h = hashlib.sha256()
    for fn in files:
        h.update(os.path.basename(fn).encode() + b"\0")
        with open(fn, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()

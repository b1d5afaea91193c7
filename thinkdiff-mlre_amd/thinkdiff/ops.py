"""`torch.ops.thinkdiff_hip.*`: the custom-op layer of the MI355X hot path (last row).

The ops are registered by a shared library -- `lib/libthinkdiff_torch_ops.so`, built by `make` from `csrc_torch/torch_ops.cpp`
(TORCH_LIBRARY + TORCH_LIBRARY_IMPL for the "CUDA" = HIP dispatch key) -- whose kernels hand raw device pointers to the C ABI of
`libthinkdiff_hip.so` (include/thinkdiff_hip.h).  Conventions: tensors are borrowed (caller owns, device-resident, innermost
stride 1), outputs are allocated by the PyTorch caching allocator on the current HIP stream, nothing synchronises, a rejected
argument is a `RuntimeError` carrying `td_last_error()`, one process per GPU.  There is NO CPU / Meta / composite kernel: calling
an op with host tensors fails in the dispatcher instead of quietly computing somewhere else, and importing this module without the
library raises.

    import thinkdiff.ops                      # loads the library (idempotent)
    y = torch.ops.thinkdiff_hip.linear(x, w, b, 0, None, None)

Op                         replaces in the reference's stack (details: include/thinkdiff_hip.h)
linear / aligner_mlp2x     nn.Linear (+bias/act/gate/residual), the ThinkDiff aligner mm_projector
attention                  F.scaled_dot_product_attention on token-major fused projections (joint or causal GQA)
norm_rows                  LayerNorm / RMSNorm rows (+ adaLN modulation)
qk_norm_rope_              per-head QK-RMSNorm + rotary embedding, in place
euler_step_                FlowMatchEulerDiscreteScheduler.step, in place
flux_pack_latents / flux_unpack_latents, cls_avgpool2, sample_top_p
flux_forward_ / flux_denoise_ / flux_denoise_multi_   FluxTransformer2DModel.forward and the pipeline's denoising loop, on a prepared engine
                           (`engine` = the td_flux* handle thinkdiff.models.flux_transformer holds); tensors are checked against the
                           extents the prepared context expects (td_flux_prepared_shape)
vae_decode_u8              AutoencoderKL.decode + VaeImageProcessor.postprocess on a td_vae* engine
attention_fp8              the joint attention with QK^T / P.V on the e4m3 MFMA
"""
import os

import torch

from . import _hip

OPS_LIB_PATH = os.path.join(os.path.dirname(_hip.LIB_PATH), "libthinkdiff_torch_ops.so")

# the schemas csrc_torch/torch_ops.cpp defines (tests compare them with what the dispatcher reports)
SCHEMAS = {
    "linear": "(Tensor x, Tensor w, Tensor? bias, int act, Tensor? gate, Tensor? res) -> Tensor",
    "aligner_mlp2x": "(Tensor x, Tensor w0, Tensor b0, Tensor w2, Tensor b2, Tensor norm_w, float eps, bool fp32_norm) -> Tensor",
    "attention": "(Tensor q, Tensor k, Tensor v, int Hq, int Hkv, float scale, bool causal) -> Tensor",
    "norm_rows": "(Tensor x, bool rms, float eps, Tensor? w, int split, Tensor? shiftA, Tensor? scaleA, Tensor? shiftB, Tensor? scaleB) -> Tensor",
    "qk_norm_rope_": "(Tensor(a!) qkv, int Hq, int Hk, int q_col, int k_col, Tensor cos, Tensor sin, int split, Tensor? wqA, Tensor? wkA, Tensor? wqB, Tensor? wkB, float eps, bool rotate_half) -> Tensor(a!)",
    "euler_step_": "(Tensor(a!) x, Tensor v, float dt) -> Tensor(a!)",
    "flux_pack_latents": "(Tensor latents) -> Tensor",
    "flux_unpack_latents": "(Tensor packed, int C, int H, int W, float div, float add) -> Tensor",
    "cls_avgpool2": "(Tensor tokens) -> Tensor",
    "sample_top_p": "(Tensor logits, float temperature, float top_p, int seed, int offset) -> Tensor",
    "flux_forward_": "(int engine, Tensor latents, int step, Tensor(a!) velocity) -> Tensor(a!)",
    "flux_denoise_": "(int engine, Tensor(a!) latents, float[] sigmas) -> Tensor(a!)",
    "flux_denoise_multi_": "(int[] engines, Tensor(a!)[] latents, float[] sigmas, int[] streams) -> ()",
    "vae_decode_u8": "(int engine, Tensor packed, int h, int w, float scaling_factor, float shift_factor) -> Tensor",
    "attention_fp8": "(Tensor q, Tensor k, Tensor v, int H, float scale) -> Tensor",
}

_loaded = False


def register():
    """Load libthinkdiff_torch_ops.so once per process (its static initialisers define the `thinkdiff_hip` namespace)."""
    global _loaded
    if not _loaded:
        if not os.path.exists(OPS_LIB_PATH):
            raise _hip.ThinkDiffHipError(f"{OPS_LIB_PATH} not found: build it with `make -C thinkdiff-mlre_amd` "
                                         "(or __graft_entry__.build()); there is no Python-side stand-in for the op layer")
        torch.ops.load_library(OPS_LIB_PATH)
        _loaded = True
    return torch.ops.thinkdiff_hip


register()

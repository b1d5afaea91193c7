"""`torch.ops.thinkdiff_hip.*`: the custom-op layer of the MI355X hot path (last row).

Each op is a schema registered with `torch.library` whose only kernel is the "CUDA" (= HIP on ROCm) dispatch entry that hands
raw device pointers to the C ABI of libthinkdiff_hip.so (include/thinkdiff_hip.h) through `thinkdiff._hip`.  Conventions:
tensors are borrowed (caller owns, device-resident, innermost stride 1), outputs are allocated by the PyTorch caching
allocator on the current HIP stream, nothing synchronises, errors surface as `RuntimeError` (ThinkDiffHipError), one process
per GPU.  There is NO CPU / Meta / composite kernel: calling an op with host tensors fails in the dispatcher
("could not run ... with arguments from the 'CPU' backend") instead of quietly computing somewhere else.

    import thinkdiff.ops                      # registers the namespace (idempotent)
    y = torch.ops.thinkdiff_hip.linear(x, w, b, 0, None, None)

Op                         replaces in the reference's stack (details: include/thinkdiff_hip.h)
linear / aligner_mlp2x     nn.Linear (+bias/act/gate/residual), the ThinkDiff aligner mm_projector
attention                  F.scaled_dot_product_attention on token-major fused projections (joint or causal GQA)
norm_rows                  LayerNorm / RMSNorm rows (+ adaLN modulation)
qk_norm_rope_              per-head QK-RMSNorm + rotary embedding, in place
euler_step_                FlowMatchEulerDiscreteScheduler.step, in place
flux_pack_latents / flux_unpack_latents, cls_avgpool2, sample_top_p
"""
from typing import Optional

import torch

from . import _hip

_LIB = None
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
}


def _attention(q, k, v, Hq: int, Hkv: int, scale: float, causal: bool):
    out = torch.empty(q.shape[0], q.shape[1], Hq * 128, dtype=torch.bfloat16, device=q.device)
    return _hip.attention(q, k, v, out, Hq, Hkv, scale, causal)


def _norm_rows(x, rms: bool, eps: float, w: Optional[torch.Tensor], split: int, shiftA, scaleA, shiftB, scaleB):
    return _hip.norm_rows(x, None, rms, eps, w, split, shiftA, scaleA, shiftB, scaleB)


def _qk_norm_rope_(qkv, Hq: int, Hk: int, q_col: int, k_col: int, cos, sin, split: int, wqA, wkA, wqB, wkB, eps: float, rotate_half: bool):
    return _hip.qk_norm_rope(qkv, Hq, Hk, q_col, k_col, cos, sin, split, wqA, wkA, wqB, wkB, eps, rotate_half)


_IMPLS = {
    "linear": lambda x, w, bias, act, gate, res: _hip.linear(x, w, bias, act, gate, res),
    "aligner_mlp2x": lambda x, w0, b0, w2, b2, nw, eps, f32: _hip.aligner_mlp2x(x, w0, b0, w2, b2, nw, eps, f32),
    "attention": _attention,
    "norm_rows": _norm_rows,
    "qk_norm_rope_": _qk_norm_rope_,
    "euler_step_": lambda x, v, dt: _hip.euler_step(x, v, dt),
    "flux_pack_latents": lambda lat: _hip.flux_pack_latents(lat),
    "flux_unpack_latents": lambda p, C, H, W, div, add: _hip.flux_unpack_latents(p, C, H, W, div, add),
    "cls_avgpool2": lambda t: _hip.cls_avgpool2(t),
    "sample_top_p": lambda lg, T, p, seed, off: _hip.sample_top_p(lg, T, p, seed, off),
}


def register():
    """Define the `thinkdiff_hip` namespace once per process; returns the torch.library.Library handle."""
    global _LIB
    if _LIB is None:
        lib = torch.library.Library("thinkdiff_hip", "DEF")
        for name, schema in SCHEMAS.items():
            lib.define(name + schema)
            lib.impl(name, _IMPLS[name], "CUDA")
        _LIB = lib
    return _LIB


register()

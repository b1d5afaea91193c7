"""FLUX.1 MMDiT on the HIP engine (libthinkdiff_hip.so `td_flux_*`).

Host-side mirror of the object the reference drivers reach through `diffusion_pipe.transformer`
([ext] diffusers 0.31.0 `FluxTransformer2DModel`): same config keys, same state-dict names, same
`forward(hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids,
guidance)` meaning.  All compute happens in the C++/HIP engine; this class only owns the handle,
moves checkpoints into the engine's fused weight arena and converts arguments to device pointers.
"""
import ctypes
import dataclasses
import glob
import json
import os
from typing import Dict, Optional, Sequence

import torch

from .. import _hip
from ..ops import register as _register_ops

_OPS = _register_ops()      # torch.ops.thinkdiff_hip: the denoise loop is dispatched as custom ops over the C ABI (GPU kernels only, no fallback)


@dataclasses.dataclass
class FluxTransformerConfig:
    """Keys of [ext] FLUX.1-dev transformer/config.json."""
    patch_size: int = 1
    in_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    guidance_embeds: bool = True
    axes_dims_rope: Sequence[int] = (16, 56, 56)

    @property
    def inner_dim(self):
        return self.attention_head_dim * self.num_attention_heads


def effective_scalar(value: float, dtype: torch.dtype) -> float:
    """What the sinusoidal embedding finally sees for `timestep`/`guidance` in the reference pipeline:
    cast to the latents dtype, /1000 in the pipeline, *1000 in the transformer, all in `dtype`
    ([ext] pipeline_flux.py `timestep / 1000`, transformer_flux.py `timestep.to(dtype) * 1000`)."""
    x = torch.tensor([value], dtype=torch.float32).to(dtype)
    return float(((x / 1000).to(dtype) * 1000).float())


class FluxTransformer2DModel:
    dtype = torch.bfloat16

    def __init__(self, config: Optional[FluxTransformerConfig] = None, max_img_tokens: int = 4096,
                 max_txt_tokens: int = 512, max_steps: int = 64, device="cuda", **config_kwargs):
        self.config = config or FluxTransformerConfig(**config_kwargs)
        c = self.config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _hip.ThinkDiffHipError("FluxTransformer2DModel runs on the MI355X HIP engine only (device='cuda')")
        self._L = _hip.lib()
        cc = _hip.TdFluxConfig(c.in_channels, c.num_layers, c.num_single_layers, c.num_attention_heads,
                               c.attention_head_dim, c.joint_attention_dim, c.pooled_projection_dim,
                               int(c.guidance_embeds), 4, (ctypes.c_int * 3)(*c.axes_dims_rope), 10000.0)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(self._L.td_flux_create(ctypes.byref(cc), max_img_tokens, max_txt_tokens, max_steps, ctypes.byref(h)))
        self._h = h
        self.max_img_tokens, self.max_txt_tokens, self.max_steps = max_img_tokens, max_txt_tokens, max_steps
        self._n_steps = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.td_flux_destroy(h)

    def fork(self) -> "FluxTransformer2DModel":
        """A second context over the same weights (own workspace / conditioning / schedule) for images in flight on
        another stream.  Keeps a reference to the parent, which owns the weights and the precision setting."""
        child = object.__new__(type(self))
        child.__dict__.update({k: v for k, v in self.__dict__.items() if k != "_h"})
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(self._L.td_flux_fork(self._h, ctypes.byref(h)))
        child._h, child._parent, child._n_steps = h, self, 0
        return child

    @staticmethod
    def denoise_multi(contexts, latents, sigmas: Sequence[float], streams):
        """td_flux_denoise for several prepared contexts at once, context k on streams[k] (torch.cuda.Stream)."""
        _OPS.flux_denoise_multi_([int(m._h.value) for m in contexts], list(latents), [float(s) for s in sigmas], [int(st.cuda_stream) for st in streams])
        return latents

    # ---- parameters ---------------------------------------------------------------------------------
    def param_table(self) -> Dict[str, int]:
        n = self._L.td_flux_num_params(self._h)
        buf = ctypes.create_string_buffer(256)
        cnt = ctypes.c_int64()
        out = {}
        for i in range(n):
            _hip.check(self._L.td_flux_param_info(self._h, i, buf, 256, ctypes.byref(cnt)))
            out[buf.value.decode()] = cnt.value
        return out

    def num_parameters(self) -> int:
        return sum(self.param_table().values())

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        table = self.param_table()
        missing = [k for k in table if k not in sd]
        unexpected = [k for k in sd if k not in table]
        if strict and (missing or unexpected):
            raise KeyError(f"FluxTransformer2DModel.load_state_dict: missing={missing[:4]}.. unexpected={unexpected[:4]}..")
        for name, t in sd.items():
            if name not in table:
                continue
            d = t.to(device=self.device, dtype=torch.bfloat16).contiguous()
            _hip.check(self._L.td_flux_load_param(self._h, name.encode(), _hip.ptr(d), d.numel(), _hip.stream_ptr()))
            torch.cuda.current_stream().synchronize()  # `d` may be a temporary
        return missing, unexpected

    @classmethod
    def from_pretrained(cls, path: str, subfolder: str = "transformer", **kw):
        """Local directories only (there is no hub access): <path>/<subfolder>/{config.json,*.safetensors}."""
        from safetensors import safe_open
        root = os.path.join(path, subfolder) if os.path.isdir(os.path.join(path, subfolder)) else path
        with open(os.path.join(root, "config.json")) as fh:
            raw = json.load(fh)
        fields = {f.name for f in dataclasses.fields(FluxTransformerConfig)}
        model = cls(FluxTransformerConfig(**{k: v for k, v in raw.items() if k in fields}), **kw)
        seen = set()
        for fn in sorted(glob.glob(os.path.join(root, "*.safetensors"))):
            with safe_open(fn, framework="pt") as fh:
                part = {k: fh.get_tensor(k) for k in fh.keys()}
            model.load_state_dict(part, strict=False)
            seen.update(part)
        missing = [k for k in model.param_table() if k not in seen]
        if missing:
            raise KeyError(f"checkpoint at {root} lacks {len(missing)} tensors, e.g. {missing[:3]}")
        return model

    def init_random(self, seed: int = 0, std: float = 0.02):
        """Synthetic full-shape checkpoint generated on the device (throughput runs)."""
        _hip.check(self._L.td_flux_init_random(self._h, seed, std, _hip.stream_ptr()))
        return self

    FP8_GEMMS = {"qkv": 1, "out": 2, "ff1": 4, "ff2": 8, "single_in": 16, "single_out": 32}     # TD_FP8_* of include/thinkdiff_hip.h

    def set_attention(self, mode: str = "bf16"):
        """Arithmetic of the joint attention: "bf16" (default, the reference graph's) or "fp8" (QK^T and P.V on the e4m3 matrix
        instruction, td_flux_set_attention(TD_ATTENTION_FP8)); independent of set_precision, meant for the 8-bit modes."""
        _hip.check(self._L.td_flux_set_attention(self._h, {"bf16": 0, "fp8": 1}[mode]))
        self.attention = mode
        return self

    def set_precision(self, precision: str = "bf16", fp8_gemms=None, act_scales: str = "dynamic", smoothing: bool = False):
        """"bf16" (default) or "fp8": e4m3 operands for the block GEMMs (weights quantised per output channel from the
        parameters as loaded now -- call after load_state_dict / init_random; activations per token on the fly).
        fp8_gemms: None = every block Linear, or the classes that take the fp8 path (names of FP8_GEMMS, or the bit mask);
        the others stay bf16."""
        code = {"bf16": 0, "bfloat16": 0, "fp8": 1, "fp8_e4m3": 1, "float8_e4m3fn": 1, "int8": 2, "w8a8": 2}[str(precision).replace("torch.", "")]
        _hip.check(self._L.td_flux_set_precision(self._h, code, _hip.stream_ptr()))
        mask = 63 if fp8_gemms is None else (int(fp8_gemms) if isinstance(fp8_gemms, int) else sum(self.FP8_GEMMS[str(n)] for n in fp8_gemms))
        _hip.check(self._L.td_flux_set_fp8_gemms(self._h, mask))
        # int8 only: "history" = per-token scales of the MLP operands from the previous denoise step (td_flux_set_act_scales)
        _hip.check(self._L.td_flux_set_act_scales(self._h, {"dynamic": 0, "history": 1}[act_scales] if code == 2 else 0))
        # int8 only: per-channel smoothing of outlier-carrying activations, calibrated on the first forward (td_flux_set_smoothing)
        _hip.check(self._L.td_flux_set_smoothing(self._h, 1 if (smoothing and code == 2) else 0))
        self.precision, self.fp8_gemms, self.act_scales = ("bf16", "fp8", "int8")[code], mask, (act_scales if code == 2 else "dynamic")
        self.smoothing = bool(smoothing and code == 2)
        return self

    # ---- conditioning / schedule ----------------------------------------------------------------------
    def set_condition(self, prompt_embeds, pooled, img_ids, txt_ids=None):
        assert prompt_embeds.dim() == 2 and pooled.dim() == 1, "one prompt per call: [T,joint], [pooled]"
        pe = prompt_embeds.to(self.device, torch.bfloat16).contiguous()
        po = pooled.to(self.device, torch.bfloat16).contiguous()
        ii = img_ids.to(self.device, torch.float32).contiguous()
        ti = None if txt_ids is None else txt_ids.to(self.device, torch.float32).contiguous()
        _hip.check(self._L.td_flux_set_condition(self._h, _hip.ptr(pe), pe.shape[0], _hip.ptr(po), _hip.ptr(ti),
                                                 _hip.ptr(ii), ii.shape[0], _hip.stream_ptr()))
        self._n_img = ii.shape[0]
        torch.cuda.current_stream().synchronize()

    def set_timesteps(self, t_eff: Sequence[float], g_eff: float = 0.0):
        arr = (ctypes.c_float * len(t_eff))(*[float(t) for t in t_eff])
        _hip.check(self._L.td_flux_set_timesteps(self._h, ctypes.cast(arr, ctypes.c_void_p), len(t_eff), float(g_eff), _hip.stream_ptr()))
        self._n_steps = len(t_eff)

    def forward_step(self, latents, step: int, out=None):
        assert latents.dtype == torch.bfloat16 and latents.is_contiguous() and latents.shape == (self._n_img, self.config.in_channels)
        if out is None:
            out = torch.empty_like(latents)
        return _OPS.flux_forward_(int(self._h.value), latents, int(step), out)

    def denoise(self, latents, sigmas: Sequence[float]):
        """In-place Euler flow-matching loop over the prepared timesteps (len(sigmas) == n_steps + 1)."""
        assert latents.dtype == torch.bfloat16 and latents.is_contiguous() and latents.shape == (self._n_img, self.config.in_channels)
        return _OPS.flux_denoise_(int(self._h.value), latents, [float(s) for s in sigmas])

    # ---- per-launch HIP-event trace (bench.py roofline leg) ------------------------------------------
    TRACE_CATEGORIES = ("gemm_256x256", "gemm_other", "attention", "layernorm_modulate", "qk_rmsnorm_rope", "gemm_288x192")

    def trace_begin(self, max_launches: int):
        _hip.check(self._L.td_flux_trace_begin(self._h, max_launches))

    def trace_end(self):
        n = len(self.TRACE_CATEGORIES)
        counts, ms, fl = (ctypes.c_int64 * n)(), (ctypes.c_double * n)(), (ctypes.c_double * n)()
        _hip.check(self._L.td_flux_trace_end(self._h, _hip.stream_ptr(), ctypes.cast(counts, ctypes.c_void_p),
                                             ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(fl, ctypes.c_void_p)))
        return {c: {"launches": int(counts[i]), "ms": float(ms[i]), "flops": float(fl[i])}
                for i, c in enumerate(self.TRACE_CATEGORIES)}

    # ---- diffusers-style call -------------------------------------------------------------------------
    def forward(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids=None,
                guidance=None, return_dict: bool = False, **_ignored):
        """[ext] FluxTransformer2DModel.forward semantics; batch is looped (conditions differ per sample)."""
        B = hidden_states.shape[0]
        outs = []
        for b in range(B):
            self.set_condition(encoder_hidden_states[b], pooled_projections[b], img_ids, txt_ids)
            t = float(timestep[b] if timestep.dim() else timestep)
            g = float(guidance[b] if guidance.dim() else guidance) if guidance is not None else 0.0
            # timestep arrives as t/1000 in the latents dtype; the transformer multiplies by 1000 in that dtype
            te = float((torch.tensor([t]).to(hidden_states.dtype) * 1000).float())
            ge = float((torch.tensor([g]).to(hidden_states.dtype) * 1000).float())
            self.set_timesteps([te], ge)
            outs.append(self.forward_step(hidden_states[b].to(torch.bfloat16).contiguous(), 0))
        out = torch.stack(outs)
        return (out,)

    __call__ = forward

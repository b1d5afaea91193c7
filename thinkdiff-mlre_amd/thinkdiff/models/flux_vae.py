"""FLUX VAE decoder on the HIP engine (`td_vae_*`): stands in for `diffusion_pipe.vae` ([ext] diffusers 0.31.0
`AutoencoderKL`, decode path only) + `VaeImageProcessor.postprocess`.  Parameter names = diffusers state dict."""
import ctypes
import dataclasses
import glob
import json
import os
from typing import Dict, Sequence

import torch

from .. import _hip
from ..ops import register as _register_ops

_OPS = _register_ops()      # torch.ops.thinkdiff_hip (the uint8 decode goes through it)


@dataclasses.dataclass
class AutoencoderKLConfig:
    """Decoder-side keys of [ext] FLUX.1-dev vae/config.json."""
    latent_channels: int = 16
    out_channels: int = 3
    block_out_channels: Sequence[int] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.3611
    shift_factor: float = 0.1159


class AutoencoderKLDecoder:
    dtype = torch.bfloat16

    def __init__(self, config: AutoencoderKLConfig = None, max_latent_size=(128, 128), device="cuda"):
        self.config = config or AutoencoderKLConfig()
        c = self.config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _hip.ThinkDiffHipError("AutoencoderKLDecoder runs on the MI355X HIP engine only (device='cuda')")
        self._L = _hip.lib()
        boc = list(c.block_out_channels) + [0] * (4 - len(c.block_out_channels))
        cc = _hip.TdVaeConfig(c.latent_channels, c.out_channels, len(c.block_out_channels), (ctypes.c_int * 4)(*boc),
                              c.layers_per_block, c.norm_num_groups)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(self._L.td_vae_create(ctypes.byref(cc), max_latent_size[0], max_latent_size[1], ctypes.byref(h)))
        self._h = h
        self.upscale = 2 ** (len(c.block_out_channels) - 1)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.td_vae_destroy(h)

    def param_table(self) -> Dict[str, int]:
        buf, cnt, out = ctypes.create_string_buffer(256), ctypes.c_int64(), {}
        for i in range(self._L.td_vae_num_params(self._h)):
            _hip.check(self._L.td_vae_param_info(self._h, i, buf, 256, ctypes.byref(cnt)))
            out[buf.value.decode()] = cnt.value
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        table = self.param_table()
        missing = [k for k in table if k not in sd]
        if strict and missing:
            raise KeyError(f"AutoencoderKLDecoder.load_state_dict: missing {missing[:4]}..")
        for name, t in sd.items():
            if name in table:     # encoder.* / quant_conv.* keys of a full checkpoint are ignored
                d = t.to(device=self.device, dtype=torch.bfloat16).contiguous()
                _hip.check(self._L.td_vae_load_param(self._h, name.encode(), _hip.ptr(d), d.numel(), _hip.stream_ptr()))
                torch.cuda.current_stream().synchronize()
        return missing

    @classmethod
    def from_pretrained(cls, path: str, subfolder: str = "vae", **kw):
        from safetensors import safe_open
        root = os.path.join(path, subfolder) if os.path.isdir(os.path.join(path, subfolder)) else path
        with open(os.path.join(root, "config.json")) as fh:
            raw = json.load(fh)
        fields = {f.name for f in dataclasses.fields(AutoencoderKLConfig)}
        m = cls(AutoencoderKLConfig(**{k: v for k, v in raw.items() if k in fields}), **kw)
        seen = set()
        for fn in sorted(glob.glob(os.path.join(root, "*.safetensors"))):
            with safe_open(fn, framework="pt") as fh:
                part = {k: fh.get_tensor(k) for k in fh.keys()}
            m.load_state_dict(part, strict=False)
            seen.update(part)
        # every decoder tensor must have arrived: an older VAE export (attention weights named query/key/value/proj_attn)
        # would otherwise leave parts of the weight arena uninitialised and decode garbage without an error
        missing = [k for k in m.param_table() if k not in seen]
        if missing:
            raise KeyError(f"VAE checkpoint at {root} lacks {len(missing)} decoder tensors, e.g. {missing[:3]}")
        return m

    def init_random(self, seed: int = 0, std: float = 0.0):
        """Seeded synthetic decoder; std <= 0 (default): 1 / sqrt(fan_in) weights, so decoded images have contrast."""
        _hip.check(self._L.td_vae_init_random(self._h, seed, std, _hip.stream_ptr()))
        return self

    @torch.no_grad()
    def decode_packed(self, packed_latents, h: int, w: int, output_type: str = "pil"):
        """packed [(h/2)(w/2), 4C] bf16 (the denoised FLUX latents) -> image.  output_type: "pil" | "np" (uint8 HWC)
        | "pt" (bf16 [3,H,W], the raw vae.decode output)."""
        x = packed_latents.to(self.device, torch.bfloat16).contiguous()
        H, W = h * self.upscale, w * self.upscale
        if output_type == "pt":
            chw = torch.empty(3, H, W, dtype=torch.bfloat16, device=self.device)
            _hip.check(self._L.td_vae_decode(self._h, _hip.ptr(x), h, w, self.config.scaling_factor, self.config.shift_factor, None, _hip.ptr(chw), _hip.stream_ptr()))
            return chw
        u8 = _OPS.vae_decode_u8(int(self._h.value), x, int(h), int(w), float(self.config.scaling_factor), float(self.config.shift_factor))
        if output_type == "np":
            return u8
        from PIL import Image
        return Image.fromarray(u8.cpu().numpy())

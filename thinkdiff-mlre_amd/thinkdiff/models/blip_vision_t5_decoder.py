"""ThinkDiff-CLIP model on the HIP path: vision tokens -> 2x2 pooling -> aligner.

Mirror of reference thinkdiff/models/blip_vision_t5_decoder.py: `build_vision_projector` (:31-61),
`BlipVisionT5DecoderForConditionalGeneration.from_config` (:501-563) and `.forward_encoder`
(:566-643).  The aligner and the token pooling run in libthinkdiff_hip.so (td_aligner_mlp2x_bf16,
td_cls_avgpool2_bf16).  The EVA-ViT-g tower that produces the 257 vision tokens is
`vision_towers.HipBlip2VisionModel` (loaded by `from_config` from a local `blip2_pretrained_model_name_or_path`
directory, or attached by `providers.load_vision`); `vision_model` may be any callable pixel_values ->
[B,257,1408] device tensor, and `image_embeds=` bypasses it.
"""
import re
from types import SimpleNamespace

import torch

from .. import _hip
from ..ops import register as _register_ops
from ..common.registry import registry
from .base_model import BaseModel
_OPS = _register_ops()      # torch.ops.thinkdiff_hip: the custom-op layer over the C ABI (GPU kernels only, no fallback)


class IdentityMap:
    def __call__(self, x, *a, **k):
        return x


class HipVisionProjector:
    """`nn.Sequential(Linear, GELU, Linear, T5LayerNorm)` replacement; state-dict keys 0/2/3 as in the reference."""

    def __init__(self, mm_hidden_size, hidden_size, projector_type="mlp2x_gelu_t5_norm", device="cuda", fp32_norm=False):
        m = re.match(r"^mlp(\d+)x_gelu_t5_norm$", projector_type)
        if not m or int(m.group(1)) != 2:
            raise _hip.ThinkDiffHipError(
                f"projector '{projector_type}': only mlp2x_gelu_t5_norm (every shipped ThinkDiff config) has a HIP path")
        self.mm_hidden_size, self.hidden_size, self.fp32_norm = mm_hidden_size, hidden_size, fp32_norm
        z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=device)
        self.params = {"0.weight": z(hidden_size, mm_hidden_size), "0.bias": z(hidden_size),
                       "2.weight": z(hidden_size, hidden_size), "2.bias": z(hidden_size),
                       "3.weight": torch.ones(hidden_size, dtype=torch.bfloat16, device=device)}

    def state_dict(self):
        return dict(self.params)

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self.params if k not in sd]
        if strict and missing:
            raise KeyError(f"mm_projector: missing {missing}")
        for k in self.params:
            if k in sd:
                assert tuple(sd[k].shape) == tuple(self.params[k].shape), (k, sd[k].shape)
                self.params[k].copy_(sd[k].to(self.params[k].device, torch.bfloat16))
        return missing

    def __call__(self, x):
        """x [..., mm_hidden] device bf16 -> [..., hidden]"""
        lead = x.shape[:-1]
        x2 = x.reshape(-1, self.mm_hidden_size).to(torch.bfloat16).contiguous()
        p = self.params
        # the aligner runs as a PyTorch custom op (thinkdiff/ops.py: TORCH_LIBRARY over td_aligner_mlp2x_bf16; GPU kernel only)
        y = _OPS.aligner_mlp2x(x2, p["0.weight"], p["0.bias"], p["2.weight"], p["2.bias"], p["3.weight"], 1e-6, bool(self.fp32_norm))
        return y.reshape(*lead, self.hidden_size)


def build_vision_projector(config, device="cuda"):
    projector_type = getattr(config, "mm_projector_type", "linear")
    if projector_type == "identity":
        return IdentityMap()
    return HipVisionProjector(config.mm_hidden_size, config.hidden_size, projector_type, device=device)


@registry.register_model("blip-vision-t5-decoder")
class BlipVisionT5DecoderForConditionalGeneration(BaseModel):
    PRETRAINED_MODEL_CONFIG_DICT = {"pretrain_blip_vision_t5_decoder": "configs/models/blip_vision_t5_decoder.yaml"}

    def __init__(self, mm_hidden_size=1408, hidden_size=4096, mm_projector_type="mlp2x_gelu_t5_norm",
                 vision_downsample_factor=2, vision_model=None, device="cuda"):
        self.config = SimpleNamespace(mm_hidden_size=mm_hidden_size, hidden_size=hidden_size,
                                      mm_projector_type=mm_projector_type,
                                      vision_downsample_factor=vision_downsample_factor, use_return_dict=True)
        self._device = torch.device(device)
        self.vision_model = vision_model
        self.mm_projector = build_vision_projector(self.config, device=device)

    @classmethod
    def from_config(cls, cfg):
        """Keys read: mm_projector_type, vision_downsample_factor, ckpt (reference :512-563).  `ckpt` must be a
        local file holding {'model': {'mm_projector.*': ...}}; it is loaded with strict=False like the reference."""
        model = cls(mm_projector_type=cfg.get("mm_projector_type", "mlp2x_gelu_t5_norm"),
                    vision_downsample_factor=cfg.get("vision_downsample_factor", None),
                    device=cfg.get("device", "cuda"))
        import os
        blip_dir = cfg.get("blip2_pretrained_model_name_or_path", "")
        if blip_dir and os.path.isdir(blip_dir):
            # reference :517-527 loads Blip2VisionModel weights from this checkpoint; here a local directory only
            from .vision_towers import HipBlip2VisionModel
            model.vision_model = HipBlip2VisionModel.from_pretrained(blip_dir, device=cfg.get("device", "cuda"))
        ckpt_path = cfg.get("ckpt", "")
        if ckpt_path:
            if os.path.isfile(ckpt_path):
                print(f"Load Checkpoint: {ckpt_path}")
                ckpt = torch.load(ckpt_path, map_location="cpu")
                model.load_state_dict(ckpt["model"], strict=False)
            else:
                print(f"ckpt {ckpt_path!r} not found: aligner keeps its initial weights")
        return model

    def load_state_dict(self, sd, strict=False):
        sub = {k[len("mm_projector."):]: v for k, v in sd.items() if k.startswith("mm_projector.")}
        return self.mm_projector.load_state_dict(sub, strict=strict)

    @torch.no_grad()
    def forward_encoder(self, pixel_values=None, input_ids=None, attention_mask=None, image_embeds=None, **_ignored):
        """pixel_values [B,3,224,224] (or image_embeds [B,257,1408]) -> [B,65,4096]; input_ids/attention_mask are
        accepted and ignored exactly as in the reference (:566-643)."""
        if image_embeds is None:
            if self.vision_model is None:
                raise _hip.ThinkDiffHipError("forward_encoder: no vision tower loaded; pass image_embeds=[B,257,1408]")
            image_embeds = self.vision_model(pixel_values)
            image_embeds = image_embeds[0] if isinstance(image_embeds, (tuple, list)) else image_embeds
        x = image_embeds.to(self._device, torch.bfloat16)
        if self.config.vision_downsample_factor is not None:
            if self.config.vision_downsample_factor != 2:
                raise _hip.ThinkDiffHipError("vision_downsample_factor: only 2 (all shipped configs) has a HIP path")
            x = torch.stack([_OPS.cls_avgpool2(x[b].contiguous()) for b in range(x.shape[0])])
        return self.mm_projector(x)

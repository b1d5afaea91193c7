"""ORACLE (test infrastructure, not product code): CPU restatement of the ThinkDiff aligner and the
ThinkDiff-CLIP token pooling.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

Follows reference thinkdiff/models/blip_vision_t5_decoder.py:
  :31-61   build_vision_projector  ("mlp{n}x_gelu_t5_norm": Linear, then (GELU, Linear, T5LayerNorm) x (n-1))
  :620-637 forward_encoder tail    (CLS split, 16x16 -> 8x8 bilinear(align_corners=False), re-attach CLS)
  :641     mm_projector(image_embeds)
and transformers T5LayerNorm (fp32 variance, cast to weight dtype, weight * x).

Pinned: tests/test_oracle_cpu.py checks this restatement against the very torch modules the reference
instantiates (nn.Linear / nn.GELU / transformers T5LayerNorm in an nn.Sequential, F.interpolate).
"""
import re
from typing import Dict

import torch
import torch.nn.functional as F


def projector_depth(projector_type: str) -> int:
    m = re.match(r"^mlp(\d+)x_gelu(_t5_norm)?$", projector_type)
    if not m:
        raise ValueError(f"Unknown projector type: {projector_type}")
    return int(m.group(1))


def param_shapes(mm_hidden: int, hidden: int, projector_type: str = "mlp2x_gelu_t5_norm") -> Dict[str, tuple]:
    """State-dict names of the nn.Sequential: index 0 Linear; per extra depth d: GELU(3d-2), Linear(3d-1), Norm(3d)."""
    n = projector_depth(projector_type)
    s = {"mm_projector.0.weight": (hidden, mm_hidden), "mm_projector.0.bias": (hidden,)}
    for d in range(1, n):
        s[f"mm_projector.{3 * d - 1}.weight"] = (hidden, hidden)
        s[f"mm_projector.{3 * d - 1}.bias"] = (hidden,)
        s[f"mm_projector.{3 * d}.weight"] = (hidden,)
    return s


def init_weights(mm_hidden: int, hidden: int, seed: int = 0, dtype=torch.bfloat16, projector_type="mlp2x_gelu_t5_norm"):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in param_shapes(mm_hidden, hidden, projector_type).items():
        if len(shp) == 1 and k.endswith("weight"):
            sd[k] = (1.0 + 0.1 * torch.randn(shp, generator=g)).to(dtype)
        else:
            sd[k] = (0.02 * torch.randn(shp, generator=g)).to(dtype)
    return sd


def t5_layer_norm(x: torch.Tensor, w: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    var = x.to(torch.float32).pow(2).mean(-1, keepdim=True)
    x = x * torch.rsqrt(var + eps)
    if w.dtype in (torch.float16, torch.bfloat16):
        x = x.to(w.dtype)
    return w * x


def mm_projector(sd: Dict[str, torch.Tensor], x: torch.Tensor, projector_type: str = "mlp2x_gelu_t5_norm") -> torch.Tensor:
    n = projector_depth(projector_type)
    y = F.linear(x, sd["mm_projector.0.weight"], sd["mm_projector.0.bias"])
    for d in range(1, n):
        y = F.gelu(y)  # nn.GELU() default = exact erf
        y = F.linear(y, sd[f"mm_projector.{3 * d - 1}.weight"], sd[f"mm_projector.{3 * d - 1}.bias"])
        if "t5_norm" in projector_type:
            y = t5_layer_norm(y, sd[f"mm_projector.{3 * d}.weight"])
    return y


def pool_vision_tokens(image_embeds: torch.Tensor, factor: int = 2) -> torch.Tensor:
    """[B, 1+G*G, C] -> [B, 1+(G/f)^2, C]  (blip_vision_t5_decoder.py:620-637)."""
    cls, grid = image_embeds[:, 0:1, :], image_embeds[:, 1:, :]
    h = w = int(grid.size(1) ** 0.5)
    g = grid.reshape(grid.shape[0], h, w, grid.shape[-1]).permute(0, 3, 1, 2)
    g = F.interpolate(g, size=(h // factor, w // factor), mode="bilinear", align_corners=False)
    g = g.permute(0, 2, 3, 1).reshape(grid.shape[0], -1, grid.shape[-1])
    return torch.cat([cls, g], dim=1)


def forward_encoder_tail(sd, image_embeds: torch.Tensor, factor: int = 2, projector_type="mlp2x_gelu_t5_norm"):
    return mm_projector(sd, pool_vision_tokens(image_embeds, factor), projector_type)

"""ORACLE (test infrastructure, not product code): CPU restatement of the FLUX VAE decoder
(`AutoencoderKL.decode`) and of the pipeline tail around it.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.

**Parity unpinned**: the code lives in `diffusers==0.31.0` (reference requirements.txt:34; not vendored, not
installed) and the reference has no tests for it.  Restated from the published diffusers 0.31.0 sources
([ext] models/autoencoders/vae.py `Decoder`, models/unets/unet_2d_blocks.py `UNetMidBlock2D` / `UpDecoderBlock2D`,
models/resnet.py `ResnetBlock2D`, models/upsampling.py `Upsample2D`, models/attention_processor.py `Attention`
(heads=1), image_processor.py `VaeImageProcessor.postprocess`), anchored on the reference's call site
scripts/test/test_blip_vision_t5_decoder_flux_text.py:234-247 (`.images[0].save(...)`).
FLUX.1-dev vae/config.json: block_out_channels (128,256,512,512), layers_per_block 2, latent_channels 16,
norm_num_groups 32, mid_block_add_attention, no quant/post_quant conv, scaling 0.3611, shift 0.1159.
"""
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class VaeConfig:
    latent_channels: int = 16
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.3611
    shift_factor: float = 0.1159


def tiny_config():
    return VaeConfig(block_out_channels=(64, 128))


def _resnet_shapes(s, p, cin, cout):
    s[p + "norm1.weight"] = (cin,); s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3); s[p + "conv1.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,); s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3); s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "conv_shortcut.weight"] = (cout, cin, 1, 1); s[p + "conv_shortcut.bias"] = (cout,)


def param_shapes(cfg: VaeConfig) -> Dict[str, tuple]:
    s: Dict[str, tuple] = {}
    chans = list(reversed(cfg.block_out_channels))
    cmid = chans[0]
    s["decoder.conv_in.weight"] = (cmid, cfg.latent_channels, 3, 3); s["decoder.conv_in.bias"] = (cmid,)
    _resnet_shapes(s, "decoder.mid_block.resnets.0.", cmid, cmid)
    _resnet_shapes(s, "decoder.mid_block.resnets.1.", cmid, cmid)
    a = "decoder.mid_block.attentions.0."
    s[a + "group_norm.weight"] = (cmid,); s[a + "group_norm.bias"] = (cmid,)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        s[a + n + ".weight"] = (cmid, cmid); s[a + n + ".bias"] = (cmid,)
    prev = cmid
    for b, co in enumerate(chans):
        for r in range(cfg.layers_per_block + 1):
            _resnet_shapes(s, f"decoder.up_blocks.{b}.resnets.{r}.", prev if r == 0 else co, co)
        if b != len(chans) - 1:
            s[f"decoder.up_blocks.{b}.upsamplers.0.conv.weight"] = (co, co, 3, 3)
            s[f"decoder.up_blocks.{b}.upsamplers.0.conv.bias"] = (co,)
        prev = co
    s["decoder.conv_norm_out.weight"] = (chans[-1],); s["decoder.conv_norm_out.bias"] = (chans[-1],)
    s["decoder.conv_out.weight"] = (cfg.out_channels, chans[-1], 3, 3); s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


def init_weights(cfg: VaeConfig, seed: int = 0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in param_shapes(cfg).items():
        if "norm" in k and k.endswith("weight"):
            sd[k] = (1.0 + 0.05 * torch.randn(shp, generator=g)).to(dtype)
        elif len(shp) == 4:
            fan_in = shp[1] * shp[2] * shp[3]
            sd[k] = (torch.randn(shp, generator=g) / fan_in ** 0.5).to(dtype)
        elif len(shp) == 2:
            sd[k] = (torch.randn(shp, generator=g) / shp[1] ** 0.5).to(dtype)
        else:
            sd[k] = (0.02 * torch.randn(shp, generator=g)).to(dtype)
    return sd


def _gn(sd, p, x, groups):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps=1e-6)


def _resnet(sd, p, x, groups):
    h = F.conv2d(F.silu(_gn(sd, p + "norm1", x, groups)), sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(sd, p + "norm2", h, groups)), sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if p + "conv_shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + "conv_shortcut.weight"], sd[p + "conv_shortcut.bias"])
    return x + h   # output_scale_factor = 1


def _mid_attention(sd, p, x, groups):
    B, C, H, W = x.shape
    res = x
    t = _gn(sd, p + "group_norm", x.view(B, C, H * W), groups).transpose(1, 2)   # [B, HW, C]
    q = F.linear(t, sd[p + "to_q.weight"], sd[p + "to_q.bias"])[:, None]
    k = F.linear(t, sd[p + "to_k.weight"], sd[p + "to_k.bias"])[:, None]
    v = F.linear(t, sd[p + "to_v.weight"], sd[p + "to_v.bias"])[:, None]
    o = F.scaled_dot_product_attention(q, k, v)[:, 0].to(q.dtype)
    o = F.linear(o, sd[p + "to_out.0.weight"], sd[p + "to_out.0.bias"])
    return o.transpose(1, 2).reshape(B, C, H, W) + res   # residual_connection, rescale_output_factor = 1


def decode(sd, cfg: VaeConfig, z: torch.Tensor) -> torch.Tensor:
    """z [B, latent, h, w] (already z/scaling + shift) -> [B, 3, 8h, 8w]"""
    g = cfg.norm_groups
    chans = list(reversed(cfg.block_out_channels))
    x = F.conv2d(z, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _resnet(sd, "decoder.mid_block.resnets.0.", x, g)
    x = _mid_attention(sd, "decoder.mid_block.attentions.0.", x, g)
    x = _resnet(sd, "decoder.mid_block.resnets.1.", x, g)
    for b in range(len(chans)):
        for r in range(cfg.layers_per_block + 1):
            x = _resnet(sd, f"decoder.up_blocks.{b}.resnets.{r}.", x, g)
        if b != len(chans) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{b}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{b}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, g))
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def latents_to_image(sd, cfg: VaeConfig, packed: torch.Tensor, h: int, w: int):
    """FluxPipeline tail: packed [B, (h/2)(w/2), 64] -> (decoded [B,3,H,W], uint8 [B,H,W,3])."""
    from .flux_ref import unpack_latents
    z = unpack_latents(packed, h, w)
    z = (z / cfg.scaling_factor) + cfg.shift_factor
    img = decode(sd, cfg, z)
    den = (img / 2 + 0.5).clamp(0, 1)                       # VaeImageProcessor.denormalize
    u8 = (den.float().permute(0, 2, 3, 1) * 255).round().to(torch.uint8)   # pt_to_numpy + numpy_to_pil
    return img, u8

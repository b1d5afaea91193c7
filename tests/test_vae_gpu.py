"""GPU parity of the VAE decoder kernels / engine (td_conv3x3_nhwc_bf16, td_groupnorm_nhwc_bf16, td_vae_decode)
against the CPU oracle (oracle/vae_ref.py: F.conv2d / F.group_norm / SDPA in bf16 = the reference's arithmetic).

Tolerances: conv / groupnorm single ops <= 2^-6 of the output scale (bf16 output, fp32 accumulate, different
summation order); the tiny 2-block decoder end to end: relative RMSE <= 3e-2 vs the bf16 oracle and uint8 pixels
within 1e-2 RMSE on the [0,1] scale (the north-star pixel bar).
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import flux_ref as R
from oracle import vae_ref as V

pytestmark = pytest.mark.gpu


def _close(got, ref, tol):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max() / ref.abs().max()
    assert err < tol, f"rel-to-scale err {err:.3e}"


@pytest.mark.parametrize("H,W,Cin,Cout,up,res", [(16, 16, 64, 64, False, False), (24, 40, 128, 256, False, True),
                                                 (32, 32, 64, 128, True, False), (18, 22, 64, 8, False, False)])
def test_conv3x3_implicit_gemm(hip, H, W, Cin, Cout, up, res):
    g = torch.Generator().manual_seed(H * W + Cin)
    Hin, Win = (H // 2, W // 2) if up else (H, W)
    x = torch.randn(1, Cin, Hin, Win, generator=g).bfloat16()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5).bfloat16()
    b = torch.randn(Cout, generator=g).bfloat16()
    r = torch.randn(1, Cout, H, W, generator=g).bfloat16() if res else None
    xin = F.interpolate(x.float(), scale_factor=2.0, mode="nearest").bfloat16() if up else x
    ref = F.conv2d(xin.float(), w.float(), b.float(), padding=1).bfloat16()
    if res:
        ref = (ref.float() + r.float()).bfloat16()
    nhwc = lambda t: t[0].permute(1, 2, 0).reshape(-1, t.shape[1]).contiguous()
    wp = hip.conv3x3_pack_weight(w.cuda())
    y = hip.conv3x3_nhwc(nhwc(x).cuda(), wp, b.cuda(), H, W, Cout, res=nhwc(r).cuda() if res else None, upsample2x=up)
    torch.cuda.synchronize()
    _close(y, nhwc(ref), 2.0 ** -6)


@pytest.mark.parametrize("P,C,silu", [(256, 64, True), (4096, 128, False), (1000, 512, True)])
def test_groupnorm_silu_nhwc(hip, P, C, silu):
    g = torch.Generator().manual_seed(P + C)
    x = (torch.randn(P, C, generator=g) * 2 + 0.3).bfloat16()
    ga, be = (1 + 0.1 * torch.randn(C, generator=g)).bfloat16(), (0.1 * torch.randn(C, generator=g)).bfloat16()
    ref = F.group_norm(x.t()[None], 32, ga, be, eps=1e-6)
    if silu:
        ref = F.silu(ref)
    y = hip.groupnorm_nhwc(x.cuda(), ga.cuda(), be.cuda(), 32, 1e-6, silu)
    torch.cuda.synchronize()
    _close(y, ref[0].t(), 2.0 ** -6)


def _tiny_vae(seed):
    from thinkdiff.models.flux_vae import AutoencoderKLConfig, AutoencoderKLDecoder
    cfg = V.tiny_config()
    sd = V.init_weights(cfg, seed=seed)
    m = AutoencoderKLDecoder(AutoencoderKLConfig(block_out_channels=cfg.block_out_channels), max_latent_size=(16, 16))
    m.load_state_dict(sd)
    return cfg, sd, m


@pytest.mark.parametrize("h,w", [(8, 8), (16, 12)])
def test_vae_decode_matches_oracle(hip, h, w):
    cfg, sd, m = _tiny_vae(seed=h)
    g = torch.Generator().manual_seed(w)
    packed = (torch.randn(1, (h // 2) * (w // 2), 64, generator=g) * 0.8).bfloat16()
    ref_img, ref_u8 = V.latents_to_image(sd, cfg, packed, h, w)
    img = m.decode_packed(packed[0].cuda(), h, w, output_type="pt")
    u8 = m.decode_packed(packed[0].cuda(), h, w, output_type="np")
    torch.cuda.synchronize()
    assert img.shape == (3, 2 * h, 2 * w) and u8.shape == (2 * h, 2 * w, 3) and u8.dtype == torch.uint8
    rel = float((img.float().cpu() - ref_img[0].float()).pow(2).mean().sqrt() / ref_img.float().pow(2).mean().sqrt())
    px = float(((u8.float().cpu() - ref_u8[0].float()) / 255).pow(2).mean().sqrt())
    print(f"vae tiny {h}x{w}: rel-RMSE {rel:.4f}, pixel RMSE {px:.5f}")
    assert rel < 3e-2 and px < 1e-2


def test_pipeline_returns_pil_image(hip):
    """FluxPipelineRewritePrompt(...).images[0] is a PIL image once a VAE is attached (the drivers .save() it)."""
    from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig
    cfgv, sdv, vae = _tiny_vae(seed=3)
    fc = R.tiny_config(num_layers=1, num_single_layers=1)
    tr = FluxTransformer2DModel(FluxTransformerConfig(num_layers=1, num_single_layers=1, num_attention_heads=fc.num_attention_heads,
                                                      joint_attention_dim=fc.joint_attention_dim, pooled_projection_dim=fc.pooled_projection_dim),
                                max_img_tokens=64, max_txt_tokens=32, max_steps=4)
    tr.load_state_dict(R.init_weights(fc, seed=1))
    pipe = FluxPipelineRewritePrompt(transformer=tr, vae=vae)
    pipe.vae_scale_factor = 4   # 2-block tiny VAE: image = 2 x latent, latent = 2 x packed grid
    g = torch.Generator().manual_seed(0)
    out = pipe(prompt_embeds=torch.randn(1, 16, fc.joint_attention_dim, generator=g).bfloat16().cuda(),
               pooled_prompt_embeds=torch.randn(1, fc.pooled_projection_dim, generator=g).bfloat16().cuda(),
               height=32, width=32, num_inference_steps=2, guidance_scale=3.5)
    assert out.images[0].size == (32, 32) and out.images[0].mode == "RGB"


def test_vae_full_architecture_small_image(hip):
    """The FLUX.1 VAE decoder architecture at full width (block_out_channels 128/256/512/512, 2 layers per block, the
    512-channel single-head mid-block attention) on a 16x16 latent -> 128x128 image, against the oracle."""
    from thinkdiff.models.flux_vae import AutoencoderKLConfig, AutoencoderKLDecoder
    cfg = V.VaeConfig()
    sd = V.init_weights(cfg, seed=3)
    m = AutoencoderKLDecoder(AutoencoderKLConfig(), max_latent_size=(16, 16))
    m.load_state_dict(sd)
    h = w = 16
    g = torch.Generator().manual_seed(11)
    packed = (torch.randn(1, (h // 2) * (w // 2), 64, generator=g) * 0.8).bfloat16()
    ref_img, ref_u8 = V.latents_to_image(sd, cfg, packed, h, w)
    img = m.decode_packed(packed[0].cuda(), h, w, output_type="pt")
    u8 = m.decode_packed(packed[0].cuda(), h, w, output_type="np")
    torch.cuda.synchronize()
    assert img.shape == (3, 8 * h, 8 * w) and u8.shape == (8 * h, 8 * w, 3)
    rel = float((img.float().cpu() - ref_img[0].float()).pow(2).mean().sqrt() / ref_img.float().pow(2).mean().sqrt())
    px = float(((u8.float().cpu() - ref_u8[0].float()) / 255).pow(2).mean().sqrt())
    print(f"vae full architecture 16x16 latent: rel-RMSE {rel:.4f}, pixel RMSE {px:.5f}")
    assert rel < 3e-2 and px < 1e-2

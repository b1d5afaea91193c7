"""Probe, don't assume: is `diffusers` importable where the tests run (the build container, the GPU box)?

oracle/flux_ref.py and oracle/vae_ref.py restate the diffusers 0.31.0 graphs the reference calls (requirements.txt:34) and are
"parity unpinned" because the package is absent from /root/reference and from this image.  If a box does have it, this test
pins the restatement against the real modules on a tiny random configuration (fp32, CPU: exact arithmetic of the same graph) --
the transformer forward, the scheduler's sigma schedule and the VAE decoder -- and says so; if not, it records the absence and
skips.  Either way the log carries one line `diffusers: <version | absent>`.  Nothing of the reference is involved.
"""
import pytest
import torch


def _probe():
    try:
        import diffusers
        return diffusers
    except Exception as e:  # noqa: BLE001
        print(f"diffusers: absent ({type(e).__name__}: {e})")
        return None


def test_diffusers_probe_and_pin_if_present():
    d = _probe()
    if d is None:
        pytest.skip("diffusers: absent -- oracle/flux_ref.py and oracle/vae_ref.py stay 'parity unpinned'")
    print(f"diffusers: {d.__version__}")
    try:
        _pin(d)
    except (TypeError, AttributeError, ImportError) as e:        # a diffusers whose constructors differ from 0.31.0: say so, do not guess
        pytest.skip(f"diffusers {d.__version__} is present but its API differs from 0.31.0 ({type(e).__name__}: {e}); oracle stays unpinned")


def _pin(d):
    from oracle import flux_ref as R
    cfg = R.tiny_config(num_layers=2, num_single_layers=2)
    sd = R.init_weights(cfg, seed=5, dtype=torch.float32)
    m = d.FluxTransformer2DModel(patch_size=1, in_channels=cfg.in_channels, num_layers=cfg.num_layers, num_single_layers=cfg.num_single_layers,
                                 attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads,
                                 joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim,
                                 guidance_embeds=True, axes_dims_rope=tuple(cfg.axes_dims_rope)).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    g = torch.Generator().manual_seed(0)
    h2 = w2 = 6
    T = 11
    lat = torch.randn(1, h2 * w2, 64, generator=g)
    pe = torch.randn(1, T, cfg.joint_attention_dim, generator=g)
    pool = torch.randn(1, cfg.pooled_projection_dim, generator=g)
    t, gd = torch.tensor([0.61]), torch.tensor([3.5])
    img_ids, txt_ids = R.latent_image_ids(h2, w2), torch.zeros(T, 3)
    with torch.no_grad():
        want = m(hidden_states=lat, encoder_hidden_states=pe, pooled_projections=pool, timestep=t, img_ids=img_ids, txt_ids=txt_ids,
                 guidance=gd, return_dict=False)[0]
        got = R.transformer_forward(sd, cfg, lat, pe, pool, t, img_ids, txt_ids, gd)
    err = float((got - want).abs().max() / want.abs().max())
    print(f"oracle/flux_ref.transformer_forward vs diffusers.FluxTransformer2DModel (fp32, tiny): max rel err {err:.2e}")
    assert err < 1e-4
    sch = d.FlowMatchEulerDiscreteScheduler(shift=3.0, use_dynamic_shifting=True, base_shift=0.5, max_shift=1.15, base_image_seq_len=256,
                                            max_image_seq_len=4096)
    import numpy as np
    n, S = 28, 4096
    sch.set_timesteps(sigmas=np.linspace(1.0, 1 / n, n), mu=R.calculate_shift(S))
    assert np.allclose(sch.sigmas.numpy(), R.make_sigmas(n, S), atol=1e-6)
    from oracle import vae_ref as V
    vcfg = V.tiny_config()
    vsd = V.init_weights(vcfg, seed=2, dtype=torch.float32)
    vae = d.AutoencoderKL(in_channels=3, out_channels=vcfg.out_channels, latent_channels=vcfg.latent_channels,
                          block_out_channels=tuple(vcfg.block_out_channels), layers_per_block=vcfg.layers_per_block,
                          norm_num_groups=vcfg.norm_groups, down_block_types=("DownEncoderBlock2D",) * len(vcfg.block_out_channels),
                          up_block_types=("UpDecoderBlock2D",) * len(vcfg.block_out_channels), use_quant_conv=False, use_post_quant_conv=False).eval()
    missing, unexpected = vae.load_state_dict(vsd, strict=False)
    assert not unexpected and all(k.startswith("encoder.") for k in missing), (missing[:4], unexpected[:4])
    z = torch.randn(1, vcfg.latent_channels, 8, 8, generator=g)
    with torch.no_grad():
        verr = float((V.decode(vsd, vcfg, z) - vae.decode(z, return_dict=False)[0]).abs().max())
    print(f"oracle/vae_ref.decode vs diffusers.AutoencoderKL.decode (fp32, tiny): max abs err {verr:.2e}")
    assert verr < 1e-4


@pytest.mark.gpu
def test_diffusers_probe_on_the_gpu_box():
    """The same probe under `-m gpu`, so that the GPU box's log says which it is."""
    d = _probe()
    if d is None:
        pytest.skip("diffusers: absent on the GPU box")
    print(f"diffusers: {d.__version__} on the GPU box -- the pinning test above applies there")

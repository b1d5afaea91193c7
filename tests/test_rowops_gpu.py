"""GPU parity of the HBM-bound row kernels and the aligner against the CPU oracle / golden vectors.

Tolerances: these kernels replicate the reference's rounding points, so they must agree with the bf16
oracle to within one bf16 ulp of the value (rtol 2^-7, atol noted per test); index kernels are bit-exact.
"""
import os

import pytest
import torch

from oracle import aligner_ref as A
from oracle import flux_ref as R

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _close(got, ref, rtol=2.0 ** -7, atol=1e-2):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape
    bad = (got - ref).abs() > atol + rtol * ref.abs()
    assert not bad.any(), f"{int(bad.sum())} / {bad.numel()} mismatches, max abs err {(got - ref).abs().max():.4g}"


@pytest.mark.parametrize("rows,D,split", [(449, 3072, 193), (7, 512, 0), (300, 4096, 300)])
def test_layernorm_modulate(hip, rows, D, split):
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, D, generator=g) * 3 + 0.5).bfloat16()
    mods = [torch.randn(D, generator=g).bfloat16() * 0.3 for _ in range(4)]
    shA, scA, shB, scB = mods
    ln = torch.nn.functional.layer_norm(x, (D,), eps=1e-6)
    ref = torch.cat([ln[:split] * (1 + scA) + shA, ln[split:] * (1 + scB) + shB])
    y = hip.norm_rows(x.cuda(), split=split, shiftA=shA.cuda(), scaleA=scA.cuda(), shiftB=shB.cuda(), scaleB=scB.cuda())
    torch.cuda.synchronize()
    _close(y, ref, atol=2e-2)


def test_rmsnorm_weight(hip):
    """T5LayerNorm / Qwen2RMSNorm form."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(130, 3584, generator=g).bfloat16() * 2
    w = (1 + 0.1 * torch.randn(3584, generator=g)).bfloat16()
    y = hip.norm_rows(x.cuda(), rms=True, w=w.cuda())
    torch.cuda.synchronize()
    _close(y, A.t5_layer_norm(x, w), atol=1e-2)


@pytest.mark.parametrize("S,H,split", [(97, 4, 24), (449, 24, 193)])
def test_qk_rmsnorm_rope_flux(hip, S, H, split):
    g = torch.Generator().manual_seed(S)
    D = H * 128
    qkv = torch.randn(S, 3 * D, generator=g).bfloat16()
    w = [(1 + 0.1 * torch.randn(128, generator=g)).bfloat16() for _ in range(4)]  # added_q, added_k, q, k
    ids = torch.cat([torch.zeros(split, 3), R.latent_image_ids(16, 16)[: S - split] + torch.tensor([0.0, 3.0, 5.0])])
    cos, sin = R.rope_tables(ids)
    d = qkv.cuda()
    hip.qk_norm_rope(d, H, H, 0, D, cos.cuda(), sin.cuda(), split=split, wqA=w[0].cuda(), wkA=w[1].cuda(), wqB=w[2].cuda(), wkB=w[3].cuda())
    torch.cuda.synchronize()
    def ref_part(cols, wa, wb):
        x = qkv[:, cols].reshape(1, S, H, 128).transpose(1, 2)          # [1,H,S,128]
        n = torch.cat([R.rms_norm(x[:, :, :split], wa), R.rms_norm(x[:, :, split:], wb)], dim=2)
        return R.apply_rotary_emb(n, cos, sin).transpose(1, 2).reshape(S, D)
    _close(d[:, :D], ref_part(slice(0, D), w[0], w[2]), atol=2e-2)
    _close(d[:, D:2 * D], ref_part(slice(D, 2 * D), w[1], w[3]), atol=2e-2)
    assert torch.equal(d[:, 2 * D:].cpu(), qkv[:, 2 * D:])  # v untouched


def test_rope_table_and_sincos(hip):
    ids = torch.cat([torch.zeros(5, 3), R.latent_image_ids(64, 64)])
    cos, sin = hip.flux_rope_table(ids.cuda())
    rc, rs = R.rope_tables(ids)
    torch.cuda.synchronize()
    assert (cos.cpu() - rc).abs().max() < 2e-6 and (sin.cpu() - rs).abs().max() < 2e-6
    t = torch.tensor([1000.0, 968.0, 3504.0, 12.25, 0.0])
    e = hip.timestep_sincos(t.cuda())
    torch.cuda.synchronize()
    # fp32 device sin/cos of arguments up to 3504 rad: absolute error ~1e-4 before the bf16 rounding
    assert (e.float().cpu() - R.timestep_proj(t)).abs().max() < 1e-2


def test_euler_pack_unpack_pool_bit_exact(hip):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4096, 64, generator=g).bfloat16()
    v = torch.randn(4096, 64, generator=g).bfloat16()
    # the scheduler's own statement, tensor form ([ext] diffusers 0.31.0 FlowMatchEulerDiscreteScheduler.step): sigmas is an fp32 tensor,
    # model_output a bf16 tensor, so `(sigma_next - sigma) * model_output` is a bf16 op (scalar cast to bf16, product rounded to bf16)
    for s0, s1 in ((0.9731, 0.9615), (1.0, 0.98828), (0.0357, 0.0)):
        sig = torch.tensor([s0, s1], dtype=torch.float32)
        prod = (sig[1] - sig[0]) * v
        assert prod.dtype == torch.bfloat16
        ref = (x.to(torch.float32) + prod).to(v.dtype)
        out = hip.euler_step(x.cuda().clone(), v.cuda(), float(sig[1] - sig[0]))
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref)
        # torch on the GPU (where the reference runs configs 2-5) applies the same promotion to a 0-dim device tensor
        ref_gpu = (x.cuda().to(torch.float32) + (sig.cuda()[1] - sig.cuda()[0]) * v.cuda()).to(v.dtype)
        assert torch.equal(ref_gpu.cpu(), ref)
    lat = torch.randn(16, 32, 48, generator=g).bfloat16()
    p = hip.flux_pack_latents(lat.cuda())
    assert torch.equal(p.cpu(), R.pack_latents(lat[None])[0])
    u = hip.flux_unpack_latents(p, 16, 32, 48)
    assert torch.equal(u.cpu(), lat)
    u2 = hip.flux_unpack_latents(p, 16, 32, 48, 0.3611, 0.1159)
    assert torch.equal(u2.cpu(), (lat / 0.3611) + 0.1159)   # two bf16 torch ops, each rounding ([ext] pipeline_flux.py before vae.decode)
    tok = torch.randn(257, 1408, generator=g).bfloat16()
    pooled = hip.cls_avgpool2(tok.cuda())
    torch.cuda.synchronize()
    _close(pooled, A.pool_vision_tokens(tok[None])[0], rtol=2.0 ** -8, atol=1e-6)


@pytest.mark.parametrize("name", ["aligner_clip_bf16.pt", "aligner_lvlm7b_bf16.pt"])
def test_aligner_matches_golden(hip, name):
    """Golden output = the reference's nn.Sequential(Linear, GELU, Linear, T5LayerNorm) in bf16 on CPU."""
    fx = torch.load(os.path.join(G, name), weights_only=False)
    sd = {k: v.bfloat16() for k, v in A.init_weights(fx["mm_hidden"], fx["hidden"], seed=fx["seed"], dtype=torch.float32).items()}
    g = torch.Generator().manual_seed(fx["seed"] + 1)
    x = torch.randn(1, fx["tokens"], fx["mm_hidden"], generator=g).bfloat16()
    from thinkdiff.models.blip_vision_t5_decoder import BlipVisionT5DecoderForConditionalGeneration
    m = BlipVisionT5DecoderForConditionalGeneration(mm_hidden_size=fx["mm_hidden"], vision_downsample_factor=2 if fx["tokens"] == 257 else None)
    m.load_state_dict(sd)
    y = m.forward_encoder(image_embeds=x.cuda())
    torch.cuda.synchronize()
    assert y.shape == fx["expected"].shape
    # output of an RMS-normalised 4096-vector: |y| ~ 1; two chained bf16 GEMMs with different summation order
    err = (y.float().cpu() - fx["expected"].float())
    rel = float(err.pow(2).mean().sqrt() / fx["expected"].float().pow(2).mean().sqrt())
    assert rel < 1e-2 and err.abs().max() < 0.15, (rel, err.abs().max())

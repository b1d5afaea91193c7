"""torch.ops.thinkdiff_hip.* on the MI355X: each op is the C-ABI kernel (bit-identical to the direct binding), allocates its
output on the current stream and raises RuntimeError for arguments the kernel rejects."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_ops_match_the_c_abi_binding(hip):
    import thinkdiff.ops  # noqa: F401
    O = torch.ops.thinkdiff_hip
    g = torch.Generator().manual_seed(0)
    x = torch.randn(300, 256, generator=g).bfloat16().cuda()
    w = (torch.randn(512, 256, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(512, generator=g).bfloat16().cuda()
    y = O.linear(x, w, b, hip.ACT_GELU_TANH, None, None)
    assert torch.equal(y, hip.linear(x, w, b, hip.ACT_GELU_TANH)) and y.shape == (300, 512)
    ref = torch.nn.functional.gelu(torch.nn.functional.linear(x.float(), w.float(), b.float()).bfloat16().float(), approximate="tanh")
    assert float((y.float() - ref).abs().max() / ref.abs().max()) < 2.0 ** -6
    # attention on a fused projection (q | k | v at columns 0 / 256 / 512 of one buffer), joint (non-causal) form
    qkv = torch.randn(1, 200, 3 * 256, generator=g).bfloat16().cuda()
    q, k, v = qkv[..., :256], qkv[..., 256:512], qkv[..., 512:]
    o = O.attention(q, k, v, 2, 2, 128 ** -0.5, False)
    heads = lambda t: t.reshape(1, 200, 2, 128).transpose(1, 2).float()
    sd = torch.nn.functional.scaled_dot_product_attention(heads(q), heads(k), heads(v)).transpose(1, 2).reshape(1, 200, 256)
    assert float((o.float() - sd).abs().max()) < 2.0 ** -6 * float(sd.abs().max()) + 1e-3
    n = O.norm_rows(x.repeat(1, 2).contiguous(), True, 1e-6, None, 0, None, None, None, None)
    assert torch.equal(n, hip.norm_rows(x.repeat(1, 2).contiguous(), rms=True))
    lat = torch.randn(16, 32, 48, generator=g).bfloat16().cuda()
    p = O.flux_pack_latents(lat)
    assert torch.equal(O.flux_unpack_latents(p, 16, 32, 48, 1.0, 0.0), lat)
    xv, vv = torch.randn(4096, generator=g).bfloat16().cuda(), torch.randn(4096, generator=g).bfloat16().cuda()
    # scheduler.step's product: a 0-dim fp32 tensor ON THE DEVICE (scheduler.sigmas lives there) x a bf16 tensor = a bf16 op, scalar cast to bf16
    # (a 0-dim CPU tensor would instead enter the kernel as an fp32 scalar -- not what the pipeline does)
    want = (xv.float() + torch.tensor(-0.0116, device="cuda") * vv).bfloat16()
    assert O.euler_step_(xv, vv, -0.0116) is xv and torch.equal(xv, want)
    ids = O.sample_top_p(torch.randn(4, 4096, generator=g).bfloat16().cuda(), 0.0, 0.9, 1, 0)
    assert ids.dtype == torch.int32 and ids.shape == (4,)
    with pytest.raises(RuntimeError):
        O.linear(x[:, :96].contiguous(), w[:, :96].contiguous(), None, 0, None, None)       # K = 96 is not a multiple of 64: the kernel refuses


def test_mixed_device_and_short_arguments_are_rejected(hip):
    """The dispatcher picks the GPU kernel as soon as ANY argument is on the GPU; a host bias / gate / residual, a wrong dtype or a
    short tensor must raise instead of reaching the kernel as a host pointer or an out-of-bounds read."""
    import thinkdiff.ops  # noqa: F401
    O = torch.ops.thinkdiff_hip
    x = torch.zeros(128, 256, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(512, 256, dtype=torch.bfloat16, device="cuda")
    b = torch.zeros(512, dtype=torch.bfloat16, device="cuda")
    for bad in (dict(bias=b.cpu()), dict(bias=b.float()), dict(bias=b[:100]), dict(gate=b.cpu()), dict(gate=b[:8]),
                dict(res=torch.zeros(128, 512, dtype=torch.bfloat16)), dict(res=torch.zeros(64, 512, dtype=torch.bfloat16, device="cuda"))):
        kw = dict(bias=b, gate=None, res=None)
        kw.update(bad)
        with pytest.raises(RuntimeError):
            O.linear(x, w, kw["bias"], 0, kw["gate"], kw["res"])
    w0, w2 = torch.zeros(512, 256, dtype=torch.bfloat16, device="cuda"), torch.zeros(512, 512, dtype=torch.bfloat16, device="cuda")
    v = torch.zeros(512, dtype=torch.bfloat16, device="cuda")                # hidden = 512 (the T5-RMSNorm kernel takes multiples of 512)
    O.aligner_mlp2x(x, w0, v, w2, v, v, 1e-6, False)                      # well-formed call passes
    for args in ((x, w0, v.cpu(), w2, v, v), (x, w0, v, w2, v[:128], v), (x, w0, v, w2, v, v.float()), (x, w0[:128], v, w2, v, v),
                 (x, w0, v, w2[:, :128].contiguous(), v, v)):
        with pytest.raises(RuntimeError):
            O.aligner_mlp2x(*args, 1e-6, False)
    q = torch.zeros(1, 64, 256, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        O.attention(q, q, q, 4, 4, 0.088, False)                           # 4 heads need 512 columns
    with pytest.raises(RuntimeError):
        O.attention(q, q.cpu(), q.cpu(), 2, 2, 0.088, False)
    with pytest.raises(RuntimeError):
        O.attention(q, q.repeat(2, 1, 1), q.repeat(2, 1, 1), 2, 2, 0.088, False)   # k/v batch differs from q's
    with pytest.raises(RuntimeError):
        O.norm_rows(x.repeat(1, 2).contiguous(), False, 1e-6, None, 0, v.cpu(), v, None, None)


def test_engine_ops_are_the_engine_and_check_extents(hip):
    """flux_forward_ / flux_denoise_ / flux_denoise_multi_ / vae_decode_u8 / attention_fp8: the product models run the denoise loop through
    these ops; each is the C-ABI call (bit-identical to the direct ctypes binding) and refuses tensors that do not match what the prepared
    context expects (wrong row count, host tensor, wrong dtype, null handle)."""
    import ctypes
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import flux_ref as R
    import thinkdiff.ops  # noqa: F401
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig, effective_scalar
    O = torch.ops.thinkdiff_hip
    cfg = R.tiny_config(num_layers=1, num_single_layers=1)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=1, num_single_layers=1, num_attention_heads=cfg.num_attention_heads,
                                                     joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                               max_img_tokens=256, max_txt_tokens=64, max_steps=4)
    m.load_state_dict(R.init_weights(cfg, seed=4))
    g = torch.Generator().manual_seed(1)
    h2 = w2 = 8
    T, n = 20, 3
    lat = torch.randn(h2 * w2, 64, generator=g).bfloat16().cuda()
    m.set_condition(torch.randn(T, cfg.joint_attention_dim, generator=g).bfloat16().cuda(), torch.randn(cfg.pooled_projection_dim, generator=g).bfloat16().cuda(),
                    R.latent_image_ids(h2, w2))
    sig = [float(s) for s in R.make_sigmas(n, h2 * w2)]
    m.set_timesteps([effective_scalar(s * 1000.0, torch.bfloat16) for s in sig[:-1]], 3500.0)
    L, h = hip.lib(), int(m._h.value)
    shp = [ctypes.c_int() for _ in range(4)]
    hip.check(L.td_flux_prepared_shape(m._h, *[ctypes.byref(s) for s in shp]))
    assert [s.value for s in shp] == [h2 * w2, T, 64, n]
    # forward: op == ctypes
    v_op = O.flux_forward_(h, lat, 1, torch.empty_like(lat))
    v_c = torch.empty_like(lat)
    hip.check(L.td_flux_forward(m._h, hip.ptr(lat), 1, hip.ptr(v_c), hip.stream_ptr()))
    assert torch.equal(v_op, v_c) and torch.isfinite(v_op.float()).all() and float(v_op.float().abs().max()) > 0
    # denoise: op == ctypes == the model method; multi == single
    a, b, c = lat.clone(), lat.clone(), lat.clone()
    assert O.flux_denoise_(h, a, sig) is a
    arr = (ctypes.c_float * (n + 1))(*sig)
    hip.check(L.td_flux_denoise(m._h, hip.ptr(b), ctypes.cast(arr, ctypes.c_void_p), n, hip.stream_ptr()))
    torch.cuda.synchronize()
    O.flux_denoise_multi_([h], [c], sig, [int(torch.cuda.current_stream().cuda_stream)])
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, m.denoise(lat.clone(), sig))
    # rejected arguments
    for bad in (lat[:-1].contiguous(), lat.cpu(), lat.float(), lat.t().contiguous().t()):
        with pytest.raises(RuntimeError):
            O.flux_forward_(h, bad, 0, torch.empty_like(lat))
    with pytest.raises(RuntimeError):
        O.flux_forward_(h, lat, n, torch.empty_like(lat))                       # step outside the prepared timesteps
    with pytest.raises(RuntimeError):
        O.flux_forward_(0, lat, 0, torch.empty_like(lat))
    with pytest.raises(RuntimeError):
        O.flux_denoise_(h, lat.clone(), [1.0])
    with pytest.raises(RuntimeError):
        O.flux_denoise_multi_([h, h], [lat.clone()], sig, [0])
    # the 8-bit attention op == the ctypes binding
    qkv = torch.randn(300, 3 * 256, generator=g).bfloat16().cuda()
    o8 = O.attention_fp8(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], 2, 128 ** -0.5)
    ref8 = hip.attention_fp8(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], torch.empty(300, 256, dtype=torch.bfloat16, device="cuda"), 2)
    assert torch.equal(o8, ref8)
    with pytest.raises(RuntimeError):
        O.attention_fp8(qkv[:, :256], qkv[:, 256:512].cpu(), qkv[:, 512:].cpu(), 2, 0.088)


def test_vae_decode_op(hip):
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import vae_ref as V
    import thinkdiff.ops  # noqa: F401
    from thinkdiff.models.flux_vae import AutoencoderKLConfig, AutoencoderKLDecoder
    O = torch.ops.thinkdiff_hip
    cfg = V.tiny_config()
    vae = AutoencoderKLDecoder(AutoencoderKLConfig(block_out_channels=cfg.block_out_channels), max_latent_size=(16, 16))
    vae.load_state_dict(V.init_weights(cfg, seed=5))
    h, w = 16, 8
    packed = (torch.randn((h // 2) * (w // 2), 64, generator=torch.Generator().manual_seed(0)) * 0.8).bfloat16().cuda()
    img = O.vae_decode_u8(int(vae._h.value), packed, h, w, float(vae.config.scaling_factor), float(vae.config.shift_factor))
    assert img.dtype == torch.uint8 and img.shape == (vae.upscale * h, vae.upscale * w, 3)
    u8 = torch.empty_like(img)
    hip.check(hip.lib().td_vae_decode(vae._h, hip.ptr(packed), h, w, vae.config.scaling_factor, vae.config.shift_factor, u8.data_ptr(), None, hip.stream_ptr()))
    assert torch.equal(img, u8) and torch.equal(img, vae.decode_packed(packed, h, w, output_type="np")) and int(img.max()) > int(img.min())
    for bad in (packed[:-1].contiguous(), packed[:, :32].contiguous(), packed.cpu(), packed.float()):
        with pytest.raises(RuntimeError):
            O.vae_decode_u8(int(vae._h.value), bad, h, w, 1.0, 0.0)
    with pytest.raises(RuntimeError):
        O.vae_decode_u8(0, packed, h, w, 1.0, 0.0)

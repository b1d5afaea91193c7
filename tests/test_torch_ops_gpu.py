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
    want = (xv.float() + (-0.0116) * vv.float()).bfloat16()
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

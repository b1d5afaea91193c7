"""fp8 operand path (BASELINE config 5): quantisers bit-exact against torch.float8_e4m3fn, the fp8 GEMM against an
fp64 contraction of the very same quantised operands (so the only difference is fp32 summation order), and the
end-to-end quantisation error against the bf16 GEMM within the stated bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _quant_ref(x):
    """per-row dynamic e4m3 quantisation exactly as csrc/elementwise.hip states it (fp32 arithmetic)."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    s = torch.where(amax > 0, amax * (1.0 / 448.0), torch.ones_like(amax))
    q = (xf * (1.0 / s)[:, None]).to(torch.float8_e4m3fn)
    return q, s


def test_quant_rows_bit_exact(hip):
    from thinkdiff import _hip
    torch.manual_seed(0)
    x = (torch.randn(77, 3072, device="cuda") * torch.logspace(-3, 2, 77, device="cuda")[:, None]).bfloat16()
    x[5] = 0
    x[6, 17] = 3.0e4
    q, s = _hip.quant_rows_fp8(x)
    torch.cuda.synchronize()
    qr, sr = _quant_ref(x)
    assert torch.equal(s, sr)
    assert torch.equal(q, qr.view(torch.uint8))
    assert s[5] == 1.0 and not q[5].any()


def test_norm_rows_quant_matches_norm_then_quant(hip):
    from thinkdiff import _hip
    torch.manual_seed(1)
    S, D, split = 301, 3072, 41
    x = torch.randn(S, D, device="cuda").bfloat16()
    mods = [(torch.randn(D, device="cuda") * 0.3).bfloat16() for _ in range(4)]
    y = _hip.norm_rows(x, rms=False, eps=1e-6, split=split, shiftA=mods[0], scaleA=mods[1], shiftB=mods[2], scaleB=mods[3])
    q, s = _hip.norm_rows_quant_fp8(x, rms=False, eps=1e-6, split=split, shiftA=mods[0], scaleA=mods[1], shiftB=mods[2], scaleB=mods[3])
    torch.cuda.synchronize()
    qr, sr = _quant_ref(y)
    assert torch.equal(s, sr) and torch.equal(q, qr.view(torch.uint8))


@pytest.mark.parametrize("M,N,K,cfg", [(4289, 3072, 3072, -1), (193, 768, 1024, -1), (777, 1536, 6144, 3), (20, 512, 256, 2), (1000, 9216, 3072, 0)])
def test_linear_fp8_matches_quantised_contraction(hip, M, N, K, cfg):
    from thinkdiff import _hip
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    xq, xs = _hip.quant_rows_fp8(x)
    wq, ws = _hip.quant_rows_fp8(w)
    y = _hip.linear_fp8(xq, xs, wq, ws, b, tile_cfg=cfg)
    torch.cuda.synchronize()
    acc = xq.view(torch.float8_e4m3fn).double() @ wq.view(torch.float8_e4m3fn).double().T
    want = (acc * xs.double()[:, None] * ws.double()[None, :] + b.double()).float()
    err = (y.float() - want).abs().max() / want.abs().max()
    assert err < 2 ** -7, err             # one bf16 rounding of the output
    # and the quantisation itself: relative RMSE against the bf16 GEMM of the unquantised operands
    ref = _hip.linear(x, w, b)
    torch.cuda.synchronize()
    rel = float((y.float() - ref.float()).pow(2).mean().sqrt() / ref.float().pow(2).mean().sqrt())
    print(f"fp8 vs bf16 GEMM rel-RMSE {rel:.4f}")
    assert rel < 5e-2                     # e4m3: 3 mantissa bits on both operands -> ~3.5 % for gaussian data


def test_linear_fp8_epilogues(hip):
    from thinkdiff import _hip
    torch.manual_seed(9)
    M, N, K = 300, 1024, 512
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    g = torch.randn(N, device="cuda").bfloat16()
    r = torch.randn(M, N, device="cuda").bfloat16()
    xq, xs = _hip.quant_rows_fp8(x)
    wq, ws = _hip.quant_rows_fp8(w)
    lin = (xq.view(torch.float8_e4m3fn).float() @ wq.view(torch.float8_e4m3fn).float().T) * xs[:, None] * ws[None, :] + b.float()
    lin = lin.bfloat16()
    got = _hip.linear_fp8(xq, xs, wq, ws, b, gate=g, res=r)
    want = ((lin * g).float() + r.float())
    got2 = _hip.linear_fp8(xq, xs, wq, ws, b, act=_hip.ACT_GELU_TANH)
    want2 = torch.nn.functional.gelu(lin.float(), approximate="tanh")
    torch.cuda.synchronize()
    assert (got.float() - want).abs().max() <= 2 ** -6 * want.abs().max()
    assert (got2.float() - want2).abs().max() <= 2 ** -6 * want2.abs().max()

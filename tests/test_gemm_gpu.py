"""GPU parity of the bf16 MFMA GEMM (td_linear_bf16) against an fp32 torch CPU reference.

Tolerance: inputs are bf16-exact on both sides and accumulation is fp32, so the only differences are
summation order and the final bf16 rounding: |err| <= 2^-8 relative to the row scale (stated per test).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_linear(x, w, b=None, act=0, gate=None, res=None):
    y = x.float() @ w.float().t()
    if b is not None:
        y = y + b.float()
    y = y.bfloat16().float()
    if act == 1:
        y = torch.nn.functional.gelu(y, approximate="tanh").bfloat16().float()
    elif act == 2:
        y = torch.nn.functional.gelu(y).bfloat16().float()
    elif act == 3:
        y = torch.nn.functional.silu(y).bfloat16().float()
    if gate is not None:
        y = (y * gate.float()).bfloat16().float()
    if res is not None:
        y = y + res.float()
    return y.bfloat16()


def _close(got, ref, scale_tol=2.0 ** -7):
    got, ref = got.float().cpu(), ref.float()
    denom = ref.abs().max().clamp_min(1e-6)
    err = (got - ref).abs().max() / denom
    assert torch.isfinite(got).all()
    assert err < scale_tol, f"max rel-to-scale error {err:.3e}"


@pytest.mark.parametrize("M,N,K", [
    (256, 256, 64), (256, 256, 128), (512, 768, 3072), (193, 3072, 4096), (449, 9216, 3072),
    (4289, 3072, 3072), (1, 3072, 256), (28, 18432, 3072), (65, 4096, 1408), (4096, 64, 3072),
    (300, 24, 64), (17, 8, 128),
])
def test_linear_plain(hip, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    x = (torch.randn(M, K, generator=g)).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = (torch.randn(N, generator=g)).bfloat16()
    y = hip.linear(x.cuda(), w.cuda(), b.cuda())
    torch.cuda.synchronize()
    _close(y, _ref_linear(x, w, b))


def test_linear_asymmetric_identity(hip):
    """A = I with an asymmetric W catches a transposed C write (guide 3, 'A=I-check')."""
    K = 256
    x = torch.eye(K).bfloat16()
    w = (torch.arange(K * K).reshape(K, K) % 251).float().bfloat16()  # W[n,k], exact in bf16
    y = hip.linear(x.cuda(), w.cuda())
    torch.cuda.synchronize()
    assert torch.equal(y.cpu().float(), w.float().t())


@pytest.mark.parametrize("act", [1, 2, 3])
def test_linear_activations(hip, act):
    g = torch.Generator().manual_seed(act)
    M, N, K = 300, 512, 256
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.1).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    y = hip.linear(x.cuda(), w.cuda(), b.cuda(), act=act)
    torch.cuda.synchronize()
    _close(y, _ref_linear(x, w, b, act=act), 2.0 ** -6)


def test_linear_gate_residual_inplace(hip):
    g = torch.Generator().manual_seed(11)
    M, N, K = 449, 3072, 768
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    gate = torch.randn(N, generator=g).bfloat16()
    h = torch.randn(M, N, generator=g).bfloat16()
    hd = h.cuda()
    hip.linear(x.cuda(), w.cuda(), b.cuda(), gate=gate.cuda(), res=hd, out=hd)
    torch.cuda.synchronize()
    _close(hd, _ref_linear(x, w, b, gate=gate, res=h))


def test_linear_strided_and_split(hip):
    """Single-block fused projection: cols [0,n_split) -> qkv, rest -> GELU into a wider buffer."""
    g = torch.Generator().manual_seed(5)
    M, K, n_split, n_mlp = 300, 256, 768, 1024
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(n_split + n_mlp, K, generator=g) * 0.1).bfloat16()
    b = torch.randn(n_split + n_mlp, generator=g).bfloat16()
    qkv = torch.zeros(M, n_split, dtype=torch.bfloat16, device="cuda")
    cat = torch.zeros(M, 256 + n_mlp, dtype=torch.bfloat16, device="cuda")
    hip.linear_split(x.cuda(), w.cuda(), b.cuda(), qkv, 0, cat[:, 256:], 1, n_split)
    torch.cuda.synchronize()
    _close(qkv, _ref_linear(x, w[:n_split], b[:n_split]))
    _close(cat[:, 256:], _ref_linear(x, w[n_split:], b[n_split:], act=1), 2.0 ** -6)
    assert torch.count_nonzero(cat[:, :256]) == 0
    # strided A: read the GELU half back as the A operand of a second GEMM
    w2 = (torch.randn(512, n_mlp, generator=g) * 0.05).bfloat16()
    y = hip.linear(cat[:, 256:], w2.cuda())
    torch.cuda.synchronize()
    _close(y, _ref_linear(cat[:, 256:].cpu(), w2))


def test_linear_rejects_bad_k(hip):
    x = torch.zeros(4, 40, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(8, 40, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(hip.ThinkDiffHipError):
        hip.linear(x, w)


@pytest.mark.parametrize("cfg", [-1, 0, 3, 10, 33])
@pytest.mark.parametrize("M0,M1,N,K", [(1024, 193, 768, 512), (300, 65, 3072, 256), (4096, 193, 3072, 3072)])
def test_linear_grouped_two_problems(hip, M0, M1, N, K, cfg):
    """Image-stream + text-stream projection of a FLUX double block in one launch, every tile config."""
    g = torch.Generator().manual_seed(M0 + M1 + cfg)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()
    x0, x1 = mk(M0, K), mk(M1, K)
    w0, w1 = (mk(N, K).float() * 0.05).bfloat16(), (mk(N, K).float() * 0.05).bfloat16()
    b0, b1, g0, g1 = mk(N), mk(N), mk(N), mk(N)
    h0, h1 = mk(M0, N), mk(M1, N)
    d0, d1 = h0.cuda(), h1.cuda()
    hip.linear_grouped2(x0.cuda(), w0.cuda(), b0.cuda(), d0, x1.cuda(), w1.cuda(), b1.cuda(), d1,
                        gate0=g0.cuda(), res0=d0, gate1=g1.cuda(), res1=d1, tile_cfg=cfg)
    torch.cuda.synchronize()
    _close(d0, _ref_linear(x0, w0, b0, gate=g0, res=h0))
    _close(d1, _ref_linear(x1, w1, b1, gate=g1, res=h1))


@pytest.mark.parametrize("cfg", [0, 3])
@pytest.mark.parametrize("M,N,K", [(258, 768, 512), (2, 512, 256), (512 + 17, 1024, 1024), (576 + 64, 384, 256), (4354, 768, 384), (256 + 65, 512, 128)])
def test_linear_ragged_last_tile(hip, M, N, K, cfg):
    """The last row of tiles holds 1 ... 64 rows (config 5: 258 text rows = one 256-row tile + 2; 4 354 joint rows = 17 tiles + 2): those tiles
    run the ragged loop (only the m-tiles that hold rows are multiplied), with either big tile shape; 65 rows stay on the main loop."""
    g = torch.Generator().manual_seed(M * 3 + N + cfg)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    gate = torch.randn(N, generator=g).bfloat16()
    h = torch.randn(M, N, generator=g).bfloat16()
    y = h.cuda()
    hip.linear_grouped2(x.cuda(), w.cuda(), b.cuda(), y, None, None, None, None, gate0=gate.cuda(), res0=y, tile_cfg=cfg)
    torch.cuda.synchronize()
    _close(y, _ref_linear(x, w, b, gate=gate, res=h))


@pytest.mark.parametrize("cfg", [0, 3])
def test_linear_grouped_ragged_text_rows(hip, cfg):
    """Config 5's double-stream shape in one grouped launch: 4096 image rows + 258 text rows (the text problem's second tile holds 2 rows)."""
    M0, M1, N, K = 4096, 258, 768, 512
    g = torch.Generator().manual_seed(cfg + 5)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()
    x0, x1 = mk(M0, K), mk(M1, K)
    w0, w1 = (mk(N, K).float() * 0.05).bfloat16(), (mk(N, K).float() * 0.05).bfloat16()
    b0, b1 = mk(N), mk(N)
    d0, d1 = torch.empty(M0, N, dtype=torch.bfloat16, device="cuda"), torch.empty(M1, N, dtype=torch.bfloat16, device="cuda")
    hip.linear_grouped2(x0.cuda(), w0.cuda(), b0.cuda(), d0, x1.cuda(), w1.cuda(), b1.cuda(), d1, act=1, tile_cfg=cfg)
    torch.cuda.synchronize()
    _close(d0, _ref_linear(x0, w0, b0, act=1), 2.0 ** -6)
    _close(d1, _ref_linear(x1, w1, b1, act=1), 2.0 ** -6)


@pytest.mark.parametrize("split", ["auto", "2", "4"])
@pytest.mark.parametrize("M0,M1,N,K", [(4096, 193, 12288, 256), (4096, 258, 9216, 128), (4354, 0, 21504, 128), (4096, 193, 4352, 512)])
def test_linear_tail_split_is_bit_identical(hip, monkeypatch, M0, M1, N, K, split):
    """Tile counts that leave a mostly empty last round of the 256 CUs (816 = 3.19 rounds, 648 = 2.5, 1 512 = 5.9 with ragged tiles, 289): the launcher
    can cut the last tiles into 2 or 4 row sub-tiles (csrc/gemm_bf16.hip, TAIL; opt-in through TD_GEMM_TAIL since the whole-image A/B went against
    it -- see the launcher's comment).  Every output element is still one workgroup's full contraction, so the
    result must equal the unsplit launch BIT FOR BIT -- with the grouped two-problem form, ragged last tiles, the gate / residual and GELU epilogues,
    int8 and e4m3 operands."""
    g = torch.Generator().manual_seed(M0 + N + K)
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16()
    x0, w0, b0, g0, h0 = mk(M0, K).cuda(), (mk(N, K).float() * 0.05).bfloat16().cuda(), mk(N).cuda(), mk(N).cuda(), mk(M0, N).cuda()
    two = M1 > 0
    x1, w1, b1, g1, h1 = (mk(M1, K).cuda(), (mk(N, K).float() * 0.05).bfloat16().cuda(), mk(N).cuda(), mk(N).cuda(), mk(M1, N).cuda()) if two else (None,) * 5

    def run():
        d0, d1 = h0.clone(), (h1.clone() if two else None)
        hip.linear_grouped2(x0, w0, b0, d0, x1, w1, b1, d1, gate0=g0, res0=d0, gate1=g1 if two else None, res1=d1, tile_cfg=0)
        e0 = torch.empty_like(h0)
        hip.linear_grouped2(x0, w0, b0, e0, None, None, None, None, act=hip.ACT_GELU_TANH, tile_cfg=0)
        xq, xs = hip.quant_rows_int8(x0)
        wq, ws = hip.quant_rows_int8(w0)
        y8 = hip.linear_int8(xq, xs, wq, ws, b0, act=hip.ACT_GELU_TANH, tile_cfg=0)
        xf, xfs = hip.quant_rows_fp8(x0)
        wf, wfs = hip.quant_rows_fp8(w0)
        yf = hip.linear_fp8(xf, xfs, wf, wfs, b0, tile_cfg=0)
        torch.cuda.synchronize()
        return d0, d1, e0, y8, yf
    monkeypatch.delenv("TD_GEMM_TAIL", raising=False)
    base = run()                                    # the default: one tile per workgroup
    monkeypatch.setenv("TD_GEMM_TAIL", split)       # "auto": the launcher's own choice of 1 / 2 / 4
    got = run()
    for a, b in zip(got, base):
        if a is not None:
            assert torch.equal(a, b)
    _close(got[0], _ref_linear(x0.cpu(), w0.cpu(), b0.cpu(), gate=g0.cpu(), res=h0.cpu()))


@pytest.mark.parametrize("M,N,K", [(4289, 3072, 1024), (449, 9216, 512), (100, 192, 64), (289, 200, 128)])
def test_linear_tile_288x192(hip, M, N, K):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16()
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.linear_grouped2(x.cuda(), w.cuda(), b.cuda(), y, None, None, None, None, act=1, tile_cfg=3)
    torch.cuda.synchronize()
    _close(y, _ref_linear(x, w, b, act=1), 2.0 ** -6)


@pytest.mark.parametrize("M,N,K", [(1, 3584, 18944), (1, 4608, 3584), (2, 152064, 256), (3, 1024, 3072), (5, 40, 1408), (8, 3072, 768), (1, 8, 64),
                                   (16, 3584, 18944), (13, 4608, 3584), (9, 37888, 1536), (6, 48, 64), (16, 151936, 1536),
                                   (17, 4608, 3584), (32, 1536, 8960), (33, 3584, 18944), (48, 151936, 1536), (64, 17920, 1536), (40, 48, 64),
                                   (50, 32784, 128), (64, 1536, 1536), (5, 16, 4096), (64, 2048, 1536), (16, 1536, 8960)])
@pytest.mark.parametrize("mode", ["bias", "act", "gate_res", "none"])
def test_skinny_m_weight_stream_kernel(hip, M, N, K, mode):
    """M <= 64 routes to csrc/gemv_bf16.hip (dot-product stream up to 4 rows, matrix-core stream with 1-4 activation blocks
    above; two weight blocks per workgroup from 2048 blocks on, odd block counts included): the tile kernel's epilogue semantics."""
    g = torch.Generator().manual_seed(M * 11 + N + K)
    x = torch.randn(M, K, generator=g).bfloat16()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16()
    b = torch.randn(N, generator=g).bfloat16() if mode != "none" else None
    gate = torch.randn(N, generator=g).bfloat16() if mode == "gate_res" else None
    res = torch.randn(M, N, generator=g).bfloat16() if mode == "gate_res" else None
    act = 1 if mode == "act" else 0
    y = hip.linear(x.cuda(), w.cuda(), None if b is None else b.cuda(), act=act,
                   gate=None if gate is None else gate.cuda(), res=None if res is None else res.cuda())
    torch.cuda.synchronize()
    _close(y, _ref_linear(x, w, b, act=act, gate=gate, res=res))


def test_skinny_m_split_k_repeats_bit_for_bit(hip):
    """Few weight blocks: K is split over workgroups and the last arriver adds the partial tiles in index order, so repeated and
    concurrent launches (each stream has its own hand-off workspace) give the same bits; the counters re-arm themselves."""
    M, N, K = 64, 1536, 8960
    g = torch.Generator().manual_seed(8)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    res = torch.randn(M, N, generator=g).bfloat16().cuda()
    first = hip.linear(x, w, None, res=res)
    torch.cuda.synchronize()
    _close(first, _ref_linear(x.cpu(), w.cpu(), res=res.cpu()))
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = []
    for rep in range(8):
        for st in streams:
            with torch.cuda.stream(st):
                outs.append(hip.linear(x, w, None, res=res))
    torch.cuda.synchronize()
    assert all(torch.equal(o, first) for o in outs)


def test_skinny_m_split_output_and_strided_rows(hip):
    """The Qwen2 decode form: q -> scratch, k|v -> cache rows (split output), x rows with a stride."""
    M, K, n_split, N = 2, 512, 1024, 1536
    g = torch.Generator().manual_seed(5)
    xbuf = torch.randn(M, K + 64, generator=g).bfloat16().cuda()
    x = xbuf[:, :K]
    w = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    y0 = torch.zeros(M, n_split, dtype=torch.bfloat16, device="cuda")
    y1 = torch.zeros(M, N - n_split + 32, dtype=torch.bfloat16, device="cuda")
    hip.linear_split(x, w, b, y0, 0, y1[:, :N - n_split], 3, n_split)
    torch.cuda.synchronize()
    ref = _ref_linear(x.cpu(), w.cpu(), b.cpu())
    _close(y0, ref[:, :n_split])
    _close(y1[:, :N - n_split], torch.nn.functional.silu(ref[:, n_split:].float()).bfloat16())
    assert not y1[:, N - n_split:].any()


@pytest.mark.parametrize("M,N,K,cfg,parts", [
    (256, 1536, 8960, 1, 10), (256, 1536, 8960, 1, -1), (200, 3584, 18944, -1, -1), (65, 1536, 1536, 1, 6), (129, 2048, 1536, 0, 4),
    (256, 3584, 3584, -1, -1), (100, 512, 4096, 2, 8), (300, 1536, 2048, -1, 2),
])
def test_linear_split_k(hip, M, N, K, cfg, parts):
    """td_linear_splitk_bf16 (TdGemmParams::split_k): K split over workgroups, fp32 partial sums added in index order by a second launch.  The
    decode shapes of 65..256 sequences (2B: 1536 / 8960, 7B: 3584 / 18944), forced and automatic part counts, every tile config, bias + in-place
    residual: against fp32 math (the GEMM tolerance), against the unsplit launch (summation order only), and bit-equal between two runs."""
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    ref = _ref_linear(x.cpu(), w.cpu(), b.cpu(), res=r.cpu())
    y1 = hip.linear_splitk(x, w, b, res=r, out=r.clone(), split_k=parts, tile_cfg=cfg)          # (in place on a copy of the residual, as the engine does)
    y2 = hip.linear_splitk(x, w, b, res=r, out=r.clone(), split_k=parts, tile_cfg=cfg)
    y0 = hip.linear(x, w, b, res=r)
    torch.cuda.synchronize()
    _close(y1, ref)
    assert torch.equal(y1, y2)
    _close(y1, y0.cpu(), 2.0 ** -7)          # (one bf16 ulp of the largest output where the two summation orders round apart)


def test_linear_split_k_two_outputs_and_refusals(hip):
    """The q | k|v projection of a wide decode step: columns >= n_split go to the second buffer (n_split not a tile multiple: the reduction splits, not
    the tiles); a part count that does not divide the k-tiles is refused, -1 on a form the reduction does not cover falls back to one launch."""
    g = torch.Generator().manual_seed(5)
    M, N, K, ns = 130, 1536 + 512, 1536, 1536
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    q = torch.zeros(M, ns, dtype=torch.bfloat16, device="cuda")
    kv = torch.zeros(M, N - ns, dtype=torch.bfloat16, device="cuda")
    hip.linear_splitk(x, w, b, out=q, out1=kv, n_split=ns, split_k=6, tile_cfg=1)
    torch.cuda.synchronize()
    ref = _ref_linear(x.cpu(), w.cpu(), b.cpu())
    _close(q, ref[:, :ns])
    _close(kv, ref[:, ns:], 2.0 ** -6)
    with pytest.raises(hip.ThinkDiffHipError):
        hip.linear_splitk(x, w, b, split_k=5, tile_cfg=1)       # 24 k-tiles are not divisible by 5


@pytest.mark.parametrize("M,N,K", [(256, 1536, 8960), (200, 3584, 3584), (65, 512, 1024), (130, 4096, 512)])
def test_linear_split_k_with_fused_rmsnorm(hip, M, N, K):
    """norm_w of td_linear_splitk_bf16: the reduction launch also writes Qwen2RMSNorm of each finished row.  Both outputs bit-equal to the split
    Linear followed by td_norm_rows_bf16 (same summation order and rounding points), whether the planner splits K or not."""
    g = torch.Generator().manual_seed(M * 3 + N)
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    r = torch.randn(M, N, generator=g).bfloat16().cuda()
    nw = (1.0 + 0.1 * torch.randn(N, generator=g)).bfloat16().cuda()
    y0 = hip.linear_splitk(x, w, None, res=r, out=r.clone())
    n0 = hip.norm_rows(y0, rms=True, eps=1e-6, w=nw)
    n1 = torch.empty_like(y0)
    y1 = hip.linear_splitk(x, w, None, res=r, out=r.clone(), norm_w=nw, norm_out=n1, norm_eps=1e-6)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1) and torch.equal(n0, n1)
    _close(y1, _ref_linear(x.cpu(), w.cpu(), res=r.cpu()))


@pytest.mark.parametrize("M", [17, 33, 64, 65])
def test_linear_split_k_small_m_boundaries(hip, M):
    """The planner's lower edge: 16 < M <= 64 asked for with split_k = -1 takes the tile kernels (a caller that sets split_k wants them), M = 65 is
    the first wide decode step; all against fp32 math, with and without the fused RMSNorm."""
    g = torch.Generator().manual_seed(M)
    N, K = 1536, 4096
    x = torch.randn(M, K, generator=g).bfloat16().cuda()
    w = (torch.randn(N, K, generator=g) * 0.03).bfloat16().cuda()
    b = torch.randn(N, generator=g).bfloat16().cuda()
    nw = (1.0 + 0.1 * torch.randn(N, generator=g)).bfloat16().cuda()
    y = hip.linear_splitk(x, w, b)
    n1 = torch.empty_like(y)
    y1 = hip.linear_splitk(x, w, b, norm_w=nw, norm_out=n1)
    torch.cuda.synchronize()
    _close(y, _ref_linear(x.cpu(), w.cpu(), b.cpu()))
    assert torch.equal(y, y1) and torch.equal(n1, hip.norm_rows(y, rms=True, eps=1e-6, w=nw))

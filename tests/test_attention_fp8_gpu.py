"""GPU parity of the 8-bit joint attention (td_attention_fp8, csrc/attention_fp8.hip).

Three layers:
 * the pack kernel is integer work and is checked BIT-EXACTLY: the e4m3 payloads and E8M0 scale bytes it leaves in the workspace
   (q8 row-major, the swizzled K tiles, the transposed / key-permuted V^T tiles) against oracle/flux_ref.py's `_e8m0_quant`;
 * the attention output against the oracle's restatement of the same arithmetic (`_attention_fp8`): e4m3 operands and
   probabilities, exact accumulation.  Both round a probability relative to an INTEGER power-of-two reference, so the rounding
   does not depend on the kernel's tiling; what is left is accumulation order, exp2's last bit and the bf16 output: tolerance
   5e-3 of the output's RMS (measured 2e-4 ... 1e-3);
 * and against the exact fp32 softmax(QK^T)V: 8e-2 of the output's RMS on random operands (measured 5.3e-2 ... 5.6e-2, the e4m3
   floor: three 3-bit roundings -- q.k, P, V -- with nothing coherent to average against; the integer form of P adds 1.9 % rms).
"""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import flux_ref as R  # noqa: E402

pytestmark = pytest.mark.gpu


_PROB = {"form": "linear"}


@pytest.fixture(params=[0, 1, 2], ids=["8wave", "4wave", "8wave_exp2"], autouse=True)
def all_forms(request, hip):
    """The shipped kernel (8-wave workgroups, probabilities by integer conversion), the 4-wave A/B form (td_attention_set_variant
    bit 0) and the exp2 form of the probabilities (bit 1; the oracle's FP8_ATTENTION_PROB = "exp2") must pass every case."""
    prev = hip.lib().td_attention_set_variant(request.param)
    _PROB["form"] = "exp2" if request.param & 2 else "linear"
    yield
    hip.lib().td_attention_set_variant(prev)
    _PROB["form"] = "linear"


def _layout(Sq, Skv, H):
    up = lambda x: (x + 255) & ~255  # noqa: E731
    nt = (Skv + 63) // 64
    o, L = 0, {}
    for name, n in (("q8", Sq * H * 128), ("qs", Sq * H), ("k8", H * nt * 8192), ("ks", H * nt * 64), ("v8", H * nt * 8192), ("vs", H * nt)):
        L[name] = o
        o += up(n)
    L["total"], L["nt"] = o, nt
    return L


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def _run(hip, qkv, H, scale=None):
    S = qkv.shape[0]
    D = H * 128
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    L = _layout(S, S, H)
    ws = torch.zeros(L["total"], dtype=torch.uint8, device="cuda")
    out = torch.zeros(S, D, dtype=torch.bfloat16, device="cuda")
    hip.attention_fp8(q, k, v, out, H, scale=scale, workspace=ws)
    torch.cuda.synchronize()
    return out, ws.cpu(), L


def _exact(qkv, H):
    S = qkv.shape[0]
    x = qkv.float().cpu().view(S, 3, H, 128).permute(1, 2, 0, 3)
    s = x[0] @ x[1].transpose(-1, -2) / math.sqrt(128)
    return (torch.softmax(s, dim=-1) @ x[2]).transpose(0, 1).reshape(S, H * 128)


def _restated(qkv, H):
    S = qkv.shape[0]
    x = qkv.view(S, 3, H, 128).permute(1, 2, 0, 3)[:, None]
    prev, R.FP8_ATTENTION = (R.FP8_ATTENTION, R.FP8_ATTENTION_PROB), True
    R.FP8_ATTENTION_PROB = _PROB["form"]
    try:
        return R._attention(x[0], x[1], x[2])[0]
    finally:
        R.FP8_ATTENTION, R.FP8_ATTENTION_PROB = prev


@pytest.mark.parametrize("S,H", [(64, 1), (300, 2), (449, 4), (1000, 3)])
def test_pack_kernel_is_bit_exact(hip, S, H):
    g = torch.Generator().manual_seed(S * 7 + H)
    qkv = (torch.randn(S, 3 * H * 128, generator=g) * torch.linspace(0.05, 8.0, 3 * H * 128)[None, :]).bfloat16()   # column-dependent magnitudes: distinct scales
    _, ws, L = _run(hip, qkv.cuda(), H)
    nt = L["nt"]
    x = qkv.float().view(S, 3, H, 128)
    # q: times fp32(fp32(128^-0.5) * fp32(log2 e)), one fp32 rounding per element, as the kernel does
    c = torch.tensor(128 ** -0.5, dtype=torch.float32) * torch.tensor(1.4426950408889634, dtype=torch.float32)
    q8, sq = R._e8m0_quant(x[:, 0] * c, (2,))
    if _PROB["form"] == "linear":
        sq = sq * 8          # the integer form of the probabilities wants the scores times 8: a factor of q's power-of-two scale
    got_q = ws[L["q8"]:L["q8"] + S * H * 128].view(torch.float8_e4m3fn).float().view(S, H, 128)
    got_sq = torch.exp2(ws[L["qs"]:L["qs"] + S * H].float().view(S, H, 1) - 127)
    assert torch.equal(got_sq, sq) and torch.equal(got_q, q8)
    # k: tile [64 keys][128 B], 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7); scales at [tile][key & 31][key >> 5]
    k8, sk = R._e8m0_quant(x[:, 1], (2,))
    kt = ws[L["k8"]:L["k8"] + H * nt * 8192].view(H, nt, 64, 8, 16)
    r = torch.arange(64)
    ch = torch.arange(8)[None, :] ^ ((r[:, None] >> 1) & 7)                    # [row, logical chunk] -> stored chunk
    kt = torch.gather(kt, 3, ch[None, None, :, :, None].expand(H, nt, 64, 8, 16)).reshape(H, nt * 64, 128)[:, :S]
    assert torch.equal(kt.contiguous().view(torch.float8_e4m3fn).float().transpose(0, 1), k8)
    ks = ws[L["ks"]:L["ks"] + H * nt * 64].view(H, nt, 32, 2).transpose(2, 3).reshape(H, nt * 64)[:, :S]
    assert torch.equal(torch.exp2(ks.float().transpose(0, 1)[..., None] - 127), sk)
    # v: tile [128 d][64 B]; key kappa of a tile at byte 32 h + 16 kb + reg, chunk c of row d at chunk c ^ ((d >> 2) & 3)
    pad = (-S) % 64
    vg = torch.nn.functional.pad(x[:, 2], (0, 0, 0, 0, 0, pad)).view(nt, 64, H, 128)
    v8, sv = R._e8m0_quant(vg, (1, 3))
    vt = ws[L["v8"]:L["v8"] + H * nt * 8192].view(H, nt, 128, 4, 16)
    d = torch.arange(128)
    chv = torch.arange(4)[None, :] ^ ((d[:, None] >> 2) & 3)
    vt = torch.gather(vt, 3, chv[None, None, :, :, None].expand(H, nt, 128, 4, 16)).reshape(H, nt, 128, 64)
    kap = torch.arange(64)
    k32 = kap & 31
    pos = 32 * ((k32 >> 2) & 1) + 16 * (kap >> 5) + (k32 & 3) + 4 * (k32 >> 3)
    vt = vt[..., pos].contiguous().view(torch.float8_e4m3fn).float()           # [H, nt, d, key]
    assert torch.equal(vt.permute(1, 3, 0, 2), v8)
    assert torch.equal(torch.exp2(ws[L["vs"]:L["vs"] + H * nt].float().view(H, nt).transpose(0, 1)[:, None, :, None] - 127), sv)


@pytest.mark.parametrize("S,H", [(64, 1), (300, 2), (449, 4), (1000, 3), (4289, 2)])
def test_fp8_attention_vs_restatement_and_exact(hip, S, H):
    g = torch.Generator().manual_seed(S + H)
    qkv = torch.randn(S, 3 * H * 128, generator=g).bfloat16()
    out, _, _ = _run(hip, qkv.cuda(), H)
    assert torch.isfinite(out.float()).all()
    e_rest = _rel(out, _restated(qkv, H))
    e_exact = _rel(out, _exact(qkv, H))
    print(f"S={S} H={H}: vs restatement {e_rest:.3e}, vs exact {e_exact:.3e}")
    assert e_rest < 5e-3 and e_exact < 8e-2


@pytest.mark.parametrize("Sq,Skv,H", [(300, 500, 2), (500, 300, 3), (1, 64, 1), (257, 65, 2)])
def test_fp8_attention_different_query_and_key_counts(hip, Sq, Skv, H):
    """td_attention_fp8 takes any (Sq, Skv) pair: more query tiles than key tiles (the pack pass then has query-only tiles), fewer,
    a single query row, a ragged key tail of one row."""
    g = torch.Generator().manual_seed(Sq * 1000 + Skv)
    D = H * 128
    q = torch.randn(Sq, D, generator=g).bfloat16()
    kv = torch.randn(Skv, 2 * D, generator=g).bfloat16()
    out = torch.zeros(Sq, D, dtype=torch.bfloat16, device="cuda")
    kvd = kv.cuda()
    hip.attention_fp8(q.cuda(), kvd[:, :D], kvd[:, D:], out, H)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    heads = lambda t, n: t.view(n, H, 128).transpose(0, 1)[None]  # noqa: E731
    prev, R.FP8_ATTENTION = (R.FP8_ATTENTION, R.FP8_ATTENTION_PROB), True
    R.FP8_ATTENTION_PROB = _PROB["form"]
    try:
        ref = R._attention(heads(q, Sq), heads(kv[:, :D].contiguous(), Skv), heads(kv[:, D:].contiguous(), Skv))[0]
    finally:
        R.FP8_ATTENTION, R.FP8_ATTENTION_PROB = prev
    x = torch.softmax(heads(q, Sq)[0].float() @ heads(kv[:, :D].contiguous(), Skv)[0].float().transpose(-1, -2) / math.sqrt(128), dim=-1) @ heads(kv[:, D:].contiguous(), Skv)[0].float()
    e_rest, e_exact = _rel(out, ref), _rel(out, x.transpose(0, 1).reshape(Sq, D))
    print(f"Sq={Sq} Skv={Skv} H={H}: vs restatement {e_rest:.3e}, vs exact {e_exact:.3e}")
    assert e_rest < 5e-3 and e_exact < 8e-2


def test_fp8_attention_sharp_rows_and_large_magnitudes(hip):
    """Rows whose softmax is nearly one-hot (|q.k| large: the reference point moves often) and v far from unit scale."""
    S, H = 520, 2
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(S, 3 * H * 128, generator=g)
    qkv[:, :2 * H * 128] *= 4.0
    qkv[:, 2 * H * 128:] *= 37.0
    qkv = qkv.bfloat16()
    out, _, _ = _run(hip, qkv.cuda(), H)
    e_rest, e_exact = _rel(out, _restated(qkv, H)), _rel(out, _exact(qkv, H))
    print(f"sharp: vs restatement {e_rest:.3e}, vs exact {e_exact:.3e}")
    assert e_rest < 5e-3 and e_exact < 2.5e-1       # (logits of std 23: their e4m3 error alone is ~0.8 in the exponent)


@pytest.mark.parametrize("S,H", [(4289, 24), (4354, 24), (1000, 80)])
def test_fp8_attention_more_items_than_cus(hip, S, H):
    """The FLUX shapes: items split across persistent workgroups (the hand-off), compared on the device with the restatement."""
    g = torch.Generator(device="cuda").manual_seed(S)
    qkv = torch.randn(S, 3 * H * 128, generator=g, device="cuda").bfloat16()
    out, _, _ = _run(hip, qkv, H)
    ref = _restated(qkv, H)
    e = _rel(out, ref)
    rows = ((out.float() - ref.float()).pow(2).mean(dim=1).sqrt() / ref.float().pow(2).mean().sqrt()).max().item()
    print(f"S={S} H={H}: vs restatement {e:.3e}, worst row {rows:.3e}")
    assert e < 5e-3 and rows < 3e-2
    out2, _, _ = _run(hip, qkv, H)
    assert _rel(out2, out) < 2e-3        # (the merge order of a split item is fixed; which owner rounds last is not)


@pytest.mark.parametrize("S,H,split", [(64, 1, 0), (300, 2, 40), (449, 4, 449), (1000, 3, 193)])
def test_fused_qk_norm_rope_pack_is_bit_identical(hip, S, H, split):
    """td_attention_fp8_qk_rope (QK-RMSNorm + RoPE inside the pack pass, raw projections in) against td_qk_norm_rope_bf16 followed by
    td_attention_fp8: the same packed bytes and scale bytes in the workspace, the same output, and an untouched projection buffer."""
    g = torch.Generator().manual_seed(S * 3 + H)
    D = H * 128
    qkv = (torch.randn(S, 3 * D, generator=g) * torch.linspace(0.1, 4.0, 3 * D)[None, :]).bfloat16().cuda()
    ids = (torch.arange(S)[:, None] * torch.tensor([0.0, 1.0, 3.0])).float().contiguous().cuda()      # a distinct angle set per token
    cos, sin = hip.flux_rope_table(ids)
    w = [(1.0 + 0.2 * torch.randn(128, generator=g)).bfloat16().cuda() for _ in range(4)]
    L = _layout(S, S, H)
    # two passes
    a = qkv.clone()
    hip.qk_norm_rope(a, H, H, 0, D, cos, sin, split=split, wqA=w[0], wkA=w[1], wqB=w[2], wkB=w[3])
    ws_a = torch.zeros(L["total"], dtype=torch.uint8, device="cuda")
    out_a = torch.zeros(S, D, dtype=torch.bfloat16, device="cuda")
    hip.attention_fp8(a[:, :D], a[:, D:2 * D], a[:, 2 * D:], out_a, H, workspace=ws_a)
    # one pass
    b = qkv.clone()
    ws_b = torch.zeros(L["total"], dtype=torch.uint8, device="cuda")
    out_b = torch.zeros(S, D, dtype=torch.bfloat16, device="cuda")
    hip.attention_fp8_qk_rope(b, out_b, H, cos, sin, split=split, wqA=w[0], wkA=w[1], wqB=w[2], wkB=w[3], workspace=ws_b)
    torch.cuda.synchronize()
    assert torch.equal(b, qkv), "the fused form must leave the projections alone"
    assert not torch.equal(a, qkv)
    for name, n in (("q8", S * D), ("qs", S * H), ("k8", H * L["nt"] * 8192), ("ks", H * L["nt"] * 64), ("v8", H * L["nt"] * 8192), ("vs", H * L["nt"])):
        assert torch.equal(ws_a[L[name]:L[name] + n], ws_b[L[name]:L[name] + n]), f"packed {name} differs"
    assert torch.equal(out_a, out_b)
    # no norm weights: RoPE only
    a = qkv.clone()
    hip.qk_norm_rope(a, H, H, 0, D, cos, sin)
    hip.attention_fp8(a[:, :D], a[:, D:2 * D], a[:, 2 * D:], out_a, H, workspace=ws_a)
    hip.attention_fp8_qk_rope(b, out_b, H, cos, sin, workspace=ws_b)
    torch.cuda.synchronize()
    assert torch.equal(ws_a[:L["vs"]], ws_b[:L["vs"]]) and torch.equal(out_a, out_b)


def test_engine_fused_rope_is_bit_identical(hip, monkeypatch):
    """The engine in 8-bit attention mode skips td_qk_norm_rope and lets the pack pass do it (q pre-multiplied by scale x log2 e in
    front of its one bf16 rounding on both paths): same bits as the two-pass form (TD_ATTN8_NO_FUSE)."""
    if _PROB["form"] != "linear":
        pytest.skip("the engine runs the shipped (integer-conversion) form only")
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig
    cfg = R.tiny_config(num_layers=2, num_single_layers=2)
    sd = R.init_weights(cfg, seed=8)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=2, num_single_layers=2, num_attention_heads=cfg.num_attention_heads,
                                                     joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                               max_img_tokens=256, max_txt_tokens=64, max_steps=4)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(10)
    h2, w2, T = 10, 14, 37
    lat = torch.randn(1, h2 * w2, 64, generator=g).bfloat16().cuda()
    pe = torch.randn(1, T, cfg.joint_attention_dim, generator=g).bfloat16().cuda()
    pool = torch.randn(1, cfg.pooled_projection_dim, generator=g).bfloat16().cuda()
    img_ids, txt_ids = R.latent_image_ids(h2, w2), torch.zeros(T, 3)
    t, gd = torch.tensor([0.41]).bfloat16().cuda(), torch.tensor([3.5])
    m.set_attention("fp8")
    fused = m.forward(lat, pe, pool, t, img_ids, txt_ids, gd)[0].clone()
    monkeypatch.setenv("TD_ATTN8_NO_FUSE", "1")
    two_pass = m.forward(lat, pe, pool, t, img_ids, txt_ids, gd)[0].clone()
    monkeypatch.delenv("TD_ATTN8_NO_FUSE")
    m.set_attention("bf16")
    torch.cuda.synchronize()
    assert torch.isfinite(fused.float()).all() and torch.equal(fused, two_pass)


def test_engine_history_reference_points(hip, monkeypatch):
    """From the second denoise step on, every row of the 8-bit attention starts from the reference point its largest score of the previous step gives
    (TdAttnParams::ref_in / ref_out) instead of from its first tile's maximum.  References are integers, so a probability is rounded the same way
    wherever its row's reference sits; what changes is where the tail is cut.  A 4-step denoise with the history against the same without it
    (TD_ATTN8_NO_HREF) and against the oracle's switch."""
    if _PROB["form"] != "linear":
        pytest.skip("the engine runs the shipped (integer-conversion) form only")
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig, effective_scalar
    cfg = R.tiny_config(num_layers=2, num_single_layers=2)
    sd = R.init_weights(cfg, seed=12)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=2, num_single_layers=2, num_attention_heads=cfg.num_attention_heads,
                                                     joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                               max_img_tokens=512, max_txt_tokens=64, max_steps=8)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(13)
    h2, w2, T, n = 20, 18, 45, 4                      # S = 405: 7 key tiles, ragged
    lat = torch.randn(1, h2 * w2, 64, generator=g).bfloat16()
    pe = torch.randn(1, T, cfg.joint_attention_dim, generator=g).bfloat16()
    pool = torch.randn(1, cfg.pooled_projection_dim, generator=g).bfloat16()
    R.FP8_ATTENTION = True
    try:
        ref = R.denoise(sd, cfg, lat, pe, pool, h2, w2, n, guidance_scale=3.5)[0]
    finally:
        R.FP8_ATTENTION = False
    sig = R.make_sigmas(n, h2 * w2)
    m.set_condition(pe[0].cuda(), pool[0].cuda(), R.latent_image_ids(h2, w2))
    m.set_timesteps([effective_scalar(float(s) * 1000.0, torch.bfloat16) for s in sig[:-1]], float((torch.tensor([3.5]).bfloat16() * 1000).float()))
    m.set_attention("fp8")
    outs = {}
    for name in ("history", "plain", "history_again"):
        if name == "plain":
            monkeypatch.setenv("TD_ATTN8_NO_HREF", "1")
        else:
            monkeypatch.delenv("TD_ATTN8_NO_HREF", raising=False)
        x = lat[0].cuda().contiguous()
        m.denoise(x, sig)
        torch.cuda.synchronize()
        outs[name] = x.float().cpu()
    monkeypatch.delenv("TD_ATTN8_NO_HREF", raising=False)
    m.set_attention("bf16")
    assert all(torch.isfinite(v).all() for v in outs.values())
    e_h, e_p, d = _rel(outs["history"], ref), _rel(outs["plain"], ref), _rel(outs["history"], outs["plain"])
    print(f"4-step tiny denoise, 8-bit attention: history~oracle {e_h:.4f}  plain~oracle {e_p:.4f}  history~plain {d:.4f}")
    assert e_h < 2e-2 and e_p < 2e-2 and d < 1e-2
    assert _rel(outs["history_again"], outs["history"]) < 2e-3      # (a fresh image starts without history: the first step is the plain path again)


def test_history_references_are_dropped_when_the_token_layout_changes(hip, monkeypatch):
    """A forward at step 1 right after the conditioning changed (another text length): the references of the previous layout must not be read --
    the result equals the plain path's bit for bit."""
    if _PROB["form"] != "linear":
        pytest.skip("the engine runs the shipped (integer-conversion) form only")
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig, effective_scalar
    cfg = R.tiny_config(num_layers=1, num_single_layers=1)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=1, num_single_layers=1, num_attention_heads=cfg.num_attention_heads,
                                                     joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                               max_img_tokens=256, max_txt_tokens=64, max_steps=4)
    m.load_state_dict(R.init_weights(cfg, seed=14))
    g = torch.Generator().manual_seed(15)
    h2 = w2 = 12
    lat = torch.randn(h2 * w2, 64, generator=g).bfloat16().cuda()
    pool = torch.randn(cfg.pooled_projection_dim, generator=g).bfloat16().cuda()
    sig = [1.0, 0.8, 0.5, 0.0]
    m.set_attention("fp8")

    def prepare(T):
        m.set_condition(torch.randn(T, cfg.joint_attention_dim, generator=torch.Generator().manual_seed(T)).bfloat16().cuda(), pool, R.latent_image_ids(h2, w2))
        m.set_timesteps([effective_scalar(s * 1000.0, torch.bfloat16) for s in sig[:-1]], 3500.0)

    prepare(30)
    m.forward_step(lat, 0)                                # leaves references for (T = 30, step 0)
    prepare(41)                                           # another layout: S changes
    a = m.forward_step(lat, 1).clone()                    # step 1 follows step 0, but on the old layout
    monkeypatch.setenv("TD_ATTN8_NO_HREF", "1")
    b = m.forward_step(lat, 1).clone()
    monkeypatch.delenv("TD_ATTN8_NO_HREF")
    m.set_attention("bf16")
    torch.cuda.synchronize()
    assert torch.isfinite(a.float()).all() and torch.equal(a, b)


def test_engine_fp8_attention_matches_the_oracle_switch(hip):
    """td_flux_set_attention(TD_ATTENTION_FP8) on a tiny config (two double + two single blocks, one forward): the engine against
    oracle/flux_ref.py with FP8_ATTENTION -- as close as the bf16 engine is to the bf16 oracle -- and the distance between the two
    attention arithmetics the same on both sides."""
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig
    cfg = R.tiny_config(num_layers=2, num_single_layers=2)
    sd = R.init_weights(cfg, seed=6)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=2, num_single_layers=2, num_attention_heads=cfg.num_attention_heads,
                                                     joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                               max_img_tokens=256, max_txt_tokens=64, max_steps=4)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(9)
    h2 = w2 = 12
    T = 40
    lat = torch.randn(1, h2 * w2, 64, generator=g).bfloat16()
    pe = torch.randn(1, T, cfg.joint_attention_dim, generator=g).bfloat16()
    pool = torch.randn(1, cfg.pooled_projection_dim, generator=g).bfloat16()
    img_ids, txt_ids = R.latent_image_ids(h2, w2), torch.zeros(T, 3)
    t, gd = torch.tensor([0.7324]), torch.tensor([3.5])
    with torch.no_grad():
        ref16 = R.transformer_forward(sd, cfg, lat, pe, pool, t.bfloat16(), img_ids.bfloat16(), txt_ids.bfloat16(), gd)
        R.FP8_ATTENTION = True
        try:
            ref8 = R.transformer_forward(sd, cfg, lat, pe, pool, t.bfloat16(), img_ids.bfloat16(), txt_ids.bfloat16(), gd)
        finally:
            R.FP8_ATTENTION = False
    out16 = m.forward(lat.cuda(), pe.cuda(), pool.cuda(), t.bfloat16().cuda(), img_ids, txt_ids, gd)[0].clone()
    if _PROB["form"] != "linear":
        pytest.skip("the engine runs the shipped (integer-conversion) form only")
    m.set_attention("fp8")
    out8 = m.forward(lat.cuda(), pe.cuda(), pool.cuda(), t.bfloat16().cuda(), img_ids, txt_ids, gd)[0].clone()
    m.set_attention("bf16")
    torch.cuda.synchronize()
    e16, e88, d_hip, d_ref = _rel(out16, ref16[0]), _rel(out8, ref8[0]), _rel(out8, out16), _rel(ref8[0], ref16[0])
    print(f"tiny config: hip~bf16-oracle {e16:.4f}  hip-attn8~oracle-attn8 {e88:.4f}  attn8~bf16 hip {d_hip:.4f} oracle {d_ref:.4f}")
    assert e16 < 2e-2 and e88 < 2e-2
    assert d_hip < 6e-2 and abs(d_hip - d_ref) < 0.5 * d_ref + 2e-3

"""int8 operand path (TD_PRECISION_INT8, round 3): the quantiser bit-exact against its torch statement, the int8 GEMM against the
exact integer contraction of the very same quantised operands (int32 accumulation is exact, so only the dequantisation and the
bf16 output rounding remain), the end-to-end quantisation error against the bf16 GEMM, and the FLUX engine in int8 mode against
the oracle's int8 switch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _quant_ref(x):
    """per-row symmetric int8 exactly as csrc/elementwise.hip states it (fp32 arithmetic, round half to even)."""
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    s = torch.where(amax > 0, amax * (1.0 / 127.0), torch.ones_like(amax))
    return torch.round(xf * (1.0 / s)[:, None]).clamp(-127, 127).to(torch.int8), s


def test_quant_rows_int8_bit_exact(hip):
    torch.manual_seed(0)
    x = (torch.randn(77, 3072, device="cuda") * torch.logspace(-3, 2, 77, device="cuda")[:, None]).bfloat16()
    x[5] = 0
    x[6, 17] = 3.0e4
    q, s = hip.quant_rows_int8(x)
    torch.cuda.synchronize()
    qr, sr = _quant_ref(x)
    assert torch.equal(s, sr)
    assert torch.equal(q, qr)
    assert s[5] == 1.0 and not q[5].any() and int(q.abs().max()) == 127


@pytest.mark.parametrize("M,N,K,cfg", [(4289, 3072, 3072, -1), (193, 768, 1024, -1), (777, 1536, 6144, 3), (20, 512, 256, 2), (1000, 9216, 3072, 0),
                                       (258, 1024, 512, 0), (4354, 768, 1024, 0), (576 + 3, 768, 512, 3), (2, 512, 256, 0)])      # (ragged last tiles: csrc/gemm_bf16.hip)
def test_linear_int8_matches_integer_contraction(hip, M, N, K, cfg):
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    xq, xs = hip.quant_rows_int8(x)
    wq, ws = hip.quant_rows_int8(w)
    y = hip.linear_int8(xq, xs, wq, ws, b, tile_cfg=cfg)
    torch.cuda.synchronize()
    acc = xq.double() @ wq.double().T                                 # exact: |sum| < 2^31 for K <= 133k
    want = (acc * xs.double()[:, None] * ws.double()[None, :] + b.double()).float()
    err = (y.float() - want).abs().max() / want.abs().max()
    assert err < 2 ** -7, err             # one bf16 rounding of the output
    ref = hip.linear(x, w, b)
    torch.cuda.synchronize()
    rel = float((y.float() - ref.float()).pow(2).mean().sqrt() / ref.float().pow(2).mean().sqrt())
    print(f"int8 vs bf16 GEMM rel-RMSE {rel:.4f}")
    assert rel < 1.5e-2                   # uniform step max/127 on both operands: ~1 % for gaussian data (e4m3: ~5 %)


def test_linear_int8_epilogues(hip):
    torch.manual_seed(9)
    M, N, K = 300, 1024, 512
    x = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda").bfloat16()
    g = torch.randn(N, device="cuda").bfloat16()
    r = torch.randn(M, N, device="cuda").bfloat16()
    xq, xs = hip.quant_rows_int8(x)
    wq, ws = hip.quant_rows_int8(w)
    lin = ((xq.float() @ wq.float().T) * xs[:, None] * ws[None, :] + b.float()).bfloat16()
    got = hip.linear_int8(xq, xs, wq, ws, b, gate=g, res=r)
    want = ((lin * g).float() + r.float())
    got2 = hip.linear_int8(xq, xs, wq, ws, b, act=hip.ACT_GELU_TANH)
    want2 = torch.nn.functional.gelu(lin.float(), approximate="tanh")
    torch.cuda.synchronize()
    assert (got.float() - want).abs().max() <= 2 ** -6 * want.abs().max()
    assert (got2.float() - want2).abs().max() <= 2 ** -6 * want2.abs().max()


def test_int8_mode_matches_int8_oracle(hip):
    """The engine in int8 mode against oracle/flux_ref.py with INT8_BLOCK_LINEARS (tiny config, one forward)."""
    from oracle import flux_ref as R
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig
    cfg = R.tiny_config(num_layers=2, num_single_layers=2)
    sd = R.init_weights(cfg, seed=4)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=2, num_single_layers=2, num_attention_heads=cfg.num_attention_heads,
                                                     joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                               max_img_tokens=256, max_txt_tokens=64, max_steps=4)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(2)
    h2 = w2 = 12
    T = 40
    lat = torch.randn(1, h2 * w2, 64, generator=g).bfloat16()
    pe = torch.randn(1, T, cfg.joint_attention_dim, generator=g).bfloat16()
    pool = torch.randn(1, cfg.pooled_projection_dim, generator=g).bfloat16()
    img_ids, txt_ids = R.latent_image_ids(h2, w2), torch.zeros(T, 3)
    t, gd = torch.tensor([0.7324]), torch.tensor([3.5])
    with torch.no_grad():
        ref16 = R.transformer_forward(sd, cfg, lat, pe, pool, t.bfloat16(), img_ids.bfloat16(), txt_ids.bfloat16(), gd)
        R.INT8_BLOCK_LINEARS = True
        try:
            ref8 = R.transformer_forward(sd, cfg, lat, pe, pool, t.bfloat16(), img_ids.bfloat16(), txt_ids.bfloat16(), gd)
        finally:
            R.INT8_BLOCK_LINEARS = False
    rel = lambda a, b: float((a.float().cpu() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt())
    out16 = m.forward(lat.cuda(), pe.cuda(), pool.cuda(), t.bfloat16().cuda(), img_ids, txt_ids, gd)[0].clone()
    m.set_precision("int8")
    out8 = m.forward(lat.cuda(), pe.cuda(), pool.cuda(), t.bfloat16().cuda(), img_ids, txt_ids, gd)[0].clone()
    m.set_precision("bf16")
    torch.cuda.synchronize()
    e16, e88, d_hip, d_ref = rel(out16, ref16[0]), rel(out8, ref8[0]), rel(out8, out16.cpu()), rel(ref8[0], ref16[0])
    print(f"tiny config: hip~bf16-oracle {e16:.4f}  hip-int8~oracle-int8 {e88:.4f}  int8~bf16 hip {d_hip:.4f} oracle {d_ref:.4f}")
    assert e16 < 2e-2 and e88 < 2e-2
    assert d_hip < 3e-2 and abs(d_hip - d_ref) < 0.5 * d_ref + 2e-3


def test_int8_history_scales_track_the_dynamic_path(hip):
    """td_flux_set_act_scales(1): from the second denoise step on, the MLP operands are quantised inside the producing GEMM epilogue under
    the previous step's per-token maxima x 1.25.  A 6-step tiny denoise must stay close to the per-step-measured ("dynamic") int8 path --
    far closer than int8 is to bf16 -- and a repeated run must reproduce itself bit for bit (the history is rebuilt from step 0)."""
    from oracle import flux_ref as R
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig, effective_scalar
    cfg = R.tiny_config(num_layers=2, num_single_layers=3)
    sd = R.init_weights(cfg, seed=8)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=2, num_single_layers=3, num_attention_heads=cfg.num_attention_heads,
                                                     joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                               max_img_tokens=1024, max_txt_tokens=64, max_steps=8)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    h2 = w2 = 24
    T, n = 40, 6
    lat = torch.randn(h2 * w2, 64, generator=g).bfloat16().cuda()
    pe = torch.randn(T, cfg.joint_attention_dim, generator=g).bfloat16().cuda()
    pool = torch.randn(cfg.pooled_projection_dim, generator=g).bfloat16().cuda()
    sig = R.make_sigmas(n, h2 * w2)
    m.set_condition(pe, pool, R.latent_image_ids(h2, w2))
    m.set_timesteps([effective_scalar(float(s) * 1000.0, torch.bfloat16) for s in sig[:-1]], 3500.0)
    outs = {}
    for name, kw in (("bf16", dict(precision="bf16")), ("dynamic", dict(precision="int8")), ("history", dict(precision="int8", act_scales="history")),
                     ("history2", dict(precision="int8", act_scales="history"))):
        m.set_precision(**kw)
        x = lat.clone()
        m.denoise(x, sig)
        torch.cuda.synchronize()
        outs[name] = x.float().cpu()
    m.set_precision("bf16")
    rel = lambda a, b: float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
    d_int8, d_hist, d_hd = rel(outs["dynamic"], outs["bf16"]), rel(outs["history"], outs["bf16"]), rel(outs["history"], outs["dynamic"])
    print(f"6-step tiny denoise: int8 dynamic~bf16 {d_int8:.4f}  int8 history~bf16 {d_hist:.4f}  history~dynamic {d_hd:.4f}")
    assert torch.isfinite(outs["history"]).all() and torch.equal(outs["history"], outs["history2"])
    assert 0 < d_hd and d_hist < 1.5 * d_int8 + 1e-3


def test_int8_smoothing_neutralises_outlier_channels_and_is_inert_without_them(hip):
    """td_flux_set_smoothing(1) on a tiny model.  (a) A checkpoint without outlier channels: after the calibration forward (bf16 path) the smoothed
    int8 forward equals the plain int8 forward BIT FOR BIT -- no channel is flagged, every factor is 1, the replicated-channel columns are zero.
    (b) The same checkpoint with a trained DiT's statistics injected -- residual-stream channels x32 out of both embedders and the blocks' output
    Linears, MLP intermediate channels x32 with their consuming columns / 32 (tests/full_depth_common.py::stress_plan in miniature): plain int8 loses
    the bulk of every row to the outliers' step, the smoothed form (outlier channels divided by powers of two, replicated in the contraction or
    folded into their small weight columns) stays close to bf16."""
    from oracle import flux_ref as R
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig
    cfg = R.tiny_config(num_layers=2, num_single_layers=2)
    D, Mh = cfg.inner_dim, 4 * cfg.inner_dim
    g = torch.Generator().manual_seed(5)
    h2 = w2 = 16
    T = 40
    lat = torch.randn(1, h2 * w2, 64, generator=g).bfloat16().cuda()
    pe = torch.randn(1, T, cfg.joint_attention_dim, generator=g).bfloat16().cuda()
    pool = torch.randn(1, cfg.pooled_projection_dim, generator=g).bfloat16().cuda()
    img_ids, txt_ids = R.latent_image_ids(h2, w2), torch.zeros(T, 3)
    t, gd = torch.tensor([0.7324]).bfloat16().cuda(), torch.tensor([3.5])
    rel = lambda a, b: float((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt())

    def model(sd):
        m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=2, num_single_layers=2, num_attention_heads=cfg.num_attention_heads,
                                                         joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                                   max_img_tokens=256, max_txt_tokens=64, max_steps=4)
        m.load_state_dict(sd)
        return m

    def run(m, **kw):
        m.set_precision(**kw)
        outs = [m.forward(lat, pe, pool, t, img_ids, txt_ids, gd)[0].clone() for _ in range(2)]      # with smoothing, the first forward calibrates
        torch.cuda.synchronize()
        return outs[1]
    plain = R.init_weights(cfg, seed=6)
    m = model(plain)
    b16, i8, i8s = run(m, precision="bf16"), run(m, precision="int8"), run(m, precision="int8", smoothing=True)
    assert torch.equal(i8s, i8), "without outlier channels the smoothed form must be the plain one"
    hot = {k: v.clone() for k, v in plain.items()}
    res_ch, mlp_ch = (3, 200, 411), (5, 700, 1500, 2040)
    for name in hot:
        lin = name.rsplit(".", 1)[0]
        if lin in ("x_embedder", "context_embedder") or lin.endswith(("attn.to_out.0", "attn.to_add_out", "ff.net.2", "ff_context.net.2", "proj_out")):
            for c in res_ch:
                if hot[name].shape[0] == D:
                    hot[name][c] *= 32.0 if lin.endswith("embedder") else 8.0
        if lin.endswith(("ff.net.0.proj", "ff_context.net.0.proj", "proj_mlp")):
            for c in mlp_ch:
                hot[name][c] *= 32.0
        if lin.endswith(("ff.net.2", "ff_context.net.2")) and name.endswith(".weight"):
            for c in mlp_ch:
                hot[name][:, c] /= 32.0
        if lin.endswith("proj_out") and lin.startswith("single") and name.endswith(".weight"):
            for c in mlp_ch:
                hot[name][:, D + c] /= 32.0
    m = model(hot)
    b16, i8, i8s, i8sh = run(m, precision="bf16"), run(m, precision="int8"), run(m, precision="int8", smoothing=True), None
    e_plain, e_smooth = rel(i8, b16), rel(i8s, b16)
    print(f"tiny model with outlier channels: int8~bf16 {e_plain:.4f}, smoothed int8~bf16 {e_smooth:.4f}")
    assert torch.isfinite(i8s.float()).all()
    assert e_smooth < 0.5 * e_plain and e_smooth < 3e-2
    m.set_precision("bf16")


def test_history_is_per_image_and_per_configuration(hip):
    """History scales (int8) and history reference points (e4m3 attention) are the previous step's state OF THE SAME IMAGE UNDER THE SAME NUMERIC
    CONFIGURATION.  A forward at step k > 0 that follows ANOTHER image's step k - 1 (partial schedules, img2img, two conditions per step), or a change
    of precision / Linear classes / scale mode / attention mode between two steps, must run the self-contained (dynamic) path: bit-equal to a context
    that never had a history.  (Before round 4 it inherited the other image's per-token maxima as clip scales, and after a precision switch scales of
    ~1e-32 from all-zero maxima.)"""
    from oracle import flux_ref as R
    from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig, effective_scalar
    cfg = R.tiny_config(num_layers=2, num_single_layers=2)
    sd = R.init_weights(cfg, seed=12)
    m = FluxTransformer2DModel(FluxTransformerConfig(num_layers=2, num_single_layers=2, num_attention_heads=cfg.num_attention_heads,
                                                     joint_attention_dim=cfg.joint_attention_dim, pooled_projection_dim=cfg.pooled_projection_dim),
                               max_img_tokens=1024, max_txt_tokens=64, max_steps=8)
    m.load_state_dict(sd)
    g = torch.Generator().manual_seed(4)
    h2 = w2 = 24
    T, n = 40, 6
    mk = lambda *s: torch.randn(*s, generator=g).bfloat16().cuda()
    latA, latB, peA, peB, poolA, poolB = mk(h2 * w2, 64), mk(h2 * w2, 64) * 3, mk(T, cfg.joint_attention_dim), mk(T, cfg.joint_attention_dim) * 2, mk(cfg.pooled_projection_dim), mk(cfg.pooled_projection_dim)
    sig = R.make_sigmas(n, h2 * w2)
    ts = [effective_scalar(float(s) * 1000.0, torch.bfloat16) for s in sig[:-1]]

    def cond(pe, pool):
        m.set_condition(pe, pool, R.latent_image_ids(h2, w2))
        m.set_timesteps(ts, 3500.0)
    for kw, att in ((dict(precision="int8", act_scales="history"), "bf16"), (dict(precision="int8", act_scales="history"), "fp8"), (dict(precision="bf16"), "fp8")):
        # reference: image B's step 3 on a context with no history at all (dynamic scales / first-tile references are what a history-less step uses)
        m.set_precision(**kw)
        m.set_attention(att)
        cond(peB, poolB)
        want = m.forward_step(latB, 3).clone()
        # image A runs steps 0..2, then image B's conditioning arrives and ITS step 3 is asked for
        cond(peA, poolA)
        for i in range(3):
            m.forward_step(latA, i)
        cond(peB, poolB)
        got = m.forward_step(latB, 3).clone()
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"{kw} / attention {att}: step 3 of a new image used the previous image's history"
        # a configuration switch between two consecutive steps of one image: step 3 after steps 0..2 in ANOTHER mode == a history-less step 3
        cond(peB, poolB)
        m.set_precision("bf16")
        m.set_attention("bf16")
        for i in range(3):
            m.forward_step(latB, i)
        m.set_precision(**kw)
        m.set_attention(att)
        got2 = m.forward_step(latB, 3).clone()
        torch.cuda.synchronize()
        assert torch.isfinite(got2.float()).all() and torch.equal(got2, want), f"{kw} / attention {att}: history survived a precision / attention switch"
    m.set_precision("bf16")
    m.set_attention("bf16")

"""Full-depth parity: the whole FLUX.1-dev-shaped model (19 double + 38 single blocks, 11.9 B parameters), 1024 x 1024, through
the VAE to uint8 pixels, at the headline configuration's STATED LENGTH -- 28 Euler steps.

The expected values are committed fixtures, tests/golden/full_depth_{cfg2_T193,cfg5_T258}.pt, made once in the build container by
tests/golden/make_full_depth_golden.py: oracle/flux_ref.py (torch CPU bf16 = the reference pipeline's arithmetic) + oracle/vae_ref.py,
latents after steps 1, 2, 4, 8, 14, 21, 28 and the uint8 image.  Weights, latents and prompt embeddings come from the integer
generator of tests/full_depth_common.py, regenerated here ON THE DEVICE bit for bit (the checkpoint's checksum is compared with
the one stored in the fixture), so nothing of 24 GB travels and the oracle does not run on the GPU box.

 * bf16 (BASELINE config 2, T = 193; and config 5's T = 258): HIP latents against the fixture after every stored step; pixels
   <= 1e-2 RMSE on [0,1] (the north-star bar, BASELINE.json) after the full 28 steps.
 * fp8 (BASELINE config 5): every policy bench.py reports -- all block Linears, single-stream blocks only -- against THE FIXTURE
   IMAGE (the oracle), not against HIP bf16.
 * config 5's ragged shape: one double + one single block at full width with T = 258 text tokens (S = 4354), live oracle.
"""
import json
import os
import time

import pytest
import torch

from oracle import flux_ref as R
from oracle import vae_ref as V
import full_depth_common as C

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _rel_rmse(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def _px_rmse(u8, ref_u8):
    return float(((u8.float().cpu() - ref_u8.float().cpu()) / 255).pow(2).mean().sqrt())


def _record(key, value):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    fn = os.path.join(out, "full_depth_parity.json")
    data = json.load(open(fn)) if os.path.exists(fn) else {}
    from thinkdiff._hip import kernel_source_digest
    head = ""
    try:      # (the GPU box's snapshot has no .git: the commit is then named by whoever copies the record into profiles/)
        import subprocess
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:  # noqa: BLE001
        pass
    # which kernel sources these figures were measured on: bench.py quotes the record only when it runs the same sources
    data["_meta"] = {"kernel_source_sha256": kernel_source_digest(), "git_head": head or data.get("_meta", {}).get("git_head", ""),
                     "recorded_utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime())}
    data[key] = value
    with open(fn, "w") as fh:
        json.dump(data, fh, indent=1)


def _fixture(job):
    fn = os.path.join(GOLD, f"full_depth_{job}.pt")
    assert os.path.exists(fn), f"{fn} missing: run tests/golden/make_full_depth_golden.py in the build container"
    return torch.load(fn)


def _checksum(tensors):
    return int(sum(int(t.view(torch.int16).to(torch.int64).sum()) for t in tensors) & ((1 << 63) - 1))


class _FullModel:
    """FLUX.1-dev-shaped transformer + VAE on the device; `ensure(profile)` (re)draws the fixtures' checkpoint of that profile tensor by tensor
    (tests/full_depth_common.py: "plain" = i.i.d. 0.02 IH4, "stress" = the heavy-tailed one) and returns its checksum."""

    def __init__(self):
        from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt
        from thinkdiff.models.flux_transformer import FluxTransformer2DModel
        from thinkdiff.models.flux_vae import AutoencoderKLDecoder
        self.tr = FluxTransformer2DModel(max_img_tokens=4096, max_txt_tokens=512, max_steps=32)
        vsd = C.draw_vae_weights(V.param_shapes(V.VaeConfig()), device="cuda")
        vae = AutoencoderKLDecoder()
        vae.load_state_dict(vsd)
        self.vck = _checksum(vsd.values())
        self.pipe = FluxPipelineRewritePrompt(transformer=self.tr, vae=vae)
        self.pipe.images_in_flight = 1
        self.profile, self.ck = None, None

    def ensure(self, profile):
        if self.profile == profile:
            return self.ck
        t0 = time.time()
        self.tr.set_precision("bf16")
        self.tr.set_attention("bf16")
        names, ck, n = set(), 0, 0
        for name, t in C.draw_flux_weights(R.param_shapes(R.FluxConfig()), device="cuda", profile=profile):
            self.tr.load_state_dict({name: t}, strict=False)
            ck += int(t.view(torch.int16).to(torch.int64).sum())
            n += t.numel()
            names.add(name)
        torch.cuda.synchronize()
        assert names == set(self.tr.param_table())
        self.profile, self.ck = profile, ck & ((1 << 63) - 1)
        print(f"[full depth] '{profile}' checkpoint: {n / 1e9:.2f} B parameters regenerated on the device in {time.time() - t0:.0f} s, checksum {self.ck:#x}")
        return self.ck


@pytest.fixture(scope="module")
def full_model(hip):
    return _FullModel()


def _hip_trajectory(pipe, T, seed, steps, side=128, profile="plain", prompt_embeds=None):
    """28 Euler steps on the HIP engine; returns {step: packed latents} for `steps` and the final latents."""
    from thinkdiff.models.flux_transformer import effective_scalar
    raw, pe, pool = C.pipeline_inputs(T, seed, device="cuda", side=side, profile=profile)
    if prompt_embeds is not None:
        pe = prompt_embeds
    lat = R.pack_latents(raw.cpu()).cuda()
    tr = pipe.transformer
    sig = R.make_sigmas(28, (side // 2) ** 2)
    tr.set_condition(pe[0], pool[0], R.latent_image_ids(side // 2, side // 2))
    tr.set_timesteps([effective_scalar(float(s) * 1000.0, torch.bfloat16) for s in sig[:-1]],
                     float((torch.tensor([3.5]).bfloat16() * 1000).float()))
    x = lat[0].contiguous().clone()
    sig_t = torch.from_numpy(sig).cuda()
    got = {}
    for i in range(28):
        v = tr.forward_step(x, i)
        x = (x.float() + (sig_t[i + 1] - sig_t[i]) * v).bfloat16()       # scheduler.step, tensor form: == td_euler_step_bf16 bit for bit (test_rowops_gpu)
        if i + 1 in steps:
            got[i + 1] = x.clone()
    return got, x, (lat, pe, pool, sig)


@pytest.mark.parametrize("job", ["cfg2_T193", "cfg5_T258", "lvlm512_T128", "stress_T258"])
def test_full_depth_bf16_28_steps_vs_oracle_fixture(full_model, job):
    """cfg2 / cfg5: 1024 x 1024 (S = 4289 / 4354: the persistent stream-K attention).  lvlm512: the 512 x 512 the LVLM multi-image drivers
    render (S_img = 1024, T = 128: 120 attention items < 256 CUs -> the plain-grid kernel, ragged GEMM tiles) at full depth and length.
    stress: config 5's shape on the heavy-tailed checkpoint (tests/full_depth_common.py::stress_plan: residual-stream and MLP outlier channels,
    QK-norm gains 1.5 / 2.3 -> peaked softmax rows, the bounded attention form in ~40 % of the blocks and the running-maximum form in the rest,
    per-Linear weight scale 0.5x .. 2x, a heavy-tailed prompt embedding)."""
    fx = _fixture(job)
    pipe, ck, vck = full_model.pipe, full_model.ensure(fx.get("profile", "plain")), full_model.vck
    side = int(fx.get("side", 128))
    assert fx["weights_checksum"] == ck and fx["vae_checksum"] == vck, "the device did not regenerate the fixture's checkpoint"
    tr = pipe.transformer
    tr.set_precision("bf16")
    got, x, (lat, pe, pool, sig) = _hip_trajectory(pipe, fx["T"], fx["seed"], set(fx["steps"]), side, fx.get("profile", "plain"))
    # the fused in-engine loop (what the pipeline and bench.py run) gives the same latents as the stepwise form
    x2 = lat[0].contiguous().clone()
    tr.denoise(x2, sig)
    torch.cuda.synchronize()
    assert torch.equal(x2, x)
    errs = {s: _rel_rmse(got[s], fx["latents"][k]) for k, s in enumerate(fx["steps"])}
    u8 = pipe.vae.decode_packed(x2, side, side, output_type="np")
    torch.cuda.synchronize()
    assert u8.shape == (8 * side, 8 * side, 3)
    px = _px_rmse(u8, fx["image_u8"])
    # the VAE alone: HIP decode of the ORACLE's final latents against the oracle's image
    u8_o = pipe.vae.decode_packed(fx["latents"][-1].cuda().contiguous(), side, side, output_type="np")
    px_vae = _px_rmse(u8_o, fx["image_u8"])
    print(f"[full depth] {job} bf16, 28 steps vs the oracle fixture: latent rel-RMSE per step {', '.join(f'{s}: {e:.4f}' for s, e in errs.items())}; "
          f"pixel RMSE {px:.5f} on [0,1] (VAE alone on the oracle's latents: {px_vae:.5f})")
    _record(f"bf16_vs_oracle_{job}", {"steps": 28, "T": fx["T"], "latent_rel_rmse": errs, "pixel_rmse": px, "vae_only_pixel_rmse": px_vae,
                                      "oracle_seconds": fx["oracle_seconds"], "oracle_threads": fx["oracle_threads"]})
    assert px_vae < 1e-2
    assert errs[1] < 1e-2 and errs[2] < 2e-2
    assert px < float(os.environ.get("TD_BF16_PIXEL_BAR", "1e-2")), f"bf16 pixels {px:.4f} from the 28-step oracle fixture exceed the 1e-2 bar"


# name = <Linear operands>[_smooth][_history][_attn8]: per-channel smoothing (td_flux_set_smoothing), MLP scales from the previous step
# (td_flux_set_act_scales), joint attention on the e4m3 MFMA (td_flux_set_attention)
POLICIES = (("bf16", None), ("fp8", None), ("fp8_single", ["single_in", "single_out"]), ("int8", None), ("int8_history", None),
            ("bf16_attn8", None), ("int8_history_attn8", None), ("int8_smooth", None), ("int8_smooth_history", None), ("int8_smooth_history_attn8", None))


def apply_policy(tr, prec, gemms=None):
    tr.set_precision(prec.split("_")[0], fp8_gemms=gemms, act_scales="history" if "_history" in prec else "dynamic", smoothing="_smooth" in prec)
    tr.set_attention("fp8" if prec.endswith("_attn8") else "bf16")


def _grade_policies(full_model, job):
    """Every 8-bit policy bench.py lists, 28 steps through the pipeline, graded against the ORACLE's image of fixture `job`."""
    fx = _fixture(job)
    profile = fx.get("profile", "plain")
    assert fx["weights_checksum"] == full_model.ensure(profile)
    pipe = full_model.pipe
    raw, pe, pool = C.pipeline_inputs(fx["T"], fx["seed"], device="cuda", profile=profile)
    lat = R.pack_latents(raw.cpu()).cuda()
    tr = pipe.transformer
    res = {}
    for prec, gemms in POLICIES:
        apply_policy(tr, prec, gemms)
        out = pipe(prompt_embeds=pe, pooled_prompt_embeds=pool, height=1024, width=1024, num_inference_steps=28, guidance_scale=3.5,
                   latents=lat, output_type="latent").images[0].clone()
        u8 = pipe.vae.decode_packed(out, 128, 128, output_type="np").clone()
        torch.cuda.synchronize()
        assert torch.isfinite(out.float()).all(), f"{prec}: non-finite latents on the '{profile}' checkpoint"
        res[prec] = {"latent_rel_rmse_vs_oracle": _rel_rmse(out, fx["latents"][-1]), "pixel_rmse_vs_oracle": _px_rmse(u8, fx["image_u8"]), "u8": u8}
    tr.set_precision("bf16")
    tr.set_attention("bf16")
    for k, _ in POLICIES[1:]:
        res[k]["pixel_rmse_vs_hip_bf16"] = _px_rmse(res[k]["u8"], res["bf16"]["u8"])
    for k, v in res.items():
        v.pop("u8")
        print(f"[full depth] {job} {k:18s} 28 steps: " + ", ".join(f"{a} {b:.5f}" for a, b in v.items()))
    _record(f"fp8_policies_vs_oracle_{job}", res)
    return res


def test_full_depth_fp8_policies_vs_oracle_fixture(full_model):
    """Config 5 (T = 258), the plain checkpoint: every 8-bit policy bench.py reports, graded against the ORACLE's image."""
    res = _grade_policies(full_model, "cfg5_T258")
    # the ordering that must hold whatever the absolute level: more fp8 Linears -> further from the oracle
    assert res["bf16"]["pixel_rmse_vs_oracle"] <= res["fp8_single"]["pixel_rmse_vs_oracle"] <= res["fp8"]["pixel_rmse_vs_oracle"]
    assert res["fp8"]["pixel_rmse_vs_oracle"] < float(os.environ.get("TD_FP8_PIXEL_BAR", "3e-2")), "all-fp8 pixels left the recorded level"
    # the 8-bit mode that holds the north-star bar with EVERY block Linear quantised: symmetric int8 (uniform step, exact accumulation)
    assert res["int8"]["pixel_rmse_vs_oracle"] < 1e-2, f"int8 pixels {res['int8']['pixel_rmse_vs_oracle']:.4f} from the oracle fixture exceed the 1e-2 bar"
    # ... also with the MLP operands quantised in the producing epilogues under the previous step's per-token scales (td_flux_set_act_scales)
    assert res["int8_history"]["pixel_rmse_vs_oracle"] < 1e-2, f"int8 (history scales) pixels {res['int8_history']['pixel_rmse_vs_oracle']:.4f} exceed the 1e-2 bar"
    # ... and with the joint attention on the e4m3 MFMA as well (td_flux_set_attention): alone on the bf16 Linears, and under the whole 8-bit path
    assert res["bf16_attn8"]["pixel_rmse_vs_oracle"] < 1e-2, f"bf16 Linears + e4m3 attention: pixels {res['bf16_attn8']['pixel_rmse_vs_oracle']:.4f} exceed the 1e-2 bar"
    assert res["int8_history_attn8"]["pixel_rmse_vs_oracle"] < 1e-2, f"int8 (history) + e4m3 attention: pixels {res['int8_history_attn8']['pixel_rmse_vs_oracle']:.4f} exceed the 1e-2 bar"
    # ... and the smoothed forms are the plain ones here (no channel of this checkpoint is an outlier): same level
    for k in ("int8_smooth", "int8_smooth_history", "int8_smooth_history_attn8"):
        assert res[k]["pixel_rmse_vs_oracle"] < 1e-2, f"{k}: {res[k]['pixel_rmse_vs_oracle']:.4f} exceeds the 1e-2 bar on the plain fixture"


def test_full_depth_8bit_policies_on_the_stress_checkpoint(full_model):
    """The same grading on the heavy-tailed checkpoint and prompt (fixture stress_T258): which 8-bit policies survive outlier channels and peaked
    softmax rows is RECORDED (gpurun_out/full_depth_parity.json -> profiles/), and bench.py quotes a policy as in tolerance only if it is inside the
    bar on BOTH fixtures.  Asserted: every policy stays finite, bf16 holds the bar, and the policy bench.py ships as its 8-bit line holds it too."""
    res = _grade_policies(full_model, "stress_T258")
    assert res["bf16"]["pixel_rmse_vs_oracle"] < 1e-2
    # what round 4 found and the smoothing exists for: per-token int8 collapses under outlier channels (2.9e-2), e4m3 does not care about them (its
    # own mantissa noise, 1.6e-2), and int8 with per-channel smoothing + outlier-channel replication is back at the plain checkpoint's level
    assert res["int8"]["pixel_rmse_vs_oracle"] > 1.5e-2, "plain int8 is expected to break on this checkpoint: if it does not, the fixture lost its outliers"
    for k in ("int8_smooth", "int8_smooth_history", "int8_smooth_history_attn8", "bf16_attn8"):
        assert res[k]["pixel_rmse_vs_oracle"] < 1e-2, f"{k}: {res[k]['pixel_rmse_vs_oracle']:.4f} on the heavy-tailed fixture exceeds the 1e-2 bar"
    shipped = os.environ.get("TD_SHIPPED_8BIT_POLICY", "int8_smooth_history_attn8")      # bench.py's default 8-bit policy (--workload config5)
    assert res[shipped]["pixel_rmse_vs_oracle"] < 1e-2, f"{shipped}: {res[shipped]['pixel_rmse_vs_oracle']:.4f} on the stress fixture"


def test_gemm_launch_forms_do_not_change_the_engine(full_model, monkeypatch):
    """Full size, 3 denoise steps per policy: the (opt-in) tail split of the 256 x 256 GEMM tile (csrc/gemm_bf16.hip) against the plain one-tile-per-workgroup
    launch -- bit-identical latents in bf16, int8 (dynamic and history scales: the int8-output epilogue) and with the e4m3 attention."""
    fx = _fixture("cfg5_T258")
    assert fx["weights_checksum"] == full_model.ensure("plain")
    pipe = full_model.pipe
    raw, pe, pool = C.pipeline_inputs(fx["T"], fx["seed"], device="cuda")
    lat = R.pack_latents(raw.cpu()).cuda()
    tr = pipe.transformer

    def run(prec):
        apply_policy(tr, prec)
        out = pipe(prompt_embeds=pe, pooled_prompt_embeds=pool, height=1024, width=1024, num_inference_steps=3, guidance_scale=3.5,
                   latents=lat, output_type="latent").images[0].clone()
        torch.cuda.synchronize()
        return out
    for prec in ("bf16", "int8", "int8_history", "int8_history_attn8", "int8_smooth_history_attn8"):
        monkeypatch.delenv("TD_GEMM_TAIL", raising=False)
        base = run(prec)
        monkeypatch.setenv("TD_GEMM_TAIL", "auto")
        got = run(prec)
        monkeypatch.delenv("TD_GEMM_TAIL")
        assert torch.equal(got, base), f"{prec}: the tail-split GEMM launches changed the latents"
    tr.set_precision("bf16")
    tr.set_attention("bf16")


def test_config3_lvlm_full_size_vs_oracle_fixture(full_model):
    """BASELINE config 3 at full size (reference scripts/test/test_mllama_t5_decoder_flux.py:143-196, configs/test_thinkdiff_lvlm_ccsbu_image_text.yaml:24-35):
    a 28-layer Qwen2-VL-7B-shaped decoder (7.6 B parameters, regenerated on the device) takes one image + instruction request -- 103 prompt rows with
    64 spliced vision tokens and 2-D M-RoPE streams -- and is teacher-forced through max_tokens = min_tokens = 128 KV-cached decode steps;
    `model.norm` hidden states of the output tokens -> aligner (fp32 T5LayerNorm) -> prompt_embeds [1, 128, 4096] -> FLUX 1024 x 1024, 28 steps,
    guidance 3.5 (T = 128, S = 4224) -> VAE -> uint8, all against tests/golden/full_depth_cfg3_lvlm7b.pt.  Bars: pixels <= 1e-2 RMSE (the north-star
    bar); decoder hidden states no further from exact arithmetic than 1.5 x the bf16 oracle is (the fixture carries both)."""
    from oracle import aligner_ref as A
    from oracle import qwen2vl_ref as Q
    from thinkdiff.models.mllama_vllm_t5_embed_decoder_2 import MllamaVllmT5EmbedDecoderForConditionalGeneration_5
    from thinkdiff.models.qwen2_vl import Qwen2VLTextEngine
    fx = _fixture("cfg3_lvlm7b")
    qcfg = Q.Qwen2Config()
    t0 = time.time()
    m = MllamaVllmT5EmbedDecoderForConditionalGeneration_5(vllm_config={"max_model_len": 8192, "max_tokens": 128, "min_tokens": 128, "temperature": 0.6, "top_p": 0.9,
                                                                        "ignore_eos": True, "max_num_seqs": 1})      # text_config default = Qwen2-VL-7B
    qck = 0
    for name, t in C.draw_qwen_weights(Q.param_shapes(qcfg), device="cuda"):
        m.mllama.load_state_dict({name: t}, strict=False)
        qck += int(t.view(torch.int16).to(torch.int64).sum())
    asd = C.draw_aligner_weights(A.param_shapes(qcfg.hidden, 4096), device="cuda")
    m.load_state_dict(asd)
    torch.cuda.synchronize()
    assert qck & ((1 << 63) - 1) == fx["qwen_checksum"] and _checksum(asd.values()) == fx["aligner_checksum"], "the device did not regenerate the fixture's LVLM checkpoint"
    print(f"[config 3] Qwen2-VL-7B-shaped decoder + aligner regenerated on the device in {time.time() - t0:.0f} s")
    rq = C.lvlm_request(qcfg.vocab, qcfg.hidden, device="cuda")
    assert torch.equal(Qwen2VLTextEngine.mrope_position_ids(rq["prompt_ids"], rq["grid"]), rq["position_ids"])      # the product's host logic gives the fixture's M-RoPE streams
    emb = m.mllama.embed_tokens(rq["prompt_ids"])
    emb[(torch.tensor(rq["prompt_ids"]) == C.IMAGE_PAD).cuda()] = rq["vision_rows"]
    req = {"prompt_token_ids": rq["prompt_ids"], "inputs_embeds": emb, "position_ids": rq["position_ids"]}
    # the decoder alone: prompt and output hidden states
    g = m.mllama.generate(rq["prompt_ids"], m.mllama_sampling_params, position_ids=rq["position_ids"], inputs_embeds=emb, forced_output_ids=rq["forced_ids"])
    torch.cuda.synchronize()
    e_p, e_o = _rel_rmse(g["prompt_hidden_states"], fx["prompt_hidden"]), _rel_rmse(g["hidden_states"], fx["output_hidden"])
    # ... and against the same graph in exact arithmetic (tests/golden/add_lvlm_fp32_reference.py): 28 random-weight layers amplify rounding noise, so
    # two correct bf16 implementations end up several per cent apart; what a correct engine cannot be is further from the exact result than the bf16
    # oracle itself is (x 1.5: the criterion of tests/test_qwen2_gpu.py at small size)
    x_p, x_o = _rel_rmse(g["prompt_hidden_states"], fx["prompt_hidden_fp32"]), _rel_rmse(g["hidden_states"], fx["output_hidden_fp32"])
    r_p, r_o = _rel_rmse(fx["prompt_hidden"], fx["prompt_hidden_fp32"]), _rel_rmse(fx["output_hidden"], fx["output_hidden_fp32"])
    assert g["hidden_states"].shape == (128, qcfg.hidden) and g["token_ids"] == rq["forced_ids"]
    # the product call: get_embed(output_embed) = hidden states of the 128 output tokens through the aligner
    t0 = time.time()
    embs, texts = m.get_embed([req], embedding_type="output_embed", need_process=False, forced_output_ids=[rq["forced_ids"]], max_new_tokens=128)
    torch.cuda.synchronize()
    t_embed = time.time() - t0
    assert embs[0].shape == (128, 4096) and texts == [" ".join(map(str, rq["forced_ids"]))]
    e_a, x_a, r_a = _rel_rmse(embs[0], fx["aligner_out"]), _rel_rmse(embs[0], fx["aligner_out_fp32"]), _rel_rmse(fx["aligner_out"], fx["aligner_out_fp32"])
    print(f"[config 3] hidden states, HIP vs the bf16 oracle: prompt {e_p:.4f}, 128 output tokens {e_o:.4f}, aligner output {e_a:.4f}; HIP vs exact arithmetic: "
          f"{x_p:.4f}, {x_o:.4f}, {x_a:.4f}; the bf16 oracle vs exact arithmetic: {r_p:.4f}, {r_o:.4f}, {r_a:.4f}; get_embed {t_embed:.2f} s")
    # sampled (not forced) generation at the config's temperature / top-p: 128 tokens, finite hidden states (the reference samples; parity is teacher-forced)
    smp = m.mllama.generate(rq["prompt_ids"], m.mllama_sampling_params, position_ids=rq["position_ids"], inputs_embeds=emb, generator=torch.Generator(device="cuda").manual_seed(42))
    assert len(smp["token_ids"]) == 128 and torch.isfinite(smp["hidden_states"].float()).all()
    # FLUX on the aligner's output, 1024 x 1024, T = 128
    pipe, ck = full_model.pipe, full_model.ensure("plain")
    assert fx["weights_checksum"] == ck and fx["vae_checksum"] == full_model.vck
    pipe.transformer.set_precision("bf16")
    got, x, (lat, _, pool, sig) = _hip_trajectory(pipe, fx["T"], fx["seed"], set(fx["steps"]), 128, prompt_embeds=embs[0][None].contiguous())
    errs = {s: _rel_rmse(got[s], fx["latents"][k]) for k, s in enumerate(fx["steps"])}
    # ... and through the pipeline's own call, as the driver makes it (encode_prompt(prompt="", prompt_embeds=...) -> diffusion_pipe(prompt_embeds, pooled))
    out = pipe(prompt_embeds=embs[0][None].contiguous(), pooled_prompt_embeds=pool, height=1024, width=1024, num_inference_steps=28, guidance_scale=3.5,
               latents=lat, output_type="latent").images[0]
    torch.cuda.synchronize()
    assert torch.equal(out, x)
    u8 = pipe.vae.decode_packed(x, 128, 128, output_type="np")
    px = _px_rmse(u8, fx["image_u8"])
    print(f"[config 3] FLUX 1024^2 T=128 on the aligner output: latent rel-RMSE per step {', '.join(f'{s}: {e:.4f}' for s, e in errs.items())}; pixel RMSE {px:.5f}")
    _record("config3_lvlm7b_vs_oracle", {"prompt_hidden_rel_rmse": e_p, "output_hidden_rel_rmse": e_o, "aligner_out_rel_rmse": e_a, "get_embed_seconds": t_embed,
                                         "vs_exact_arithmetic": {"hip": [x_p, x_o, x_a], "bf16_oracle": [r_p, r_o, r_a]},
                                         "latent_rel_rmse": errs, "pixel_rmse": px, "oracle_seconds": fx["oracle_seconds"]})
    for name, x, r in (("prompt hidden states", x_p, r_p), ("output hidden states", x_o, r_o), ("aligner output", x_a, r_a)):
        assert x < 1.5 * r + 2e-3, f"config 3 {name}: HIP is {x:.4f} from exact arithmetic, the bf16 oracle only {r:.4f}"
    assert max(e_p, e_o, e_a) < 2.5 * max(r_p, r_o, r_a), "HIP and the bf16 oracle are further apart than two bf16 roundings of the same graph can be"
    assert px < 1e-2, f"config 3 pixels {px:.4f} from the oracle fixture exceed the 1e-2 bar"


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

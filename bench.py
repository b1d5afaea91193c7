#!/usr/bin/env python3
"""Headline benchmark: images/sec, ThinkDiff-CLIP -> FLUX.1-dev, 1024x1024, 28 steps (BASELINE.json).

One "step" = one image: the reference driver's `diffusion_pipe(prompt_embeds=[1,193,4096],
pooled_prompt_embeds=[1,768], height=width=1024, num_inference_steps=28, guidance_scale=3.5)` call
(scripts/test/test_blip_vision_t5_decoder_flux_text.py:234-242) on synthetic inputs and a seeded
random-init checkpoint of the full FLUX.1-dev shape (11.9 B params), inputs resident in HBM.
Every rank generates its own image (the reference's multi-GPU semantics: same prompt list, seed+rank),
so scaling is weak and there is no data-path collective.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
sys.path.insert(0, ROOT)

HEIGHT = WIDTH = 1024
NUM_STEPS = 28
T_TXT = 193            # 65 aligner tokens + 128 T5 tokens
GUIDANCE = 3.5
BF16_DENSE_PEAK_TFLOPS = 2500.0   # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
FP8_DENSE_PEAK_TFLOPS = 5000.0
RESULT_OUT = sys.stdout            # the stream the ONE JSON line goes to (main() re-points it at the process's original stdout)
FP8_DTYPE = "fp8_e4m3 block-GEMM operands, fp32 accumulate, bf16 elsewhere"
INT8_DTYPE = "int8 block-GEMM operands (symmetric W8A8), exact int32 accumulate, bf16 elsewhere"


def flux_flops_per_forward(s_img: int, s_txt: int) -> float:
    """SURVEY.md 8(d) algorithmic work of one FluxTransformer2DModel.forward, B=1 (FLUX.1-dev dims)."""
    D, M = 3072, 12288
    S = s_img + s_txt
    per_tok = 4 * D * D + 2 * D * M            # 113.25 M MACs per token per (double-stream) block
    lin = 2 * s_img * 64 * D + 2 * s_txt * 4096 * D + 19 * 2 * per_tok * S + 38 * 2 * per_tok * S + 2 * s_img * D * 64
    attn = (19 + 38) * 4 * S * S * D
    return lin + attn + 2.9e9                  # + modulation / time-MLP GEMVs


def cpu_baseline(threads: int):
    """Times the oracle (oracle/flux_ref.py = the reference's own bf16 torch-CPU arithmetic) on a bounded
    sample -- one double-stream + one single-stream FLUX.1-dev block -- and extrapolates to one image."""
    from oracle import flux_ref as R
    torch.set_num_threads(threads)
    cfg = R.FluxConfig()
    D = cfg.inner_dim
    # calibrate the host so the sample stays ~10-30 s
    a = torch.randn(1024, D).bfloat16(); w = torch.randn(D, D).bfloat16()
    torch.nn.functional.linear(a, w)
    t0 = time.time(); torch.nn.functional.linear(a, w); cal = time.time() - t0
    tfs = 2 * 1024 * D * D / cal / 1e12
    s_img = 4096
    full_blocks_flop = 2 * 2 * (4 * D * D + 2 * D * 4 * D) * (s_img + T_TXT) + 2 * 4 * (s_img + T_TXT) ** 2 * D
    while s_img > 256 and full_blocks_flop * (s_img + T_TXT) / (4096 + T_TXT) / (tfs * 1e12) > 30.0:
        s_img //= 2
    g = torch.Generator().manual_seed(0)
    shapes = {k: v for k, v in R.param_shapes(cfg).items()
              if k.startswith("transformer_blocks.0.") or k.startswith("single_transformer_blocks.0.")}
    sd = {k: (0.02 * torch.randn(v, generator=g)).bfloat16() for k, v in shapes.items()}
    S = s_img + T_TXT
    hidden = torch.randn(1, s_img, D, generator=g).bfloat16()
    enc = torch.randn(1, T_TXT, D, generator=g).bfloat16()
    temb = torch.randn(1, D, generator=g).bfloat16()
    side = int(s_img ** 0.5)
    cos, sin = R.rope_tables(torch.cat([torch.zeros(T_TXT, 3), R.latent_image_ids(side, s_img // side)]), cfg.axes_dims_rope)
    with torch.no_grad():
        e2, h2 = R.double_block(sd, cfg, 0, hidden, enc, temb, cos, sin)   # warm-up (page-in, oneDNN primitives)
        joint = torch.cat([e2, h2], dim=1)
        R.single_block(sd, cfg, 0, joint, temb, cos, sin)
        # a FIXED number of repetitions and the BEST time of each block: other tenants of the host only ever add time, so the
        # minimum is the figure that reproduces (a time-boxed mean moved by 30 % with the host's load, the median by 17 %);
        # ~10-30 s of CPU work in all
        reps, td, ts = 9, [], []
        for _ in range(reps):
            t0 = time.time(); R.double_block(sd, cfg, 0, hidden, enc, temb, cos, sin); td.append(time.time() - t0)
            t0 = time.time(); R.single_block(sd, cfg, 0, joint, temb, cos, sin); ts.append(time.time() - t0)
        t_d, t_s = min(td), min(ts)
    # scale the two block times to the cfg-2 token count by algorithmic FLOPs (exact when s_img == 4096)
    def blk_flop(si, dbl):
        s = si + T_TXT
        return 2 * (4 * D * D + 2 * D * 4 * D) * s + 4 * s * s * D
    k = blk_flop(4096, True) / blk_flop(s_img, True)
    sec_per_image = NUM_STEPS * (19 * t_d + 38 * t_s) * k
    return {
        "value": 1.0 / sec_per_image, "unit": "images/s", "cores": threads, "kind": "port",
        "sample": (f"oracle/flux_ref.py (torch CPU bf16, the reference pipeline's arithmetic): 1 double-stream block "
                   f"({t_d:.2f} s) + 1 single-stream block ({t_s:.2f} s) of FLUX.1-dev at S_img={s_img}, T={T_TXT}, best of {reps} reps, {threads} threads; "
                   f"extrapolated x(19, 38) blocks x {NUM_STEPS} steps" + ("" if s_img == 4096 else f" x{k:.2f} FLOP ratio to S_img=4096")
                   + f" = {sec_per_image:.0f} s/image" + _whole_run_note()),
    }


def _whole_run_note():
    """The one directly measured whole run of this workload on a CPU: the 28-step, full-depth oracle trajectory that became the
    parity fixture (tests/golden/make_full_depth_golden.py, build container).  Quoted beside the extrapolated sample as a cross-check."""
    fn = os.path.join(ROOT, "tests", "golden", "full_depth_cfg2_T193.pt")
    try:
        fx = torch.load(fn)
        return (f"; cross-check: the whole 28-step 1024x1024 oracle run measured once in the build container took {fx['oracle_seconds']:.0f} s "
                f"on {fx['oracle_threads']} cores (tests/golden/full_depth_cfg2_T193.pt)")
    except Exception:  # noqa: BLE001
        return ""


def fp8_leg(pipe, G, rank, steps=2):
    """BASELINE config 5's shape on this rank after the headline: two-image composition (T = 2 x 65 aligner + 128 T5 = 258 text
    tokens, S = 4354), e4m3 operands for every block GEMM, `G` images in flight, `steps` timed steps after one warm-up.
    Not the headline (`value` above is bf16, the reference's precision); recorded so that the driver's run carries the fp8 rate."""
    from thinkdiff import _hip
    T5 = 258
    g = torch.Generator().manual_seed(4242 + rank)
    pe = (0.1 * torch.randn(G, T5, 4096, generator=g)).bfloat16().cuda()
    pooled = torch.randn(G, 768, generator=g).bfloat16().cuda()
    raw = torch.randn(G, 16, HEIGHT // 8, WIDTH // 8, generator=g).bfloat16().cuda()
    packed = torch.stack([_hip.flux_pack_latents(raw[i]) for i in range(G)])
    tr = pipe.transformer

    def run(n):
        return pipe(prompt_embeds=pe[:n], pooled_prompt_embeds=pooled[:n], num_images_per_prompt=1, height=HEIGHT, width=WIDTH,
                    num_inference_steps=NUM_STEPS, guidance_scale=GUIDANCE, latents=packed[:n], output_type="pil").images

    def measure(fp8_gemms, precision="fp8", act_scales="dynamic", attention="bf16", smoothing=False):
        tr.set_precision(precision, fp8_gemms=fp8_gemms, act_scales=act_scales, smoothing=smoothing)
        tr.set_attention(attention)
        run(G)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = run(G)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        run(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run(1)
        torch.cuda.synchronize()
        single = time.perf_counter() - t1
        assert len(out) == G and out[0].size == (WIDTH, HEIGHT)
        return el, single

    el, single = measure(None)                                     # every block Linear in fp8
    el_s, single_s = measure(["single_in", "single_out"])          # the 38 single-stream blocks in fp8, the 19 double-stream ones in bf16
    el_i, single_i = measure(None, "int8")                         # every block Linear on symmetric int8 operands (TD_PRECISION_INT8)
    el_h, single_h = measure(None, "int8", "history")              # ... the MLP operands quantised in the producing epilogues (previous step's scales)
    el_a, single_a = measure(None, "int8", "history", "fp8")       # ... and QK^T / P.V of the joint attention on the e4m3 MFMA (td_flux_set_attention)
    el_m, single_m = measure(None, "int8", "history", "fp8", True)  # ... and per-channel smoothing with outlier-channel replication (td_flux_set_smoothing): what survives heavy tails
    tr.set_precision("bf16")
    tr.set_attention("bf16")
    fl = NUM_STEPS * flux_flops_per_forward(4096, T5)
    def entry(key, elapsed, single_s, dtype, what, peak=FP8_DENSE_PEAK_TFLOPS):
        par = _policy_parity(key)
        return {"value": steps * G / elapsed, "unit": "images/s/GPU", "one_image_in_flight": 1.0 / single_s, "dtype": dtype, "what": what,
                "whole_step_tflops_per_gpu": fl * G / (elapsed / steps) / 1e12, "frac_of_8bit_dense_peak": fl * G / (elapsed / steps) / 1e12 / peak,
                "parity": par, "inside_1e-2_bar": bool(par["inside_1e-2_bar_on_every_fixture"])}

    e4m3 = "fp8_e4m3 block-GEMM operands (per-channel weight / per-token activation scales), fp32 accumulate, bf16 elsewhere"
    i8 = "int8 block-GEMM operands (symmetric, per-channel weight / per-token activation scales), exact int32 accumulate (v_mfma_i32_16x16x64_i8, the fp8 MFMA rate)"
    policies = {
        "e4m3_all_block_linears": entry("fp8", el, single, e4m3, "td_flux_set_precision(TD_PRECISION_FP8_E4M3), every block Linear"),
        "e4m3_single_stream_blocks_only": entry("fp8_single", el_s, single_s, e4m3, "the 38 single-stream blocks in e4m3, the 19 double-stream blocks in bf16 (td_flux_set_fp8_gemms)"),
        "int8_all_block_linears": entry("int8", el_i, single_i, i8 + ", bf16 elsewhere", "td_flux_set_precision(TD_PRECISION_INT8), every block Linear, activation scales measured on the spot"),
        "int8_history_scales": entry("int8_history", el_h, single_h, i8 + ", bf16 elsewhere",
                                     "the line above + td_flux_set_act_scales(1): per-token scales of the MLP operands from the previous denoise step x 1.25, int8 written by the "
                                     "producing GEMM / attention epilogues (no quantisation pass over them)"),
        "int8_history_scales_e4m3_attention": entry("int8_history_attn8", el_a, single_a, i8 + "; QK^T and P.V of the joint attention on v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 q / k / v / "
                                                    "probabilities under power-of-two scales, fp32 accumulate and softmax state); bf16 elsewhere",
                                                    "the line above + td_flux_set_attention(TD_ATTENTION_FP8) (csrc/attention_fp8.hip; QK-RMSNorm + RoPE inside its pack pass)"),
        "int8_smoothed_history_scales_e4m3_attention": entry("int8_smooth_history_attn8", el_m, single_m, i8 + ", outlier channels of the LayerNorm outputs / MLP intermediates divided by powers of two and "
                                                             "replicated in the contraction or folded into their small weight columns; e4m3 joint attention; bf16 elsewhere",
                                                             "the line above + td_flux_set_smoothing(1): calibrated on the first forward, LayerNorm-fed Linears contract over K + 128 channels"),
    }
    inside = {k: v for k, v in policies.items() if v["inside_1e-2_bar"]}
    best = max(inside or policies, key=lambda k: policies[k]["value"])
    head = dict(policies[best])
    head.update({"policy": best, "steps": steps, "ms_per_step": steps * G / head["value"] / steps * 1e3, "images_per_step": G,
                 "workload": "BASELINE config 5 shape per GPU: ThinkDiff-CLIP two-image composition, T_txt=258 (2 x 65 aligner + 128 T5), joint S=4354, "
                             "1024x1024, 28 steps, FLUX.1-dev shape, denoise + VAE decode + uint8/PIL",
                 "selection": "`value` = the fastest 8-bit policy whose committed full-depth parity record (made on THESE kernel sources) is inside the 1e-2 "
                              "pixel-RMSE bar against the 28-step oracle image on EVERY fixture -- the plain and the heavy-tailed checkpoint (none inside: the fastest, "
                              "`inside_1e-2_bar` false); every measured policy is listed under `policies`",
                 "policies": policies})
    return head


def _parity_record():
    """The newest committed full-depth parity record (profiles/r*_full_depth_parity.json, written by tests/test_flux_full_depth_gpu.py
    on the GPU box): the 8-bit lines quote their vs-oracle pixel RMSE from there instead of carrying hand-typed figures.  The record names
    the kernel sources it was measured on (sha256 over csrc/, thinkdiff/_hip.py::kernel_source_digest); a record made on other sources is
    marked STALE and vouches for nothing."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_full_depth_parity.json")), reverse=True):
        with open(fn) as fh:
            d = json.load(fh)
        if "fp8_policies_vs_oracle_cfg5_T258" in d:
            d["_file"] = os.path.relpath(fn, ROOT)
            return d
    return {}


def _policy_parity(key):
    """What the committed record says about 8-bit policy `key` (a name of tests/test_flux_full_depth_gpu.py::POLICIES): pixel RMSE on [0,1] against
    the 28-step ORACLE image on each full-depth fixture -- the plain i.i.d. checkpoint (cfg5_T258) and the heavy-tailed one (stress_T258: outlier
    channels, peaked softmax rows).  QUOTED from the record, not measured in this run; both checkpoints are synthetic (no trained FLUX weights
    can be fetched here)."""
    from thinkdiff._hip import kernel_source_digest
    par = _parity_record()
    meta = par.get("_meta", {})
    fresh = bool(meta) and meta.get("kernel_source_sha256") == kernel_source_digest()
    fixtures = {}
    for job in ("cfg5_T258", "stress_T258"):
        v = par.get(f"fp8_policies_vs_oracle_{job}", {}).get(key or "", {}).get("pixel_rmse_vs_oracle")
        fixtures[job] = {"pixel_rmse_vs_oracle": v, "hip_bf16_itself": par.get(f"fp8_policies_vs_oracle_{job}", {}).get("bf16", {}).get("pixel_rmse_vs_oracle")}
    inside = fresh and all(f["pixel_rmse_vs_oracle"] is not None and f["pixel_rmse_vs_oracle"] <= 1e-2 for f in fixtures.values())
    return {"policy": key, "fixtures": fixtures, "inside_1e-2_bar_on_every_fixture": bool(inside),
            "record": par.get("_file"), "record_git_head": meta.get("git_head"), "record_kernel_sources": "match this tree" if fresh else "STALE or absent: the record was made on other kernel sources and is not quoted as evidence",
            "note": "quoted from the committed record of tests/test_flux_full_depth_gpu.py (asserted green in the same round's GPU suite), not measured in this run; "
                    "tolerance verified on SYNTHETIC checkpoints only (plain i.i.d. and heavy-tailed stress profile), not on trained FLUX.1-dev weights"}


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` ("td_gemm_bf16_nt_kernel<8,4>" ...) from the newest committed PMC passes.  The kernel's launch forms (the
    trailing template arguments: tail split 1 / 2 / 4) are separate entries of the record: their launch-weighted mean is what one launch moved."""
    import glob
    want = kernel.replace(" ", "").rstrip(">") + ",false,false,false"      # bf16, no conv: "td_gemm_bf16_nt_kernel<8,4,false,false,false" (+ ">" or ",TAIL>")
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")), reverse=True):
        with open(fn) as fh:
            prof = json.load(fh)
        hit = [v for name, v in prof["kernels"].items() if want in name.replace(" ", "")]
        n = sum(v["launches"] for v in hit)
        if n:
            return {"traffic": sum(v["hbm_bytes_per_launch"] * v["launches"] for v in hit) / n, "traffic_unit": "bytes/launch",
                    "traffic_source": f"{os.path.basename(fn)}: {prof['source']}; {prof['correction']}"}
    return {"traffic": None}


def launch_ranks(n: int, argv, deadline_s: float = 3000.0) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves -- one child process per GPU, env contract of
    torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), rendezvous on 127.0.0.1.  The parent has
    made no GPU call and makes none (a process that initialised the GPU must not start others on this pool), never execs, relays
    rank 0's JSON line and returns non-zero if any rank failed (the others are then terminated by PID) -- or if the ranks are still
    running `deadline_s` seconds after the start (a rendezvous that never completes, a stalled RCCL init): they are then terminated by
    PID as well, so a hung job never outlives its launcher."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    rc = 0
    live = set(range(n))
    t_start = time.monotonic()
    while live and rc == 0:
        if time.monotonic() - t_start > deadline_s:
            rc = 124
            print(f"bench.py: ranks {sorted(live)} still running after {deadline_s:.0f} s (--timeout): terminating them", file=sys.stderr)
            break
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
        time.sleep(0.05)
    for r in live:                                               # a rank failed: stop exactly the children we started
        procs[r].terminate()
    for r in live:
        try:
            procs[r].wait(timeout=30)
        except subprocess.TimeoutExpired:
            procs[r].kill()
    for line in procs[0].stdout.read().splitlines():             # the result line to stdout; library chatter (gloo / RCCL banners) to stderr
        print(line, file=sys.stdout if (rc == 0 and line.startswith("{")) else sys.stderr, flush=True)
    return rc


def _per_rank(dist, dev, world, **seconds):
    """{name: [seconds of rank 0, 1, ...]}: every rank's own figures gathered for rank 0's line (MAX alone hides which rank was slow, and
    whether the time went into the run or into initialisation)."""
    names = sorted(seconds)
    t = torch.tensor([seconds[k] for k in names], dtype=torch.float64, device=dev)
    if dist is None:
        return {k: [float(v)] for k, v in zip(names, t.tolist())}
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return {k: [float(o[i]) for o in out] for i, k in enumerate(names)}


class _DryRunPipeline:
    """--dry-run: a stand-in for the HIP pipeline so the launcher / rendezvous / sharding / gather / JSON path of this file can
    be exercised on a GPU-less box (tests/test_bench_cpu.py).  It renders nothing and its rate means nothing."""
    transformer = None

    def __call__(self, prompt_embeds, **kw):
        from types import SimpleNamespace
        from PIL import Image
        time.sleep(0.01 * prompt_embeds.shape[0])
        return SimpleNamespace(images=[Image.new("RGB", (WIDTH, HEIGHT)) for _ in range(prompt_embeds.shape[0])])


def config5_leg(a, pipe, dist, rank, world, dev):
    """--workload config5 (BASELINE config 5): ONE batch of `--prompts` (64) two-image-composition jobs sharded over the ranks,
    the drivers' `run.shard_prompts` path (thinkdiff/runners/dp_inference.py): rank 0 plans the job list (one seed per job, so
    an image does not depend on the world size) -> broadcast_work_list (RCCL one-to-all) -> jobs[rank::world] -> every rank
    renders its share, `--in-flight` images at a time -> gather_results (all-to-one) of (job, bytes rendered) in job order.
    A step = the whole batch; value = prompts / wall time, scaling "strong".  Reference semantics:
    runs/test_thinkdiff_clip_two_images.sh, scripts/test/test_mllama_t5_decoder_flux.py:57-65 (every rank the whole list)."""
    from thinkdiff.runners.dp_inference import broadcast_work_list, gather_results, shard
    T5, G = 258, max(1, a.in_flight)

    def fence():
        if dev != "cpu":
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if dev != "cpu":
            torch.cuda.synchronize()

    def render(jobs):
        done = []
        for i in range(0, len(jobs), G):
            chunk = jobs[i:i + G]
            pe, pooled, lat = [], [], []
            for _, seed in chunk:
                g = torch.Generator().manual_seed(seed)
                pe.append(0.1 * torch.randn(T5, 4096, generator=g))
                pooled.append(torch.randn(768, generator=g))
                lat.append(torch.randn(16, HEIGHT // 8, WIDTH // 8, generator=g))
            pe, pooled, lat = (torch.stack(x).bfloat16().to(dev) for x in (pe, pooled, lat))
            if dev != "cpu":
                from thinkdiff import _hip
                lat = torch.stack([_hip.flux_pack_latents(lat[k]) for k in range(len(chunk))])
            imgs = pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, num_images_per_prompt=1, height=HEIGHT, width=WIDTH,
                        num_inference_steps=NUM_STEPS, guidance_scale=GUIDANCE, latents=lat, output_type="pil").images
            assert len(imgs) == len(chunk) and imgs[0].size == (WIDTH, HEIGHT)
            done += [(j, WIDTH * HEIGHT * 3) for j, _ in chunk]
        return done

    for _ in range(a.warmup):                                       # warm-up = one chunk per rank, not a whole batch
        render([(-1, 7 + rank)] * G)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        plan = [(j, 4242 + j) for j in range(a.prompts)] if rank == 0 else None
        jobs = shard(broadcast_work_list(plan), rank, world)
        got = gather_results(render(jobs))
        if rank == 0:
            assert [j for j, _ in got] == list(range(a.prompts)), "gathered results are not the planned job list"
    fence()
    return time.perf_counter() - t0


def side_workload(a, dist, rank, world, dev, t_proc0=None):
    """Everything that is not the driver's headline run: --workload config5 (real pipeline, fp8 by default) and --dry-run of either
    workload (stub pipeline on the CPU, gloo).  Same protocol as the headline: W warm-up steps, K timed steps between
    barrier + synchronize on both sides, MAX over ranks, one JSON line from rank 0."""
    if a.dry_run:
        pipe = _DryRunPipeline()
    else:
        from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt
        pipe = FluxPipelineRewritePrompt.from_random(seed=1234, max_img_tokens=4096, max_txt_tokens=512, max_steps=32)
        pipe.transformer.set_precision(a.precision, act_scales=a.act_scales if a.precision == "int8" else "dynamic", smoothing=a.smoothing == "on")
        pipe.transformer.set_attention(a.attention)
        pipe.images_in_flight = max(1, a.in_flight)
    G = max(1, a.in_flight)
    t_init = time.perf_counter() - (t_proc0 if t_proc0 is not None else time.perf_counter())
    if a.workload == "config5":
        elapsed = config5_leg(a, pipe, dist, rank, world, dev)
        images, T, scaling = a.steps * a.prompts, 258, "strong"
        workload = (f"BASELINE config 5: ThinkDiff-CLIP two-image composition, ONE batch of {a.prompts} prompts sharded over {world} rank(s) "
                    "(rank 0 plans -> broadcast -> jobs[rank::world] -> gather), T_txt=258 (2 x 65 aligner + 128 T5), joint S=4354, FLUX.1-dev shape, "
                    f"1024x1024, 28 steps, denoise + VAE decode + uint8/PIL, {G} images in flight per rank; a step = the whole batch")
    else:                                                            # dry run of the headline protocol
        pe = torch.zeros(G, T_TXT, 4096)

        def fence():
            if dist is not None:
                dist.barrier()
        for _ in range(a.warmup):
            pipe(prompt_embeds=pe)
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = pipe(prompt_embeds=pe).images
        fence()
        elapsed = time.perf_counter() - t0
        assert len(out) == G
        images, T, scaling = world * a.steps * G, T_TXT, "weak"
        workload = "BASELINE config 2 protocol with a stub pipeline"
    ranks = _per_rank(dist, dev, world, init_s=t_init, timed_s=elapsed)
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        fl = NUM_STEPS * flux_flops_per_forward(4096, T)
        res = {"metric": "images/sec (1024², 28 steps) ThinkDiff-CLIP FLUX.1 at 1/2/4/8 MI355X", "value": images / elapsed, "unit": "images/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
               "scaling": scaling, "vs_baseline": None,
               "dtype": {"bf16": "bf16", "fp8": FP8_DTYPE, "int8": INT8_DTYPE}[a.precision] + ("; per-channel smoothing with outlier-channel replication" if a.smoothing == "on" else "")
                        + ("; joint attention QK^T / P.V on the e4m3 MFMA" if a.attention == "fp8" else ""),
               "data": "synthetic" if not a.dry_run else "none (dry run: stub pipeline on the CPU, gloo; the rate is meaningless)",
               "config": {"workload": workload, "precision": a.precision, "act_scales": a.act_scales if a.precision == "int8" else None,
                          "smoothing": a.smoothing if a.precision == "int8" else None, "attention": a.attention,
                          "images_per_rank_per_step": G if a.workload == "config2" else None,
                          "prompts": a.prompts if a.workload == "config5" else None,
                          "parallelism": f"dp{world} (" + ("sharded job list" if a.workload == "config5" else "independent images, seed+rank") + ")"},
               "per_rank_seconds": ranks, "dry_run": bool(a.dry_run)}
        if not a.dry_run:
            res["whole_step_tflops"] = fl * images / elapsed / 1e12
            if a.precision != "bf16":
                res["parity"] = _policy_parity({("fp8", "dynamic", "bf16", "off"): "fp8", ("int8", "dynamic", "bf16", "off"): "int8", ("int8", "history", "bf16", "off"): "int8_history",
                                                ("int8", "history", "fp8", "off"): "int8_history_attn8", ("int8", "dynamic", "bf16", "on"): "int8_smooth",
                                                ("int8", "history", "bf16", "on"): "int8_smooth_history",
                                                ("int8", "history", "fp8", "on"): "int8_smooth_history_attn8"}.get((a.precision, a.act_scales, a.attention, a.smoothing)))
        print(json.dumps(res), file=RESULT_OUT, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    t_proc0 = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE under an external launcher, else 1)")
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 3; 1 for --workload config5)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-trace", action="store_true", help="skip the per-launch HIP-event trace (roofline leg)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="independent images advanced concurrently per rank (engine contexts on separate streams); a step = this many images")
    ap.add_argument("--precision", choices=("bf16", "fp8", "int8"), default=None,
                    help="operand type of the block GEMMs; bf16 = the headline (reference precision), fp8 = e4m3, int8 = symmetric W8A8 "
                         "(--workload config5 defaults to int8 + --act-scales history + --attention fp8: the 8-bit policy inside the 1e-2 pixel bar)")
    ap.add_argument("--act-scales", choices=("dynamic", "history"), default=None,
                    help="--precision int8 only: per-token activation scales measured on the spot, or taken from the previous denoise step (td_flux_set_act_scales)")
    ap.add_argument("--smoothing", choices=("on", "off"), default=None,
                    help="--precision int8 only: per-channel smoothing with outlier-channel replication, calibrated on the first forward (td_flux_set_smoothing)")
    ap.add_argument("--attention", choices=("bf16", "fp8"), default=None,
                    help="arithmetic of the joint attention: bf16 = the reference graph's (headline), fp8 = QK^T / P.V on the e4m3 MFMA "
                         "(td_flux_set_attention; meant for the 8-bit precisions)")
    ap.add_argument("--no-fp8-leg", action="store_true",
                    help="skip the short fp8 measurement of BASELINE config 5's shape that the default bf16 run appends as the `fp8` sub-object")
    ap.add_argument("--workload", choices=("config2", "config5"), default="config2",
                    help="config2 = the headline (every rank its own images, weak scaling); config5 = one batch of --prompts two-image "
                         "compositions sharded over the ranks (strong scaling)")
    ap.add_argument("--prompts", type=int, default=64, help="batch size of --workload config5")
    ap.add_argument("--timeout", type=float, default=3000.0,
                    help="bare `--gpus N` launcher only: seconds after which ranks that are still running are terminated and the run fails (rc 124)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: gloo backend and a stub pipeline; exercises the launcher, rendezvous, sharding and JSON path only")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 1 if a.workload == "config5" else 3
    if a.precision is None and a.workload == "config5":
        # BASELINE config 5 names the 8-bit MFMA path; its default here is the 8-bit policy that holds the reference tolerance (6.4e-3 pixel RMSE
        # against the 28-step oracle fixture): int8 block Linears under history scales + the e4m3 joint attention.  `--precision fp8` = all-e4m3 Linears (1.7e-2).
        a.precision = "int8"
        a.act_scales = a.act_scales or "history"
        a.attention = a.attention or "fp8"
        a.smoothing = a.smoothing or "on"
    a.precision = a.precision or "bf16"
    a.act_scales = a.act_scales or "dynamic"
    a.attention = a.attention or "bf16"
    a.smoothing = (a.smoothing or "off") if a.precision == "int8" else "off"

    if a.gpus is None:                                            # under torch.distributed.run the world size is the launcher's
        a.gpus = int(os.environ.get("WORLD_SIZE", "1")) if "RANK" in os.environ else 1
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:], a.timeout))      # BEFORE anything touches the GPU in this process
    # ONE line on stdout: libraries write banners there (RCCL prints its version block at communicator creation, gloo its
    # connection notes), so this process keeps the real stdout for the result and points fd 1 at stderr for everything else.
    global RESULT_OUT
    sys.stdout.flush()
    RESULT_OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start `python bench.py --gpus N` bare (it spawns its ranks) or under "
                         "torch.distributed.run --nproc-per-node N")
    if os.environ.get("TD_BENCH_FAIL_RANK") == str(rank):       # tests/test_bench_cpu.py: a rank that dies before the rendezvous
        raise SystemExit(f"rank {rank}: failing on request (TD_BENCH_FAIL_RANK)")
    if os.environ.get("TD_BENCH_HANG_RANK") == str(rank):       # tests/test_bench_cpu.py: a rank that never reaches the rendezvous
        time.sleep(3600)
    dev = "cpu" if a.dry_run else "cuda"
    if not a.dry_run:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("TD_BENCH_FORCE_DIST"):   # the env switch lets a 1-GPU box rehearse the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if a.dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if a.dry_run or a.workload == "config5":
        return side_workload(a, dist, rank, world, dev, t_proc0)

    from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt
    pipe = FluxPipelineRewritePrompt.from_random(seed=1234, max_img_tokens=4096, max_txt_tokens=512, max_steps=32)
    tr = pipe.transformer
    tr.set_precision(a.precision, act_scales=a.act_scales if a.precision == "int8" else "dynamic", smoothing=a.smoothing == "on")
    tr.set_attention(a.attention)

    # synthetic inputs (SURVEY.md 8d cfg 2), seed + rank as the reference drivers do
    g = torch.Generator().manual_seed(42 + rank)
    G = max(1, a.in_flight)
    pipe.images_in_flight = G
    prompt_embeds = (0.1 * torch.randn(G, T_TXT, 4096, generator=g)).bfloat16().cuda()     # G different prompts
    pooled = torch.randn(G, 768, generator=g).bfloat16().cuda()
    raw = torch.randn(G, 16, HEIGHT // 8, WIDTH // 8, generator=g).bfloat16().cuda()
    from thinkdiff import _hip
    packed = torch.stack([_hip.flux_pack_latents(raw[i]) for i in range(G)])
    torch.cuda.synchronize()

    def one_image(n=G):
        return pipe(prompt_embeds=prompt_embeds[:n], pooled_prompt_embeds=pooled[:n], num_images_per_prompt=1,
                    height=HEIGHT, width=WIDTH, num_inference_steps=NUM_STEPS, guidance_scale=GUIDANCE,
                    latents=packed[:n], output_type="pil").images

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_init = time.perf_counter() - t_proc0                            # imports, rendezvous, weights, inputs
    for _ in range(a.warmup):
        out = one_image()
    fence()
    t_warm = time.perf_counter() - t_proc0 - t_init
    trace = not a.no_trace
    t0 = time.perf_counter()
    for i in range(a.steps):
        if trace and G == 1 and i == a.steps - 1:
            # one image in flight: the per-launch HIP-event trace brackets every kernel of the LAST timed image
            # (tracing all K would cost ~4 % of `value`: two event records per launch)
            tr.trace_begin(NUM_STEPS * 520 + 64)
        out = one_image()
    fence()
    elapsed = time.perf_counter() - t0
    cats = tr.trace_end() if (trace and G == 1) else None
    assert len(out) == G and out[0].size == (WIDTH, HEIGHT) and out[0].mode == "RGB", "pipeline did not return decoded images"
    single = None
    if G > 1 and rank == 0:
        # With several images in flight the kernels of different images overlap on the chip, so a per-launch event
        # interval is no longer that kernel's own duration: the roofline leg (and the one-image-at-a-time rate) are
        # taken on images run alone, right after the timed region, same process, same weights and inputs.
        one_image(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_image(1)
        torch.cuda.synchronize()
        single = time.perf_counter() - t1
        if trace:
            tr.trace_begin(NUM_STEPS * 520 + 64)
            one_image(1)
            cats = tr.trace_end()

    ranks = _per_rank(dist, "cuda", world, init_s=t_init, warmup_s=t_warm, timed_s=elapsed)
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = world * a.steps * G / elapsed
        flops_img = NUM_STEPS * flux_flops_per_forward(4096, T_TXT)
        res = {
            "metric": "images/sec (1024², 28 steps) ThinkDiff-CLIP FLUX.1 at 1/2/4/8 MI355X",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp8": "fp8_e4m3 block-GEMM operands (per-channel weight / per-token activation scales), fp32 accumulate, bf16 elsewhere",
                      "int8": INT8_DTYPE}[a.precision],
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE config 2: ThinkDiff-CLIP single image+text, FLUX.1-dev shape (11.9 B params, seeded "
                             "random init), 1024x1024, 28 Euler steps, T_txt=193 (65 aligner + 128 T5), joint S=4289, "
                             "guidance 3.5; step = `images_per_rank_per_step` independent images per rank (different prompts and latents, "
                             "advanced concurrently on separate streams over one set of weights), each from HBM-resident prompt_embeds/pooled/latents through the 28-step "
                             "denoise loop, VAE decode (FLUX.1-dev VAE shape, seeded random init) and uint8 conversion to a host "
                             "PIL image -- the reference driver's diffusion_pipe(...).images[0]"),
                "precision": a.precision, "attention": a.attention, "images_per_rank_per_step": G,
                "parallelism": f"dp{world} (independent images, seed+rank)",
                "algorithmic_pflop_per_image": flops_img / 1e15,
                "weights_note": ("all 11.9 B parameters drawn N(0, 0.02).  Lines recorded before this note existed (round 1: 0.703, round 2 up to "
                                 "0.736) ran with every block weight at zero -- a truncated fill launch -- which let the chip clock ~20 % higher; "
                                 "DESIGN.md section 5"),
            },
            "whole_step_tflops_per_gpu": flops_img * G / (elapsed / a.steps) / 1e12,
            "per_rank_seconds": ranks,      # init (imports, rendezvous, weights), warm-up and the timed region of every rank: `value` uses the MAX of timed_s
        }
        if single is not None:
            res["one_image_in_flight"] = {"value": 1.0 / single, "unit": "images/s/GPU", "ms_per_image": single * 1e3}
        if cats:
            kern = {"gemm_256x256": "td_gemm_bf16_nt_kernel<8,4>", "gemm_288x192": "td_gemm_bf16_nt_kernel<9,3>"}
            dom = max(kern, key=lambda k: cats[k]["ms"])      # the GEMM tile variant with the most device time
            gm = cats[dom]
            ach = gm["flops"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
            peak = BF16_DENSE_PEAK_TFLOPS if a.precision == "bf16" else FP8_DENSE_PEAK_TFLOPS
            res["roofline"] = {
                "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                "frac": ach / peak, "traffic": None,
                "kernel": kern[dom] if a.precision == "bf16" else kern[dom].replace(">", "," + a.precision + ">"),
                "launches": gm["launches"], "avg_launch_us": gm["ms"] * 1e3 / max(gm["launches"], 1),
                "avg_flops_per_launch": gm["flops"] / max(gm["launches"], 1),
            }
            # HBM bytes per launch of the same kernel from the committed PMC passes of this command (rocprofv3 cannot run
            # inside the timed process): profiles/r*_hbm_traffic.json, made by tools/pmc_traffic.py
            if a.precision == "bf16":
                res["roofline"].update(_pmc_traffic(kern[dom]))
            res["roofline"]["sampled"] = ("every launch of the kernel in the last image of the timed region (HIP events on the launch stream)" if G == 1 else
                                          "every launch of the kernel in one image run alone right after the timed region (HIP events on the launch stream); "
                                          "inside the timed region kernels of the images in flight overlap, so per-launch intervals are not kernel durations")
            res["kernel_ms_per_image"] = {k: v["ms"] for k, v in cats.items()}
            at = cats["attention"]
            res["attention_tflops"] = at["flops"] / (at["ms"] * 1e-3) / 1e12 if at["ms"] > 0 else 0.0
            res["attention_roofline"] = {"bound": "mfma", "achieved": res["attention_tflops"], "peak": peak, "unit": "TFLOP/s",
                                         "frac": res["attention_tflops"] / peak,
                                         "kernel": "td_attn_fwd_d128_streamk_kernel<8,true,true,true,true>" if a.attention == "bf16" else "td_attn_fp8_pack_kernel + td_attn_fwd_d128_fp8_kernel<8,true,true>",
                                         "launches": at["launches"], "avg_launch_us": at["ms"] * 1e3 / max(at["launches"], 1)}
        if world == 1 and a.precision == "bf16" and not a.no_fp8_leg:
            res["fp8"] = fp8_leg(pipe, G, rank)
        if world == 1 and not a.no_cpu_baseline:
            # the 1-GPU box's CPU share is 16 cores; never oversubscribe past the affinity mask
            res["cpu_baseline"] = cpu_baseline(min(len(os.sched_getaffinity(0)), 16))
        print(json.dumps(res), file=RESULT_OUT, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Generates tests/golden/full_depth_{cfg2_T193,cfg5_T258,lvlm512_T128,stress_T258,cfg3_lvlm7b}.pt -- run ONCE, in the build container (CPU only, ~40 GB of RAM):

    python tests/golden/make_full_depth_golden.py [job ...]

The headline configuration at its stated length (BASELINE config 2: 1024 x 1024, 28 steps, guidance 3.5, T = 193 prompt tokens =
65 aligner + 128 T5; reference configs/test_thinkdiff_clip_image_text.yaml:91-95, call site
scripts/test/test_blip_vision_t5_decoder_flux_text.py:234-242) and config 5's token count (T = 258 = 2 x 65 + 128;
scripts/test/test_blip_vision_t5_decoder_flux.py:220-228), through oracle/flux_ref.py (bf16 = the reference's arithmetic, every
op rounding) and oracle/vae_ref.py, on the FLUX.1-dev-shaped synthetic checkpoint of tests/full_depth_common.py.

Stored per job: the packed latents [4096, 64] bf16 after Euler steps GOLDEN_STEPS, the uint8 image [1024, 1024, 3], and the
generator's identity (seed, a checksum of the drawn weights) so the GPU test can prove it regenerated the same checkpoint.
The oracle is "parity unpinned" (diffusers is absent): these are vectors of the restated algorithm, not of diffusers itself.
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

from oracle import aligner_ref as A       # noqa: E402
from oracle import flux_ref as R          # noqa: E402
from oracle import qwen2vl_ref as Q       # noqa: E402
from oracle import vae_ref as V           # noqa: E402
import full_depth_common as C              # noqa: E402


def checksum(sd):
    """Order-independent 64-bit sum of every parameter's bit pattern (int16 view, exact in int64)."""
    return int(sum(int(t.view(torch.int16).to(torch.int64).sum()) for t in sd.values()) & ((1 << 63) - 1))


def lvlm_stage():
    """BASELINE config 3's front half (reference scripts/test/test_mllama_t5_decoder_flux.py:143-157 -> thinkdiff/models/
    mllama_vllm_t5_embed_decoder_2.py:1019-1118 get_embed, embedding_type "output_embed"): the 28-layer Qwen2-VL-7B-shaped decoder
    over [prompt ‖ 128 teacher-forced output tokens] in one causal pass (oracle/qwen2vl_ref.py, bf16), `model.norm` hidden states of
    the output tokens -> the aligner with its T5LayerNorm in fp32 (the reference keeps the aligner's parameters fp32 under bf16
    autocast, :884) -> prompt_embeds [1, 128, 4096].  Returns the fixture entries and the prompt embeddings."""
    qcfg = Q.Qwen2Config()
    t0 = time.time()
    qsd = dict(C.draw_qwen_weights(Q.param_shapes(qcfg)))
    qck = checksum(qsd)
    print(f"Qwen2-VL-7B-shaped decoder drawn in {time.time() - t0:.0f} s: {sum(t.numel() for t in qsd.values()) / 1e9:.2f} B parameters, checksum {qck:#x}", flush=True)
    rq = C.lvlm_request(qcfg.vocab, qcfg.hidden)
    ids = torch.tensor(rq["prompt_ids"] + rq["forced_ids"])
    n_p, n_o = len(rq["prompt_ids"]), len(rq["forced_ids"])
    emb = torch.nn.functional.embedding(ids, qsd["model.embed_tokens.weight"])
    emb[ids == C.IMAGE_PAD] = rq["vision_rows"]                  # (the forced ids are < 151 000: no placeholder among them)
    nxt = int(rq["position_ids"].max()) + 1                      # generation continues one past the largest prompt position (all three streams)
    pos = torch.cat([rq["position_ids"], (nxt + torch.arange(n_o, dtype=torch.int32))[None].expand(3, n_o)], dim=1)
    t0 = time.time()
    with torch.no_grad():
        hid, _ = Q.text_model_hidden(qsd, qcfg, pos, inputs_embeds=emb)
    print(f"decoder pass over {n_p} + {n_o} tokens: {time.time() - t0:.0f} s", flush=True)
    del qsd
    asd = C.draw_aligner_weights(A.param_shapes(qcfg.hidden, 4096))
    h_out = hid[n_p:]
    with torch.no_grad():
        y = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(h_out, asd["mm_projector.0.weight"], asd["mm_projector.0.bias"])),
                                       asd["mm_projector.2.weight"], asd["mm_projector.2.bias"])
        pe = A.t5_layer_norm(y.float(), asd["mm_projector.3.weight"].float()).bfloat16()
    entries = {"qwen_checksum": qck, "aligner_checksum": checksum(asd), "n_prompt": n_p, "prompt_hidden": hid[:n_p].contiguous(),
               "output_hidden": h_out.contiguous(), "aligner_out": pe.contiguous()}
    return entries, pe[None].contiguous()


def main(jobs):
    torch.set_num_threads(int(os.environ.get("TD_GOLDEN_THREADS", "8")))
    C.use_fast_host_generator()      # tests/golden/hashgen.c, checked against the torch form first
    cfg, vcfg = R.FluxConfig(), V.VaeConfig()
    lvlm = {job: lvlm_stage() for job in jobs if C.GOLDEN_JOBS[job].get("lvlm")}      # before the 24 GB FLUX checkpoint is resident
    vsd = C.draw_vae_weights(V.param_shapes(vcfg))
    vck = checksum(vsd)
    sd, ck, drawn = None, None, None
    for job in sorted(jobs, key=lambda j: C.GOLDEN_JOBS[j].get("profile", "plain")):      # one draw per checkpoint profile
        spec = C.GOLDEN_JOBS[job]
        side, profile = spec["side"], spec.get("profile", "plain")
        if drawn != profile:
            del sd
            t0 = time.time()
            sd = dict(C.draw_flux_weights(R.param_shapes(cfg), profile=profile))
            ck, drawn = checksum(sd), profile
            print(f"'{profile}' weights drawn in {time.time() - t0:.0f} s: {sum(t.numel() for t in sd.values()) / 1e9:.2f} B parameters, checksum {ck:#x} / vae {vck:#x}", flush=True)
        raw, pe, pool = C.pipeline_inputs(spec["T"], spec["seed"], side=side, profile=profile)
        extra = {}
        if spec.get("lvlm"):
            extra, pe = lvlm[job]                   # the aligner's output IS the prompt embedding (encode_prompt(prompt="", prompt_embeds=...): CLIP("") pooled = `pool`)
            assert pe.shape == (1, spec["T"], 4096)
        lat = R.pack_latents(raw)
        trace, t0 = [], time.time()

        class Progress(list):
            def append(self, x):
                super().append(x)
                print(f"  {job}: step {len(self)} / 28 at {time.time() - t0:.0f} s", flush=True)
        trace = Progress()
        with torch.no_grad():
            out = R.denoise(sd, cfg, lat, pe, pool, side // 2, side // 2, 28, guidance_scale=3.5, trace=trace)
            _, u8 = V.latents_to_image(vsd, vcfg, out, side, side)
        fx = {**extra, "job": job, "T": spec["T"], "seed": spec["seed"], "side": side, "profile": profile, "weights_checksum": ck, "vae_checksum": vck,
              "steps": list(C.GOLDEN_STEPS), "latents": torch.stack([trace[s - 1][0] for s in C.GOLDEN_STEPS]).contiguous(),
              "image_u8": u8[0].contiguous(), "oracle_seconds": time.time() - t0, "oracle_threads": torch.get_num_threads(),
              "torch": str(torch.__version__)}
        torch.save(fx, os.path.join(HERE, f"full_depth_{job}.pt"))
        print(f"{job}: written ({fx['oracle_seconds']:.0f} s)", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(C.GOLDEN_JOBS))

"""Not a test: a one-off accuracy probe, run on the GPU box before the 8-bit attention kernel was written.

    python tests/probe_attention_fp8.py [cfg2_T193] [steps]

Runs oracle/flux_ref.py's denoise loop ON THE DEVICE (torch ops, the oracle's own arithmetic) on the full-depth synthetic
checkpoint with FP8_ATTENTION off / on, alone and together with INT8_BLOCK_LINEARS, and prints each image's pixel RMSE against
the committed CPU fixture.  It answers one question -- how much of the 1e-2 pixel budget an e4m3 QK^T / P.V costs over 28 steps
x 57 blocks -- without any product code in the loop.  Output: gpurun_out/attn8_probe.json.
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path[:0] = [ROOT, HERE]

from oracle import flux_ref as R          # noqa: E402
from oracle import vae_ref as V           # noqa: E402
import full_depth_common as C              # noqa: E402


def main():
    job = sys.argv[1] if len(sys.argv) > 1 else "cfg2_T193"
    which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bf16", "attn8", "int8", "int8+attn8"]
    dev = "cuda"
    fx = torch.load(os.path.join(HERE, "golden", f"full_depth_{job}.pt"))
    spec = C.GOLDEN_JOBS[job]
    side = spec["side"]
    cfg, vcfg = R.FluxConfig(), V.VaeConfig()
    sd = dict(C.draw_flux_weights(R.param_shapes(cfg), device=dev))
    vsd = C.draw_vae_weights(V.param_shapes(vcfg), device=dev)
    raw, pe, pool = C.pipeline_inputs(spec["T"], spec["seed"], device=dev, side=side)
    lat = R.pack_latents(raw)
    ref = fx["image_u8"].float() / 255.0
    res = {}
    for name in which:
        R.FP8_ATTENTION = "attn8" in name
        R.FP8_ATTENTION_PROB = "exp2" if "attn8exp" in name else "linear"
        R.INT8_BLOCK_LINEARS = "int8" in name
        t0 = time.time()
        trace = []
        with torch.no_grad():
            out = R.denoise(sd, cfg, lat, pe, pool, side // 2, side // 2, 28, guidance_scale=3.5, trace=trace)
            _, u8 = V.latents_to_image(vsd, vcfg, out, side, side)
        img = u8[0].float().cpu() / 255.0
        px = float((img - ref).pow(2).mean().sqrt())
        lt = []
        for j, s in enumerate(fx["steps"]):
            a, b = trace[s - 1][0].float().cpu(), fx["latents"][j].float()
            lt.append(float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()))
        res[name] = {"pixel_rmse_vs_fixture": px, "latent_rel_rmse": lt, "seconds": time.time() - t0}
        print(name, f"px {px:.3e}", " ".join(f"{x:.2e}" for x in lt), f"{time.time() - t0:.0f} s", flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        json.dump(res, open(os.path.join(ROOT, "gpurun_out", "attn8_probe.json"), "w"), indent=1)
    R.FP8_ATTENTION = R.INT8_BLOCK_LINEARS = False
    R.FP8_ATTENTION_PROB = "linear"


if __name__ == "__main__":
    main()

"""Generates tests/golden/full_depth_{cfg2_T193,cfg5_T258}.pt -- run ONCE, in the build container (CPU only, ~25 GB of RAM):

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

from oracle import flux_ref as R          # noqa: E402
from oracle import vae_ref as V           # noqa: E402
import full_depth_common as C              # noqa: E402


def checksum(sd):
    """Order-independent 64-bit sum of every parameter's bit pattern (int16 view, exact in int64)."""
    return int(sum(int(t.view(torch.int16).to(torch.int64).sum()) for t in sd.values()) & ((1 << 63) - 1))


def main(jobs):
    torch.set_num_threads(int(os.environ.get("TD_GOLDEN_THREADS", "8")))
    cfg, vcfg = R.FluxConfig(), V.VaeConfig()
    t0 = time.time()
    sd = dict(C.draw_flux_weights(R.param_shapes(cfg)))
    vsd = C.draw_vae_weights(V.param_shapes(vcfg))
    ck, vck = checksum(sd), checksum(vsd)
    print(f"weights drawn in {time.time() - t0:.0f} s: {sum(t.numel() for t in sd.values()) / 1e9:.2f} B parameters, checksum {ck:#x} / vae {vck:#x}", flush=True)
    for job in jobs:
        spec = C.GOLDEN_JOBS[job]
        side = spec["side"]
        raw, pe, pool = C.pipeline_inputs(spec["T"], spec["seed"], side=side)
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
        fx = {"job": job, "T": spec["T"], "seed": spec["seed"], "side": side, "weight_seed": C.WEIGHT_SEED, "weights_checksum": ck, "vae_checksum": vck,
              "steps": list(C.GOLDEN_STEPS), "latents": torch.stack([trace[s - 1][0] for s in C.GOLDEN_STEPS]).contiguous(),
              "image_u8": u8[0].contiguous(), "oracle_seconds": time.time() - t0, "oracle_threads": torch.get_num_threads(),
              "torch": str(torch.__version__)}
        torch.save(fx, os.path.join(HERE, f"full_depth_{job}.pt"))
        print(f"{job}: written ({fx['oracle_seconds']:.0f} s)", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or list(C.GOLDEN_JOBS))

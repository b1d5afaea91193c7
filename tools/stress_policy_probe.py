"""Every 8-bit policy on the heavy-tailed checkpoint (tests/full_depth_common.py, profile "stress"), 28 steps at config 5's shape, against HIP bf16 on the
same checkpoint -- a quick look that needs no oracle fixture (tests/test_flux_full_depth_gpu.py grades against the oracle's image).
usage: python tools/stress_policy_probe.py [profile=stress] [steps=28] [only policies containing this substring]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "thinkdiff-mlre_amd")]
import full_depth_common as C                      # noqa: E402
from oracle import flux_ref as R                    # noqa: E402  (parameter shapes / pack_latents only)
import test_flux_full_depth_gpu as T                # noqa: E402

profile = sys.argv[1] if len(sys.argv) > 1 else "stress"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 28
fm = T._FullModel()
fm.ensure(profile)
spec = C.GOLDEN_JOBS["stress_T258" if profile == "stress" else "cfg5_T258"]
raw, pe, pool = C.pipeline_inputs(spec["T"], spec["seed"], device="cuda", profile=profile)
lat = R.pack_latents(raw.cpu()).cuda()
tr, pipe = fm.tr, fm.pipe
res = {}
for prec, gemms in T.POLICIES:
    if len(sys.argv) > 3 and sys.argv[3] not in prec and prec != "bf16":
        continue
    T.apply_policy(tr, prec, gemms)
    out = pipe(prompt_embeds=pe, pooled_prompt_embeds=pool, height=1024, width=1024, num_inference_steps=steps, guidance_scale=3.5,
               latents=lat, output_type="latent").images[0].clone()
    u8 = pipe.vae.decode_packed(out, 128, 128, output_type="np").clone()
    torch.cuda.synchronize()
    res[prec] = (out, u8)
    b_out, b_u8 = res["bf16"]
    print(f"{profile:7s} {prec:20s} finite {bool(torch.isfinite(out.float()).all())}  latents vs HIP bf16 {T._rel_rmse(out, b_out):.5f}  pixels vs HIP bf16 {T._px_rmse(u8, b_u8):.5f}", flush=True)

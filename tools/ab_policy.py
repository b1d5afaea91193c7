"""Same-box, same-process A/B of an environment switch on whole images: config 5's shape (T = 258), 2 images in flight, a policy of
tests/test_flux_full_depth_gpu.py::POLICIES; the switch (e.g. TD_GEMM_NO_TAIL=1) is read per launch, so both arms share weights, clocks and thermals.
usage: python tools/ab_policy.py POLICY ENVVAR [rounds=3] [images in flight=2]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "thinkdiff-mlre_amd")]
from thinkdiff import _hip                                     # noqa: E402
from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt   # noqa: E402
import test_flux_full_depth_gpu as T                             # noqa: E402

policy, var = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
G = int(sys.argv[4]) if len(sys.argv) > 4 else 2
pipe = FluxPipelineRewritePrompt.from_random(seed=1234, max_img_tokens=4096, max_txt_tokens=512, max_steps=32)
pipe.images_in_flight = G
T.apply_policy(pipe.transformer, policy)
g = torch.Generator().manual_seed(1)
pe = (0.1 * torch.randn(2, 258, 4096, generator=g)).bfloat16().cuda()
pooled = torch.randn(2, 768, generator=g).bfloat16().cuda()
raw = torch.randn(2, 16, 128, 128, generator=g).bfloat16().cuda()
packed = torch.stack([_hip.flux_pack_latents(raw[i]) for i in range(2)])


def run():
    return pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, num_images_per_prompt=1, height=1024, width=1024, num_inference_steps=28, guidance_scale=3.5,
                latents=packed, output_type="pil").images


run()
res = {"off": [], "on": []}
for _ in range(rounds):
    for arm in ("off", "on"):
        if arm == "on":
            os.environ[var] = os.environ.get("TD_AB_VALUE", "1")
        else:
            os.environ.pop(var, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(); run()
        torch.cuda.synchronize()
        res[arm].append(4 / (time.perf_counter() - t0))
os.environ.pop(var, None)
print(f"{policy} ({G} in flight): {var} unset " + " ".join(f"{v:.4f}" for v in res["off"]) + f" images/s | {var}=1 " + " ".join(f"{v:.4f}" for v in res["on"]), flush=True)

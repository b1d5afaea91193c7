"""Which block Linears run in fp8: pixel deviation from the bf16 image and time per image for every policy of interest.

Full FLUX.1-dev-shaped model (19 + 38 blocks, seeded N(0, 0.02) weights -- the weights of tests/test_flux_full_depth_gpu.py), T = 258
(BASELINE config 5's token count), 1024^2, 28 steps, VAE to uint8; one image at a time.  Prints a table and writes
gpurun_out/fp8_policy_sweep.json."""
import itertools, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
from thinkdiff import _hip
from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt
from thinkdiff.models.flux_transformer import FluxTransformer2DModel

pipe = FluxPipelineRewritePrompt.from_random(seed=20251004, max_img_tokens=4096, max_txt_tokens=512, max_steps=32)
pipe.images_in_flight = 1
tr = pipe.transformer
T = 258
g = torch.Generator().manual_seed(43)
raw = torch.randn(1, 16, 128, 128, generator=g).bfloat16().cuda()
pe = (0.1 * torch.randn(1, T, 4096, generator=g)).bfloat16().cuda()
pool = torch.randn(1, 768, generator=g).bfloat16().cuda()
lat = torch.stack([_hip.flux_pack_latents(raw[0])])
kw = dict(prompt_embeds=pe, pooled_prompt_embeds=pool, height=1024, width=1024, num_inference_steps=28, guidance_scale=3.5)


def render():
    out = pipe(output_type="latent", latents=lat.clone(), **kw).images[0]
    render.latents = out.float().clone()
    return pipe.vae.decode_packed(out, 128, 128, output_type="np").float()


def timed():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    u8 = render()
    torch.cuda.synchronize()
    return u8, time.perf_counter() - t0


names = list(FluxTransformer2DModel.FP8_GEMMS)
tr.set_precision("bf16"); render()
ref, t_bf16 = timed()
ref_lat = render.latents
print(f"bf16 image: mean {float(ref.mean()):.1f} std {float(ref.std()):.1f} min {float(ref.min()):.0f} max {float(ref.max()):.0f}; latents std {float(ref_lat.std()):.3f}")
rows = [{"policy": "bf16", "mask": 0, "pixel_rmse": 0.0, "s_per_image": t_bf16}]
policies = [("all", 63), ("mlp only (ff1 ff2 + single in/out)", 4 | 8 | 16 | 32), ("double + single MLP-side without attention-side qkv/out", 4 | 8 | 16 | 32),
            ("double blocks only", 15), ("single blocks only", 48), ("ff1+ff2", 12), ("single_in", 16), ("single_out", 32), ("single_in+ff1", 20),
            ("qkv+out", 3), ("ff1 only", 4), ("ff2 only", 8), ("all but single_out", 63 - 32), ("all but single_in", 63 - 16), ("all but ff2+single_out", 63 - 8 - 32),
            ("ff1+single_in (the K = 3072 producers)", 4 | 16), ("qkv+ff1+single_in (LayerNorm-fed)", 1 | 4 | 16)]
seen = set()
for label, mask in policies:
    if mask in seen:
        continue
    seen.add(mask)
    tr.set_precision("fp8", fp8_gemms=mask)
    render()
    u8, t = timed()
    px = float(((u8 - ref) / 255).pow(2).mean().sqrt())
    lr = float((render.latents - ref_lat).pow(2).mean().sqrt() / ref_lat.pow(2).mean().sqrt())
    rows.append({"policy": label, "mask": mask, "classes": [n for n in names if mask & FluxTransformer2DModel.FP8_GEMMS[n]], "pixel_rmse": px, "latent_rel_rmse": lr, "s_per_image": t})
    print(f"{label:60s} mask {mask:2d}  pixel RMSE vs bf16 {px:.5f}  latent rel-RMSE {lr:.4f}   {t:.3f} s/image  ({1 / t:.3f} images/s)", flush=True)
print(f"bf16: {t_bf16:.3f} s/image ({1 / t_bf16:.3f} images/s)")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "fp8_policy_sweep.json"), "w"), indent=1)

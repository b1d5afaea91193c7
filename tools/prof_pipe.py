"""Where does a 3-images-in-flight pipeline call spend its time?  (host-side stage timing with device syncs)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
from thinkdiff import _hip
from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt
from thinkdiff.models.flux_transformer import FluxTransformer2DModel

pipe = FluxPipelineRewritePrompt.from_random(seed=1234, max_img_tokens=4096, max_txt_tokens=256, max_steps=32)
G = 3
pipe.images_in_flight = G
g = torch.Generator().manual_seed(0)
pe = (0.1 * torch.randn(G, 193, 4096, generator=g)).bfloat16().cuda()
pool = torch.randn(G, 768, generator=g).bfloat16().cuda()
raw = torch.randn(G, 16, 128, 128, generator=g).bfloat16().cuda()
packed = torch.stack([_hip.flux_pack_latents(raw[i]) for i in range(G)])

def call(ot):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = pipe(prompt_embeds=pe, pooled_prompt_embeds=pool, height=1024, width=1024, num_inference_steps=28, guidance_scale=3.5,
               latents=packed.clone(), output_type=ot).images
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3

call("pil")
for ot in ("latent", "pt", "pil", "latent", "pil"):
    print(f"output_type={ot:7s}: {call(ot):8.1f} ms per call of {G} images", flush=True)
# denoise only
ctxs = pipe._contexts(G)
sig = pipe.scheduler.sigmas(28, 4096)
lat = [packed[i].clone() for i in range(G)]
torch.cuda.synchronize(); t0 = time.perf_counter()
FluxTransformer2DModel.denoise_multi(ctxs, lat, sig, pipe._streams[:G])
torch.cuda.synchronize(); print(f"denoise_multi alone: {(time.perf_counter()-t0)*1e3:8.1f} ms")
t0 = time.perf_counter()
for i in range(G):
    pipe.vae.decode_packed(lat[i], 128, 128, output_type="pil")
torch.cuda.synchronize(); print(f"3 x vae decode + PIL: {(time.perf_counter()-t0)*1e3:8.1f} ms")

"""Run-to-run bit identity of the full-size FLUX forward / denoise (diagnostic)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd")); sys.path.insert(0, ROOT)
from thinkdiff.models.flux_transformer import FluxTransformer2DModel, FluxTransformerConfig, effective_scalar
from oracle import flux_ref as R
import numpy as np

layers = int(os.environ.get("L", "19")); singles = int(os.environ.get("SL", "38"))
tr = FluxTransformer2DModel(FluxTransformerConfig(num_layers=layers, num_single_layers=singles), max_img_tokens=4096, max_txt_tokens=512, max_steps=8).init_random(7)
g = torch.Generator().manual_seed(1)
T = 193
lat = torch.randn(4096, 64, generator=g).bfloat16().cuda()
pe = (0.1 * torch.randn(T, 4096, generator=g)).bfloat16().cuda()
pool = torch.randn(768, generator=g).bfloat16().cuda()
n = 2
sig = R.make_sigmas(n, 4096)
tr.set_condition(pe, pool, R.latent_image_ids(64, 64))
tr.set_timesteps([effective_scalar(float(s) * 1000.0, torch.bfloat16) for s in sig[:-1]], 3500.0)
outs = [tr.forward_step(lat, 0).clone() for _ in range(4)]
torch.cuda.synchronize()
for k in range(1, 4):
    d = (outs[k] != outs[0])
    print(f"forward run {k} vs 0: {int(d.sum())} differing elements of {d.numel()}; rows touched {int(d.any(1).sum())}")
outs1 = [tr.forward_step(lat, 1).clone() for _ in range(2)]
print("step-1 forward repeat:", int((outs1[0] != outs1[1]).sum()))
res = []
for rep in range(3):
    x = lat.clone()
    tr.denoise(x, sig)
    res.append(x.clone())
torch.cuda.synchronize()
print("denoise repeat:", [int((res[k] != res[0]).sum()) for k in (1, 2)])
x = lat.clone()
for i in range(n):
    v = tr.forward_step(x, i)
    x = (x.float() + float(sig[i + 1] - sig[i]) * v.float()).bfloat16()
print("stepwise(torch euler) vs denoise:", int((x != res[0]).sum()))
from thinkdiff import _hip
x = lat.clone()
for i in range(n):
    v = tr.forward_step(x, i)
    _hip.euler_step(x, v, float(sig[i + 1] - sig[i]))
print("stepwise(hip euler) vs denoise:", int((x != res[0]).sum()))
x = lat.clone(); v = tr.forward_step(x, 0); a = (x.float() + float(sig[1] - sig[0]) * v.float()).bfloat16(); b = _hip.euler_step(x.clone(), v, float(sig[1] - sig[0]))
print("euler torch vs hip after one step:", int((a != b).sum()))

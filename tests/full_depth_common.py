"""Test infrastructure shared by tests/golden/make_full_depth_golden.py (build container, CPU) and
tests/test_flux_full_depth_gpu.py (GPU box): the FLUX.1-dev-shaped synthetic checkpoint and the pipeline inputs of the
28-step fixtures, reproducible bit for bit on any device.

Weights come from a counter-based integer generator evaluated with torch integer ops (plumbing): element i of the
checkpoint (one running index over all tensors, in `oracle.flux_ref.param_shapes` order) is
    z = splitmix64(i + seed * 0x9E3779B97F4A7C15);  u = sum of the four 16-bit fields of z - 2 * 65535   (Irwin-Hall(4), exact)
    w = bf16(fp32(u) * fp32(std / (65536 / sqrt(3))))
Integer arithmetic wraps identically on the host and on the GPU and the only roundings are one fp32 multiply and one bf16
cast, so the GPU test regenerates on the device exactly the weights the fixture was made with on the host -- no 24 GB file,
no host generation + upload.  |w| <= 3.46 std; excess kurtosis -0.3: close enough to N(0, std) for a synthetic checkpoint.
"""
import math

import torch

WEIGHT_SEED = 20261004
GOLDEN_STEPS = (1, 2, 4, 8, 14, 21, 28)
_M64 = (1 << 64) - 1


def _s64(v):
    v &= _M64
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(z, k):
    return (z >> k) & ((1 << (64 - k)) - 1)


def hash_normal(n, start, seed, std, device, mean=0.0, chunk=None):
    """bf16 tensor [n]: elements start .. start+n-1 of the stream `seed`."""
    chunk = chunk or (1 << 24 if str(device).startswith("cuda") else 1 << 20)      # host: stay inside the caches
    out = torch.empty(n, dtype=torch.bfloat16, device=device)
    scale = torch.tensor(std / (65536.0 / math.sqrt(3.0)), dtype=torch.float32, device=device)
    off = _s64(seed * 0x9E3779B97F4A7C15)
    c1, c2 = _s64(0xBF58476D1CE4E5B9), _s64(0x94D049BB133111EB)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        z = torch.arange(start + a, start + b, dtype=torch.int64, device=device) + off
        z = (z ^ _lsr(z, 30)) * c1
        z = (z ^ _lsr(z, 27)) * c2
        z = z ^ _lsr(z, 31)
        u = (z & 0xFFFF) + ((z >> 16) & 0xFFFF) + ((z >> 32) & 0xFFFF) + ((z >> 48) & 0xFFFF) - 2 * 65535
        w = u.to(torch.float32) * scale
        if mean:
            w = w + mean
        out[a:b] = w.to(torch.bfloat16)
    return out


def draw_flux_weights(param_shapes, seed=WEIGHT_SEED, device="cpu"):
    """Yields (name, bf16 tensor): Linear weights / biases ~ 0.02 * IH4, QK-RMSNorm weights ~ 1 + 0.1 * IH4."""
    start = 0
    for name, shape in param_shapes.items():
        n = math.prod(shape)
        if ".norm_" in name and name.endswith(".weight") and len(shape) == 1:
            t = hash_normal(n, start, seed, 0.1, device, mean=1.0)
        else:
            t = hash_normal(n, start, seed, 0.02, device)
        start += n
        yield name, t.view(*shape)


def pipeline_inputs(T, seed, device="cpu", side=128):
    """Inputs of one job from the same generator (streams seed, seed+1, seed+2): raw latents [1,16,128,128] ~ IH4(0,1) (where the
    drivers draw randn with their seed-42 generator), T prompt-embedding rows ~ 0.1 * IH4, the pooled CLIP vector ~ IH4."""
    raw = hash_normal(16 * side * side, 0, seed, 1.0, device).view(1, 16, side, side)
    pe = hash_normal(T * 4096, 0, seed + 1, 0.1, device).view(1, T, 4096)
    pool = hash_normal(768, 0, seed + 2, 1.0, device).view(1, 768)
    return raw, pe, pool


def draw_vae_weights(param_shapes, seed=WEIGHT_SEED + 7, device="cpu"):
    """The VAE decoder checkpoint with oracle.vae_ref.init_weights' scaling rules, from the integer generator."""
    sd, start = {}, 0
    for k, shp in param_shapes.items():
        n = math.prod(shp)
        if "norm" in k and k.endswith("weight"):
            t = hash_normal(n, start, seed, 0.05, device, mean=1.0)
        elif len(shp) == 4:
            t = hash_normal(n, start, seed, 1.0 / (shp[1] * shp[2] * shp[3]) ** 0.5, device)
        elif len(shp) == 2:
            t = hash_normal(n, start, seed, 1.0 / shp[1] ** 0.5, device)
        else:
            t = hash_normal(n, start, seed, 0.02, device)
        start += n
        sd[k] = t.view(*shp)
    return sd


# side = latent height = width (image = 8 x side): 128 = the 1024^2 of configs 2 / 5; 64 = the 512^2 the LVLM multi-image drivers render
# (reference scripts/test/test_mllama_t5_decoder_flux_multi_image.py:258-259), T = 128 aligner tokens there
GOLDEN_JOBS = {"cfg2_T193": dict(T=193, seed=42, side=128), "cfg5_T258": dict(T=258, seed=43, side=128), "lvlm512_T128": dict(T=128, seed=44, side=64)}

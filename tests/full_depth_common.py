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


_FAST_HOST = None      # tests/golden/hashgen.c through ctypes: set by use_fast_host_generator() in the build container's fixture generator only


def use_fast_host_generator(build_dir="/tmp"):
    """Compiles tests/golden/hashgen.c (gcc + OpenMP), proves it bit-identical to the torch form below on samples of every kind of stream the
    fixtures draw, and routes host-side hash_normal calls through it.  The torch form stays the definition (the GPU tests use it on the device)."""
    global _FAST_HOST
    import ctypes
    import os
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hashgen.c")
    so = os.path.join(build_dir, "td_hashgen.so")
    subprocess.run(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so], check=True)
    lib = ctypes.CDLL(so)
    lib.hash_normal_bf16.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float, ctypes.c_float, ctypes.c_int]
    lib.hash_normal_bf16.restype = None
    for (n, start, seed, std, mean) in ((100003, 0, WEIGHT_SEED, 0.02, 0.0), (70001, 11_900_000_000, STRESS_SEED, 0.028, 0.0), (4099, 123456789, 7, 0.1, 1.0),
                                        (5003, 5, 45, 1.0, 0.0), (3001, 9, QWEN_SEED, 0.05, 1.0), (2003, 77, 3, 0.04, 2.3)):
        want = hash_normal(n, start, seed, std, "cpu", mean=mean)
        _FAST_HOST = lib
        got = hash_normal(n, start, seed, std, "cpu", mean=mean)
        _FAST_HOST = None
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), "tests/golden/hashgen.c disagrees with the torch generator"
    _FAST_HOST = lib


def hash_normal(n, start, seed, std, device, mean=0.0, chunk=None):
    """bf16 tensor [n]: elements start .. start+n-1 of the stream `seed`."""
    if _FAST_HOST is not None and str(device) == "cpu":
        out = torch.empty(n, dtype=torch.bfloat16)
        scale = float(torch.tensor(std / (65536.0 / math.sqrt(3.0)), dtype=torch.float32))
        _FAST_HOST.hash_normal_bf16(out.data_ptr(), n, start, seed & _M64, scale, float(torch.tensor(mean, dtype=torch.float32)), 1 if mean else 0)
        return out
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


def _mix(*keys):
    """splitmix64 of a tuple of small ints, in Python integers: the per-block choices of the stress profile (host side, any device)."""
    z = 0
    for k in keys:
        z = (z + k + 0x9E3779B97F4A7C15) & _M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        z ^= z >> 31
    return z


STRESS_SEED = WEIGHT_SEED + 101
STRESS_STD = (0.01, 0.014, 0.02, 0.028, 0.04)          # per-Linear weight std: 0.5x .. 2x of the plain profile's 0.02
STRESS_RESIDUAL_CHANNELS = (7, 515, 1802, 2950)        # "massive activation" channels of the residual stream (x32 out of both embedders)


def stress_plan(name, D=3072):
    """What the stress profile does to parameter `name` -- a pure function of the name (so the host and the device draw the same checkpoint).

    A trained DiT differs from an i.i.d. Gaussian one in the ways that matter to 8-bit arithmetic and to the attention's bounded form:
      * a few residual-stream channels carry values tens of times the rest (they survive LayerNorm and sit in every q/k/v/MLP input row),
      * the MLP intermediates (`ff.net.2`, `ff_context.net.2`, `proj_out` inputs) have outlier channels paired with small weight columns,
      * QK-RMSNorm gains well above 1 make the softmax rows peaked (scores ~ N(0, (g_q g_k)^2)),
      * weight scale varies from layer to layer.
    Returns dict(std=..., mean=..., row_pow2={row: k}, col_pow2={col: k}): rows / columns multiplied by 2^k AFTER the draw (exact in bf16)."""
    plan = dict(std=0.02, mean=0.0, rows={}, cols={})
    parts = name.split(".")
    blk = int(parts[1]) if parts[0] in ("transformer_blocks", "single_transformer_blocks") else -1
    kind = 0 if parts[0] == "transformer_blocks" else 1
    if name in ("x_embedder.weight", "x_embedder.bias", "context_embedder.weight", "context_embedder.bias"):
        plan["rows"] = {c: 5 for c in STRESS_RESIDUAL_CHANNELS}
        return plan
    if blk < 0:
        return plan
    if ".norm_" in name and name.endswith(".weight"):            # QK-RMSNorm gains: 40 % of the blocks ~1.5 (the bounded attention form still
        hot = _mix(kind, blk, 1) % 5 < 3                          # applies: 16.65 g_q g_k <= 48), 60 % ~2.3 (scores ~N(0, 5^2): running-maximum form)
        plan.update(mean=2.3 if hot else 1.5, std=0.1 if hot else 0.04)
        return plan
    if ".norm" in name:                                          # adaLN modulation Linears: plain
        return plan
    lin = name.rsplit(".", 1)[0]
    plan["std"] = STRESS_STD[_mix(kind, blk, sum(map(ord, lin))) % 5]
    # every block keeps feeding the massive residual channels: x8 on those output rows of the Linears that write the residual stream
    if lin.endswith(("attn.to_out.0", "attn.to_add_out", "ff.net.2", "ff_context.net.2", "proj_out")):
        plan["rows"] = {c: 3 for c in STRESS_RESIDUAL_CHANNELS}
    # MLP outlier channels: 8 intermediate channels x2^5 / x2^6 in two thirds of the blocks, the consuming columns / 2^k
    if _mix(kind, blk, 2) % 3 != 0:
        M = 4 * D
        ch = {int(_mix(kind, blk, 3, j) % M): 5 + int(_mix(kind, blk, 4, j) % 2) for j in range(8)}
        if lin.endswith(("ff.net.0.proj", "ff_context.net.0.proj", "proj_mlp")):
            plan["rows"] = ch
        elif lin.endswith(("ff.net.2", "ff_context.net.2")) and name.endswith(".weight"):
            plan["cols"] = {c: -k for c, k in ch.items()}
        elif lin.endswith("proj_out") and name.endswith(".weight"):
            plan["cols"] = {D + c: -k for c, k in ch.items()}
    return plan


def draw_flux_weights(param_shapes, seed=None, device="cpu", profile="plain"):
    """Yields (name, bf16 tensor).  profile "plain": Linear weights / biases ~ 0.02 * IH4, QK-RMSNorm weights ~ 1 + 0.1 * IH4 (seed WEIGHT_SEED);
    "stress": the heavy-tailed checkpoint of `stress_plan` (seed STRESS_SEED)."""
    assert profile in ("plain", "stress")
    seed = seed if seed is not None else (WEIGHT_SEED if profile == "plain" else STRESS_SEED)
    start = 0
    D = param_shapes["x_embedder.weight"][0]
    for name, shape in param_shapes.items():
        n = math.prod(shape)
        if profile == "stress":
            pl = stress_plan(name, D)
            t = hash_normal(n, start, seed, pl["std"], device, mean=pl["mean"]).view(*shape)
            for r, k in pl["rows"].items():
                if r < shape[0]:
                    t[r] *= 2.0 ** k                           # a power of two: exact in bf16 on any device
            for c, k in pl["cols"].items():
                if len(shape) == 2 and c < shape[1]:
                    t[:, c] *= 2.0 ** k
        elif ".norm_" in name and name.endswith(".weight") and len(shape) == 1:
            t = hash_normal(n, start, seed, 0.1, device, mean=1.0)
        else:
            t = hash_normal(n, start, seed, 0.02, device)
        start += n
        yield name, t.view(*shape)


def pipeline_inputs(T, seed, device="cpu", side=128, profile="plain"):
    """Inputs of one job from the same generator (streams seed, seed+1, seed+2): raw latents [1,16,128,128] ~ IH4(0,1) (where the
    drivers draw randn with their seed-42 generator), T prompt-embedding rows ~ 0.1 * IH4, the pooled CLIP vector ~ IH4.
    profile "stress": heavy-tailed prompt embeddings -- 0.026 z^3 (excess kurtosis ~40, same RMS ~0.1; two fp32 multiplies, identical on
    any device) with three channels x16 (the outlier channels aligner / T5 outputs carry)."""
    raw = hash_normal(16 * side * side, 0, seed, 1.0, device).view(1, 16, side, side)
    if profile == "stress":
        z = hash_normal(T * 4096, 0, seed + 1, 1.0, device).float()
        pe = ((z * z) * (z * 0.026)).to(torch.bfloat16).view(1, T, 4096)
        for c in (11, 1733, 3999):
            pe[:, :, c] *= 16.0
    else:
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


# ---- BASELINE config 3 (ThinkDiff-LVLM): Qwen2-VL-7B-shaped decoder + aligner in front of FLUX ------------------------------------------------
QWEN_SEED = WEIGHT_SEED + 300
ALIGNER_SEED = WEIGHT_SEED + 301
LVLM_REQUEST_SEED = WEIGHT_SEED + 302
IMAGE_PAD, VISION_START, VISION_END = 151655, 151652, 151653      # Qwen2-VL special token ids (config.json: image_token_id, vision_start / end)
LVLM_GRID = (1, 16, 16)                                             # one 224 x 224 image: 16 x 16 patches -> 8 x 8 = 64 merged vision tokens
LVLM_N_OUT = 128                                                    # vllm_config.max_tokens = min_tokens = 128 (configs/test_thinkdiff_lvlm_ccsbu_image_text.yaml:29-30)


def draw_qwen_weights(param_shapes, seed=QWEN_SEED, device="cpu"):
    """Yields (name, bf16 tensor) in `oracle.qwen2vl_ref.param_shapes` order: RMSNorm gains ~ 1 + 0.05 IH4, token embeddings ~ 0.5 IH4
    (oracle.qwen2vl_ref.init_weights' scaling rules), everything else ~ 0.02 IH4."""
    start = 0
    for name, shape in param_shapes.items():
        n = math.prod(shape)
        if name.endswith("norm.weight") or "layernorm" in name:
            t = hash_normal(n, start, seed, 0.05, device, mean=1.0)
        elif "embed_tokens" in name:
            t = hash_normal(n, start, seed, 0.5, device)
        else:
            t = hash_normal(n, start, seed, 0.02, device)
        start += n
        yield name, t.view(*shape)


def draw_aligner_weights(param_shapes, seed=ALIGNER_SEED, device="cpu"):
    """`oracle.aligner_ref.param_shapes` order: Linear weights / biases ~ 0.02 IH4, the T5LayerNorm gain ~ 1 + 0.1 IH4."""
    sd, start = {}, 0
    for name, shape in param_shapes.items():
        n = math.prod(shape)
        gain = len(shape) == 1 and name.endswith("weight")
        sd[name] = hash_normal(n, start, seed, 0.1 if gain else 0.02, device, mean=1.0 if gain else 0.0).view(*shape)
        start += n
    return sd


def lvlm_request(vocab=152064, hidden=3584, seed=LVLM_REQUEST_SEED, device="cpu"):
    """One image + instruction request in the shape the reference's chat template gives it (thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:1040-1075):
    14 system-prompt tokens, <|vision_start|>, 64 <|image_pad|> rows, <|vision_end|>, 20 instruction tokens, 3 assistant-header tokens; then the
    128 token ids the decoder is teacher-forced to emit.  Token ids are hash draws below the special-token range; the 64 merged vision
    tokens (what the ViT's merger would splice over the placeholders) are ~0.5 IH4 rows from the same generator.
    Returns dict(prompt_ids, forced_ids, vision_rows [64, hidden] bf16, position_ids int32 [3, n_prompt], grid)."""
    def ids(n, stream):
        z = hash_normal(n, 0, seed + stream, 1.0, "cpu").view(torch.int16).to(torch.int64)      # any deterministic integers
        return [int(v) % 151000 for v in (z * 40503 + 12345)]
    t, h, w = LVLM_GRID
    n_img = t * (h // 2) * (w // 2)
    prompt = ids(14, 1) + [VISION_START] + [IMAGE_PAD] * n_img + [VISION_END] + ids(20, 2) + ids(3, 3)
    forced = ids(LVLM_N_OUT, 4)
    vision = hash_normal(n_img * hidden, 0, seed + 5, 0.5, device).view(n_img, hidden)
    # M-RoPE streams ([ext] transformers Qwen2VLForConditionalGeneration.get_rope_index): text counts up on all three, the image block keeps t and
    # walks its merged grid, the text behind it resumes at max + 1
    pos = torch.zeros(3, len(prompt), dtype=torch.int32)
    a = 15
    pos[:, :a] = torch.arange(a, dtype=torch.int32)
    gh, gw = h // 2, w // 2
    pos[0, a:a + n_img] = a
    pos[1, a:a + n_img] = a + torch.arange(gh, dtype=torch.int32).repeat_interleave(gw)
    pos[2, a:a + n_img] = a + torch.arange(gw, dtype=torch.int32).repeat(gh)
    rest = len(prompt) - a - n_img
    pos[:, a + n_img:] = a + max(gh, gw) + torch.arange(rest, dtype=torch.int32)
    return dict(prompt_ids=prompt, forced_ids=forced, vision_rows=vision, position_ids=pos, grid=[list(LVLM_GRID)])


# side = latent height = width (image = 8 x side): 128 = the 1024^2 of configs 2 / 5; 64 = the 512^2 the LVLM multi-image drivers render
# (reference scripts/test/test_mllama_t5_decoder_flux_multi_image.py:258-259), T = 128 aligner tokens there
GOLDEN_JOBS = {"cfg2_T193": dict(T=193, seed=42, side=128), "cfg5_T258": dict(T=258, seed=43, side=128), "lvlm512_T128": dict(T=128, seed=44, side=64),
               # config 5's shape on the heavy-tailed checkpoint and prompt (stress_plan): the fixture every 8-bit policy is graded on as well
               "stress_T258": dict(T=258, seed=45, side=128, profile="stress"),
               # BASELINE config 3 at full size: 28-layer Qwen2-VL-7B-shaped decoder (teacher-forced, 128 output tokens) -> aligner -> FLUX 1024^2, T = 128
               "cfg3_lvlm7b": dict(T=128, seed=46, side=128, lvlm=True)}

"""ORACLE (test infrastructure, not product code): CPU restatement of the Qwen2-VL text decoder up to
`model.norm` -- the hidden state ThinkDiff-LVLM captures and feeds to its aligner.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference path: the vLLM fork the reference calls (`self.mllama.generate(..., return_hidden_states=True)`,
thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:1083-1089, `embedding_layer_name: "model.norm"`) is not in
/root/reference nor installed.  The per-layer math is restated from the installed `transformers`
`modeling_qwen2_vl.py` (:96-110 RMSNorm, :117-168 rotary, :180-222 M-RoPE, :453-466 MLP, :469-556 attention,
:559-625 layer), which SURVEY.md 8a row A7 identifies as the same computation.

Pinned: tests/test_oracle_cpu.py runs this file against `transformers` `Qwen2VLTextModel` itself (tiny config,
fp32: <= 1e-5; bf16: bit-level agreement of the rounding points) and against tests/golden/qwen2vl_tiny.pt.
"""
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class Qwen2Config:
    hidden: int = 3584
    num_layers: int = 28
    num_heads: int = 28
    num_kv_heads: int = 4
    head_dim: int = 128
    intermediate: int = 18944
    vocab: int = 152064
    tie_embeddings: bool = False
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    rms_eps: float = 1e-6
    rope_theta: float = 1e6


def tiny_config(**kw):
    base = dict(hidden=512, num_layers=2, num_heads=4, num_kv_heads=2, intermediate=1024, vocab=320)
    base.update(kw)
    return Qwen2Config(**base)


def param_shapes(cfg: Qwen2Config) -> Dict[str, tuple]:
    D, hd = cfg.hidden, cfg.head_dim
    s = {"model.embed_tokens.weight": (cfg.vocab, D), "model.norm.weight": (D,)}
    if not cfg.tie_embeddings:
        s["lm_head.weight"] = (cfg.vocab, D)
    for i in range(cfg.num_layers):
        p = f"model.layers.{i}."
        s[p + "self_attn.q_proj.weight"] = (cfg.num_heads * hd, D)
        s[p + "self_attn.q_proj.bias"] = (cfg.num_heads * hd,)
        for n in ("k_proj", "v_proj"):
            s[p + f"self_attn.{n}.weight"] = (cfg.num_kv_heads * hd, D)
            s[p + f"self_attn.{n}.bias"] = (cfg.num_kv_heads * hd,)
        s[p + "self_attn.o_proj.weight"] = (D, cfg.num_heads * hd)
        s[p + "mlp.gate_proj.weight"] = (cfg.intermediate, D)
        s[p + "mlp.up_proj.weight"] = (cfg.intermediate, D)
        s[p + "mlp.down_proj.weight"] = (D, cfg.intermediate)
        s[p + "input_layernorm.weight"] = (D,)
        s[p + "post_attention_layernorm.weight"] = (D,)
    return s


def init_weights(cfg: Qwen2Config, seed: int = 0, std: float = 0.02, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in param_shapes(cfg).items():
        if k.endswith("norm.weight") or "layernorm" in k:
            sd[k] = (1.0 + 0.05 * torch.randn(shp, generator=g)).to(dtype)
        elif "embed_tokens" in k:
            sd[k] = torch.randn(shp, generator=g).to(dtype) * 0.5
        else:
            sd[k] = (std * torch.randn(shp, generator=g)).to(dtype)
    return sd


def rms_norm(x, w, eps):
    """modeling_qwen2_vl.py:96-110"""
    dt = x.dtype
    x = x.to(torch.float32)
    x = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps)
    return w * x.to(dt)


def mrope_cos_sin(position_ids: torch.Tensor, cfg: Qwen2Config, dtype):
    """position_ids int [3, n] -> cos, sin [n, 128] in `dtype`, M-RoPE sections merged
    (modeling_qwen2_vl.py:117-168 rotary forward + :213-218 section merge)."""
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float) / cfg.head_dim))
    freqs = position_ids[:, :, None].float() * inv_freq[None, None, :]          # [3, n, 64]
    emb = torch.cat((freqs, freqs), dim=-1)                                     # [3, n, 128]
    cos, sin = emb.cos().to(dtype), emb.sin().to(dtype)
    sec = list(cfg.mrope_section) * 2
    cos = torch.cat([m[i % 3] for i, m in enumerate(cos.split(sec, dim=-1))], dim=-1)
    sin = torch.cat([m[i % 3] for i, m in enumerate(sin.split(sec, dim=-1))], dim=-1)
    return cos, sin


def rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def decoder_layer(sd, cfg: Qwen2Config, i: int, h, cos, sin, past_k=None, past_v=None):
    """h [n, D]; returns (h, k, v) with k/v [Hkv, n_total, 128] (modeling_qwen2_vl.py:469-625, eager attention)."""
    p = f"model.layers.{i}."
    n = h.shape[0]
    Hq, Hkv, hd = cfg.num_heads, cfg.num_kv_heads, cfg.head_dim
    x = rms_norm(h, sd[p + "input_layernorm.weight"], cfg.rms_eps)
    q = F.linear(x, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"]).view(n, Hq, hd).transpose(0, 1)
    k = F.linear(x, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"]).view(n, Hkv, hd).transpose(0, 1)
    v = F.linear(x, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"]).view(n, Hkv, hd).transpose(0, 1)
    q = q * cos[None] + rotate_half(q) * sin[None]
    k = k * cos[None] + rotate_half(k) * sin[None]
    if past_k is not None:
        k, v = torch.cat([past_k, k], dim=1), torch.cat([past_v, v], dim=1)
    kk = k.repeat_interleave(Hq // Hkv, dim=0)
    vv = v.repeat_interleave(Hq // Hkv, dim=0)
    s = torch.matmul(q, kk.transpose(1, 2)) * (hd ** -0.5)
    nt = k.shape[1]
    mask = torch.arange(nt)[None, :] > (torch.arange(n)[:, None] + (nt - n))
    s = s.masked_fill(mask[None], torch.finfo(s.dtype).min)
    a = torch.softmax(s, dim=-1, dtype=torch.float32).to(q.dtype)
    o = torch.matmul(a, vv).transpose(0, 1).reshape(n, Hq * hd)
    h = h + F.linear(o, sd[p + "self_attn.o_proj.weight"])
    x = rms_norm(h, sd[p + "post_attention_layernorm.weight"], cfg.rms_eps)
    m = F.linear(F.silu(F.linear(x, sd[p + "mlp.gate_proj.weight"])) * F.linear(x, sd[p + "mlp.up_proj.weight"]),
                 sd[p + "mlp.down_proj.weight"])
    return h + m, k, v


def text_model_hidden(sd, cfg: Qwen2Config, position_ids, token_ids=None, inputs_embeds=None, past=None):
    """Returns (model.norm(h) [n, D], kv list).  `past`: list of (k, v) per layer for KV-cached continuation."""
    h = F.embedding(token_ids, sd["model.embed_tokens.weight"]) if inputs_embeds is None else inputs_embeds
    cos, sin = mrope_cos_sin(position_ids, cfg, h.dtype)
    kv = []
    for i in range(cfg.num_layers):
        pk, pv = past[i] if past is not None else (None, None)
        h, k, v = decoder_layer(sd, cfg, i, h, cos, sin, pk, pv)
        kv.append((k, v))
    return rms_norm(h, sd["model.norm.weight"], cfg.rms_eps), kv


def lm_logits(sd, cfg: Qwen2Config, hidden_last):
    w = sd["model.embed_tokens.weight"] if cfg.tie_embeddings else sd["lm_head.weight"]
    return F.linear(hidden_last, w)


def text_position_ids(n: int, start: int = 0) -> torch.Tensor:
    """Text-only M-RoPE ids: the three streams are identical (modeling_qwen2_vl.py get_rope_index, text branch)."""
    return (torch.arange(n) + start)[None, :].expand(3, n).contiguous().to(torch.int32)

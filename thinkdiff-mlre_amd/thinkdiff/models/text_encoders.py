"""T5 encoder and CLIP text encoder on the HIP ops -- the two modules `encode_prompt` calls
(thinkdiff/models/flux_prompt.py:88-104 -> [ext] FluxPipeline `_get_clip_prompt_embeds` -> transformers
CLIPTextModel.pooler_output, `_get_t5_prompt_embeds` -> transformers T5EncoderModel(...)[0]).

Orchestration is host Python (one-off per prompt, ~1.2 TFLOP); every tensor op is a libthinkdiff_hip.so call:
GEMMs with fused bias / activation / residual, LayerNorm / T5-RMSNorm rows, the D = 128 fused attention with every
head zero-padded from 64 to 128 columns (the padding lives in the fused projection weights, built at load time), the
T5 relative-position bias as an additive fp32 score bias.  State-dict names are the Hugging Face ones.
"""
import glob
import json
import math
import os
from typing import Dict, Optional

import torch

from .. import _hip

HP = 128  # padded head width of the attention kernel


def _pad_heads_rows(w: torch.Tensor, H: int, hd: int) -> torch.Tensor:
    """[H*hd, K] -> [H*128, K]: head h's rows land at [h*128, h*128+hd), the rest are zero."""
    out = torch.zeros(H * HP, w.shape[1], dtype=w.dtype, device=w.device)
    out.view(H, HP, -1)[:, :hd] = w.view(H, hd, -1)
    return out


def _pad_heads_vec(b: Optional[torch.Tensor], H: int, hd: int, device) -> torch.Tensor:
    out = torch.zeros(H * HP, dtype=torch.bfloat16, device=device)
    if b is not None:
        out.view(H, HP)[:, :hd] = b.view(H, hd)
    return out


def _pad_heads_cols(w: torch.Tensor, H: int, hd: int) -> torch.Tensor:
    """o_proj [D, H*hd] -> [D, H*128]."""
    out = torch.zeros(w.shape[0], H * HP, dtype=w.dtype, device=w.device)
    out.view(w.shape[0], H, HP)[:, :, :hd] = w.view(w.shape[0], H, hd)
    return out


def _read_dir(path: str, subfolder: str):
    """(config dict, state dict) of a local Hugging Face model directory (config.json + *.safetensors)."""
    from safetensors import safe_open
    root = os.path.join(path, subfolder) if os.path.isdir(os.path.join(path, subfolder)) else path
    with open(os.path.join(root, "config.json")) as fh:
        cfg = json.load(fh)
    sd = {}
    for fn in sorted(glob.glob(os.path.join(root, "*.safetensors"))):
        with safe_open(fn, framework="pt") as fh:
            sd.update({k: fh.get_tensor(k) for k in fh.keys()})
    if not sd:
        raise FileNotFoundError(f"no *.safetensors under {root}")
    return cfg, sd


def _random_sd(shapes: Dict[str, tuple], seed: int, device, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Synthetic checkpoint drawn on the device: N(0, std) matrices / embeddings, unit norm weights, small biases."""
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for name, shape in shapes.items():
        if name.endswith(("norm.weight", "norm1.weight", "norm2.weight", "ln_q.weight", "layernorm.weight")):
            sd[name] = torch.ones(shape, dtype=torch.bfloat16, device=device)
        else:
            sd[name] = (torch.randn(shape, generator=g, device=device, dtype=torch.float32) * std).to(torch.bfloat16)
    return sd


class HashTokenizer:
    """Asset-free stand-in for the CLIP / T5 tokenizers (synthetic runs only): one deterministic id per
    whitespace-separated word, EOS (= vocab - 1, the largest id) then padding to `max_length`."""

    def __init__(self, vocab_size: int, pad_id: int = 0):
        self.vocab_size, self.pad_id = vocab_size, pad_id

    def __call__(self, prompt, padding="max_length", max_length=77, truncation=True, return_tensors="pt", **_kw):
        import zlib
        rows = []
        for text in ([prompt] if isinstance(prompt, str) else prompt):
            ids = [1 + zlib.crc32(w.encode()) % (self.vocab_size - 2) for w in text.split()][: max_length - 1]
            ids.append(self.vocab_size - 1)
            rows.append(ids + [self.pad_id] * (max_length - len(ids)))
        return type("Encoding", (), {"input_ids": torch.tensor(rows, dtype=torch.long)})()


class _EncoderOutput(tuple):
    """What the pipeline reads from a transformers ModelOutput: `[0]` and `.pooler_output`."""
    last_hidden_state = property(lambda self: self[0])
    pooler_output = property(lambda self: self[1])


class _Base:
    dtype = torch.bfloat16

    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _hip.ThinkDiffHipError("text encoders run on the MI355X HIP ops only (device='cuda')")
        _hip.lib()

    def _dev(self, t):
        return t.to(self.device, torch.bfloat16).contiguous()


class HipT5Encoder(_Base):
    """transformers T5EncoderModel (flan-t5-xxl: d_model 4096, 24 layers, 64 heads x 64, d_ff 10240, gated-gelu)."""

    def __init__(self, sd: Dict[str, torch.Tensor], num_heads: int, d_kv: int, num_buckets: int = 32, max_distance: int = 128,
                 eps: float = 1e-6, device="cuda"):
        super().__init__(device)
        self.H, self.hd, self.eps = num_heads, d_kv, eps
        self.num_buckets, self.max_distance = num_buckets, max_distance
        g = lambda k: self._dev(sd[k])
        self.embed = g("shared.weight") if "shared.weight" in sd else g("encoder.embed_tokens.weight")
        self.vocab, self.D = self.embed.shape
        self.rel_bias = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"].float().to(self.device)  # [buckets, H]
        self.layers = []
        i = 0
        while f"encoder.block.{i}.layer.0.SelfAttention.q.weight" in sd:
            p = f"encoder.block.{i}.layer."
            qkv = torch.cat([_pad_heads_rows(g(p + f"0.SelfAttention.{n}.weight"), self.H, self.hd) for n in "qkv"])
            self.layers.append(dict(
                ln1=g(p + "0.layer_norm.weight"), qkv=qkv.contiguous(),
                o=_pad_heads_cols(g(p + "0.SelfAttention.o.weight"), self.H, self.hd).contiguous(),
                ln2=g(p + "1.layer_norm.weight"),
                wi=torch.cat([g(p + "1.DenseReluDense.wi_0.weight"), g(p + "1.DenseReluDense.wi_1.weight")]).contiguous(),
                wo=g(p + "1.DenseReluDense.wo.weight")))
            i += 1
        self.final_ln = g("encoder.final_layer_norm.weight")
        self._bias_cache = {}

    def _position_bias(self, S: int) -> torch.Tensor:
        """[H, S, S] fp32: T5Attention.compute_bias, bidirectional buckets (modeling_t5.py `_relative_position_bucket`)."""
        if S not in self._bias_cache:
            ctx = torch.arange(S)[:, None]
            mem = torch.arange(S)[None, :]
            rel = mem - ctx
            nb = self.num_buckets // 2
            buckets = (rel > 0).long() * nb
            rel = rel.abs()
            max_exact = nb // 2
            is_small = rel < max_exact
            large = max_exact + (torch.log(rel.float().clamp_min(1) / max_exact) / math.log(self.max_distance / max_exact)
                                 * (nb - max_exact)).long()
            large = torch.min(large, torch.full_like(large, nb - 1))
            buckets = buckets + torch.where(is_small, rel, large)
            self._bias_cache[S] = self.rel_bias[buckets.to(self.device)].permute(2, 0, 1).contiguous()
        return self._bias_cache[S]

    @classmethod
    def from_random(cls, d_model=4096, num_layers=24, num_heads=64, d_kv=64, d_ff=10240, vocab_size=32128, seed=0, device="cuda"):
        """Synthetic flan-t5-xxl-shaped encoder (defaults) drawn on the device."""
        shapes = {"shared.weight": (vocab_size, d_model), "encoder.final_layer_norm.weight": (d_model,),
                  "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight": (32, num_heads)}
        for i in range(num_layers):
            p = f"encoder.block.{i}.layer."
            shapes.update({p + "0.SelfAttention.q.weight": (num_heads * d_kv, d_model), p + "0.SelfAttention.k.weight": (num_heads * d_kv, d_model),
                           p + "0.SelfAttention.v.weight": (num_heads * d_kv, d_model), p + "0.SelfAttention.o.weight": (d_model, num_heads * d_kv),
                           p + "0.layer_norm.weight": (d_model,), p + "1.layer_norm.weight": (d_model,),
                           p + "1.DenseReluDense.wi_0.weight": (d_ff, d_model), p + "1.DenseReluDense.wi_1.weight": (d_ff, d_model),
                           p + "1.DenseReluDense.wo.weight": (d_model, d_ff)})
        return cls(_random_sd(shapes, seed, torch.device(device)), num_heads, d_kv, device=device)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: str = "text_encoder_2", device="cuda"):
        cfg, sd = _read_dir(path, subfolder)
        if cfg.get("feed_forward_proj", "gated-gelu") != "gated-gelu":
            raise _hip.ThinkDiffHipError("only the gated-gelu T5 v1.1 / flan-t5 feed-forward is implemented")
        return cls(sd, cfg["num_heads"], cfg["d_kv"], cfg.get("relative_attention_num_buckets", 32),
                   cfg.get("relative_attention_max_distance", 128), cfg.get("layer_norm_epsilon", 1e-6), device)

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, output_hidden_states: bool = False, **_kw):
        """input_ids [B, S] -> (last_hidden_state [B, S, D],)  (no attention mask: FluxPipeline passes none)."""
        return _EncoderOutput((self.encode(input_ids), None))

    @torch.no_grad()
    def encode(self, input_ids: torch.Tensor) -> torch.Tensor:
        outs = []
        for ids in input_ids:
            S = ids.shape[0]
            assert S % 4 == 0, "sequence length must be a multiple of 4 (FLUX pads T5 prompts to max_sequence_length)"
            h = self.embed[ids.to(self.device).long()].contiguous()
            bias = self._position_bias(S)
            for L in self.layers:
                x = _hip.layernorm(h, L["ln1"], None, self.eps, rms=True)
                qkv = _hip.linear(x, L["qkv"])
                a = _hip.attention_padded(qkv, self.H, 1.0, causal=False, bias=bias)     # T5: no 1/sqrt(d) scaling
                h = _hip.linear(a, L["o"], res=h)
                x = _hip.layernorm(h, L["ln2"], None, self.eps, rms=True)
                gu = _hip.linear(x, L["wi"])
                h = _hip.linear(_hip.glu_mul(gu, _hip.ACT_GELU_TANH), L["wo"], res=h)
            outs.append(_hip.layernorm(h, self.final_ln, None, self.eps, rms=True))
        return torch.stack(outs)


class HipCLIPTextEncoder(_Base):
    """transformers CLIPTextModel (CLIP-L: 12 layers, d 768, 12 heads x 64, quick_gelu MLP 3072, causal)."""

    def __init__(self, sd: Dict[str, torch.Tensor], num_heads: int, eps: float = 1e-5, eos_token_id: int = 2, act: str = "quick_gelu",
                 device="cuda"):
        super().__init__(device)
        pre = "text_model." if "text_model.embeddings.token_embedding.weight" in sd else ""   # checkpoint files carry the prefix
        g = lambda k: self._dev(sd[pre + k])
        self.tok = g("embeddings.token_embedding.weight")
        self.pos = g("embeddings.position_embedding.weight")
        self.D = self.tok.shape[1]
        self.H, self.hd, self.eps, self.eos = num_heads, self.D // num_heads, eps, eos_token_id
        self.act = {"quick_gelu": _hip.ACT_QUICK_GELU, "gelu": _hip.ACT_GELU_ERF}[act]
        self.layers = []
        i = 0
        while f"{pre}encoder.layers.{i}.self_attn.q_proj.weight" in sd:
            p = f"encoder.layers.{i}."
            qkv_w = torch.cat([_pad_heads_rows(g(p + f"self_attn.{n}_proj.weight"), self.H, self.hd) for n in "qkv"])
            qkv_b = torch.cat([_pad_heads_vec(g(p + f"self_attn.{n}_proj.bias"), self.H, self.hd, self.device) for n in "qkv"])
            self.layers.append(dict(
                ln1w=g(p + "layer_norm1.weight"), ln1b=g(p + "layer_norm1.bias"), qkv_w=qkv_w.contiguous(), qkv_b=qkv_b.contiguous(),
                o_w=_pad_heads_cols(g(p + "self_attn.out_proj.weight"), self.H, self.hd).contiguous(), o_b=g(p + "self_attn.out_proj.bias"),
                ln2w=g(p + "layer_norm2.weight"), ln2b=g(p + "layer_norm2.bias"),
                fc1_w=g(p + "mlp.fc1.weight"), fc1_b=g(p + "mlp.fc1.bias"), fc2_w=g(p + "mlp.fc2.weight"), fc2_b=g(p + "mlp.fc2.bias")))
            i += 1
        self.fln_w, self.fln_b = g("final_layer_norm.weight"), g("final_layer_norm.bias")

    @classmethod
    def from_random(cls, hidden=768, num_layers=12, num_heads=12, intermediate=3072, vocab_size=49408, max_positions=77, seed=0, device="cuda"):
        """Synthetic CLIP-L-shaped text model (defaults) drawn on the device."""
        shapes = {"embeddings.token_embedding.weight": (vocab_size, hidden), "embeddings.position_embedding.weight": (max_positions, hidden),
                  "final_layer_norm.weight": (hidden,), "final_layer_norm.bias": (hidden,)}
        for i in range(num_layers):
            p = f"encoder.layers.{i}."
            for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
                shapes.update({p + f"self_attn.{n}.weight": (hidden, hidden), p + f"self_attn.{n}.bias": (hidden,)})
            shapes.update({p + "layer_norm1.weight": (hidden,), p + "layer_norm1.bias": (hidden,), p + "layer_norm2.weight": (hidden,),
                           p + "layer_norm2.bias": (hidden,), p + "mlp.fc1.weight": (intermediate, hidden), p + "mlp.fc1.bias": (intermediate,),
                           p + "mlp.fc2.weight": (hidden, intermediate), p + "mlp.fc2.bias": (hidden,)})
        return cls(_random_sd(shapes, seed, torch.device(device)), num_heads, device=device)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: str = "text_encoder", device="cuda"):
        cfg, sd = _read_dir(path, subfolder)
        cfg = cfg.get("text_config", cfg)
        return cls(sd, cfg["num_attention_heads"], cfg.get("layer_norm_eps", 1e-5), cfg.get("eos_token_id", 2),
                   cfg.get("hidden_act", "quick_gelu"), device)

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, output_hidden_states: bool = False, **_kw):
        """input_ids [B, S] -> (last_hidden_state [B,S,D], pooler_output [B,D])."""
        hs, pooled = [], []
        for ids in input_ids:
            S = ids.shape[0]
            h = _hip.add_rows(self.tok[ids.to(self.device).long()].contiguous(), self.pos[:S])
            for L in self.layers:
                x = _hip.layernorm(h, L["ln1w"], L["ln1b"], self.eps)
                qkv = _hip.linear(x, L["qkv_w"], L["qkv_b"])
                a = _hip.attention_padded(qkv, self.H, self.hd ** -0.5, causal=True)
                h = _hip.linear(a, L["o_w"], L["o_b"], res=h)
                x = _hip.layernorm(h, L["ln2w"], L["ln2b"], self.eps)
                h = _hip.linear(_hip.linear(x, L["fc1_w"], L["fc1_b"], act=self.act), L["fc2_w"], L["fc2_b"], res=h)
            h = _hip.layernorm(h, self.fln_w, self.fln_b, self.eps)
            hs.append(h)
            ids_c = ids.cpu()
            idx = int(ids_c.argmax()) if self.eos == 2 else int((ids_c == self.eos).int().argmax())   # CLIPTextTransformer pooling
            pooled.append(h[idx])
        return _EncoderOutput((torch.stack(hs), torch.stack(pooled)))

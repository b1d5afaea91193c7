"""Vision towers upstream of the aligner, on the HIP ops.

* `HipBlip2VisionModel` -- EVA-ViT-g as wrapped by transformers `Blip2VisionModel` (39 layers, D 1408, 16 heads x 88,
  MLP 6144, 14x14 patches of a 224^2 image -> 257 tokens): `self.vision_model(pixel_values)[0]` in
  reference thinkdiff/models/blip_vision_t5_decoder.py:611-618.
* `HipQwen2VisionTransformer` -- the Qwen2-VL ViT (32 blocks, D 1280, 16 heads x 80, 2-D rotary, full attention per
  image, 2x2 patch merger) that vLLM runs inside `self.mllama.generate` (mllama_vllm_t5_embed_decoder_2.py:1083-1089);
  oracle = transformers `Qwen2VisionTransformerPretrainedModel`.

Heads of width 88 / 80 are zero-padded to the attention kernel's 128 columns inside the fused projection weights
(built once at load): q.k is unchanged by zero columns and the padded value columns meet zero rows of the output
projection.  GEMM K must be a multiple of 64, so the patch operands (588 / 1176 wide) are zero-padded to 640 / 1216.
"""
from typing import Dict, Sequence

import torch

from .. import _hip
from .text_encoders import HP, _Base, _EncoderOutput, _pad_heads_cols, _pad_heads_rows, _pad_heads_vec, _random_sd, _read_dir


def _round64(k: int) -> int:
    return (k + 63) // 64 * 64


def _pad_k(w: torch.Tensor) -> torch.Tensor:
    out = torch.zeros(w.shape[0], _round64(w.shape[1]), dtype=w.dtype, device=w.device)
    out[:, :w.shape[1]] = w
    return out


class HipBlip2VisionModel(_Base):
    def __init__(self, sd: Dict[str, torch.Tensor], num_heads: int = 16, patch_size: int = 14, eps: float = 1e-6, device="cuda"):
        super().__init__(device)
        pre = "vision_model." if "vision_model.embeddings.class_embedding" in sd else ""
        g = lambda k: self._dev(sd[pre + k])
        self.cls = g("embeddings.class_embedding").view(1, -1)
        self.pos = g("embeddings.position_embedding").view(-1, self.cls.shape[1])
        self.D = self.cls.shape[1]
        self.H, self.hd, self.p, self.eps = num_heads, self.D // num_heads, patch_size, eps
        self.patch_w = _pad_k(g("embeddings.patch_embedding.weight").view(self.D, -1)).contiguous()
        self.patch_b = g("embeddings.patch_embedding.bias")
        self.layers = []
        i = 0
        while f"{pre}encoder.layers.{i}.self_attn.qkv.weight" in sd:
            p = f"encoder.layers.{i}."
            w = g(p + "self_attn.qkv.weight")                                   # rows ordered [3][H][hd]
            if (pre + p + "self_attn.qkv.bias") in sd:
                b = g(p + "self_attn.qkv.bias")
            elif (pre + p + "self_attn.q_bias") in sd:                          # on-disk BLIP-2 checkpoints: k has no bias
                b = torch.cat([g(p + "self_attn.q_bias"), torch.zeros(self.D, dtype=torch.bfloat16, device=self.device), g(p + "self_attn.v_bias")])
            else:
                b = torch.zeros(3 * self.D, dtype=torch.bfloat16, device=self.device)
            self.layers.append(dict(
                ln1w=g(p + "layer_norm1.weight"), ln1b=g(p + "layer_norm1.bias"),
                qkv_w=_pad_heads_rows(w, 3 * self.H, self.hd).contiguous(), qkv_b=_pad_heads_vec(b, 3 * self.H, self.hd, self.device),
                o_w=_pad_heads_cols(g(p + "self_attn.projection.weight"), self.H, self.hd).contiguous(), o_b=g(p + "self_attn.projection.bias"),
                ln2w=g(p + "layer_norm2.weight"), ln2b=g(p + "layer_norm2.bias"),
                fc1_w=g(p + "mlp.fc1.weight"), fc1_b=g(p + "mlp.fc1.bias"), fc2_w=g(p + "mlp.fc2.weight"), fc2_b=g(p + "mlp.fc2.bias")))
            i += 1
        self.post_w, self.post_b = g("post_layernorm.weight"), g("post_layernorm.bias")

    @classmethod
    def from_random(cls, hidden=1408, num_layers=39, num_heads=16, intermediate=6144, image_size=224, patch_size=14, seed=0, device="cuda"):
        """Synthetic EVA-ViT-g-shaped tower (defaults = Salesforce/blip2-flan-t5-xxl vision_config) drawn on the device."""
        n_pos = (image_size // patch_size) ** 2 + 1
        shapes = {"embeddings.class_embedding": (1, 1, hidden), "embeddings.position_embedding": (1, n_pos, hidden),
                  "embeddings.patch_embedding.weight": (hidden, 3, patch_size, patch_size), "embeddings.patch_embedding.bias": (hidden,),
                  "post_layernorm.weight": (hidden,), "post_layernorm.bias": (hidden,)}
        for i in range(num_layers):
            p = f"encoder.layers.{i}."
            shapes.update({p + "self_attn.qkv.weight": (3 * hidden, hidden), p + "self_attn.qkv.bias": (3 * hidden,),
                           p + "self_attn.projection.weight": (hidden, hidden), p + "self_attn.projection.bias": (hidden,),
                           p + "layer_norm1.weight": (hidden,), p + "layer_norm1.bias": (hidden,), p + "layer_norm2.weight": (hidden,),
                           p + "layer_norm2.bias": (hidden,), p + "mlp.fc1.weight": (intermediate, hidden), p + "mlp.fc1.bias": (intermediate,),
                           p + "mlp.fc2.weight": (hidden, intermediate), p + "mlp.fc2.bias": (hidden,)})
        return cls(_random_sd(shapes, seed, torch.device(device)), num_heads, patch_size, device=device)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: str = "", device="cuda"):
        cfg, sd = _read_dir(path, subfolder)
        cfg = cfg.get("vision_config", cfg)
        return cls(sd, cfg.get("num_attention_heads", 16), cfg.get("patch_size", 14), cfg.get("layer_norm_eps", 1e-6), device)

    @torch.no_grad()
    def __call__(self, pixel_values: torch.Tensor, **_kw):
        """pixel_values [B,3,224,224] (fp32 or bf16) -> ([B,257,D] post-layernormed hidden states, pooled [B,D])."""
        outs = []
        for img in pixel_values:
            img = img.to(self.device)
            img = img.contiguous() if img.dtype in (torch.float32, torch.bfloat16) else img.float().contiguous()
            patches = _hip.patchify(img, self.p, self.patch_w.shape[1])
            n = patches.shape[0]
            if n + 1 != self.pos.shape[0]:
                raise _hip.ThinkDiffHipError(f"image gives {n} patches; position table holds {self.pos.shape[0] - 1} (interpolation is not implemented)")
            h = torch.empty(n + 1, self.D, dtype=torch.bfloat16, device=self.device)
            h[0] = self.cls[0]
            _hip.linear(patches, self.patch_w, self.patch_b, out=h[1:])
            h = _hip.add_rows(h, self.pos)
            for L in self.layers:
                x = _hip.layernorm(h, L["ln1w"], L["ln1b"], self.eps)
                a = _hip.attention_padded(_hip.linear(x, L["qkv_w"], L["qkv_b"]), self.H, self.hd ** -0.5)
                h = _hip.linear(a, L["o_w"], L["o_b"], res=h)
                x = _hip.layernorm(h, L["ln2w"], L["ln2b"], self.eps)
                h = _hip.linear(_hip.linear(x, L["fc1_w"], L["fc1_b"], act=_hip.ACT_GELU_ERF), L["fc2_w"], L["fc2_b"], res=h)
            outs.append(_hip.layernorm(h, self.post_w, self.post_b, self.eps))
        hs = torch.stack(outs)
        pooled = torch.stack([_hip.layernorm(o[:1].contiguous(), self.post_w, self.post_b, self.eps)[0] for o in outs])
        return _EncoderOutput((hs, pooled))


def vision_position_ids(grid_thw: Sequence[Sequence[int]], merge: int = 2) -> torch.Tensor:
    """[S, 2] (row, column) of every patch in the processor's merge-window order: patches are emitted
    block-by-block (merge x merge neighbours adjacent), frames repeat the grid ([ext] Qwen2-VL rot_pos_emb)."""
    out = []
    for t, h, w in grid_thw:
        t, h, w = int(t), int(h), int(w)
        hp = torch.arange(h)[:, None].expand(h, w).reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        wp = torch.arange(w)[None, :].expand(h, w).reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        out.append(torch.stack([hp, wp], -1).repeat(t, 1))
    return torch.cat(out)


class HipQwen2VisionTransformer(_Base):
    def __init__(self, sd: Dict[str, torch.Tensor], num_heads: int = 16, spatial_merge_size: int = 2, hidden_act: str = "quick_gelu",
                 rope_theta: float = 10000.0, device="cuda"):
        super().__init__(device)
        pre = next((p for p in ("model.visual.", "visual.", "") if (p + "patch_embed.proj.weight") in sd), None)
        if pre is None:
            raise KeyError("patch_embed.proj.weight not found in the state dict")
        g = lambda k: self._dev(sd[pre + k])
        w = g("patch_embed.proj.weight")
        self.D = w.shape[0]
        self.K_in = w[0].numel()
        self.patch_w = _pad_k(w.view(self.D, -1)).contiguous()
        self.H, self.hd, self.merge, self.theta = num_heads, self.D // num_heads, spatial_merge_size, rope_theta
        self.act = {"quick_gelu": _hip.ACT_QUICK_GELU, "gelu": _hip.ACT_GELU_ERF, "gelu_pytorch_tanh": _hip.ACT_GELU_TANH, "silu": _hip.ACT_SILU}[hidden_act]
        self.layers = []
        i = 0
        while f"{pre}blocks.{i}.attn.qkv.weight" in sd:
            p = f"blocks.{i}."
            self.layers.append(dict(
                ln1w=g(p + "norm1.weight"), ln1b=g(p + "norm1.bias"),
                qkv_w=_pad_heads_rows(g(p + "attn.qkv.weight"), 3 * self.H, self.hd).contiguous(),
                qkv_b=_pad_heads_vec(g(p + "attn.qkv.bias"), 3 * self.H, self.hd, self.device),
                o_w=_pad_heads_cols(g(p + "attn.proj.weight"), self.H, self.hd).contiguous(), o_b=g(p + "attn.proj.bias"),
                ln2w=g(p + "norm2.weight"), ln2b=g(p + "norm2.bias"),
                fc1_w=g(p + "mlp.fc1.weight"), fc1_b=g(p + "mlp.fc1.bias"), fc2_w=g(p + "mlp.fc2.weight"), fc2_b=g(p + "mlp.fc2.bias")))
            i += 1
        self.lnq_w, self.lnq_b = g("merger.ln_q.weight"), g("merger.ln_q.bias")
        self.m0_w, self.m0_b = g("merger.mlp.0.weight"), g("merger.mlp.0.bias")
        self.m2_w, self.m2_b = g("merger.mlp.2.weight"), g("merger.mlp.2.bias")

    @property
    def padded_patch_dim(self) -> int:
        """Columns of the patch-embedding GEMM operand (C T p p rounded up to the k-tile)."""
        return self.patch_w.shape[1]

    @classmethod
    def from_random(cls, embed_dim=1280, depth=32, num_heads=16, mlp_ratio=4, out_hidden=3584, patch_size=14, temporal_patch_size=2,
                    merge=2, seed=0, device="cuda"):
        """Synthetic Qwen2-VL-7B-shaped tower (defaults; the 2B model has out_hidden 1536) drawn on the device."""
        shapes = {"patch_embed.proj.weight": (embed_dim, 3, temporal_patch_size, patch_size, patch_size),
                  "merger.ln_q.weight": (embed_dim,), "merger.ln_q.bias": (embed_dim,),
                  "merger.mlp.0.weight": (embed_dim * merge * merge, embed_dim * merge * merge), "merger.mlp.0.bias": (embed_dim * merge * merge,),
                  "merger.mlp.2.weight": (out_hidden, embed_dim * merge * merge), "merger.mlp.2.bias": (out_hidden,)}
        for i in range(depth):
            p = f"blocks.{i}."
            shapes.update({p + "attn.qkv.weight": (3 * embed_dim, embed_dim), p + "attn.qkv.bias": (3 * embed_dim,),
                           p + "attn.proj.weight": (embed_dim, embed_dim), p + "attn.proj.bias": (embed_dim,),
                           p + "norm1.weight": (embed_dim,), p + "norm1.bias": (embed_dim,), p + "norm2.weight": (embed_dim,), p + "norm2.bias": (embed_dim,),
                           p + "mlp.fc1.weight": (embed_dim * mlp_ratio, embed_dim), p + "mlp.fc1.bias": (embed_dim * mlp_ratio,),
                           p + "mlp.fc2.weight": (embed_dim, embed_dim * mlp_ratio), p + "mlp.fc2.bias": (embed_dim,)})
        return cls(_random_sd(shapes, seed, torch.device(device)), num_heads, merge, device=device)

    @classmethod
    def from_pretrained(cls, path: str, subfolder: str = "", device="cuda"):
        cfg, sd = _read_dir(path, subfolder)
        cfg = cfg.get("vision_config", cfg)
        sd = {k: v for k, v in sd.items() if "visual." in k or k.startswith(("patch_embed", "blocks", "merger"))}
        return cls(sd, cfg.get("num_heads", 16), cfg.get("spatial_merge_size", 2), cfg.get("hidden_act", "quick_gelu"), device=device)

    @torch.no_grad()
    def __call__(self, hidden_states: torch.Tensor, grid_thw, **_kw):
        """hidden_states: the image processor's flattened patches [S, C*T*p*p]; grid_thw [n_img, 3].
        Returns (last_hidden_state [S, D], pooler_output = merged tokens [S / merge^2, out_dim])."""
        grid = [[int(v) for v in row] for row in (grid_thw.tolist() if torch.is_tensor(grid_thw) else grid_thw)]
        src = hidden_states.to(self.device)
        src = src.contiguous() if src.dtype in (torch.float32, torch.bfloat16) else src.float().contiguous()
        S = src.shape[0]
        assert S == sum(t * h * w for t, h, w in grid)
        if src.dtype == torch.bfloat16 and src.shape[1] == self.patch_w.shape[1] != self.K_in:
            padded = src               # already the GEMM operand (td_qwen2_patchify_u8: device-side preprocessing)
        else:
            assert src.shape[1] == self.K_in
            padded = _hip.cast_pad_rows(src, self.patch_w.shape[1])
        h = _hip.linear(padded, self.patch_w)
        cos, sin = _hip.vision_rope_table(vision_position_ids(grid, self.merge).to(self.device, torch.int32).contiguous(), self.hd, self.theta)
        seg, a0 = [], 0
        for t, gh, gw in grid:                     # full attention inside each frame (cu_seqlens of the reference tower)
            for _ in range(t):
                seg.append((a0, a0 + gh * gw))
                a0 += gh * gw
        # several frames: one packed variable-length launch per layer instead of one launch per frame
        starts = torch.tensor([s0 for s0, _ in seg] + [S], dtype=torch.int32).to(self.device) if len(seg) > 1 else None
        max_len = max(s1 - s0 for s0, s1 in seg)
        for L in self.layers:
            x = _hip.layernorm(h, L["ln1w"], L["ln1b"], 1e-6)
            qkv = _hip.linear(x, L["qkv_w"], L["qkv_b"])
            _hip.rope_half(qkv, 2 * self.H, self.hd, cos, sin)                # q and k heads are adjacent 128-wide slots
            if starts is None:
                a = _hip.attention_padded(qkv, self.H, self.hd ** -0.5)
            else:
                a = _hip.attention_padded_varlen(qkv, self.H, self.hd ** -0.5, starts, max_len)
            h = _hip.linear(a, L["o_w"], L["o_b"], res=h)
            x = _hip.layernorm(h, L["ln2w"], L["ln2b"], 1e-6)
            h = _hip.linear(_hip.linear(x, L["fc1_w"], L["fc1_b"], act=self.act), L["fc2_w"], L["fc2_b"], res=h)
        m = _hip.layernorm(h, self.lnq_w, self.lnq_b, 1e-6).view(S // self.merge ** 2, -1)
        merged = _hip.linear(_hip.linear(m, self.m0_w, self.m0_b, act=_hip.ACT_GELU_ERF), self.m2_w, self.m2_b)
        return _EncoderOutput((h, merged))

"""Generates the golden vectors under tests/golden/ (run in the build container: `python tests/golden/make_golden.py`).

Pinned vectors come from the torch / transformers modules the reference itself instantiates
(nn.Linear, nn.GELU, transformers T5LayerNorm, F.interpolate); FLUX vectors come from oracle/flux_ref.py
itself (no diffusers here: they pin the oracle against drift, not against the reference -- "parity
unpinned", see the oracle header).  Weights are never stored: every fixture records the seed and the
expected output only; `transformers.__version__` and `torch.__version__` are recorded per file.
"""
import os
import sys

import torch
import torch.nn as nn
import transformers
from transformers.models.t5.modeling_t5 import T5LayerNorm

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import aligner_ref as A  # noqa: E402
from oracle import flux_ref as R  # noqa: E402

META = {"torch": torch.__version__, "transformers": transformers.__version__}


def aligner_case(mm_hidden, hidden, tokens, seed, dtype):
    """expected = the reference's nn.Sequential(Linear, GELU, Linear, T5LayerNorm) after 2x2 bilinear pooling."""
    sd = A.init_weights(mm_hidden, hidden, seed=seed, dtype=torch.float32)
    seq = nn.Sequential(nn.Linear(mm_hidden, hidden), nn.GELU(), nn.Linear(hidden, hidden), T5LayerNorm(hidden))
    seq.load_state_dict({k.replace("mm_projector.", ""): v for k, v in sd.items()})
    seq = seq.to(dtype)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(1, tokens, mm_hidden, generator=g).to(dtype)
    with torch.no_grad():
        if tokens == 257:   # CLIP path: CLS + 16x16 grid -> CLS + 8x8
            cls, grid = x[:, :1], x[:, 1:]
            gg = grid.reshape(1, 16, 16, mm_hidden).permute(0, 3, 1, 2)
            gg = torch.nn.functional.interpolate(gg, size=(8, 8), mode="bilinear", align_corners=False)
            xin = torch.cat([cls, gg.permute(0, 2, 3, 1).reshape(1, 64, mm_hidden)], dim=1)
        else:
            xin = x
        y = seq(xin)
    return {"mm_hidden": mm_hidden, "hidden": hidden, "tokens": tokens, "seed": seed, "dtype": str(dtype),
            "expected": y.clone(), **META}


def flux_case(seed):
    cfg = R.tiny_config(num_layers=1, num_single_layers=2)
    sd = R.init_weights(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    h2, w2, T = 4, 6, 16
    lat = torch.randn(1, h2 * w2, 64, generator=g).bfloat16()
    pe = torch.randn(1, T, cfg.joint_attention_dim, generator=g).bfloat16()
    pool = torch.randn(1, cfg.pooled_projection_dim, generator=g).bfloat16()
    with torch.no_grad():
        fwd = R.transformer_forward(sd, cfg, lat, pe, pool, torch.tensor([0.5]).bfloat16(),
                                    R.latent_image_ids(h2, w2).bfloat16(), torch.zeros(T, 3).bfloat16(), torch.tensor([3.5]))
        den = R.denoise(sd, cfg, lat, pe, pool, h2, w2, 3)
    return {"seed": seed, "h2": h2, "w2": w2, "T": T, "layers": 1, "singles": 2, "forward": fwd.clone(), "denoise3": den.clone(),
            "pinned_by": "oracle itself (diffusers 0.31.0 not available): drift check only", **META}


def qwen2_case(seed):
    """expected = transformers Qwen2VLTextModel(tiny).last_hidden_state (eager attention), fp32 and bf16."""
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLTextConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLTextModel
    from oracle import qwen2vl_ref as Q
    cfg = Q.tiny_config()
    hc = Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
                           num_key_value_heads=cfg.num_kv_heads, intermediate_size=cfg.intermediate, vocab_size=cfg.vocab,
                           rms_norm_eps=1e-6, max_position_embeddings=1024, bos_token_id=None, eos_token_id=None, pad_token_id=None,
                           rope_parameters={"rope_type": "default", "rope_theta": 1e6, "mrope_section": [16, 24, 24]})
    hc._attn_implementation = "eager"
    m = Qwen2VLTextModel(hc).eval()
    sd = Q.init_weights(cfg, seed=seed, dtype=torch.float32)
    m.load_state_dict({k[len("model."):]: v for k, v in sd.items() if k.startswith("model.")}, strict=False)
    n = 41
    ids = torch.randint(0, cfg.vocab, (1, n), generator=torch.Generator().manual_seed(seed + 1))
    pos = torch.stack([torch.arange(n), torch.arange(n) // 3 + 2, (torch.arange(n) * 2) % 11])[:, None, :]
    with torch.no_grad():
        h32 = m(input_ids=ids, position_ids=pos).last_hidden_state[0].clone()
        h16 = m.bfloat16()(input_ids=ids, position_ids=pos).last_hidden_state[0].clone()
    return {"seed": seed, "n": n, "ids": ids[0].clone(), "pos": pos[:, 0].clone(), "hidden_fp32": h32, "hidden_bf16": h16, **META}


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.save(aligner_case(1408, 4096, 257, 11, torch.float32), os.path.join(HERE, "aligner_clip_fp32.pt"))
    torch.save(aligner_case(1408, 4096, 257, 12, torch.bfloat16), os.path.join(HERE, "aligner_clip_bf16.pt"))
    torch.save(aligner_case(3584, 4096, 128, 13, torch.bfloat16), os.path.join(HERE, "aligner_lvlm7b_bf16.pt"))
    torch.save(flux_case(21), os.path.join(HERE, "flux_tiny_oracle.pt"))
    torch.save(qwen2_case(31), os.path.join(HERE, "qwen2vl_tiny.pt"))
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))

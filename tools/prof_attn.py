"""Runs each attention variant a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thinkdiff-mlre_amd"))
from thinkdiff import _hip
S, H = 4289, 24
qkv = torch.randn(1, S, 3 * H * 128, device="cuda").bfloat16()
out = torch.empty(1, S, H * 128, device="cuda", dtype=torch.bfloat16)
q, k, v = qkv[:, :, :H * 128], qkv[:, :, H * 128:2 * H * 128], qkv[:, :, 2 * H * 128:]
for var in (0, 1):      # 0: shipped (persistent where it applies), 1: one workgroup per item
    _hip.lib().td_attention_set_variant(var)
    for _ in range(3):
        _hip.attention(q, k, v, out, H, H)
torch.cuda.synchronize()

"""Determinism probe of the attention kernels: repeated launches on the same inputs, where do outputs differ."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
from thinkdiff import _hip

def probe(S, H, var, reps=6):
    g = torch.Generator().manual_seed(S + H)
    qkv = torch.randn(1, S, 3 * H * 128, generator=g).bfloat16().cuda()
    q, k, v = qkv[:, :, :H * 128], qkv[:, :, H * 128:2 * H * 128], qkv[:, :, 2 * H * 128:]
    L = _hip.lib()
    prev = L.td_attention_set_variant(var)
    outs = []
    for _ in range(reps):
        out = torch.zeros(1, S, H * 128, dtype=torch.bfloat16, device="cuda")
        _hip.attention(q, k, v, out, H, H)
        torch.cuda.synchronize()
        outs.append(out)
    L.td_attention_set_variant(prev)
    msg = []
    for i in range(1, reps):
        d = (outs[i].float() - outs[0].float())[0]
        nz = d.abs() > 0
        if nz.any():
            rows = nz.any(dim=1).nonzero().flatten()
            heads = nz.view(S, H, 128).any(dim=2).any(dim=0).nonzero().flatten()
            msg.append(f"rep {i}: {int(nz.sum())} elems, max |d| {float(d.abs().max()):.4g}, rows {rows[:6].tolist()}..{rows[-3:].tolist()} ({len(rows)}), heads {heads[:8].tolist()} ({len(heads)})")
    print(f"S={S} H={H} variant {var:#x}: " + ("deterministic" if not msg else "; ".join(msg)), flush=True)

for S, H in [(1000, 80), (4289, 24), (520, 50)]:
    for var in (0, 1, 2, 0x800, 0x801):
        probe(S, H, var)

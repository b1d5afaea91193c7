"""Which (query tile, head) items of the stream-K attention differ from the one-workgroup-per-item kernel (diagnostic)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thinkdiff-mlre_amd"))
from thinkdiff import _hip
S, H = int(os.environ.get("S", 4289)), int(os.environ.get("H", 24))
W = H * 128
g = torch.Generator().manual_seed(1)
qkv = torch.randn(1, S, 3 * W, generator=g).bfloat16().cuda()
q, k, v = qkv[:, :, :W], qkv[:, :, W:2 * W], qkv[:, :, 2 * W:]
outs = {}
for var in (1, 2, 0):
    _hip.lib().td_attention_set_variant(var)
    o = torch.zeros(1, S, W, device="cuda", dtype=torch.bfloat16)
    _hip.attention(q, k, v, o, H, H)
    torch.cuda.synchronize()
    outs[var] = o.float()
_hip.lib().td_attention_set_variant(0)
nq, nt, G = (S + 255) // 256, (S + 63) // 64, 256
total = nq * H * nt
for var in (2, 0):
    d = (outs[var] - outs[1]).abs()[0]
    print(f"variant {var}: max abs diff {float(d.max()):.4f}")
    bad = []
    for item in range(nq * H):
        qb, h = item % nq, item // nq
        e = float(d[qb * 256:(qb + 1) * 256, h * 128:(h + 1) * 128].max())
        if e > 0.02:
            # which ranges touch this item
            lo, hi = item * nt, (item + 1) * nt
            rs = [r for r in range(G) if total * r // G < hi and total * (r + 1) // G > lo]
            bad.append((item, qb, h, round(e, 3), rs, [(max(lo, total * r // G) - lo, min(hi, total * (r + 1) // G) - lo) for r in rs]))
    print(f"  {len(bad)} bad items of {nq * H}")
    for b in bad[:12]:
        print("   ", b)
    # rows pattern inside the first bad item
    if bad:
        item, qb, h = bad[0][:3]
        blk = d[qb * 256:(qb + 1) * 256, h * 128:(h + 1) * 128]
        print("   bad rows in first bad item (per 32-row wave):", [round(float(blk[w * 32:(w + 1) * 32].max()), 3) for w in range(8)])
        print("   bad cols (per 32):", [round(float(blk[:, c * 32:(c + 1) * 32].max()), 3) for c in range(4)])

import sys, math, torch
sys.path.insert(0, '/root/repo')
from oracle import flux_ref as R
import torch.nn.functional as F

def attn8(q, k, v, headroom, form="linear"):
    B, H, S, hd = q.shape
    c = (hd ** -0.5) * math.log2(math.e)
    out = torch.empty(B, S, H * hd)
    pad = (-S) % 64
    for h in range(H):
        q8, sq = R._e8m0_quant(q[0, h].float() * c, (1,))
        k8, sk = R._e8m0_quant(k[0, h].float(), (1,))
        vg = F.pad(v[0, h].float(), (0, 0, 0, pad)).view(-1, 64, hd)
        v8, sv = R._e8m0_quant(vg, (1, 2))
        vq = (v8 * sv).view(-1, hd)[:S]
        s = (q8 * sq) @ (k8 * sk).T
        ref = torch.ceil(s.amax(dim=1, keepdim=True))
        p = torch.round(8.0 * (s - ref) + 8 * headroom + 56).clamp_(0, 126).to(torch.uint8).view(torch.float8_e4m3fn).float()
        out[0, :, h * hd:(h + 1) * hd] = (p @ vq) / p.sum(dim=1, keepdim=True)
    return out[0]

torch.manual_seed(0)
S, H = 4289, 2
qkv = torch.randn(S, 3 * H * 128).bfloat16()
x = qkv.view(S, 3, H, 128).permute(1, 2, 0, 3)[:, None]
exact = (torch.softmax(x[0][0].float() @ x[1][0].float().transpose(-1, -2) / math.sqrt(128), dim=-1) @ x[2][0].float()).transpose(0, 1).reshape(S, H * 128)
outs = {h: attn8(x[0], x[1], x[2], h) for h in (7, 5, 3)}
def rel(a, b): return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())
def worst(a, b): return float(((a - b).pow(2).mean(dim=1).sqrt() / b.pow(2).mean().sqrt()).max())
for h in (7, 5, 3):
    print(f"headroom {h}: vs exact {rel(outs[h], exact):.4e} worst row {worst(outs[h], exact):.3e};  vs headroom 7: {rel(outs[h], outs[7]):.3e} worst row {worst(outs[h], outs[7]):.3e}")

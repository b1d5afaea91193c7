"""Runs the main FLUX GEMM shapes with cold weights (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thinkdiff-mlre_amd"))
from thinkdiff import _hip
for M, N, K, cfg in [(4289, 21504, 3072, 0), (4289, 3072, 15360, 3), (4289, 3072, 15360, 0)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    pool = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(6)]
    b = torch.randn(N, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for i in range(6):
        _hip.linear_grouped2(x, pool[i], b, y, None, None, None, None, tile_cfg=cfg)
torch.cuda.synchronize()

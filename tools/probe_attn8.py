"""Timing-only probes of the 8-bit attention kernel (a -DTD_ATTN8_PROBE build loaded through TD_HIP_LIB): what each stage of the per-tile
chain costs on the critical path.  Results of probes 1-6 are wrong by construction; the launch duration is the measurement."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
from thinkdiff import _hip

NAMES = {0: "shipped", 1: "no row-max chain", 2: "no reference logic", 3: "P not converted (P.V free of the scores)", 4: "no score MFMAs", 5: "no P.V MFMAs", 6: "no tile barrier"}


def timeit(fn, iters=10, warmup=2):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


S, H = 4289, 24
W = H * 128
pool = [torch.randn(S, 3 * W, device="cuda").bfloat16() for _ in range(6)]
out = torch.empty(S, W, device="cuda", dtype=torch.bfloat16)
L = _hip.lib()
L.td_attention_fp8_workspace_bytes.restype = __import__("ctypes").c_size_t
ws = torch.empty(int(L.td_attention_fp8_workspace_bytes(S, S, H)), dtype=torch.uint8, device="cuda")
st = {"i": 0}


def f8():
    st["i"] = (st["i"] + 1) % len(pool)
    q = pool[st["i"]]
    _hip.attention_fp8(q[:, :W], q[:, W:2 * W], q[:, 2 * W:], out, H, workspace=ws)


best = {k: 1e9 for k in NAMES}
for _ in range(4):
    for k in NAMES:
        L.td_attention_set_variant(k << 4)
        best[k] = min(best[k], timeit(f8))
L.td_attention_set_variant(0)
for k, n in NAMES.items():
    print(f"probe {k} {n:45s} {best[k]*1e6:7.1f} us (pack pass included)   {best[k]/best[0]:.3f}", flush=True)

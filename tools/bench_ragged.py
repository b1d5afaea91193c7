"""A/B of the GEMM's ragged-tile loop (TD_GEMM_NO_RAGGED) on config 5's shapes: M = 4354 joint rows, and the grouped 4096 + 258 launch."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
from thinkdiff import _hip


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


for name, M, N, K in [("single_in", 4354, 21504, 3072), ("single_out", 4354, 3072, 15360), ("qkv(joint)", 4354, 9216, 3072), ("single_in T=193", 4289, 21504, 3072)]:
    x = torch.randn(M, K, device="cuda").bfloat16()
    npool = max(2, int(0.8e9 // (N * K * 2)))
    pool = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(npool)]
    ipool = [_hip.quant_rows_int8(w) for w in pool]
    xi, xis = _hip.quant_rows_int8(x)
    b = torch.randn(N, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    st = {"i": 0}
    def f16():
        st["i"] = (st["i"] + 1) % npool
        _hip.linear(x, pool[st["i"]], b, out=y)
    def f8():
        st["i"] = (st["i"] + 1) % npool
        wq, ws = ipool[st["i"]]
        _hip.linear_int8(xi, xis, wq, ws, b, out=y)
    best = {}
    for rnd in range(4):
        for rag in (1, 0):
            if rag:
                os.environ.pop("TD_GEMM_NO_RAGGED", None)
            else:
                os.environ["TD_GEMM_NO_RAGGED"] = "1"
            for k, f in (("bf16", f16), ("int8", f8)):
                t = timeit(f, iters=10, warmup=2)
                best[k, rag] = min(best.get((k, rag), 1e9), t)
    os.environ.pop("TD_GEMM_NO_RAGGED", None)
    fl = 2.0 * M * N * K
    print(f"{name:16s} M={M} N={N} K={K}: " + "   ".join(f"{k} ragged {best[k,1]*1e6:7.1f} us / off {best[k,0]*1e6:7.1f} us (x{best[k,0]/best[k,1]:.3f}, {fl/best[k,1]/1e12:6.0f} TF/s)" for k in ("bf16", "int8")), flush=True)

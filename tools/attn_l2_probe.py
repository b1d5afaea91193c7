import sys, os, torch
sys.path.insert(0, os.path.join(os.getcwd(), "thinkdiff-mlre_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from thinkdiff import _hip
from bench_ops import timeit
S, H = 4289, 24
W = H * 128
for hkv in (24, 1):
    pool = [torch.randn(1, S, W + 2 * hkv * 128, device="cuda").bfloat16() for _ in range(6)]
    out = torch.empty(1, S, W, device="cuda", dtype=torch.bfloat16)
    st = {"i": 0}
    def f():
        st["i"] = (st["i"] + 1) % len(pool)
        q = pool[st["i"]]
        _hip.attention(q[:, :, :W], q[:, :, W:W + hkv * 128], q[:, :, W + hkv * 128:], out, H, hkv)
    best = 1e9
    for _ in range(5):
        best = min(best, timeit(f, iters=10, warmup=2))
    print(f"S={S} Hq={H} Hkv={hkv}: {best*1e3:.1f} us  {4.0*S*S*H*128/best/1e9:.0f} TF/s", flush=True)

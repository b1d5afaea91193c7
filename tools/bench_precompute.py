"""BASELINE config 4 throughput on synthetic Qwen2-VL-2B-shaped weights: MllamaVllmGenerate_1.forward over a loader batch
(image + instruction per sample, generation until max_tokens since random weights never emit EOS) -> samples/s on one GPU."""
import os, sys, time
import torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
from thinkdiff.common.config import Node
from thinkdiff.models import providers
from thinkdiff.models.mllama_vllm_generate_1 import MllamaVllmGenerate_1
from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig

max_tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 64
nseq = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tc = Qwen2VLTextConfig(hidden_size=1536, num_hidden_layers=28, num_attention_heads=12, num_key_value_heads=2, intermediate_size=8960,
                       vocab_size=151936, tie_word_embeddings=True)
m = MllamaVllmGenerate_1(tc, vllm_config={"max_model_len": 2048, "max_tokens": max_tokens, "min_tokens": 1, "ignore_eos": False, "max_num_seqs": nseq})
providers.load_lvlm_frontend(Node({"synthetic": True, "seed": 0, "synthetic_max_image_tokens": 320}), m, "cuda")
imgs = [[Image.new("RGB", (500, 375), (10 * k % 255, 80, 160))] for k in range(max(32, 2 * nseq))]
samples = {"images": imgs, "answers": ["Describe the image in one sentence."] * len(imgs)}
m.forward({"images": imgs[:nseq], "answers": samples["answers"][:nseq]})
torch.cuda.synchronize()
t0 = time.perf_counter()
out = m.forward(samples)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ntok = sum(len(t) for t in out["generated_token"]["output_token_ids"])
npr = sum(len(t) for t in out["generated_token"]["input_prompt_token_ids"])
print(f"decode batch {m.decode_batch}: {len(imgs)} samples in {dt:.2f} s = {len(imgs)/dt:.1f} samples/s; {npr/len(imgs):.0f} prompt tokens and {ntok/len(imgs):.0f} generated tokens per sample; {ntok/dt:.0f} generated tokens/s")

if os.environ.get("TD_PHASES"):       # where one chunk's time goes: request building (CPU preprocessing + ViT + splice), prefill, decode
    import ctypes
    from thinkdiff import _hip
    phases = {}
    def timed(name, fn):
        def w(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(); phases[name] = phases.get(name, 0.0) + time.perf_counter() - t
            return r
        return w
    m._requests = timed("requests (preprocess + ViT + splice)", m._requests)
    class _Vis:
        def __init__(self, v): self._v, self._f = v, timed("  of which vision tower", v.__call__)
        def __call__(self, *a, **k): return self._f(*a, **k)
        def __getattr__(self, k): return getattr(self._v, k)
    m.visual = _Vis(m.visual)
    eng = m.mllama
    eng.decode_batch = timed("decode steps", eng.decode_batch)
    L = eng._L
    class _L2:
        def __getattr__(self, k):
            f = getattr(L, k)
            return timed("prefill", f) if k == "td_qwen2_prefill_batch" else f
    eng._L = _L2()
    t0 = time.perf_counter()
    m.forward(samples)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
    print(f"phases over {len(imgs)} samples, total {tot*1e3:.0f} ms:")
    for k, v in phases.items():
        print(f"  {k}: {v*1e3:.0f} ms")

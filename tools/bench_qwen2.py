"""Qwen2-VL-7B-shaped text engine (synthetic weights): prefill and KV-cached decode timing (BASELINE config 3's
embedding extraction: ~n_prompt prefill tokens + 128 generated tokens with hidden-state capture)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thinkdiff-mlre_amd"))
from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine

e = Qwen2VLTextEngine(Qwen2VLTextConfig(), max_model_len=4096).init_random(0)
cfg = e.config
params = (cfg.num_hidden_layers * (cfg.hidden_size * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * 128 + cfg.num_attention_heads * 128 * cfg.hidden_size
          + 3 * cfg.hidden_size * cfg.intermediate_size) + 2 * cfg.vocab_size * cfg.hidden_size)
print(f"params {params/1e9:.2f} B")
for n in (64, 512, 1024):
    ids = torch.randint(0, cfg.vocab_size, (n,), dtype=torch.int32)
    pos = e.text_position_ids(n)
    e.forward(pos, ids, want_logits=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        e.forward(pos, ids, want_logits=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(f"prefill n={n:5d}: {ms:8.2f} ms   {2*params*n/ms/1e9:7.1f} TF/s", flush=True)
n0 = 512
ids = torch.randint(0, cfg.vocab_size, (n0,), dtype=torch.int32)
e.forward(e.text_position_ids(n0), ids)
tok = torch.tensor([5], dtype=torch.int32)
for i in range(4):
    e.forward(e.text_position_ids(1, n0 + i), tok, pos0=n0 + i, want_logits=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 64
for i in range(N):
    e.forward(e.text_position_ids(1, n0 + 4 + i), tok, pos0=n0 + 4 + i, want_logits=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / N * 1e3
print(f"decode (cache {n0}): {ms:.3f} ms/token -> {2*params/ms/1e6:.0f} GB/s of weights   (128 tokens = {128*ms:.0f} ms)")

# the reference's request shape: ~300-token prompt, 128 sampled tokens (T=0.6, top-p 0.9, ignore_eos), hidden states captured
from thinkdiff.models.qwen2_vl import SamplingParams
sp = SamplingParams(temperature=0.6, top_p=0.9, max_tokens=128, min_tokens=128, ignore_eos=True)
prompt = torch.randint(0, cfg.vocab_size, (300,)).tolist()
gen = torch.Generator(device="cuda").manual_seed(0)
e.generate(prompt, sp, generator=gen); torch.cuda.synchronize()
t0 = time.perf_counter()
out = e.generate(prompt, sp, generator=gen)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"generate(): 300-token prompt + 128 sampled tokens with hidden states: {dt*1e3:.0f} ms ({out['hidden_states'].shape[0]} states)")

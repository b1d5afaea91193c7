"""KV-cached decode step of the Qwen2-VL engines (synthetic weights) against the number of sequences sharing the pass over the
weights: ms per step and per sequence-token, for the 2B (precompute job) and 7B (ThinkDiff-LVLM) shapes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thinkdiff-mlre_amd"))
from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine

shapes = {"2B": Qwen2VLTextConfig(hidden_size=1536, num_hidden_layers=28, num_attention_heads=12, num_key_value_heads=2, intermediate_size=8960,
                                  vocab_size=151936, tie_word_embeddings=True),
          "7B": Qwen2VLTextConfig()}
group = [int(a[2:]) for a in sys.argv[1:] if a.startswith("G=")]      # G=<q heads per workgroup of the decode attention>: A/B of td_attention_decode_set_group
if group:
    from thinkdiff import _hip
    _hip.lib().td_attention_decode_set_group(group[0])
which = [a for a in sys.argv[1:] if not a.isdigit() and not a.startswith("G=")] or ["2B", "7B"]
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 4, 8, 16, 24, 32, 48, 64]
CACHE = int(os.environ.get("TD_BENCH_CACHE", "300"))
for name in which:
    cfg = shapes[name]
    slots = max(64, max(batches))
    e = Qwen2VLTextEngine(cfg, max_model_len=max(512, (CACHE + 64 + 63) // 64 * 64), n_slots=slots, prefill_rows=64 * 320).init_random(0)
    wbytes = 2 * (cfg.num_hidden_layers * (cfg.hidden_size * (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * 128 + cfg.num_attention_heads * 128 * cfg.hidden_size
                  + 3 * cfg.hidden_size * cfg.intermediate_size) + cfg.vocab_size * cfg.hidden_size)
    for B in batches:
        toks = [5] * B
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            N = 24
            for i in range(N):
                pos = torch.full((3, B), CACHE + i, dtype=torch.int32)
                e.decode_batch(toks, pos, [CACHE + i] * B)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / N * 1e3
        print(f"{name} decode B={B:3d}: {ms:7.3f} ms/step  {ms / B * 1e3:8.1f} us per sequence-token  {wbytes / ms / 1e9:6.2f} TB/s of weights", flush=True)
    del e
    torch.cuda.empty_cache()

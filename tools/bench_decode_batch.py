"""KV-cached decode step time vs batch (sequences per weight pass), Qwen2-VL-2B and -7B shapes, synthetic weights."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "thinkdiff-mlre_amd"))
from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine

for name, tc in (("2B", Qwen2VLTextConfig(hidden_size=1536, num_hidden_layers=28, num_attention_heads=12, num_key_value_heads=2, intermediate_size=8960, vocab_size=151936, tie_word_embeddings=True)),
                 ("7B", Qwen2VLTextConfig())):
    e = Qwen2VLTextEngine(tc, max_model_len=1024, n_slots=16).init_random(0)
    n0 = 300
    for b in range(16):
        e.forward(e.text_position_ids(n0), torch.randint(0, 1000, (n0,), dtype=torch.int32), slot=b)
    for B in (1, 2, 4, 8, 16):
        tok = torch.randint(0, 1000, (B,), dtype=torch.int32)
        for i in range(3):
            e.decode_batch(tok, [[n0 + i] * B] * 3, [n0 + i] * B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        N = 30
        for i in range(N):
            e.decode_batch(tok, [[n0 + 3 + i] * B] * 3, [n0 + 3 + i] * B)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / N * 1e3
        print(f"{name} decode batch {B:2d}: {ms:6.3f} ms/step  {B/ms*1e3:7.0f} tokens/s", flush=True)
    del e

"""GPU parity of the Qwen2-VL text-decoder engine (td_qwen2_*) against the CPU oracle (oracle/qwen2vl_ref.py,
pinned to transformers' Qwen2VLTextModel in tests/test_oracle_cpu.py).

Tolerance: same bf16 rounding points, different fp32 summation order -> relative RMSE <= 2e-2 of the hidden
state RMS vs the bf16 oracle, and no further from the fp32 oracle than 1.5x the bf16 oracle itself is.
"""
import pytest
import torch

from oracle import qwen2vl_ref as Q

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def _engine(cfg, sd, max_len=512):
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine
    e = Qwen2VLTextEngine(Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers,
                                            num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                                            intermediate_size=cfg.intermediate, vocab_size=cfg.vocab,
                                            tie_word_embeddings=cfg.tie_embeddings), max_model_len=max_len)
    e.load_state_dict(sd)
    return e


@pytest.mark.parametrize("n,tie", [(37, False), (300, True)])
def test_prefill_hidden_states_match_oracle(hip, n, tie):
    cfg = Q.tiny_config(tie_embeddings=tie)
    sd = Q.init_weights(cfg, seed=n)
    e = _engine(cfg, sd)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, cfg.vocab, (n,), generator=g).to(torch.int32)
    # non-trivial M-RoPE streams (an "image" block in the middle uses a 2-D grid)
    pos = torch.stack([torch.arange(n), torch.arange(n) // 3 + 2, (torch.arange(n) * 2) % 11]).to(torch.int32)
    ref16, _ = Q.text_model_hidden(sd, cfg, pos, token_ids=ids.long())
    ref32, _ = Q.text_model_hidden({k: v.float() for k, v in sd.items()}, cfg, pos, token_ids=ids.long())
    hid, logits = e.forward(pos, ids, want_logits=True)
    torch.cuda.synchronize()
    e16, e32, eref = _rel(hid, ref16), _rel(hid, ref32), _rel(ref16, ref32)
    print(f"rel-RMSE hip~bf16 {e16:.4f} hip~fp32 {e32:.4f} bf16~fp32 {eref:.4f}")
    assert e16 < 2e-2 and e32 < 1.5 * eref + 2e-3
    lref = Q.lm_logits(sd, cfg, ref16[-1])
    assert _rel(logits, lref) < 3e-2


def test_kv_cached_decode_equals_prefill(hip):
    """Prefill 40 tokens, then feed 9 more one at a time through the KV cache: hidden states must equal a single
    49-token prefill (and the oracle)."""
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=5)
    e = _engine(cfg, sd)
    g = torch.Generator().manual_seed(2)
    n0, n1 = 40, 9
    ids = torch.randint(0, cfg.vocab, (n0 + n1,), generator=g).to(torch.int32)
    pos = Q.text_position_ids(n0 + n1)
    full, _ = e.forward(pos, ids)
    a, _ = e.forward(pos[:, :n0], ids[:n0])
    steps = [e.forward(pos[:, n0 + i:n0 + i + 1], ids[n0 + i:n0 + i + 1], pos0=n0 + i)[0] for i in range(n1)]
    torch.cuda.synchronize()
    inc = torch.cat([a] + steps)
    assert _rel(inc, full) < 5e-3
    ref, _ = Q.text_model_hidden(sd, cfg, pos, token_ids=ids.long())
    assert _rel(inc, ref) < 2e-2


def test_get_embed_teacher_forced(hip):
    """ThinkDiff-LVLM get_embed: hidden states of forced output tokens -> aligner, all embedding_type selections."""
    from oracle import aligner_ref as A
    from thinkdiff.models.mllama_vllm_t5_embed_decoder_2 import MllamaVllmT5EmbedDecoderForConditionalGeneration_5
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=9)
    asd = A.init_weights(cfg.hidden, 4096, seed=4)
    m = MllamaVllmT5EmbedDecoderForConditionalGeneration_5(
        Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
                          num_key_value_heads=cfg.num_kv_heads, intermediate_size=cfg.intermediate, vocab_size=cfg.vocab),
        vllm_config={"max_model_len": 256, "max_tokens": 6, "min_tokens": 6})
    m.mllama.load_state_dict(sd)
    m.load_state_dict(asd)
    g = torch.Generator().manual_seed(3)
    prompt = torch.randint(0, cfg.vocab, (20,), generator=g).tolist()
    forced = torch.randint(0, cfg.vocab, (6,), generator=g).tolist()
    pos = Q.text_position_ids(26)
    ref_h, _ = Q.text_model_hidden(sd, cfg, pos, token_ids=torch.tensor(prompt + forced))
    # LVLM path: aligner Linear layers in bf16, T5LayerNorm in fp32 (weight fp32) -> cast once
    def ref_aligner(h):
        y = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(h, asd["mm_projector.0.weight"], asd["mm_projector.0.bias"])),
                                       asd["mm_projector.2.weight"], asd["mm_projector.2.bias"])
        return A.t5_layer_norm(y.float(), asd["mm_projector.3.weight"].float()).bfloat16()
    for et, sl in [("both", slice(0, 26)), ("input_embed", slice(0, 20)), ("input_no_system", slice(14, 20)), ("output_embed", slice(20, 26))]:
        embs, texts = m.get_embed([{"prompt_token_ids": prompt}], embedding_type=et, need_process=False, forced_output_ids=[forced])
        torch.cuda.synchronize()
        assert embs[0].shape == (sl.stop - sl.start, 4096) and texts == [" ".join(map(str, forced))]
        assert _rel(embs[0], ref_aligner(ref_h[sl])) < 3e-2


def test_sampling_is_seeded_and_respects_lengths(hip):
    cfg = Q.tiny_config()
    e = _engine(cfg, Q.init_weights(cfg, seed=1))
    from thinkdiff.models.qwen2_vl import SamplingParams
    sp = SamplingParams(temperature=0.6, top_p=0.9, max_tokens=8, min_tokens=8, ignore_eos=True)
    runs = []
    for _ in range(2):
        g = torch.Generator(device="cuda").manual_seed(42)
        runs.append(e.generate([1, 2, 3, 4, 5], sp, generator=g))
    assert runs[0]["token_ids"] == runs[1]["token_ids"] and len(runs[0]["token_ids"]) == 8
    assert runs[0]["hidden_states"].shape == (8, cfg.hidden) and runs[0]["prompt_hidden_states"].shape == (5, cfg.hidden)


def test_batched_decode_equals_one_sequence_at_a_time(hip):
    """generate_batch (prefill per slot, then every step advances all live sequences in one pass over the weights, with
    slot compaction as sequences finish) against generate() run per request: same hidden states, teacher-forced ids of
    different lengths, prompts of different lengths, one request with non-trivial M-RoPE streams."""
    from thinkdiff.models.qwen2_vl import SamplingParams
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=11)
    e = _engine(cfg, sd, max_len=2048)
    g = torch.Generator().manual_seed(4)
    lens, gens = [37, 5, 120, 64, 9], [12, 3, 7, 12, 1]
    reqs, forced = [], []
    for n, k in zip(lens, gens):
        ids = torch.randint(0, cfg.vocab, (n,), generator=g).tolist()
        reqs.append({"prompt_token_ids": ids})
        forced.append(torch.randint(0, cfg.vocab, (k,), generator=g).tolist())
    reqs[2]["position_ids"] = torch.stack([torch.arange(120), torch.arange(120) // 3 + 2, (torch.arange(120) * 2) % 11]).to(torch.int32)
    sp = SamplingParams(max_tokens=12, min_tokens=12, ignore_eos=True)
    single = [e.generate(r["prompt_token_ids"], sp, position_ids=r.get("position_ids"), forced_output_ids=f) for r, f in zip(reqs, forced)]
    single = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in single]
    e.set_slots(8)
    assert e.slot_len == 256
    batch = e.generate_batch(reqs, sp, forced_output_ids=forced)        # 5 x 120 padded rows fit the 2048-row workspace: batched prefill
    torch.cuda.synchronize()
    for a, b, k in zip(single, batch, gens):
        assert b["token_ids"] == a["token_ids"] and b["hidden_states"].shape == (k, cfg.hidden)
        assert b["prompt_hidden_states"].shape == a["prompt_hidden_states"].shape
        assert _rel(b["prompt_hidden_states"], a["prompt_hidden_states"]) < 5e-3          # different tile schedule (M = 600 vs M = n)
        assert _rel(b["hidden_states"], a["hidden_states"]) < 5e-3
    # sampled continuation: every sequence yields max_tokens tokens and hidden states, greedy = argmax of its own logits
    sp0 = SamplingParams(temperature=0.0, max_tokens=6, min_tokens=6, ignore_eos=True)
    gb = e.generate_batch(reqs[:3], sp0)
    e.set_slots(1)
    for r, o in zip(reqs[:3], gb):
        ref = e.generate(r["prompt_token_ids"], sp0, position_ids=r.get("position_ids"))
        assert o["token_ids"] == ref["token_ids"] and _rel(o["hidden_states"], ref["hidden_states"]) < 5e-3


def test_batched_decode_of_forty_sequences(hip):
    """More than 16 sequences per pass over the weights (td_gemv_mfma_kernel with 3 activation blocks, gated and split-output
    forms included): 40 requests of different lengths, teacher-forced, against generate() per request."""
    from thinkdiff.models.qwen2_vl import SamplingParams
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=12)
    e = _engine(cfg, sd, max_len=8192)
    g = torch.Generator().manual_seed(6)
    B = 40
    lens = [5 + (7 * i) % 60 for i in range(B)]
    gens = [1 + (5 * i) % 9 for i in range(B)]
    reqs = [{"prompt_token_ids": torch.randint(0, cfg.vocab, (n,), generator=g).tolist()} for n in lens]
    forced = [torch.randint(0, cfg.vocab, (k,), generator=g).tolist() for k in gens]
    sp = SamplingParams(max_tokens=9, min_tokens=9, ignore_eos=True)
    single = [e.generate(r["prompt_token_ids"], sp, forced_output_ids=f) for r, f in zip(reqs, forced)]
    single = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in single]
    e.set_slots(64)
    assert e.slot_len == 128
    batch = e.generate_batch(reqs, sp, forced_output_ids=forced)
    torch.cuda.synchronize()
    for a, b, k in zip(single, batch, gens):
        assert b["token_ids"] == a["token_ids"] and b["hidden_states"].shape == (k, cfg.hidden)
        assert _rel(b["prompt_hidden_states"], a["prompt_hidden_states"]) < 5e-3
        assert _rel(b["hidden_states"], a["hidden_states"]) < 5e-3


@pytest.mark.parametrize("B", [65, 150, 256])
def test_batched_decode_beyond_64_sequences(hip, B):
    """max_num_seqs = 256 of the precompute job (configs/qwen2_vl_embed_ccsbu.yaml:20): above 64 sequences a decode step runs its Linears on the tile
    kernels, K split over workgroups for the narrow outputs (TdGemmParams::split_k).  B requests of different lengths, teacher-forced, against
    generate() per request; the step is also bit-reproducible (graph replay included: steps 0 eager, 1 capture, 2+ replay)."""
    from thinkdiff.models.qwen2_vl import SamplingParams
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=21)
    e = _engine(cfg, sd, max_len=256 * 96)
    g = torch.Generator().manual_seed(8)
    lens = [3 + (11 * i) % 50 for i in range(B)]
    gens = [2 + (3 * i) % 5 for i in range(B)]
    reqs = [{"prompt_token_ids": torch.randint(0, cfg.vocab, (n,), generator=g).tolist()} for n in lens]
    forced = [torch.randint(0, cfg.vocab, (k,), generator=g).tolist() for k in gens]
    sp = SamplingParams(max_tokens=6, min_tokens=6, ignore_eos=True)
    idx = list(range(0, B, max(1, B // 24)))          # a sample of the requests is checked one at a time
    single = {}
    for i in idx:
        o = e.generate(reqs[i]["prompt_token_ids"], sp, forced_output_ids=forced[i])
        single[i] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()}
    e.set_slots(256)
    assert e.slot_len == 96
    runs = []
    for _ in range(2):
        batch = e.generate_batch(reqs, sp, forced_output_ids=forced)
        torch.cuda.synchronize()
        runs.append([{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in batch])
    for i in idx:
        a, b = single[i], runs[0][i]
        assert b["token_ids"] == a["token_ids"] and b["hidden_states"].shape == (gens[i], cfg.hidden)
        assert _rel(b["prompt_hidden_states"], a["prompt_hidden_states"]) < 5e-3
        assert _rel(b["hidden_states"], a["hidden_states"]) < 5e-3
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a["hidden_states"], b["hidden_states"])


@pytest.mark.parametrize("Hq,Hkv,G", [(7, 1, 7), (6, 1, 6), (6, 2, 3), (4, 2, 2), (4, 1, 4)])
def test_decode_attention_groups_q_heads_of_a_kv_head(hip, Hq, Hkv, G):
    """td_attention_decode_set_group: one workgroup serves G q heads of a kv head (K/V read once).  The batched decode step of a
    decoder with Hq / Hkv q heads per kv head gives the same hidden states and logits with the forced group as with one head
    per workgroup (the automatic choice needs >= 256 workgroups, i.e. large batches of a full-size model)."""
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine
    tc = Qwen2VLTextConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=Hq, num_key_value_heads=Hkv, intermediate_size=1024, vocab_size=1024)
    e = Qwen2VLTextEngine(tc, max_model_len=64, n_slots=5).init_random(3)
    g = torch.Generator().manual_seed(2)
    lens = [9, 33, 1, 20, 47]
    for b, n in enumerate(lens):
        e.forward(e.text_position_ids(n), torch.randint(0, 1024, (n,), generator=g).to(torch.int32), slot=b)
    toks = torch.randint(0, 1024, (5,), generator=g).tolist()
    pos = torch.tensor([lens] * 3, dtype=torch.int32)
    outs = {}
    for grp in (1, G):
        prev = hip.lib().td_attention_decode_set_group(grp)
        try:
            h, lg = e.decode_batch(toks, pos, lens)
            torch.cuda.synchronize()
            outs[grp] = (h.clone(), lg.clone())
        finally:
            hip.lib().td_attention_decode_set_group(prev)
    assert torch.isfinite(outs[G][0].float()).all()
    assert _rel(outs[G][0], outs[1][0]) < 2e-3 and _rel(outs[G][1], outs[1][1]) < 2e-3


@pytest.mark.parametrize("Hq,Hkv,G", [(12, 2, 1), (12, 2, 3), (6, 2, 3), (28, 4, 7), (4, 4, 1)])
def test_decode_fused_rope_and_cache_write_bit_identical(hip, Hq, Hkv, G):
    """td_qwen2_set_fused_rope: the decode attention rotates its own q heads and the new key, attends the new key / value from registers and
    writes them to the cache.  Hidden states, logits AND the cache (seen through three further steps) must be bit-identical to the form with
    the separate rope + scatter launch, for one head per workgroup and for grouped q heads, sequence lengths that put the new key in every
    key slot of the 16-slot order, and three distinct M-RoPE streams."""
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine
    tc = Qwen2VLTextConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=Hq, num_key_value_heads=Hkv, intermediate_size=1024, vocab_size=1024)
    lens = [1, 2, 15, 16, 17, 31, 63, 64, 65, 100, 130]
    outs = {}
    for fused in (False, True):
        e = Qwen2VLTextEngine(tc, max_model_len=len(lens) * 160, n_slots=len(lens)).init_random(5)
        assert e.set_fused_rope(fused) is True        # the default is the fused form
        g = torch.Generator().manual_seed(9)
        for b, n in enumerate(lens):
            e.forward(e.text_position_ids(n), torch.randint(0, 1024, (n,), generator=g).to(torch.int32), slot=b)
        prev = hip.lib().td_attention_decode_set_group(G)
        try:
            cur, steps = list(lens), []
            for step in range(4):          # step 0 eager, step 1 captures the graph, steps 2-3 replay it
                toks = torch.randint(0, 1024, (len(lens),), generator=g).tolist()
                pos = torch.tensor([cur, [c // 2 + 1 for c in cur], [(3 * c) % 7 for c in cur]], dtype=torch.int32)
                h, lg = e.decode_batch(toks, pos, cur)
                torch.cuda.synchronize()
                steps.append((h.clone(), lg.clone()))
                cur = [c + 1 for c in cur]
        finally:
            hip.lib().td_attention_decode_set_group(prev)
        outs[fused] = steps
    for (ha, la), (hb, lb) in zip(outs[False], outs[True]):
        assert torch.isfinite(hb.float()).all() and torch.equal(ha, hb) and torch.equal(la, lb)


def test_batched_prefill_in_several_passes(hip):
    """A request batch whose padded prompts exceed the activation workspace (40 x 64 rows against 256) is prefilled four
    sequences per pass into consecutive cache slots (td_qwen2_prefill_batch_at): same states as one request at a time."""
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine, SamplingParams
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=13)
    e = Qwen2VLTextEngine(Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
                                            num_key_value_heads=cfg.num_kv_heads, intermediate_size=cfg.intermediate, vocab_size=cfg.vocab,
                                            tie_word_embeddings=cfg.tie_embeddings), max_model_len=256, n_slots=40, prefill_rows=256)
    e.load_state_dict(sd)
    assert e.prefill_rows == 256 and e.slot_len == 256
    g = torch.Generator().manual_seed(9)
    B = 40
    lens = [3 + (11 * i) % 58 for i in range(B)]
    reqs = [{"prompt_token_ids": torch.randint(0, cfg.vocab, (n,), generator=g).tolist()} for n in lens]
    forced = [torch.randint(0, cfg.vocab, (2,), generator=g).tolist() for _ in range(B)]
    sp = SamplingParams(max_tokens=2, min_tokens=2, ignore_eos=True)
    batch = e.generate_batch(reqs, sp, forced_output_ids=forced)
    batch = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in batch]
    torch.cuda.synchronize()
    for r, f, b in zip(reqs, forced, batch):
        a = e.generate(r["prompt_token_ids"], sp, forced_output_ids=f)
        assert b["prompt_hidden_states"].shape == a["prompt_hidden_states"].shape
        assert _rel(b["prompt_hidden_states"], a["prompt_hidden_states"]) < 5e-3 and _rel(b["hidden_states"], a["hidden_states"]) < 5e-3


def test_get_embed_batches_requests(hip):
    """get_embed over several requests with max_num_seqs > 1 runs them through generate_batch: same aligner inputs as one
    request at a time (teacher-forced)."""
    from oracle import aligner_ref as A
    from thinkdiff.models.mllama_vllm_t5_embed_decoder_2 import MllamaVllmT5EmbedDecoderForConditionalGeneration_5
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=9)
    asd = A.init_weights(cfg.hidden, 4096, seed=4)
    tc = Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
                           num_key_value_heads=cfg.num_kv_heads, intermediate_size=cfg.intermediate, vocab_size=cfg.vocab)
    outs = {}
    for nseq in (1, 4):
        m = MllamaVllmT5EmbedDecoderForConditionalGeneration_5(tc, vllm_config={"max_model_len": 256, "max_tokens": 8, "min_tokens": 8, "max_num_seqs": nseq})
        m.mllama.load_state_dict(sd)
        m.load_state_dict(asd)
        g = torch.Generator().manual_seed(1)
        reqs = [{"prompt_token_ids": torch.randint(0, cfg.vocab, (n,), generator=g).tolist()} for n in (20, 33, 7)]
        forced = [torch.randint(0, cfg.vocab, (8,), generator=g).tolist() for _ in reqs]
        outs[nseq] = m.get_embed(reqs, embedding_type="both", need_process=False, forced_output_ids=forced)
        torch.cuda.synchronize()
    for a, b, n in zip(outs[1][0], outs[4][0], (20, 33, 7)):
        assert a.shape == b.shape == (n + 8, 4096) and _rel(b, a) < 5e-3
    assert outs[1][1] == outs[4][1]


def test_full_width_qwen2_7b_layers(hip):
    """Qwen2-VL-7B layer shapes (hidden 3584, 28 q / 4 kv heads, MLP 18944), two layers: prefill of 300 tokens with an
    image-like M-RoPE block, then decode one sequence at a time (weight-stream kernel, M = 1) and 9 sequences per pass
    (matrix-core skinny GEMM, M = 9) against the oracle on the same teacher-forced ids."""
    from thinkdiff.models.qwen2_vl import SamplingParams
    cfg = Q.tiny_config(hidden=3584, num_layers=2, num_heads=28, num_kv_heads=4, intermediate=18944, vocab=4096)
    sd = Q.init_weights(cfg, seed=31)
    e = _engine(cfg, sd, max_len=512)
    g = torch.Generator().manual_seed(6)
    n, k = 300, 6
    ids = torch.randint(0, cfg.vocab, (n + k,), generator=g)
    pos = torch.stack([torch.arange(n + k), torch.arange(n + k) // 3 + 2, (torch.arange(n + k) * 2) % 11]).to(torch.int32)
    pos[:, n:] = pos[:, :n].max() + 1 + torch.arange(k, dtype=torch.int32)          # generation continues past the largest position
    ref, _ = Q.text_model_hidden(sd, cfg, pos, token_ids=ids)
    sp = SamplingParams(max_tokens=k, min_tokens=k, ignore_eos=True)
    one = e.generate(ids[:n].tolist(), sp, position_ids=pos[:, :n], forced_output_ids=ids[n:].tolist())
    torch.cuda.synchronize()
    e_pre, e_dec = _rel(one["prompt_hidden_states"], ref[:n]), _rel(one["hidden_states"], ref[n:])
    e2 = type(e)(e.config, max_model_len=512, n_slots=9)
    e2.load_state_dict(sd)
    reqs = [{"prompt_token_ids": ids[:n].tolist(), "position_ids": pos[:, :n]}] + \
           [{"prompt_token_ids": torch.randint(0, cfg.vocab, (40 + 7 * b,), generator=g).tolist()} for b in range(8)]
    forced = [ids[n:].tolist()] + [torch.randint(0, cfg.vocab, (k,), generator=g).tolist() for _ in range(8)]
    many = e2.generate_batch(reqs, sp, forced_output_ids=forced)
    torch.cuda.synchronize()
    e_bat = _rel(many[0]["hidden_states"], ref[n:])
    ref32, _ = Q.text_model_hidden({k_: v.float() for k_, v in sd.items()}, cfg, pos, token_ids=ids)
    e32, eref = _rel(torch.cat([one["prompt_hidden_states"], one["hidden_states"]]), ref32), _rel(ref, ref32)
    print(f"7B-width layers: prefill {e_pre:.4f}  decode M=1 {e_dec:.4f}  decode M=9 {e_bat:.4f}   hip~fp32 {e32:.4f}  bf16-oracle~fp32 {eref:.4f}")
    assert e_pre < 2e-2 and e_dec < 2e-2 and e_bat < 2e-2
    assert e32 < 1.5 * eref + 2e-3
    assert all(o["hidden_states"].shape == (k, 3584) for o in many)


@pytest.mark.parametrize("B", [40, 130])
def test_full_width_qwen2_7b_layers_wide_decode(hip, B):
    """The 7B layer shapes in the decode regime the tile kernels own (from 33 sequences on this width: split-K Linears on the 64- / 128- / 256-column
    tiles, the reduction launches that also normalise 3584-wide rows, 7 query heads per decode-attention workgroup): B sequences per step, the first
    one the oracle's 300-token image-like request, teacher-forced -- its decode states against the oracle and against the one-sequence path."""
    from thinkdiff.models.qwen2_vl import SamplingParams
    cfg = Q.tiny_config(hidden=3584, num_layers=2, num_heads=28, num_kv_heads=4, intermediate=18944, vocab=4096)
    sd = Q.init_weights(cfg, seed=31)
    g = torch.Generator().manual_seed(6)
    n, k = 300, 6
    ids = torch.randint(0, cfg.vocab, (n + k,), generator=g)
    pos = torch.stack([torch.arange(n + k), torch.arange(n + k) // 3 + 2, (torch.arange(n + k) * 2) % 11]).to(torch.int32)
    pos[:, n:] = pos[:, :n].max() + 1 + torch.arange(k, dtype=torch.int32)
    ref, _ = Q.text_model_hidden(sd, cfg, pos, token_ids=ids)
    sp = SamplingParams(max_tokens=k, min_tokens=k, ignore_eos=True)
    e = _engine(cfg, sd, max_len=B * 320)
    one = e.generate(ids[:n].tolist(), sp, position_ids=pos[:, :n], forced_output_ids=ids[n:].tolist())
    one = {k_: (v.clone() if torch.is_tensor(v) else v) for k_, v in one.items()}
    e.set_slots(B)
    assert e.slot_len == 320
    reqs = [{"prompt_token_ids": ids[:n].tolist(), "position_ids": pos[:, :n]}] + \
           [{"prompt_token_ids": torch.randint(0, cfg.vocab, (5 + (13 * b) % 90,), generator=g).tolist()} for b in range(B - 1)]
    forced = [ids[n:].tolist()] + [torch.randint(0, cfg.vocab, (k,), generator=g).tolist() for _ in range(B - 1)]
    many = e.generate_batch(reqs, sp, forced_output_ids=forced)
    torch.cuda.synchronize()
    e_bat, e_one = _rel(many[0]["hidden_states"], ref[n:]), _rel(many[0]["hidden_states"], one["hidden_states"])
    print(f"7B-width layers, {B} sequences per step: decode vs oracle {e_bat:.4f}, vs the one-sequence path {e_one:.4f}")
    assert e_bat < 2e-2 and e_one < 2e-2
    assert all(o["hidden_states"].shape == (k, 3584) and torch.isfinite(o["hidden_states"].float()).all() for o in many)


def test_packed_prefill_equals_padded_and_single(hip):
    """td_qwen2_prefill_packed (prompts back to back, causal attention per packed segment, k|v rows scattered to their slots) against the right-padded
    batched prefill and against one request at a time: prompt states, first sampled logits (through greedy continuation) and the cache (through
    teacher-forced decode steps).  Lengths from 1 token to several query tiles, one request with distinct M-RoPE streams, one with inputs_embeds,
    and a workspace small enough to force several passes."""
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine, SamplingParams
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=17)
    tc = Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                           intermediate_size=cfg.intermediate, vocab_size=cfg.vocab, tie_word_embeddings=cfg.tie_embeddings)
    g = torch.Generator().manual_seed(3)
    lens = [1, 300, 7, 64, 257, 33, 512, 5, 129, 90, 2, 256]
    reqs = [{"prompt_token_ids": torch.randint(0, cfg.vocab, (n,), generator=g).tolist()} for n in lens]
    reqs[4]["position_ids"] = torch.stack([torch.arange(257), torch.arange(257) // 3 + 2, (torch.arange(257) * 2) % 11]).to(torch.int32)
    forced = [torch.randint(0, cfg.vocab, (4,), generator=g).tolist() for _ in lens]
    sp = SamplingParams(max_tokens=4, min_tokens=4, ignore_eos=True)
    outs = {}
    for mode in ("single", "padded", "packed"):
        e = Qwen2VLTextEngine(tc, max_model_len=640, n_slots=len(lens), prefill_rows=1100)        # 1656 prompt rows: two packed passes, several padded ones
        e.load_state_dict(sd)
        if mode == "single":
            e.set_slots(1)
            res = [e.generate(r["prompt_token_ids"], sp, position_ids=r.get("position_ids"), forced_output_ids=f) for r, f in zip(reqs, forced)]
            res = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()} for o in res]
        else:
            e.packed_prefill = mode == "packed"
            rr = [dict(r) for r in reqs]
            if mode == "packed":
                rr[3]["inputs_embeds"] = e.embed_tokens(rr[3]["prompt_token_ids"])        # a request that brings its own embeddings (spliced vision rows)
            res = e.generate_batch(rr, sp, forced_output_ids=forced)
        torch.cuda.synchronize()
        outs[mode] = res
    for a, b, c, n in zip(outs["single"], outs["padded"], outs["packed"], lens):
        assert c["prompt_hidden_states"].shape == (n, cfg.hidden) and c["hidden_states"].shape == (4, cfg.hidden)
        assert torch.isfinite(c["prompt_hidden_states"].float()).all()
        assert _rel(c["prompt_hidden_states"], a["prompt_hidden_states"]) < 5e-3 and _rel(c["hidden_states"], a["hidden_states"]) < 5e-3
        assert _rel(c["prompt_hidden_states"], b["prompt_hidden_states"]) < 5e-3 and _rel(c["hidden_states"], b["hidden_states"]) < 5e-3
    # sampled continuation from the packed prefill's logits: greedy = the one-request path's tokens
    sp0 = SamplingParams(temperature=0.0, max_tokens=5, min_tokens=5, ignore_eos=True)
    e = Qwen2VLTextEngine(tc, max_model_len=640, n_slots=4, prefill_rows=1100)
    e.load_state_dict(sd)
    gb = e.generate_batch(reqs[1:5], sp0)
    e.set_slots(1)
    for r, o in zip(reqs[1:5], gb):
        ref = e.generate(r["prompt_token_ids"], sp0, position_ids=r.get("position_ids"))
        assert o["token_ids"] == ref["token_ids"]


def test_packed_prefill_from_token_ids_through_the_c_abi(hip):
    """td_qwen2_prefill_packed with token ids instead of embeddings (the form a C caller uses), one and several prompts, logits of the last tokens: against
    td_qwen2_forward_slot per prompt -- bit-equal (same kernels per row: the packed form only changes which rows share a launch)."""
    import ctypes
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=19)
    tc = Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                           intermediate_size=cfg.intermediate, vocab_size=cfg.vocab, tie_word_embeddings=cfg.tie_embeddings)
    e = Qwen2VLTextEngine(tc, max_model_len=256, n_slots=4, prefill_rows=1024)
    e.load_state_dict(sd)
    L = hip.lib()
    g = torch.Generator().manual_seed(2)
    for lens in ([70], [1, 255, 64, 33]):
        ids = [torch.randint(0, cfg.vocab, (n,), generator=g).to(torch.int32) for n in lens]
        total = sum(lens)
        tok = torch.cat(ids).cuda().contiguous()
        pos = torch.cat([e.text_position_ids(n) for n in lens], dim=1).to(torch.int32).cuda().contiguous()
        hid = torch.empty(total, cfg.hidden, dtype=torch.bfloat16, device="cuda")
        lg = torch.empty(len(lens), cfg.vocab, dtype=torch.bfloat16, device="cuda")
        cl = (ctypes.c_int * len(lens))(*lens)
        hip.check(L.td_qwen2_prefill_packed(e._h, 0, len(lens), hip.ptr(tok), None, hip.ptr(pos), ctypes.cast(cl, ctypes.c_void_p), hip.ptr(hid), hip.ptr(lg), hip.stream_ptr()))
        torch.cuda.synchronize()
        r = 0
        for b, n in enumerate(lens):
            h1, l1 = e.forward(e.text_position_ids(n), ids[b], want_hidden=True, want_logits=True, slot=b)
            torch.cuda.synchronize()
            assert _rel(hid[r:r + n], h1) < 5e-3 and _rel(lg[b], l1) < 5e-3
            r += n
    # refused: more rows than the workspace, a prompt longer than its slot
    big = (ctypes.c_int * 4)(255, 255, 255, 255)
    hip.check(L.td_qwen2_prefill_packed(e._h, 0, 4, hip.ptr(torch.zeros(1020, dtype=torch.int32, device="cuda")), None,
                                        hip.ptr(torch.zeros(3, 1020, dtype=torch.int32, device="cuda")), ctypes.cast(big, ctypes.c_void_p), None, None, hip.stream_ptr()))
    with pytest.raises(hip.ThinkDiffHipError):
        bad = (ctypes.c_int * 1)(300)
        hip.check(L.td_qwen2_prefill_packed(e._h, 0, 1, hip.ptr(tok), None, hip.ptr(pos), ctypes.cast(bad, ctypes.c_void_p), None, None, hip.stream_ptr()))


def test_continuous_batching_equals_one_request_at_a_time(hip):
    """generate_continuous: 45 requests against 8 cache slots -- a finished sequence's slot goes to a waiting request (compaction + one packed prefill
    pass into the free slots), requests arriving in chunks from an iterable.  Teacher-forced outputs of 0..14 tokens (so slots free up at different
    steps, several admissions, a request with nothing to generate), one request with distinct M-RoPE streams and one with its own embeddings:
    prompt states, output states and token ids against generate() per request; then sampled continuation: reproducible under a seed, every
    request gets max_tokens tokens, and greedy sampling equals the one-request path."""
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine, SamplingParams
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=23)
    tc = Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                           intermediate_size=cfg.intermediate, vocab_size=cfg.vocab, tie_word_embeddings=cfg.tie_embeddings)
    g = torch.Generator().manual_seed(12)
    N = 45
    lens = [1 + (17 * i) % 90 for i in range(N)]
    gens = [(5 * i + 3) % 15 for i in range(N)]
    gens[7] = 0
    reqs = [{"prompt_token_ids": torch.randint(0, cfg.vocab, (n,), generator=g).tolist()} for n in lens]
    reqs[11]["position_ids"] = torch.stack([torch.arange(lens[11]), torch.arange(lens[11]) // 3 + 2, (torch.arange(lens[11]) * 2) % 11]).to(torch.int32)
    forced = [torch.randint(0, cfg.vocab, (k,), generator=g).tolist() for k in gens]
    sp = SamplingParams(max_tokens=14, min_tokens=14, ignore_eos=True)
    e1 = Qwen2VLTextEngine(tc, max_model_len=128)
    e1.load_state_dict(sd)
    single = []
    for r, f in zip(reqs, forced):
        o = e1.generate(r["prompt_token_ids"], SamplingParams(max_tokens=max(len(f), 1), min_tokens=len(f), ignore_eos=True), position_ids=r.get("position_ids"),
                        forced_output_ids=f) if f else {"prompt_hidden_states": e1.forward(e1.text_position_ids(len(r["prompt_token_ids"])),
                                                                                         torch.tensor(r["prompt_token_ids"], dtype=torch.int32))[0],
                                                      "hidden_states": torch.empty(0, cfg.hidden, dtype=torch.bfloat16, device="cuda"), "token_ids": []}
        single.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in o.items()})
    e = Qwen2VLTextEngine(tc, max_model_len=128, n_slots=8, prefill_rows=300)
    e.load_state_dict(sd)
    rr = [dict(r) for r in reqs]
    rr[20]["inputs_embeds"] = e.embed_tokens(rr[20]["prompt_token_ids"])
    chunks = [rr[0:10], rr[10:11], rr[11:30], rr[30:45]]
    out = e.generate_continuous(iter(chunks), sp, forced_output_ids=forced, admit_min=2)
    torch.cuda.synchronize()
    assert len(out) == N
    for a, b, k, n in zip(single, out, gens, lens):
        assert b["token_ids"] == a["token_ids"] and b["hidden_states"].shape == (k, cfg.hidden) and b["prompt_hidden_states"].shape == (n, cfg.hidden)
        assert _rel(b["prompt_hidden_states"], a["prompt_hidden_states"]) < 5e-3
        if k:
            assert _rel(b["hidden_states"], a["hidden_states"]) < 5e-3
    # sampled: reproducible, complete, and (greedy) equal to the one-request path
    sps = SamplingParams(temperature=0.8, top_p=0.9, max_tokens=9, min_tokens=9, ignore_eos=True)
    runs = []
    for _ in range(2):
        gg = torch.Generator(device="cuda").manual_seed(5)
        o = e.generate_continuous([dict(r) for r in reqs[:30]], sps, generator=gg)
        runs.append([x["token_ids"] for x in o])
    assert runs[0] == runs[1] and all(len(t) == 9 for t in runs[0])
    # greedy: every sampled token is the arg-max of ITS sequence's logits -- checked against lm_head applied to the hidden state the call returned for
    # that sequence and step (a logits row attached to the wrong slot after a compaction or an admission would pick a token far below the maximum;
    # a one-ulp tie between the 8-row and the 1-row Linear kernels may not flip the test, hence "within a bf16 ulp of the maximum")
    W = (sd["model.embed_tokens.weight"] if cfg.tie_embeddings else sd["lm_head.weight"]).float().cuda()
    def near_argmax(state, tok):
        ref = state.float() @ W.t()
        return float(ref[tok]) >= float(ref.max()) - 2.0 ** -6 * float(ref.abs().max())
    sp0 = SamplingParams(temperature=0.0, max_tokens=12, min_tokens=1, ignore_eos=True, stop_token_ids=list(range(0, cfg.vocab, 5)))
    o = e.generate_continuous([dict(r) for r in reqs[:30]], sp0)
    n_tok = []
    for r, x in zip(reqs[:30], o):
        t = x["token_ids"]
        n_tok.append(len(t))
        assert 1 <= len(t) <= 12 and x["hidden_states"].shape[0] == len(t)
        assert all(tt % 5 != 0 for tt in t[:-1]) and (len(t) == 12 or t[-1] % 5 == 0)          # stopped exactly at its first stop token
        assert near_argmax(x["prompt_hidden_states"][-1], t[0])
        for k in range(len(t) - 1):
            assert near_argmax(x["hidden_states"][k], t[k + 1])
    assert len(set(n_tok)) > 2, "the stop tokens end the sequences at different steps (slots are re-used mid-flight)"


def test_move_slot_and_slot_indirect_decode_bit_equal(hip):
    """td_qwen2_move_slot + td_qwen2_decode_batch_slots: a sequence prefilled into slot 5 and moved to slot 1 decodes from there exactly as it does from
    slot 5 -- alone, and as one row of a step whose rows name scattered slots (6, 1, 3) in an order that is not the slot order."""
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=29)
    tc = Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads, num_key_value_heads=cfg.num_kv_heads,
                           intermediate_size=cfg.intermediate, vocab_size=cfg.vocab, tie_word_embeddings=cfg.tie_embeddings)
    e = Qwen2VLTextEngine(tc, max_model_len=8 * 96, n_slots=8)
    e.load_state_dict(sd)
    g = torch.Generator().manual_seed(4)
    lens = {5: 40, 6: 17, 3: 70}
    for slot, n in lens.items():
        e.forward(e.text_position_ids(n), torch.randint(0, cfg.vocab, (n,), generator=g).to(torch.int32), slot=slot)
    tok = torch.randint(0, cfg.vocab, (3,), generator=g).tolist()
    pos1 = torch.tensor([[40]] * 3, dtype=torch.int32)
    h5, l5 = e.decode_batch(tok[:1], pos1, [40], slots=[5])
    h5, l5 = h5.clone(), l5.clone()
    e.move_slot(5, 1, 40)                       # (the step above appended a row to slot 5; the first 40 rows are the prompt)
    h1, l1 = e.decode_batch(tok[:1], pos1, [40], slots=[1])
    torch.cuda.synchronize()
    assert torch.equal(h1, h5) and torch.equal(l1, l5)
    # three rows, scattered slots: row 1 is the moved sequence again (its slot holds 41 rows now; decode the same token at the same place)
    e.move_slot(5, 1, 40)
    posb = torch.tensor([[17, 40, 70]] * 3, dtype=torch.int32)
    hb, lb = e.decode_batch([tok[1], tok[0], tok[2]], posb, [17, 40, 70], slots=[6, 1, 3])
    torch.cuda.synchronize()
    assert _rel(hb[1:2], h5) < 2e-3 and torch.isfinite(hb.float()).all()      # (three rows take another Linear kernel than one: not bit-equal, equal to rounding)
    with pytest.raises(hip.ThinkDiffHipError):
        e.decode_batch(tok[:2], torch.tensor([[1, 1]] * 3, dtype=torch.int32), [1, 1], slots=[2, 9])      # slot 9 of 8
    with pytest.raises(hip.ThinkDiffHipError):
        e.decode_batch(tok[:2], torch.tensor([[17, 17]] * 3, dtype=torch.int32), [17, 17], slots=[6, 6])    # two rows on one slot

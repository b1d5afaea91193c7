"""Host logic of the continuous-batching scheduler (Qwen2VLTextEngine.generate_continuous) against a stand-in engine on the CPU: the two engine
calls (_prefill_packed_slots, _decode_slots) are replaced by a deterministic toy model that CHECKS the scheduler's bookkeeping on every call --
slots distinct and inside range, the cache length the scheduler reports for a slot equal to the number of tokens that slot really holds, positions
advancing by one -- and echoes (token, slot) into the hidden rows, so a row attached to the wrong request shows up in the output.  The GPU tests
(tests/test_qwen2_gpu.py::test_continuous_batching_equals_one_request_at_a_time) cover the same scheduler on the real engine."""
import ctypes
from types import SimpleNamespace

import numpy as np
import pytest
import torch

V, D = 97, 8


def _next(tok):      # the toy model's greedy continuation
    return (int(tok) * 7 + 3) % V


def _fake_engine(n_slots, slot_len=64, prefill_rows=96):
    from thinkdiff.models import qwen2_vl as Q

    class Fake(Q.Qwen2VLTextEngine):
        def __init__(self):      # no HIP handle: only what the scheduler touches
            self.config = SimpleNamespace(hidden_size=D, vocab_size=V)
            self.device = torch.device("cpu")
            self.n_slots, self.slot_len, self.prefill_rows = n_slots, slot_len, prefill_rows
            self.cache = {}          # slot -> tokens it holds
            self.max_rows = 0        # the fullest decode step seen
            self.prefills = 0

        def __del__(self):
            pass

        def embed_tokens(self, token_ids):
            out = torch.zeros(len(token_ids), D, dtype=torch.bfloat16)
            out[:, 0] = torch.tensor(list(token_ids), dtype=torch.float32)
            return out

        def forward(self, position_ids, token_ids=None, inputs_embeds=None, pos0=0, want_hidden=True, want_logits=False, slot=0):
            assert 0 <= slot < self.n_slots        # (a request with nothing to generate: prompt states through a free slot)
            return (self.embed_tokens(token_ids.tolist()) if inputs_embeds is None else inputs_embeds), None

        def _logits_for(self, tok, row):
            row.zero_()
            row[_next(tok)] = 8.0

        def _prefill_packed_slots(self, k, slots_c, emb, pos, lens_c, hid_out, logits_out):
            self.prefills += 1
            slots, lens = list(slots_c), list(lens_c)
            assert len(set(slots)) == k and all(0 <= s < self.n_slots for s in slots)
            assert sum(lens) == emb.shape[0] <= self.prefill_rows and pos.shape == (3, sum(lens))
            r = 0
            for b, (s, n) in enumerate(zip(slots, lens)):
                toks = [int(t) for t in emb[r:r + n, 0].float().tolist()]
                assert pos[0, r:r + n].tolist() == list(range(n))
                self.cache[s] = toks
                hid_out[r:r + n] = emb[r:r + n]
                self._logits_for(toks[-1], logits_out[b])
                r += n

        def _decode_slots(self, n, slots_np, tok_dev, pos_dev, cache_np, hid_out, logits_out):
            slots = [int(x) for x in slots_np[:n]]
            assert len(set(slots)) == n and all(s in self.cache for s in slots), "a decode row names a free or repeated slot"
            self.max_rows = max(self.max_rows, n)
            for i, s in enumerate(slots):
                assert int(cache_np[i]) == len(self.cache[s]) < self.slot_len, "cache length out of step with the slot's contents"
                assert int(pos_dev[0, i]) == len(self.cache[s]) and pos_dev[:, i].unique().numel() == 1
                t = int(tok_dev[i])
                self.cache[s].append(t)
                hid_out[i].zero_()
                hid_out[i, 0], hid_out[i, 1] = float(t), float(s)
                self._logits_for(t, logits_out[i])

    Q_ops = SimpleNamespace(sample_top_p=lambda logits, temperature, top_p, key, step: logits.float().argmax(dim=1).to(torch.int32))
    return Fake(), Q, Q_ops


def _run(e, Q, reqs, sp, forced=None, **kw):
    # (a prefill simply overwrites its slot in the toy engine; a slot handed out while its sequence is still decoding shows up as a cache length that no
    # longer matches -- asserted in _decode_slots -- and as foreign tokens in that request's echoed hidden rows)
    return e.generate_continuous(reqs, sp, forced_output_ids=forced, **kw)


@pytest.mark.parametrize("max_live,admit_min,chunks", [(4, 1, 1), (4, 2, 3), (8, 1, 5), (16, 4, 2), (3, 3, 40)])
def test_forced_outputs_reach_their_requests(max_live, admit_min, chunks):
    e, Q, _ = _fake_engine(n_slots=16)
    g = torch.Generator().manual_seed(max_live * 100 + chunks)
    N = 40
    lens = [1 + (11 * i) % 30 for i in range(N)]
    gens = [(5 * i + 2) % 13 for i in range(N)]
    reqs = [{"prompt_token_ids": torch.randint(0, V, (n,), generator=g).tolist()} for n in lens]
    forced = [torch.randint(0, V, (k,), generator=g).tolist() for k in gens]
    per = (N + chunks - 1) // chunks
    src = iter([reqs[i:i + per] for i in range(0, N, per)]) if chunks > 1 else reqs
    out = _run(e, Q, src, Q.SamplingParams(max_tokens=13, min_tokens=13), forced=forced, max_live=max_live, admit_min=admit_min)
    assert len(out) == N and e.max_rows <= max_live
    for r, f, o in zip(reqs, forced, out):
        assert o["token_ids"] == f
        assert o["prompt_hidden_states"][:, 0].float().tolist() == [float(t) for t in r["prompt_token_ids"]]
        assert o["hidden_states"].shape == (len(f), D) and o["hidden_states"][:, 0].float().tolist() == [float(t) for t in f]
    if max_live < N:
        assert e.max_rows == max_live, "the batch fills up whenever requests are waiting"


@pytest.mark.parametrize("max_live", [2, 5, 16])
def test_sampled_outputs_follow_each_sequence_and_stop_rules(max_live, monkeypatch):
    e, Q, ops = _fake_engine(n_slots=16)
    monkeypatch.setattr(Q, "_OPS", ops)
    monkeypatch.setattr(Q.Qwen2VLTextEngine, "_draw_sampler_key", staticmethod(lambda generator=None: 1))
    g = torch.Generator().manual_seed(3)
    N = 30
    reqs = [{"prompt_token_ids": torch.randint(0, V, (1 + (7 * i) % 20,), generator=g).tolist()} for i in range(N)]
    stops = list(range(0, V, 4))
    sp = Q.SamplingParams(temperature=0.0, max_tokens=10, min_tokens=3, ignore_eos=False, stop_token_ids=stops)
    out = _run(e, Q, reqs, sp, eos_token_id=5, max_live=max_live)
    for r, o in zip(reqs, out):
        exp, t = [], r["prompt_token_ids"][-1]
        while len(exp) < 10:
            t = _next(t)
            exp.append(t)
            if len(exp) >= 3 and (t in stops or t == 5):
                break
        assert o["token_ids"] == exp and o["hidden_states"][:, 0].float().tolist() == [float(x) for x in exp]


def test_refusals_and_empty_input():
    e, Q, _ = _fake_engine(n_slots=4, slot_len=16, prefill_rows=16)
    from thinkdiff import _hip
    assert e.generate_continuous([], Q.SamplingParams(max_tokens=2, min_tokens=2), forced_output_ids=[]) == []
    with pytest.raises(_hip.ThinkDiffHipError):
        e.generate_continuous([{"prompt_token_ids": list(range(17))}], Q.SamplingParams(max_tokens=2, min_tokens=2), forced_output_ids=[[1, 2]])
    with pytest.raises(_hip.ThinkDiffHipError):
        e.generate_continuous([{"prompt_token_ids": [1]}], Q.SamplingParams(max_tokens=2, min_tokens=2), forced_output_ids=[[1]], max_live=9)

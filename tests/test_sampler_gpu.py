"""td_sample_top_p_bf16 against the sort-based form it replaces (softmax -> sort -> cumsum -> mask -> multinomial, the
vLLM sampler's rule; reference thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:817-823 sets temperature 0.6 / top_p 0.9).

The kernel draws from its own counter-based stream, so parity is a distribution test on fixed logits:
  * support: every sampled token lies in the nucleus computed by torch in fp64 from the same bf16 logits
    (ties at the boundary value may resolve to either tied token);
  * law: the empirical frequencies of N = 65 536 draws sit at the total-variation distance sampling noise alone predicts
    from the exact renormalised nucleus distribution (<= 1.25 x E[TV] + 3e-3), a chi-square over 16 rank-ordered equal-mass
    bins shows no systematic bias (< 50 at 15 degrees of freedom), and the head token's frequency is within 5 sigma;
  * greedy (temperature 0) = first argmax; a row whose top token alone reaches top_p always returns it;
  * determinism: the same (seed, offset) gives the same tokens, another offset gives other tokens.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _nucleus(logits_bf16_row, temperature, top_p):
    x = logits_bf16_row.double().cpu() / temperature
    p = torch.softmax(x, dim=-1)
    sp, idx = torch.sort(p, descending=True, stable=True)
    keep = (torch.cumsum(sp, 0) - sp) < top_p
    kept = torch.where(keep, sp, torch.zeros_like(sp))
    q = torch.zeros_like(p)
    q[idx] = kept / kept.sum()
    return q, float(sp[keep].min())      # law, boundary probability


@pytest.mark.parametrize("vocab,scale,temperature,top_p", [(152064, 3.0, 0.6, 0.9), (151936, 2.0, 1.0, 0.5), (4096, 4.0, 0.6, 0.9), (152064, 3.0, 0.8, 1.0)])
def test_top_p_sampler_law(hip, vocab, scale, temperature, top_p):
    g = torch.Generator().manual_seed(vocab + int(10 * top_p))
    row = (torch.randn(vocab, generator=g) * scale).bfloat16()
    N = 65536 if top_p < 1.0 else 32768
    rows = 4096
    logits = row.cuda()[None].expand(rows, vocab).contiguous()
    draws = torch.cat([hip.sample_top_p(logits, temperature, top_p, seed=1234, offset=o) for o in range(N // rows)])
    torch.cuda.synchronize()
    assert draws.dtype == torch.int32 and int(draws.min()) >= 0 and int(draws.max()) < vocab
    q, p_edge = _nucleus(row, temperature, top_p)
    p_full = torch.softmax(row.double() / temperature, -1)
    counts = torch.bincount(draws.cpu().long(), minlength=vocab).double()
    # support: sampled tokens are in the nucleus, or tie with its boundary probability
    outside = (counts > 0) & (q == 0)
    assert bool(((p_full[outside] - p_edge).abs() <= 1e-12 * p_edge).all()), "a token outside the nucleus was sampled"
    # law: tied boundary tokens are interchangeable, so compare after pooling all tokens whose probability equals the edge value
    tie = (p_full - p_edge).abs() <= 1e-12 * p_edge
    emp = counts / counts.sum()
    tv = 0.5 * ((emp[~tie] - q[~tie]).abs().sum() + abs(float(emp[tie].sum() - q[tie].sum())))
    top = int(q.argmax())
    sigma = float((q[top] * (1 - q[top]) / N).sqrt())
    print(f"vocab {vocab} T {temperature} top_p {top_p}: nucleus {int((q > 0).sum())} tokens, TV {float(tv):.4f}, head freq {float(emp[top]):.4f} vs {float(q[top]):.4f}")
    # sampling noise alone gives E[TV] = 1/2 sum_i E|emp_i - q_i| ~ 1/2 sum_i sqrt(2 q_i (1 - q_i) / (pi N))
    tv_noise = float(0.5 * (2 * q * (1 - q) / (3.141592653589793 * N)).sqrt().sum())
    assert float(tv) < 1.25 * tv_noise + 3e-3, f"TV {float(tv):.4f} vs {tv_noise:.4f} expected from sampling noise"
    assert abs(float(emp[top] - q[top])) < 5 * sigma + 1e-4
    # systematic bias (a wrong nucleus boundary, a skewed draw) shows in coarse bins: 16 rank-ordered bins of equal mass
    order = torch.argsort(q, descending=True)
    cum = torch.cumsum(q[order], 0)
    bin_of = torch.clamp((cum * 16).floor().long(), max=15)
    exp_b = torch.zeros(16, dtype=torch.float64).index_add_(0, bin_of, q[order]) * counts.sum()
    # tokens outside the law's support are boundary ties, interchangeable with the tied tokens inside it: they count in the last bin
    inside = counts[order] * (q[order] > 0)
    obs_b = torch.zeros(16, dtype=torch.float64).index_add_(0, bin_of, inside)
    obs_b[bin_of[int((q[order] > 0).sum()) - 1]] += counts[q == 0].sum()
    ok = exp_b > 0
    chi2 = float((((obs_b - exp_b) ** 2)[ok] / exp_b[ok]).sum())
    print(f"   expected TV from noise {tv_noise:.4f}; chi2 over {int(ok.sum())} equal-mass bins {chi2:.1f}")
    assert chi2 < 50.0                                         # 15 degrees of freedom: P(chi2 > 50) ~ 1e-5


def test_top_p_sampler_edge_cases(hip):
    vocab = 152064
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(3, vocab, generator=g) * 2).bfloat16()
    x[1, 777] = 30.0                                 # one dominant token: p > 0.9 on its own
    x[2, 100] = x[2, 90000] = x[2].max() + 1         # two tied maxima: greedy takes the first
    xd = x.cuda()
    greedy = hip.sample_top_p(xd, 0.0, 0.9, seed=1, offset=0).cpu()
    assert greedy.tolist() == [int(x[0].float().argmax()), 777, 100]
    for off in range(8):
        assert int(hip.sample_top_p(xd[1:2], 0.6, 0.9, seed=9, offset=off)) == 777
    a = hip.sample_top_p(xd[:1].expand(512, vocab).contiguous(), 1.0, 0.95, seed=7, offset=3).cpu()
    b = hip.sample_top_p(xd[:1].expand(512, vocab).contiguous(), 1.0, 0.95, seed=7, offset=3).cpu()
    c = hip.sample_top_p(xd[:1].expand(512, vocab).contiguous(), 1.0, 0.95, seed=7, offset=4).cpu()
    assert torch.equal(a, b) and not torch.equal(a, c) and len(set(a.tolist())) > 50
    # -inf logits (masked vocabulary) never come out; a single finite logit is always taken
    y = torch.full((1, 1024), float("-inf")).bfloat16()
    y[0, 513] = -3.0
    assert int(hip.sample_top_p(y.cuda(), 0.6, 0.9, seed=1, offset=0)) == 513
    with pytest.raises(hip.ThinkDiffHipError):
        hip.sample_top_p(torch.zeros(1, 1001, dtype=torch.bfloat16, device="cuda"), 0.6, 0.9, 1, 0)


def test_generate_is_reproducible_under_torch_seed(hip):
    """The sampler's stream is keyed from torch's generator: same manual_seed -> same sampled continuation."""
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine, SamplingParams
    eng = Qwen2VLTextEngine(Qwen2VLTextConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                                              intermediate_size=1024, vocab_size=4096), max_model_len=256).init_random(3, std=0.05)
    sp = SamplingParams(temperature=0.6, top_p=0.9, max_tokens=12, min_tokens=12)
    outs = []
    for seed in (11, 11, 12):
        torch.manual_seed(seed)
        outs.append(eng.generate(list(range(5, 25)), sp)["token_ids"])
    assert outs[0] == outs[1] and outs[0] != outs[2] and len(outs[0]) == 12
    g = torch.Generator().manual_seed(99)
    a = eng.generate(list(range(5, 25)), sp, generator=g)["token_ids"]
    b = eng.generate(list(range(5, 25)), sp, generator=torch.Generator().manual_seed(99))["token_ids"]
    assert a == b


def test_infinite_and_degenerate_rows_always_yield_a_token(hip):
    """A +inf logit holds all the mass: the sampler must return it (the first one) whatever temperature / top_p say, and rows of
    -inf / NaN must still write a valid id (such rows left out[row] unwritten and could index LDS out of range)."""
    vocab = 4096
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, vocab, generator=g).bfloat16()
    x[0, 777] = float("inf")
    x[1, 12] = float("inf"); x[1, 3000] = float("inf")
    x[2, :] = float("-inf")
    x[3, :] = float("nan")
    x[4, :] = float("-inf"); x[4, 99] = 1.5
    out = torch.full((6,), -7, dtype=torch.int32, device="cuda")
    for o in range(4):
        ids = hip.sample_top_p(x.cuda(), 0.6, 0.9, seed=3, offset=o)
        torch.cuda.synchronize()
        ids = ids.cpu()
        assert int(ids[0]) == 777 and int(ids[1]) == 12 and int(ids[4]) == 99
        assert all(0 <= int(i) < vocab for i in ids), ids
    del out

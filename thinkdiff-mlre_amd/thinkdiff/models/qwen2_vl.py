"""Qwen2-VL text decoder on the HIP engine (`td_qwen2_*`): hidden states at `model.norm` and sampling.

Stands in for the `vllm.LLM(model="Qwen/Qwen2-VL-*", return_hidden_states=True)` object of the reference
(thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py:790-816; thinkdiff/models/mllama_vllm_generate_1.py:382-413)
for the part of its behaviour the ThinkDiff path uses: `generate()` returning, per request, the prompt tokens'
and the generated tokens' hidden states at `embedding_layer_name="model.norm"` plus the generated token ids.
Image inputs: `vision_towers.HipQwen2VisionTransformer` produces the merged vision tokens; `expand_image_placeholders`,
`embed_tokens` and `mrope_position_ids` below do what vLLM's input processor and `get_rope_index` do around the decoder.
"""
import ctypes
import dataclasses
from typing import Dict, List, Optional, Sequence

import torch

from .. import _hip
from ..ops import register as _register_ops
_OPS = _register_ops()      # torch.ops.thinkdiff_hip: the custom-op layer over the C ABI (GPU kernels only, no fallback)


@dataclasses.dataclass
class Qwen2VLTextConfig:
    """Text-side keys of [ext] Qwen/Qwen2-VL-7B-Instruct config.json (2B: 1536 / 12 / 2 / 8960 / 151936 / tied)."""
    hidden_size: int = 3584
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    intermediate_size: int = 18944
    vocab_size: int = 152064
    tie_word_embeddings: bool = False
    mrope_section: Sequence[int] = (16, 24, 24)
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6


@dataclasses.dataclass
class SamplingParams:
    """The fields of vllm.SamplingParams the reference sets (mllama_vllm_t5_embed_decoder_2.py:817-823)."""
    temperature: float = 0.6
    top_p: float = 0.9
    max_tokens: int = 128
    min_tokens: int = 128
    ignore_eos: bool = True
    stop_token_ids: Optional[List[int]] = None


class Qwen2VLTextEngine:
    dtype = torch.bfloat16

    def __init__(self, config: Optional[Qwen2VLTextConfig] = None, max_model_len: int = 8192, device="cuda", n_slots: int = 1,
                 prefill_rows: Optional[int] = None, **kw):
        self.config = config or Qwen2VLTextConfig(**kw)
        c = self.config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _hip.ThinkDiffHipError("Qwen2VLTextEngine runs on the MI355X HIP engine only (device='cuda')")
        self._L = _hip.lib()
        cc = _hip.TdQwen2Config(c.hidden_size, c.num_hidden_layers, c.num_attention_heads, c.num_key_value_heads, 128,
                                c.intermediate_size, c.vocab_size, int(c.tie_word_embeddings),
                                (ctypes.c_int * 3)(*c.mrope_section), c.rms_norm_eps, c.rope_theta)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            # n_slots sequences of max_model_len tokens each (batched decode); n_slots = 1 is the single-request engine
            # prefill_rows: activation-workspace rows = the capacity (sum of padded prompt lengths) of one batched prefill
            self.prefill_rows = max(int(prefill_rows or 0), max_model_len)
            _hip.check(self._L.td_qwen2_create_ex(ctypes.byref(cc), max_model_len, int(n_slots), self.prefill_rows, ctypes.byref(h)))
        self._h = h
        self.n_slots, self.slot_len = int(n_slots), max_model_len
        self.max_model_len = max_model_len

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.td_qwen2_destroy(h)

    # ---- parameters (Hugging Face names) -----------------------------------------------------------------
    def param_table(self) -> Dict[str, int]:
        buf, cnt, out = ctypes.create_string_buffer(256), ctypes.c_int64(), {}
        for i in range(self._L.td_qwen2_num_params(self._h)):
            _hip.check(self._L.td_qwen2_param_info(self._h, i, buf, 256, ctypes.byref(cnt)))
            out[buf.value.decode()] = cnt.value
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        table = self.param_table()
        missing = [k for k in table if k not in sd]
        if strict and missing:
            raise KeyError(f"Qwen2VLTextEngine.load_state_dict: missing {missing[:4]}..")
        for name, t in sd.items():
            if name in table:
                d = t.to(device=self.device, dtype=torch.bfloat16).contiguous()
                _hip.check(self._L.td_qwen2_load_param(self._h, name.encode(), _hip.ptr(d), d.numel(), _hip.stream_ptr()))
                torch.cuda.current_stream().synchronize()
        return missing

    def load_pretrained(self, path: str):
        """Text-decoder tensors of a LOCAL Hugging Face Qwen2-VL checkpoint directory (*.safetensors; `visual.*` skipped)."""
        import glob
        import os
        from safetensors import safe_open
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise FileNotFoundError(f"no *.safetensors under {path}")
        seen = set()
        for fn in files:
            with safe_open(fn, framework="pt") as fh:
                sd = {}
                for k in fh.keys():
                    if "visual." in k:
                        continue
                    name = k.replace("model.language_model.", "model.")     # newer exports nest the decoder
                    sd[name] = fh.get_tensor(k)
                seen.update(sd)
                self.load_state_dict(sd, strict=False)
        missing = [k for k in self.param_table() if k not in seen and not (k == "lm_head.weight" and self.config.tie_word_embeddings)]
        if missing:
            raise KeyError(f"checkpoint at {path} lacks {len(missing)} decoder tensors, e.g. {missing[:3]}")
        return self

    def init_random(self, seed: int = 0, std: float = 0.02):
        _hip.check(self._L.td_qwen2_init_random(self._h, seed, std, _hip.stream_ptr()))
        return self

    # ---- one decoder pass over n new tokens ---------------------------------------------------------------
    def set_slots(self, n_slots: int):
        """Split the KV cache into `n_slots` sequences (each max_model_len // n_slots tokens) for batched decode."""
        _hip.check(self._L.td_qwen2_set_slots(self._h, int(n_slots)))
        self.n_slots, self.slot_len = int(n_slots), int(self._L.td_qwen2_slot_capacity(self._h))
        return self

    def set_fused_rope(self, on: bool) -> bool:
        """Decode step: rotary embedding + cache write inside the attention launch (default) or as a launch of their own.  Returns the
        previous setting."""
        return bool(self._L.td_qwen2_set_fused_rope(self._h, 1 if on else 0))

    def forward(self, position_ids, token_ids=None, inputs_embeds=None, pos0: int = 0, want_hidden=True, want_logits=False, slot: int = 0):
        """position_ids int32 [3,n]; token_ids int32 [n] or inputs_embeds bf16 [n,hidden].  Returns
        (hidden [n,hidden] | None, logits_last [vocab] | None)."""
        pos = position_ids.to(self.device, torch.int32).contiguous()
        n = pos.shape[1]
        tok = None if token_ids is None else token_ids.to(self.device, torch.int32).contiguous()
        emb = None if inputs_embeds is None else inputs_embeds.to(self.device, torch.bfloat16).contiguous()
        hid = torch.empty(n, self.config.hidden_size, dtype=torch.bfloat16, device=self.device) if want_hidden else None
        lg = torch.empty(self.config.vocab_size, dtype=torch.bfloat16, device=self.device) if want_logits else None
        _hip.check(self._L.td_qwen2_forward_slot(self._h, slot, _hip.ptr(tok), _hip.ptr(emb), _hip.ptr(pos), n, pos0,
                                                 _hip.ptr(hid), _hip.ptr(lg), _hip.stream_ptr()))
        return hid, lg

    def decode_batch(self, token_ids, position_ids, cache_pos: Sequence[int], want_logits=True, slots: Optional[Sequence[int]] = None):
        """One token for each of the sequences in cache slots `slots` (default 0..B-1) in a single pass over the weights.
        token_ids [B], position_ids [3,B], cache_pos[b] = tokens already cached.  -> (hidden [B,hidden], logits [B,vocab] | None)."""
        B = len(cache_pos)
        tok = torch.as_tensor(token_ids, dtype=torch.int32).to(self.device).contiguous()
        pos = torch.as_tensor(position_ids, dtype=torch.int32).to(self.device).contiguous()
        assert tok.shape == (B,) and pos.shape == (3, B)
        hid = torch.empty(B, self.config.hidden_size, dtype=torch.bfloat16, device=self.device)
        lg = torch.empty(B, self.config.vocab_size, dtype=torch.bfloat16, device=self.device) if want_logits else None
        cp = (ctypes.c_int * B)(*[int(c) for c in cache_pos])
        sl = ctypes.cast((ctypes.c_int * B)(*[int(c) for c in slots]), ctypes.c_void_p) if slots is not None else None
        _hip.check(self._L.td_qwen2_decode_batch_slots(self._h, B, sl, _hip.ptr(tok), _hip.ptr(pos), ctypes.cast(cp, ctypes.c_void_p),
                                                       _hip.ptr(hid), _hip.ptr(lg), _hip.stream_ptr()))
        return hid, lg

    # ---- multimodal prompt assembly ([ext] vLLM Qwen2-VL input processor + transformers Qwen2VLModel.get_rope_index) ----
    def embed_tokens(self, token_ids) -> torch.Tensor:
        """[n] ids -> bf16 [n, hidden] rows of model.embed_tokens."""
        tok = torch.as_tensor(token_ids, dtype=torch.int32).to(self.device).contiguous()
        out = torch.empty(tok.numel(), self.config.hidden_size, dtype=torch.bfloat16, device=self.device)
        _hip.check(self._L.td_qwen2_embed_tokens(self._h, _hip.ptr(tok), _hip.ptr(out), tok.numel(), _hip.stream_ptr()))
        return out

    @staticmethod
    def expand_image_placeholders(token_ids: Sequence[int], grid_thw, merge: int = 2, image_token_id: int = 151655) -> List[int]:
        """Each `<|image_pad|>` in the templated prompt stands for one image: repeat it t*h*w/merge^2 times (one per
        merged vision token).  Prompts whose placeholders are already expanded pass through unchanged."""
        counts = [int(t) * int(h) * int(w) // (merge * merge) for t, h, w in grid_thw]
        ids = list(token_ids)
        if sum(1 for v in ids if v == image_token_id) == sum(counts):
            return ids
        if sum(1 for v in ids if v == image_token_id) != len(counts):
            raise ValueError(f"prompt holds {sum(1 for v in ids if v == image_token_id)} image placeholders for {len(counts)} images")
        out, k = [], 0
        for v in ids:
            if v == image_token_id:
                out.extend([v] * counts[k])
                k += 1
            else:
                out.append(v)
        return out

    @staticmethod
    def mrope_position_ids(token_ids: Sequence[int], grid_thw, merge: int = 2, image_token_id: int = 151655) -> torch.Tensor:
        """int32 [3, n] (t, h, w) streams: text runs count up on all three; an image block keeps t fixed and walks
        its merged (h/merge, w/merge) grid; the text after it resumes at max + 1."""
        ids = list(token_ids)
        pos = torch.zeros(3, len(ids), dtype=torch.int32)
        i, nxt, img = 0, 0, 0
        while i < len(ids):
            if ids[i] == image_token_id:
                t, h, w = (int(v) for v in grid_thw[img])
                gh, gw = h // merge, w // merge
                n = t * gh * gw
                pos[0, i:i + n] = nxt + torch.arange(t, dtype=torch.int32).repeat_interleave(gh * gw)
                pos[1, i:i + n] = nxt + torch.arange(gh, dtype=torch.int32).repeat_interleave(gw).repeat(t)
                pos[2, i:i + n] = nxt + torch.arange(gw, dtype=torch.int32).repeat(t * gh)
                nxt += max(t, gh, gw)
                i += n
                img += 1
            else:
                pos[:, i] = nxt
                nxt += 1
                i += 1
        return pos

    @staticmethod
    def text_position_ids(n: int, start: int = 0) -> torch.Tensor:
        return (torch.arange(n, dtype=torch.int32) + start)[None, :].expand(3, n).contiguous()

    # ---- the part of LLM.generate(return_hidden_states=True) the reference consumes ------------------------
    @torch.no_grad()
    def generate(self, prompt_token_ids: Sequence[int], sampling: SamplingParams, position_ids=None, inputs_embeds=None,
                 eos_token_id: Optional[int] = None, generator: Optional[torch.Generator] = None,
                 forced_output_ids: Optional[Sequence[int]] = None):
        """Returns dict(prompt_hidden_states [n_prompt,D], hidden_states [n_gen,D], token_ids [n_gen]).

        Sampling is temperature / top-p on the device.  `forced_output_ids` teacher-forces the continuation
        (the reference samples at T=0.6, so parity of the hidden states is checked on forced ids)."""
        n_p = len(prompt_token_ids)
        pos = self.text_position_ids(n_p) if position_ids is None else position_ids
        next_pos = int(pos.max()) + 1      # M-RoPE: generation continues one past the largest prompt position
        tok = torch.tensor(list(prompt_token_ids), dtype=torch.int32)
        prompt_hidden, logits = self.forward(pos, tok if inputs_embeds is None else None, inputs_embeds, 0, True, True)
        out_tok, out_hidden = [], []
        stops = set(sampling.stop_token_ids or [])
        can_stop = forced_output_ids is None and (bool(stops) or ((not sampling.ignore_eos) and eos_token_id is not None))
        key = None if forced_output_ids is not None else self._draw_sampler_key(generator)
        if forced_output_ids is not None:
            forced_dev = torch.tensor(list(forced_output_ids), dtype=torch.int32).to(self.device)
        for step in range(sampling.max_tokens):
            if forced_output_ids is not None:
                if step >= len(forced_output_ids):
                    break
                nxt = forced_dev[step:step + 1]
            else:
                nxt = _OPS.sample_top_p(logits, float(sampling.temperature), float(sampling.top_p), int(key), int(step))     # device int32 [1], no host round trip
            out_tok.append(nxt)
            # one token against the cache: the decode step (fused rope + cache write, decode attention, gated-MLP weight stream)
            h1, lg = self.decode_batch(nxt, [[next_pos + step]] * 3, [n_p + step])
            logits = lg[0]
            out_hidden.append(h1)
            if can_stop and step + 1 >= sampling.min_tokens:      # only now does the host need to see the token
                t = int(nxt)
                if ((not sampling.ignore_eos) and eos_token_id is not None and t == eos_token_id) or t in stops:
                    break
        hs = torch.cat(out_hidden) if out_hidden else torch.empty(0, self.config.hidden_size, dtype=torch.bfloat16, device=self.device)
        out_ids = torch.cat(out_tok).tolist() if out_tok else []
        return {"prompt_hidden_states": prompt_hidden, "hidden_states": hs, "token_ids": out_ids}

    @staticmethod
    def _draw_sampler_key(generator: Optional[torch.Generator] = None) -> int:
        """64-bit key of one generate() call's sampling stream, drawn from `generator` (default: torch's global CPU generator),
        so `torch.manual_seed` / a seeded generator reproduce the sampled tokens and successive calls differ.  The HIP sampler
        derives row r's draw for generated token t from (key, t, r) (td_sample_top_p_bf16)."""
        dev = generator.device if generator is not None else "cpu"
        return int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=dev))

    packed_prefill = True   # generate_batch: prompts back to back (td_qwen2_prefill_packed); False = right-padded to the longest (td_qwen2_prefill_batch_at: A/B, tests)
    MAX_BATCH = 256     # td_qwen2_decode_batch: sequences sharing one decode step (vLLM's max_num_seqs of the precompute job)

    @torch.no_grad()
    def generate_batch(self, requests: Sequence[dict], sampling: SamplingParams, eos_token_id: Optional[int] = None,
                       generator: Optional[torch.Generator] = None, forced_output_ids: Optional[Sequence[Sequence[int]]] = None):
        """`generate` for up to min(256, n_slots) requests together: each prompt is prefilled into its own cache slot, then
        every decode step advances all unfinished sequences in one pass over the weights (what vLLM's batching gives the
        reference's precompute job).  requests: dicts with "prompt_token_ids" and optional "position_ids" / "inputs_embeds".
        Returns one generate()-style dict per request, in order."""
        B = len(requests)
        if B > min(self.MAX_BATCH, getattr(self, "n_slots", 1)):
            raise _hip.ThinkDiffHipError(f"generate_batch: {B} requests exceed min({self.MAX_BATCH}, n_slots={getattr(self, 'n_slots', 1)}); call set_slots first")
        res = [{"prompt_hidden_states": None, "hidden_states": [], "token_ids": []} for _ in range(B)]
        cache_len = [len(r["prompt_token_ids"]) for r in requests]
        poss = [self.text_position_ids(n) if r.get("position_ids") is None else r["position_ids"] for r, n in zip(requests, cache_len)]
        next_pos = [int(p.max()) + 1 for p in poss]
        L = (max(cache_len) + 7) // 8 * 8
        if B > 1 and self.packed_prefill and max(cache_len) <= min(self.slot_len, self.prefill_rows):
            # packed prefill (td_qwen2_prefill_packed): the prompts of a pass lie back to back, no padding rows -- vLLM's batching of the
            # reference's requests; a pass takes as many prompts as the activation workspace (max_num_batched_tokens) holds
            D = self.config.hidden_size
            logits = torch.empty(B, self.config.vocab_size, dtype=torch.bfloat16, device=self.device)
            b0 = 0
            while b0 < B:
                nb, rows = 0, 0
                while b0 + nb < B and nb < self.MAX_BATCH and rows + cache_len[b0 + nb] <= self.prefill_rows:
                    rows += cache_len[b0 + nb]
                    nb += 1
                emb = torch.empty(rows, D, dtype=torch.bfloat16, device=self.device)
                pos = torch.empty(3, rows, dtype=torch.int32)
                r = 0
                for j in range(nb):
                    q, n, pp = requests[b0 + j], cache_len[b0 + j], poss[b0 + j]
                    e = q.get("inputs_embeds")
                    emb[r:r + n] = self.embed_tokens(q["prompt_token_ids"]) if e is None else e.to(self.device, torch.bfloat16)
                    pos[:, r:r + n] = pp.to(torch.int32)
                    r += n
                pos = pos.to(self.device).contiguous()
                hid = torch.empty(rows, D, dtype=torch.bfloat16, device=self.device)
                lens = (ctypes.c_int * nb)(*cache_len[b0:b0 + nb])
                _hip.check(self._L.td_qwen2_prefill_packed(self._h, b0, nb, None, _hip.ptr(emb), _hip.ptr(pos), ctypes.cast(lens, ctypes.c_void_p),
                                                           _hip.ptr(hid), _hip.ptr(logits[b0:b0 + nb]), _hip.stream_ptr()))
                r = 0
                for j in range(nb):
                    res[b0 + j]["prompt_hidden_states"] = hid[r:r + cache_len[b0 + j]]
                    r += cache_len[b0 + j]
                b0 += nb
        elif B > 1 and L <= self.slot_len and 2 * L <= self.prefill_rows:
            # one pass over the weights for as many prompts as the activation workspace holds (all of them, normally): right-padded
            # to L rows each (causal attention keeps the padding inert); a larger request batch takes several such passes
            D = self.config.hidden_size
            per = min(B, self.prefill_rows // L)
            logits = torch.empty(B, self.config.vocab_size, dtype=torch.bfloat16, device=self.device)
            for b0 in range(0, B, per):
                nb = min(per, B - b0)
                emb = torch.zeros(nb * L, D, dtype=torch.bfloat16, device=self.device)
                pos = torch.zeros(3, nb * L, dtype=torch.int32)
                for j in range(nb):
                    r, n, p = requests[b0 + j], cache_len[b0 + j], poss[b0 + j]
                    e = r.get("inputs_embeds")
                    emb[j * L:j * L + n] = self.embed_tokens(r["prompt_token_ids"]) if e is None else e.to(self.device, torch.bfloat16)
                    pos[:, j * L:j * L + n] = p.to(torch.int32)
                    pos[:, j * L + n:(j + 1) * L] = next_pos[b0 + j] + torch.arange(L - n, dtype=torch.int32)
                pos = pos.to(self.device).contiguous()
                hid = torch.empty(nb * L, D, dtype=torch.bfloat16, device=self.device)
                lens = (ctypes.c_int * nb)(*cache_len[b0:b0 + nb])
                _hip.check(self._L.td_qwen2_prefill_batch_at(self._h, b0, nb, L, None, _hip.ptr(emb), _hip.ptr(pos), ctypes.cast(lens, ctypes.c_void_p),
                                                             _hip.ptr(hid), _hip.ptr(logits[b0:b0 + nb]), _hip.stream_ptr()))
                for j in range(nb):
                    res[b0 + j]["prompt_hidden_states"] = hid[j * L:j * L + cache_len[b0 + j]]
        else:
            rows = []
            for b, (r, p) in enumerate(zip(requests, poss)):
                e = r.get("inputs_embeds")
                hid, lg = self.forward(p, torch.tensor(list(r["prompt_token_ids"]), dtype=torch.int32) if e is None else None, e, 0, True, True, slot=b)
                res[b]["prompt_hidden_states"] = hid
                rows.append(lg)
            logits = torch.stack(rows)              # [B, vocab], row i belongs to the sequence in slot i
        owner = list(range(B))                      # live row -> request index
        slots = list(range(B))                      # live row -> cache slot (a finished sequence just leaves the lists: no cache rows move)
        key = None if forced_output_ids is not None else self._draw_sampler_key(generator)
        stops = set(sampling.stop_token_ids or [])
        step = 0
        # per step: the hidden rows stay ONE device tensor and the slot -> request map is remembered; the rows are handed to their requests by a
        # single gather after the loop (256 sequences x 256 steps of per-row tensor views and list appends were ~0.5 ms of host time per step,
        # during which the GPU idled: 15 % of a 3.3 ms step)
        step_hid, step_owner, step_toks = [], [], []
        eos_on = (not sampling.ignore_eos) and eos_token_id is not None
        while owner and step < sampling.max_tokens:
            n = len(owner)
            if forced_output_ids is not None:
                live = [i for i in range(n) if step < len(forced_output_ids[owner[i]])]
                if len(live) < n:                   # forced continuations of different lengths: retire the exhausted ones first
                    owner, slots = [owner[i] for i in live], [slots[i] for i in live]
                    cache_len, next_pos = [cache_len[i] for i in live], [next_pos[i] for i in live]
                    logits = logits[live]
                    if not owner:
                        break
                    n = len(owner)
                toks = [int(forced_output_ids[owner[i]][step]) for i in range(n)]
            else:
                # one launch for all live rows (td_sample_top_p_bf16); the ids come to the host once per step for the bookkeeping below
                toks = _OPS.sample_top_p(logits[:n], float(sampling.temperature), float(sampling.top_p), int(key), int(step)).tolist()
            pos = torch.tensor(next_pos[:n], dtype=torch.int32).unsqueeze(0).expand(3, n)
            hid, logits = self.decode_batch(toks, pos, cache_len[:n], slots=slots)
            step_hid.append(hid)
            step_owner.append(list(owner))
            step_toks.append(toks)
            may_stop = forced_output_ids is None and step + 1 >= sampling.min_tokens
            keep = []
            for i in range(n):
                cache_len[i] += 1
                next_pos[i] += 1
                stop = may_stop and ((eos_on and toks[i] == eos_token_id) or toks[i] in stops)
                if not stop and cache_len[i] < self.slot_len:
                    keep.append(i)
            if len(keep) < n:
                owner, slots = [owner[i] for i in keep], [slots[i] for i in keep]
                cache_len, next_pos = [cache_len[i] for i in keep], [next_pos[i] for i in keep]
                logits = logits[keep]
            step += 1
        rows = [[] for _ in range(B)]
        base = 0
        for owners, toks in zip(step_owner, step_toks):
            for i, o in enumerate(owners):
                rows[o].append(base + i)
                res[o]["token_ids"].append(toks[i])
            base += len(owners)
        D = self.config.hidden_size
        if base:
            allh = torch.cat(step_hid)
            idx = torch.tensor([r for rr in rows for r in rr], dtype=torch.int64).to(self.device)
            parts = torch.split(allh.index_select(0, idx), [len(rr) for rr in rows])
        for b, r in enumerate(res):
            r["hidden_states"] = parts[b] if base and rows[b] else torch.empty(0, D, dtype=torch.bfloat16, device=self.device)
        return res

    @torch.no_grad()
    def generate_continuous(self, request_chunks, sampling: SamplingParams, eos_token_id: Optional[int] = None,
                            generator: Optional[torch.Generator] = None, forced_output_ids: Optional[Sequence[Sequence[int]]] = None,
                            max_live: Optional[int] = None, admit_min: Optional[int] = None):
        """Continuous batching ([ext] vLLM's scheduler behind the reference's `LLM.generate` of a whole loader batch, thinkdiff/models/
        mllama_vllm_generate_1.py:585 with `max_num_seqs: 256`): any number of requests against `max_live` (default min(256, n_slots)) cache slots.
        A sequence that finishes frees its slot at once -- nothing moves: a decode step names the slot of each of its rows
        (td_qwen2_decode_batch_slots) -- and waiting requests are prefilled into the free slots (one packed pass, td_qwen2_prefill_packed_slots)
        as soon as `admit_min` of them fit (default max_live / 8: a
        prefill pass interrupts the decode steps, so it should be worth a pass), then decode with everybody else.  With outputs of different
        lengths the decode steps stay full, where generate_batch per chunk of max_live runs every chunk down to its longest sequence.
        `request_chunks`: an iterable of request lists -- pulled only when the waiting queue runs low, so a producer (the precompute model's helper
        thread) can build later chunks while earlier ones decode -- or one flat list.  Requests are numbered in arrival order; returns one
        generate()-style dict per request in that order.  forced_output_ids[i]: teacher-forced continuation of request i."""
        from collections import deque
        import numpy as np
        max_live = min(self.MAX_BATCH, getattr(self, "n_slots", 1)) if max_live is None else int(max_live)
        if max_live < 1 or max_live > min(self.MAX_BATCH, getattr(self, "n_slots", 1)):
            raise _hip.ThinkDiffHipError(f"generate_continuous: max_live={max_live} exceeds min({self.MAX_BATCH}, n_slots={getattr(self, 'n_slots', 1)}); call set_slots first")
        admit_min = max(1, max_live // 8) if admit_min is None else max(1, int(admit_min))
        if isinstance(request_chunks, (list, tuple)) and (not request_chunks or isinstance(request_chunks[0], dict)):
            request_chunks = [list(request_chunks)]
        chunk_it = iter(request_chunks)
        D, V = self.config.hidden_size, self.config.vocab_size
        requests, res, waiting = [], [], deque()
        exhausted = False

        def pull():
            nonlocal exhausted
            try:
                chunk = next(chunk_it)
            except StopIteration:
                exhausted = True
                return
            for r in chunk:
                n = len(r["prompt_token_ids"])
                if n < 1 or n > min(self.slot_len, self.prefill_rows):
                    raise _hip.ThinkDiffHipError(f"generate_continuous: a prompt of {n} tokens does not fit a slot ({self.slot_len}) / a prefill pass ({self.prefill_rows})")
                waiting.append(len(requests))
                requests.append(r)
                res.append({"prompt_hidden_states": None, "hidden_states": None, "token_ids": []})

        # per live row (numpy, rows [0, n)): request, cache slot, cached tokens, next position id, tokens generated, token budget.  The per-step
        # bookkeeping is vectorised and the arrays are handed to the C ABI as they are: at 256 sequences the Python-list form cost ~2 ms of host
        # time per step, about as much as the step itself takes on the GPU, and the GPU idles while the host prepares the next launch.
        own, slt, clen, npos, ngen, lim = (np.zeros(max_live, dtype=np.int32) for _ in range(6))
        n = 0
        free_slots = list(range(max_live - 1, -1, -1))      # (a stack: the lowest free slot is taken first)
        logits = torch.empty(max_live, V, dtype=torch.bfloat16, device=self.device)      # row i = the next-token logits of live row i
        key = None if forced_output_ids is not None else self._draw_sampler_key(generator)
        stops = np.array(sorted(set(sampling.stop_token_ids or [])), dtype=np.int64)
        eos_on = (not sampling.ignore_eos) and eos_token_id is not None
        can_stop = forced_output_ids is None and (stops.size > 0 or eos_on)
        step_hid, step_owner, step_toks = [], [], []
        gstep = 0
        slot_len, min_tok = self.slot_len, sampling.min_tokens
        vp = ctypes.c_void_p

        def budget(i):      # tokens request i may still produce
            return len(forced_output_ids[i]) if forced_output_ids is not None else sampling.max_tokens

        while True:
            free = max_live - n
            if not exhausted and len(waiting) < free:      # lazily -- the producer builds the next chunk while the current sequences decode -- but enough to fill the free slots
                pull()
                continue
            # ---- admission: one packed prefill pass into free slots ----
            if waiting and free > 0 and (n == 0 or min(free, len(waiting)) >= admit_min or (exhausted and free >= len(waiting))):
                new, rows = [], 0
                while waiting and len(new) < free and rows + len(requests[waiting[0]]["prompt_token_ids"]) <= self.prefill_rows:
                    i = waiting.popleft()
                    if budget(i) <= 0:      # nothing to generate: the prompt states only, through the first free slot
                        q = requests[i]
                        m = len(q["prompt_token_ids"])
                        e = q.get("inputs_embeds")
                        res[i]["prompt_hidden_states"], _ = self.forward(
                            self.text_position_ids(m) if q.get("position_ids") is None else q["position_ids"],
                            torch.tensor(list(q["prompt_token_ids"]), dtype=torch.int32) if e is None else None, e, 0, True, False, slot=free_slots[-1])
                        continue
                    new.append(i)
                    rows += len(requests[i]["prompt_token_ids"])
                if new:
                    k = len(new)
                    emb = torch.empty(rows, D, dtype=torch.bfloat16, device=self.device)
                    pos = torch.empty(3, rows, dtype=torch.int32)
                    lens, r0 = [], 0
                    for j, i in enumerate(new):
                        q = requests[i]
                        m = len(q["prompt_token_ids"])
                        pp = self.text_position_ids(m) if q.get("position_ids") is None else q["position_ids"]
                        e = q.get("inputs_embeds")
                        emb[r0:r0 + m] = self.embed_tokens(q["prompt_token_ids"]) if e is None else e.to(self.device, torch.bfloat16)
                        pos[:, r0:r0 + m] = pp.to(torch.int32)
                        lens.append(m)
                        npos[n + j] = int(pp.max()) + 1
                        r0 += m
                    pos = pos.to(self.device).contiguous()
                    hid = torch.empty(rows, D, dtype=torch.bfloat16, device=self.device)
                    new_slots = [free_slots.pop() for _ in new]
                    cl = (ctypes.c_int * k)(*lens)
                    sl = (ctypes.c_int * k)(*new_slots)
                    self._prefill_packed_slots(k, sl, emb, pos, cl, hid, logits[n:n + k])
                    r0 = 0
                    for i, m in zip(new, lens):
                        res[i]["prompt_hidden_states"] = hid[r0:r0 + m]
                        r0 += m
                    own[n:n + k] = new
                    slt[n:n + k] = new_slots
                    clen[n:n + k] = lens
                    ngen[n:n + k] = 0
                    lim[n:n + k] = [budget(i) for i in new]
                    n += k
                    continue      # (more may fit in another pass before the next decode step)
            if n == 0:
                if waiting or not exhausted:
                    continue
                break
            # ---- one decode step for every live sequence ----
            if forced_output_ids is not None:
                toks = np.array([forced_output_ids[own[i]][ngen[i]] for i in range(n)], dtype=np.int32)
                tok_dev = torch.from_numpy(toks).to(self.device)
            else:
                tok_dev = _OPS.sample_top_p(logits[:n], float(sampling.temperature), float(sampling.top_p), int(key), int(gstep))
                toks = tok_dev.cpu().numpy()      # (the host sees the ids once per step, for the stop rules below)
            pos_dev = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(npos[:n], (3, n)))).to(self.device)
            hid = torch.empty(n, D, dtype=torch.bfloat16, device=self.device)
            self._decode_slots(n, slt, tok_dev, pos_dev, clen, hid, logits)      # logits of row i -> logits[i]
            step_hid.append(hid)
            step_owner.append(own[:n].copy())
            step_toks.append(toks)
            clen[:n] += 1
            npos[:n] += 1
            ngen[:n] += 1
            done = (ngen[:n] >= lim[:n]) | (clen[:n] >= slot_len)
            if can_stop:
                hit = np.isin(toks, stops) if stops.size else np.zeros(n, dtype=bool)
                if eos_on:
                    hit |= toks == eos_token_id
                done |= hit & (ngen[:n] >= min_tok)
            if done.any():
                keep = np.nonzero(~done)[0]
                free_slots.extend(sorted((int(x) for x in slt[:n][done]), reverse=True))
                k = keep.size
                for arr in (own, slt, clen, npos, ngen, lim):
                    arr[:k] = arr[:n][keep]
                if k:
                    logits[:k] = logits[:n].index_select(0, torch.from_numpy(keep).to(self.device))
                n = k
            gstep += 1
        N = len(requests)
        if step_owner:
            owners_all = np.concatenate(step_owner)
            order = np.argsort(owners_all, kind="stable")      # rows grouped by request, in step order inside a request
            counts = np.bincount(owners_all, minlength=N).tolist()
            parts = torch.split(torch.cat(step_hid).index_select(0, torch.from_numpy(order).to(self.device)), counts)
            toks_sorted = np.concatenate(step_toks)[order].tolist()
            r0 = 0
            for b in range(N):
                res[b]["hidden_states"] = parts[b]
                res[b]["token_ids"] = toks_sorted[r0:r0 + counts[b]]
                r0 += counts[b]
        else:
            for r in res:
                r["hidden_states"] = torch.empty(0, D, dtype=torch.bfloat16, device=self.device)
        return res

    # the two engine calls of generate_continuous (separate methods so that the scheduler's host logic can be exercised against a stand-in engine)
    def _prefill_packed_slots(self, k, slots_c, emb, pos, lens_c, hid_out, logits_out):
        _hip.check(self._L.td_qwen2_prefill_packed_slots(self._h, k, ctypes.cast(slots_c, ctypes.c_void_p), None, _hip.ptr(emb), _hip.ptr(pos),
                                                         ctypes.cast(lens_c, ctypes.c_void_p), _hip.ptr(hid_out), _hip.ptr(logits_out), _hip.stream_ptr()))

    def _decode_slots(self, n, slots_np, tok_dev, pos_dev, cache_np, hid_out, logits_out):
        _hip.check(self._L.td_qwen2_decode_batch_slots(self._h, n, ctypes.c_void_p(slots_np.ctypes.data), _hip.ptr(tok_dev), _hip.ptr(pos_dev),
                                                       ctypes.c_void_p(cache_np.ctypes.data), _hip.ptr(hid_out), _hip.ptr(logits_out), _hip.stream_ptr()))

    def move_slot(self, src: int, dst: int, length: int):
        """Copy the first `length` cache rows of slot `src` to slot `dst` (td_qwen2_move_slot).  The decode steps name their slots, so nothing in this
        class needs it any more; it stays for callers that want to defragment a handle before re-partitioning it (set_slots)."""
        _hip.check(self._L.td_qwen2_move_slot(self._h, int(src), int(dst), int(length), _hip.stream_ptr()))


def smart_resize(height: int, width: int, factor: int = 28, min_pixels: int = 4 * 28 * 28, max_pixels: int = 16384 * 28 * 28):
    """[ext] qwen_vl_utils.smart_resize / transformers Qwen2-VL image processing: both sides to multiples of `factor`,
    area inside [min_pixels, max_pixels], aspect ratio kept as closely as the grid allows."""
    import math
    if max(height, width) / min(height, width) > 200:
        raise ValueError(f"absolute aspect ratio must be smaller than 200, got {max(height, width) / min(height, width)}")
    h_bar = max(factor, round(height / factor) * factor)
    w_bar = max(factor, round(width / factor) * factor)
    if h_bar * w_bar > max_pixels:
        beta = math.sqrt((height * width) / max_pixels)
        h_bar = max(factor, math.floor(height / beta / factor) * factor)
        w_bar = max(factor, math.floor(width / beta / factor) * factor)
    elif h_bar * w_bar < min_pixels:
        beta = math.sqrt(min_pixels / (height * width))
        h_bar = math.ceil(height * beta / factor) * factor
        w_bar = math.ceil(width * beta / factor) * factor
    return h_bar, w_bar


def process_vision_info(conversations):
    """The image half of [ext] qwen_vl_utils.process_vision_info, which the reference's multi-image drivers call on their
    chat messages (scripts/test/test_mllama_t5_decoder_flux_multi_image.py:223): every {"type": "image", "image": path | PIL,
    ["min_pixels", "max_pixels" | "resized_height", "resized_width"]} part, in message order, opened as RGB and resized
    so that both sides are multiples of 28 and the area lies inside the part's pixel budget (defaults 4*28*28 ..
    16384*28*28).  Returns (images | None, None) -- videos are not part of the ThinkDiff path."""
    from PIL import Image
    convs = conversations if conversations and isinstance(conversations[0], (list, tuple)) else [conversations]
    images = []
    for conv in convs:
        for turn in conv:
            if isinstance(turn.get("content"), str):
                continue
            for part in turn["content"]:
                if part.get("type") != "image" and "image" not in part:
                    continue
                im = part["image"]
                if isinstance(im, str):
                    im = Image.open(im[len("file://"):] if im.startswith("file://") else im)
                im = im.convert("RGB")
                if "resized_height" in part and "resized_width" in part:
                    rh, rw = smart_resize(part["resized_height"], part["resized_width"], factor=28)
                else:
                    w, h = im.size
                    rh, rw = smart_resize(h, w, factor=28, min_pixels=part.get("min_pixels", 4 * 28 * 28),
                                          max_pixels=part.get("max_pixels", 16384 * 28 * 28))
                images.append(im.resize((rw, rh)))
    return (images or None), None


SYSTEM_PROMPT = "You are a helpful assistant."   # reference mllama_vllm_t5_embed_decoder_2.py:1048, mllama_vllm_generate_1.py:551


class QwenChatFrontend:
    """What vLLM's Qwen2-VL input pipeline does around the decoder, shared by the two LVLM model mirrors: ChatML request
    building, tokenisation, the vision tower, placeholder expansion / embedding splice and M-RoPE positions.
    Expects on `self`: mllama (Qwen2VLTextEngine), mllama_processor, mllama_tokenizer, image_processor, visual, image_token_id."""

    def chat_requests(self, texts, images) -> List[dict]:
        """(instruction, [image]) pairs -> [{"prompt", "multi_modal_data": {"image": ...}}] exactly as the reference builds
        them (system turn + user turn with one image part then the text part, generation prompt appended)."""
        if self.mllama_processor is None:
            raise _hip.ThinkDiffHipError(
                "text requests need the Qwen2-VL processor / chat template loaded from a local path (or the synthetic stand-in); "
                "pass need_process=False with {'prompt_token_ids': ...} requests instead")
        msgs = [[{"role": "system", "content": SYSTEM_PROMPT},
                 {"role": "user", "content": ([{"type": "image", "image": im}] if im is not None else []) + [{"type": "text", "text": t}]}]
                for t, im in zip(texts, images)]
        prompts = self.mllama_processor.apply_chat_template(msgs, tokenize=False, add_generation_prompt=True)
        return [{"prompt": p, "multi_modal_data": {"image": im}} for p, im in zip(prompts, images)]

    def splice_images(self, ids, images):
        """Prompt ids with one placeholder per image -> (expanded ids, inputs_embeds [n, hidden], position_ids [3, n])."""
        if self.visual is None or self.image_processor is None:
            raise _hip.ThinkDiffHipError("image request: load the vision tower (visual=HipQwen2VisionTransformer...) and an "
                                         "image_processor, or supply 'inputs_embeds' and 'position_ids' with the request")
        images = list(images) if isinstance(images, (list, tuple)) else [images]
        feats = self._preprocess_on_device(images) or self.image_processor(images=images, return_tensors="pt")
        grid = feats["image_grid_thw"].tolist() if torch.is_tensor(feats["image_grid_thw"]) else feats["image_grid_thw"]
        merged = self.visual(feats["pixel_values"], grid).pooler_output
        merge = self.visual.merge
        ids = Qwen2VLTextEngine.expand_image_placeholders(ids, grid, merge, self.image_token_id)
        emb = self.mllama.embed_tokens(ids)
        mask = torch.tensor(ids) == self.image_token_id
        emb[mask.to(emb.device)] = merged
        return ids, emb, Qwen2VLTextEngine.mrope_position_ids(ids, grid, merge, self.image_token_id)

    def resolve_requests(self, reqs: Sequence[dict]) -> List[dict]:
        """resolve_request for a batch: all images of the batch go through the image processor and the vision tower in ONE
        call (the tower's GEMMs then see thousands of patch rows instead of one image's few hundred)."""
        out = [dict(r) for r in reqs]
        for r in out:
            if "prompt_token_ids" not in r:
                if self.mllama_tokenizer is None:
                    raise _hip.ThinkDiffHipError("request has no 'prompt_token_ids' and no tokenizer is loaded")
                r["prompt_token_ids"] = self.mllama_tokenizer.encode(r["prompt"], add_special_tokens=False)
        need = [i for i, r in enumerate(out) if (r.get("multi_modal_data") or {}).get("image") is not None and "inputs_embeds" not in r]
        if not need:
            return out
        if self.visual is None or self.image_processor is None:
            raise _hip.ThinkDiffHipError("image request: load the vision tower (visual=HipQwen2VisionTransformer...) and an "
                                         "image_processor, or supply 'inputs_embeds' and 'position_ids' with the request")
        per_req = []
        for i in need:
            im = out[i]["multi_modal_data"]["image"]
            per_req.append(list(im) if isinstance(im, (list, tuple)) else [im])
        flat = [im for ims in per_req for im in ims]
        feats = self._preprocess_on_device(flat)
        if feats is None and len(flat) > 2:      # resize / normalise / patchify is CPU work per image: spread it over threads (PIL and numpy drop the GIL)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(8, len(flat))) as pool:
                parts = list(pool.map(lambda im: self.image_processor(images=[im], return_tensors="pt"), flat))
            feats = {"pixel_values": torch.cat([f["pixel_values"] for f in parts]), "image_grid_thw": torch.cat([f["image_grid_thw"] for f in parts])}
        elif feats is None:
            feats = self.image_processor(images=flat, return_tensors="pt")
        grid = feats["image_grid_thw"].tolist() if torch.is_tensor(feats["image_grid_thw"]) else feats["image_grid_thw"]
        merged = self.visual(feats["pixel_values"], grid).pooler_output
        merge = self.visual.merge
        counts = [t * h * w // (merge * merge) for t, h, w in grid]
        # One embedding gather and ONE scatter of the merged vision tokens for the whole batch: the placeholder positions are known on
        # the host (they are token ids), so nothing here waits for the device -- a boolean-mask assignment per request would
        # synchronise with the vision tower once per request.
        import numpy as np
        g0 = 0
        all_ids, spans, grids = [], [], []
        for i, ims in zip(need, per_req):
            g = grid[g0:g0 + len(ims)]
            ids = Qwen2VLTextEngine.expand_image_placeholders(list(out[i]["prompt_token_ids"]), g, merge, self.image_token_id)
            spans.append((len(all_ids), len(all_ids) + len(ids)))
            all_ids += ids
            grids.append(g)
            g0 += len(ims)
        flat_ids = np.asarray(all_ids, dtype=np.int64)
        where = np.flatnonzero(flat_ids == self.image_token_id)
        if where.size != sum(counts):
            raise ValueError(f"the batch holds {where.size} image placeholder tokens for {sum(counts)} merged vision tokens")
        emb_all = self.mllama.embed_tokens(all_ids)
        emb_all.index_copy_(0, torch.from_numpy(where).to(emb_all.device), merged.to(emb_all.dtype))
        for i, (a0, a1), g in zip(need, spans, grids):
            r = out[i]
            ids = all_ids[a0:a1]
            r["prompt_token_ids"], r["inputs_embeds"] = ids, emb_all[a0:a1]
            r["position_ids"] = Qwen2VLTextEngine.mrope_position_ids(ids, g, merge, self.image_token_id)
        return out

    def _device_preprocess_plan(self):
        """Settings of the image processor when its pipeline is the PIL one of transformers (convert to RGB, smart_resize +
        PIL resize, rescale, normalize, patchify) -- then only the resize stays on the host and the rest runs as one HIP launch per
        image (td_qwen2_patchify_u8).  None for any other processor: it is called as is."""
        ip = self.image_processor
        plan = getattr(self, "_dev_pre_plan", None)
        if plan is not None and plan[0] is ip:
            return plan[1]
        out = None
        try:
            if type(ip).__name__ == "Qwen2VLImageProcessorPil" and hasattr(self.visual, "padded_patch_dim"):
                import numpy as np
                ramp = np.arange(256, dtype=np.uint8)[None, None, :].repeat(3, 0)          # [C, 1, 256] channels first, as the backend holds images
                x = ip.rescale(ramp, ip.rescale_factor) if ip.do_rescale else ramp
                x = ip.normalize(x, ip.image_mean, ip.image_std) if ip.do_normalize else x
                lut = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32).reshape(3, 256))).to(self.visual.device)
                size = ip.size
                out = dict(lut=lut, patch=int(ip.patch_size), merge=int(ip.merge_size), temporal=int(ip.temporal_patch_size), resample=int(ip.resample),
                           do_resize=bool(ip.do_resize), convert_rgb=bool(ip.do_convert_rgb), min_pixels=int(size["shortest_edge"]), max_pixels=int(size["longest_edge"]))
                if 3 * out["temporal"] * out["patch"] ** 2 > self.visual.padded_patch_dim or out["merge"] != self.visual.merge:
                    out = None
        except Exception:
            out = None
        self._dev_pre_plan = (ip, out)
        return out

    def _preprocess_on_device(self, images):
        """PIL images -> {"pixel_values": bf16 [S, Kpad] on the device, "image_grid_thw": [[1, gh, gw], ...]} or None (caller falls back
        to the processor).  Host: RGB conversion + the processor's smart_resize / PIL resize (threads); device: the rest."""
        plan = self._device_preprocess_plan()
        from PIL import Image
        if plan is None or not images or not all(isinstance(im, Image.Image) for im in images):
            return None
        if not plan["convert_rgb"] and any(im.mode != "RGB" for im in images):
            return None
        import numpy as np
        f = plan["patch"] * plan["merge"]

        def prep(im):
            im = im.convert("RGB") if im.mode != "RGB" else im
            w, h = im.size
            if plan["do_resize"]:
                h2, w2 = smart_resize(h, w, factor=f, min_pixels=plan["min_pixels"], max_pixels=plan["max_pixels"])
                im = im.resize((w2, h2), resample=plan["resample"])
            elif h % f or w % f:
                raise ValueError(f"image {w}x{h} is not a multiple of {f} and the processor does not resize")
            return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))

        if len(images) > 2:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(16, len(images))) as pool:
                arrs = list(pool.map(prep, images))
        else:
            arrs = [prep(im) for im in images]
        p = plan["patch"]
        grid = [[1, a.shape[0] // p, a.shape[1] // p] for a in arrs]
        Kpad = self.visual.padded_patch_dim
        dev = self.visual.device
        out = torch.empty(sum(g[1] * g[2] for g in grid), Kpad, dtype=torch.bfloat16, device=dev)
        r0 = 0
        for a, g in zip(arrs, grid):
            n = g[1] * g[2]
            _hip.qwen2_patchify_u8(torch.from_numpy(a).to(dev), plan["lut"], p, plan["merge"], plan["temporal"], Kpad, out=out[r0:r0 + n])
            r0 += n
        return {"pixel_values": out, "image_grid_thw": grid}

    def resolve_request(self, r: dict) -> dict:
        """Fill prompt_token_ids (tokenizer) and, for image requests, inputs_embeds + position_ids."""
        r = dict(r)
        if "prompt_token_ids" not in r:
            if self.mllama_tokenizer is None:
                raise _hip.ThinkDiffHipError("request has no 'prompt_token_ids' and no tokenizer is loaded")
            r["prompt_token_ids"] = self.mllama_tokenizer.encode(r["prompt"], add_special_tokens=False)
        mm = r.get("multi_modal_data") or {}
        if mm.get("image") is not None and "inputs_embeds" not in r:
            r["prompt_token_ids"], r["inputs_embeds"], r["position_ids"] = self.splice_images(list(r["prompt_token_ids"]), mm["image"])
        return r

"""Precompute model (BASELINE config 4): Qwen2-VL generate + hidden states for a whole loader batch.

Mirror of reference thinkdiff/models/mllama_vllm_generate_1.py `MllamaVllmGenerate_1` (:364-718): `forward(samples)`
(:625-641) -> `forward_inner` (:493-623) returns
  {"generated_text": [str], "generated_token": {"input_prompt", "input_prompt_token_ids", "output_text",
   "output_token_ids"}, "generated_embed": {"model.norm": {"output_embed": [Tensor[n_gen,D]], "input_embed":
   [Tensor[n_prompt,D]]}}}.
The vLLM engine is replaced by `Qwen2VLTextEngine` (HIP) behind `QwenChatFrontend` (chat template -> token ids -> vision
tower -> placeholder splice -> M-RoPE positions).  The chat template / tokenizer come from a local Qwen2-VL directory or the
synthetic stand-in (`providers.load_lvlm_frontend`); `request_builder(samples, i) -> {"prompt", "prompt_token_ids", optional
"inputs_embeds", "position_ids"}` overrides the whole front end.
"""
import os
from concurrent.futures import ThreadPoolExecutor
from types import SimpleNamespace
from typing import Callable, Dict, List, Optional

import torch

from .. import _hip
from ..common.registry import registry

from .base_model import BaseModel
from .qwen2_vl import QwenChatFrontend, Qwen2VLTextConfig, Qwen2VLTextEngine, SamplingParams


@registry.register_model("mllama-vllm-generate-1")
class MllamaVllmGenerate_1(QwenChatFrontend, BaseModel):
    PRETRAINED_MODEL_CONFIG_DICT = {"pretrain_mllama_vllm_generate_1": "configs/models/mllama_vllm_generate_1.yaml"}

    def __init__(self, text_config: Optional[Qwen2VLTextConfig] = None, vllm_config: Optional[dict] = None,
                 text_input_key: Optional[str] = "answers", device="cuda", tokenizer=None,
                 request_builder: Optional[Callable[[dict, int], dict]] = None):
        vc = dict(vllm_config or {})
        self.config = SimpleNamespace(vllm_config=vc, text_input_key=text_input_key)
        self._device = torch.device(device)
        # vLLM decodes `max_num_seqs` requests together (256 in configs/qwen2_vl_embed_ccsbu.yaml); the engine takes up to 256 per decode step
        self.decode_batch = max(1, min(Qwen2VLTextEngine.MAX_BATCH, int(vc.get("max_num_seqs", Qwen2VLTextEngine.MAX_BATCH))))
        self.mllama = Qwen2VLTextEngine(text_config, max_model_len=vc.get("max_model_len", 8192), device=device, n_slots=self.decode_batch,
                                        prefill_rows=min(int(vc.get("max_num_batched_tokens", 16384)), 65536) if self.decode_batch > 1 else None)      # (60 000 in the reference config: rows of one packed prefill pass)
        self.mllama_sampling_params = SamplingParams(
            temperature=vc.get("temperature", 0.6), top_p=vc.get("top_p", 0.9), max_tokens=vc.get("max_tokens", 256),
            min_tokens=vc.get("min_tokens", 1), ignore_eos=vc.get("ignore_eos", False),
            stop_token_ids=list(vc["stop_token_ids"]) if vc.get("stop_token_ids") else None)      # (a vllm.SamplingParams field; the reference's config leaves it unset)
        self.mllama_tokenizer, self.mllama_processor = tokenizer, None
        self.visual, self.image_processor, self.image_token_id = None, None, 151655
        self.request_builder = request_builder
        self.eos_token_id = vc.get("eos_token_id", None)

    @classmethod
    def from_config(cls, cfg):
        vc = cfg.get("vllm_config", {})
        vc = vc.to_dict() if hasattr(vc, "to_dict") else dict(vc)
        tc = cfg.get("text_config", None)       # optional decoder shape override; default = the Qwen2-VL-7B shape
        tc = Qwen2VLTextConfig(**(tc.to_dict() if hasattr(tc, "to_dict") else dict(tc))) if tc else None
        return cls(text_config=tc, vllm_config=vc, text_input_key=cfg.get("text_input_key", "answers"), device=cfg.get("device", "cuda"))

    def _request(self, samples: dict, i: int) -> dict:
        if self.request_builder is not None:
            return self.request_builder(samples, i)
        texts = samples["answers"] if self.config.text_input_key is None else samples[self.config.text_input_key]
        if self.mllama_tokenizer is None:
            raise _hip.ThinkDiffHipError("MllamaVllmGenerate_1: supply `request_builder` (token ids / vision embeddings) or a tokenizer")
        if self.mllama_processor is not None:      # reference :543-583: chat template + image, one request per sample
            images = samples.get("images", None)
            return self.resolve_request(self.chat_requests([texts[i]], [images[i] if images is not None else None])[0])
        prompt = texts[i]
        return {"prompt": prompt, "prompt_token_ids": self.mllama_tokenizer.encode(prompt, add_special_tokens=False)}

    def _requests(self, samples: dict, idx) -> List[dict]:
        """The requests of one decode chunk; with the chat front end loaded, their images share one vision-tower call."""
        if self.request_builder is not None or self.mllama_processor is None or self.mllama_tokenizer is None:
            return [self._request(samples, i) for i in idx]
        texts = samples["answers"] if self.config.text_input_key is None else samples[self.config.text_input_key]
        images = samples.get("images", None)
        idx = list(idx)
        return self.resolve_requests(self.chat_requests([texts[i] for i in idx], [images[i] if images is not None else None for i in idx]))

    @torch.no_grad()
    def forward_inner(self, mllama_inputs: dict, generator=None) -> Dict:
        n = len(mllama_inputs["images"]) if "images" in mllama_inputs else len(mllama_inputs["answers"])
        layer = self.config.vllm_config.get("embedding_layer_name", "model.norm")
        tok = {"input_prompt": [], "input_prompt_token_ids": [], "output_text": [], "output_token_ids": []}
        out_embed, in_embed, texts = [], [], []
        # The requests of chunk k + 1 -- chat template, image resize, device patchify, the vision tower, token embeddings: ~1.2 s per 512 samples, a
        # third of the chunk's decode time -- are built by a helper thread on a stream of its own while chunk k decodes (the decode loop is a chain
        # of small latency-bound launches that leaves most of the chip idle, and its host thread spends its time waiting for token ids).  vLLM's
        # engine overlaps its input processing with decoding in the same way.  TD_PRECOMPUTE_PREFETCH=0: one after the other (A/B).
        chunks = [range(c0, min(c0 + self.decode_batch, n)) for c0 in range(0, n, self.decode_batch)]
        prefetch = len(chunks) > 1 and os.environ.get("TD_PRECOMPUTE_PREFETCH", "1") != "0"
        pool = ThreadPoolExecutor(max_workers=1) if prefetch else None
        # Continuous batching over the whole loader batch (Qwen2VLTextEngine.generate_continuous -- a finished sequence's slot goes to a waiting request
        # at once, as in vLLM's scheduler) instead of one generate_batch per chunk of `decode_batch`, which runs every chunk down to its longest
        # sequence.  Measured on the 2B shape (profiles/r4av_job_continuous_4096.log, r4as_job_numpy_var_ab.log), model seconds per 4096 samples:
        # outputs that end at random (mean 110 of 256 tokens) 18.2 vs 24.1 with a loader batch of 4096 (16 chunks) -- but 14.0 vs 13.1 per 2048 with
        # loader batches of 1024 (4 chunks: the scheduler then waits for this model's request builder, 0.6-0.9 s per 256 requests on one helper
        # thread, more than it saves); outputs that all run to 256 tokens 27.9 vs 26.5.  Default: from 8 chunks per loader batch on (the reference's
        # loader batch is 8192 = 32 chunks, its outputs end at EOS); TD_PRECOMPUTE_CONTINUOUS=1 / 0 forces either form.  The two sample different
        # tokens (the draw of a sequence depends on the step and the row it sits in), each reproducibly under its seed.
        env_c = os.environ.get("TD_PRECOMPUTE_CONTINUOUS", "")
        continuous = env_c == "1" or (env_c != "0" and len(chunks) >= 8)
        all_reqs, all_outs = [], []
        try:
            if continuous:
                def chunk_source():
                    fut = pool.submit(self._requests_on_side_stream, mllama_inputs, chunks[0]) if prefetch else None
                    for k, idx in enumerate(chunks):
                        reqs = fut.result() if prefetch else self._requests(mllama_inputs, idx)
                        if prefetch and k + 1 < len(chunks):
                            fut = pool.submit(self._requests_on_side_stream, mllama_inputs, chunks[k + 1])
                        all_reqs.extend(reqs)
                        yield reqs
                all_outs = self.mllama.generate_continuous(chunk_source(), self.mllama_sampling_params, eos_token_id=self.eos_token_id, generator=generator,
                                                           max_live=self.decode_batch,
                                                           admit_min=int(os.environ["TD_CONTINUOUS_ADMIT_MIN"]) if os.environ.get("TD_CONTINUOUS_ADMIT_MIN") else None)
            else:
                fut = pool.submit(self._requests_on_side_stream, mllama_inputs, chunks[0]) if prefetch else None
                for k, idx in enumerate(chunks):
                    if prefetch:
                        reqs = fut.result()
                        fut = pool.submit(self._requests_on_side_stream, mllama_inputs, chunks[k + 1]) if k + 1 < len(chunks) else None
                    else:
                        reqs = self._requests(mllama_inputs, idx)
                    all_reqs.extend(reqs)
                    all_outs.extend(self.mllama.generate_batch(reqs, self.mllama_sampling_params, eos_token_id=self.eos_token_id, generator=generator))
        finally:
            if pool is not None:
                pool.shutdown(wait=True)      # (also when a chunk raised: the helper thread does not outlive the call)
        for r, o in zip(all_reqs, all_outs):
            text = self.mllama_tokenizer.decode(o["token_ids"]) if self.mllama_tokenizer is not None else " ".join(map(str, o["token_ids"]))
            tok["input_prompt"].append(r.get("prompt", ""))
            tok["input_prompt_token_ids"].append(list(r["prompt_token_ids"]))
            tok["output_text"].append(text)
            tok["output_token_ids"].append(tuple(o["token_ids"]))
            texts.append(text)
            out_embed.append(o["hidden_states"])
            in_embed.append(o["prompt_hidden_states"])
        return {"generated_text": texts, "generated_token": tok,
                "generated_embed": {layer: {"output_embed": out_embed, "input_embed": in_embed}}}

    def _requests_on_side_stream(self, inputs: dict, idx) -> List[dict]:
        """`_requests` on the helper thread: its launches go to a stream of this model's own (torch's current stream is per thread), and the thread
        returns only when they have completed, so the decode thread may use the tensors on its stream without an event."""
        with torch.no_grad(), torch.cuda.device(self._device):
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=self._device)
            with torch.cuda.stream(self._side_stream):
                reqs = self._requests(inputs, idx)
            self._side_stream.synchronize()
        return reqs

    def forward(self, samples, reduction="mean"):
        inputs = {k: v for k, v in samples.items() if k not in ("epoch", "num_iters_per_epoch", "iters")}
        return self.forward_inner(mllama_inputs=inputs)

    __call__ = forward

"""ThinkDiff-LVLM model on the HIP path: Qwen2-VL hidden states -> aligner.

Mirror of reference thinkdiff/models/mllama_vllm_t5_embed_decoder_2.py `MllamaVllmT5EmbedDecoderForConditional
Generation_5` (:779-1191): `from_config` (:838-903) and `get_embed` (:1019-1118).  The vLLM engine is replaced by
`Qwen2VLTextEngine` (libthinkdiff_hip.so `td_qwen2_*`), the aligner by `HipVisionProjector`
(`td_aligner_mlp2x_bf16`, fp32-norm variant: in the reference the aligner's parameters stay fp32 and run under
bf16 autocast, :884 and scripts/test/test_mllama_t5_decoder_flux.py:149).

Images: `visual` (vision_towers.HipQwen2VisionTransformer) + `image_processor` (the HF Qwen2-VL image processor, or any
callable images -> {"pixel_values", "image_grid_thw"}) turn `multi_modal_data["image"]` into merged vision tokens that
replace the `<|image_pad|>` rows of the prompt embedding, with M-RoPE positions from `mrope_position_ids`.
The HF chat template / tokenizer assets cannot be fetched here: requests carry token ids (`{"prompt_token_ids": [...]}`,
vLLM's TokensPrompt form) unless a tokenizer/processor loaded from a LOCAL path enables the text prompt form.
"""
from types import SimpleNamespace
from typing import List

import torch

from .. import _hip
from ..common.registry import registry
from .base_model import BaseModel
from .blip_vision_t5_decoder import HipVisionProjector
from .qwen2_vl import QwenChatFrontend, Qwen2VLTextConfig, Qwen2VLTextEngine, SamplingParams

SYSTEM_PROMPT = "You are a helpful assistant."   # reference :1048
NUM_SYSTEM_TOKENS = 14                            # reference :1107-1109 ("input_no_system" drops the first 14)


@registry.register_model("mllama-vllm-t5-embed-decoder-5")
class MllamaVllmT5EmbedDecoderForConditionalGeneration_5(QwenChatFrontend, BaseModel):
    PRETRAINED_MODEL_CONFIG_DICT = {"pretrain_mllama_vllm_t5_embed_decoder_5": "configs/models/mllama_vllm_t5_embed_decoder_5.yaml"}   # reference :781-783

    def __init__(self, text_config: Qwen2VLTextConfig = None, vllm_config: dict = None, hidden_size: int = 4096,
                 mm_projector_type: str = "mlp2x_gelu_t5_norm", device="cuda", tokenizer=None, processor=None,
                 visual=None, image_processor=None, image_token_id: int = 151655):
        vc = dict(vllm_config or {})
        self.config = SimpleNamespace(vllm_config=vc, mm_projector_type=mm_projector_type,
                                      mm_hidden_size=(text_config or Qwen2VLTextConfig()).hidden_size, hidden_size=hidden_size)
        self._device = torch.device(device)
        # vLLM decodes up to `max_num_seqs` requests together; the engine advances up to 256 per decode step
        self.decode_batch = max(1, min(Qwen2VLTextEngine.MAX_BATCH, int(vc.get("max_num_seqs", 1))))
        self.mllama = Qwen2VLTextEngine(text_config, max_model_len=vc.get("max_model_len", 8192), device=device, n_slots=self.decode_batch,
                                        prefill_rows=min(int(vc.get("max_num_batched_tokens", 16384)), 16384) if self.decode_batch > 1 else None)
        self.mllama_sampling_params = SamplingParams(
            temperature=vc.get("temperature", 0.6), top_p=vc.get("top_p", 0.9), max_tokens=vc.get("max_tokens", 128),
            min_tokens=vc.get("min_tokens", 128), ignore_eos=vc.get("ignore_eos", True))
        self.mm_projector = HipVisionProjector(self.config.mm_hidden_size, hidden_size, mm_projector_type, device=device, fp32_norm=True)
        self.mllama_tokenizer, self.mllama_processor = tokenizer, processor
        self.visual, self.image_token_id = visual, image_token_id
        self.image_processor = image_processor or getattr(processor, "image_processor", None)

    @classmethod
    def from_config(cls, cfg):
        vc = cfg.get("vllm_config", {})
        vc = vc.to_dict() if hasattr(vc, "to_dict") else dict(vc)
        tc = cfg.get("text_config", None)       # optional decoder shape override (tests); default = Qwen2-VL-7B
        tc = Qwen2VLTextConfig(**(tc.to_dict() if hasattr(tc, "to_dict") else dict(tc))) if tc else None
        model = cls(text_config=tc, vllm_config=vc, mm_projector_type=cfg.get("mm_projector_type", "mlp2x_gelu_t5_norm"),
                    device=cfg.get("device", "cuda"))
        import os
        ckpt = cfg.get("ckpt", "")
        if ckpt and os.path.isfile(ckpt):
            model.load_state_dict(torch.load(ckpt, map_location="cpu")["model"], strict=False)
        return model

    def load_state_dict(self, sd, strict=False):
        sub = {k[len("mm_projector."):]: v for k, v in sd.items() if k.startswith("mm_projector.")}
        return self.mm_projector.load_state_dict(sub, strict=strict)

    def _to_requests(self, mllama_inputs, need_process) -> List[dict]:
        if need_process:
            texts = mllama_inputs["answers"]
            return self.chat_requests(texts, mllama_inputs.get("images", [None] * len(texts)))
        return mllama_inputs if isinstance(mllama_inputs, list) else [mllama_inputs]

    def _splice_images(self, ids, images):
        return self.splice_images(ids, images)

    @torch.no_grad()
    def get_embed(self, mllama_inputs, embedding_type="both", output_len_factor=1, need_process=True,
                  forced_output_ids=None, generator=None, **generate_kwargs):
        """-> (list[Tensor[n_i, 4096]], list[str]) exactly as the reference (:1019-1118); the second list holds the
        decoded text when a tokenizer is loaded, else the generated token ids as a space-separated string."""
        reqs = self._to_requests(mllama_inputs, need_process)
        sp = self.mllama_sampling_params    # **generate_kwargs (e.g. the drivers' max_new_tokens=128) are accepted and unused, as in the reference (:1019-1118)
        reqs = self.resolve_requests(reqs)
        outs = []
        if self.decode_batch > 1 and len(reqs) > 1:
            for c0 in range(0, len(reqs), self.decode_batch):
                chunk = reqs[c0:c0 + self.decode_batch]
                forced = None if forced_output_ids is None else list(forced_output_ids[c0:c0 + self.decode_batch])
                outs.extend(self.mllama.generate_batch(chunk, sp, generator=generator, forced_output_ids=forced))
        else:
            for i, r in enumerate(reqs):
                forced = None if forced_output_ids is None else forced_output_ids[i]
                outs.append(self.mllama.generate(list(r["prompt_token_ids"]), sp, position_ids=r.get("position_ids"),
                                                 inputs_embeds=r.get("inputs_embeds"), generator=generator, forced_output_ids=forced))
        inp = [o["prompt_hidden_states"] for o in outs]
        out = [o["hidden_states"] for o in outs]
        if embedding_type == "both":
            sel = [torch.cat([a, b], dim=0) for a, b in zip(inp, out)]
        elif embedding_type == "input_embed":
            sel = inp
        elif embedding_type == "input_no_system":
            sel = [a[NUM_SYSTEM_TOKENS:] for a in inp]
        elif embedding_type == "output_embed":
            sel = out
        else:
            raise ValueError(f"unknown embedding_type {embedding_type!r}")
        texts = [self.mllama_tokenizer.decode(o["token_ids"]) if self.mllama_tokenizer is not None
                 else " ".join(map(str, o["token_ids"])) for o in outs]
        return [self.mm_projector(e) for e in sel], texts

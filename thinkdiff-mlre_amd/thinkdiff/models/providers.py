"""Where the upstream stages of the ThinkDiff-CLIP driver come from: the EVA-ViT-g vision
tower + Blip2Processor and the CLIP-L / T5-XXL text encoders.  All of them run on libthinkdiff_hip.so
(`vision_towers.HipBlip2VisionModel`, `text_encoders.HipT5Encoder / HipCLIPTextEncoder`).

* `run.local_weights.<name>` pointing at a local directory -> weights (and the HF processor / tokenizers) are read
  from disk;
* `run.synthetic: true` -> full-size synthetic weights drawn on the device plus asset-free processor / tokenizer
  stand-ins, so the whole driver (config surface, naming rules, tower, aligner, encoders, FLUX, VAE, PNG writing)
  runs the real amount of compute without any asset; `run.synthetic_tiny: true` swaps in the cheap seeded stand-ins
  below for plumbing tests.  Hub ids are never fetched.
"""
import hashlib

import torch


def _seed_from(*parts) -> int:
    return int(hashlib.sha256("|".join(map(str, parts)).encode()).hexdigest()[:8], 16)


class SyntheticVisionTower:
    """pixel_values [B,3,224,224] -> [B,257,1408], deterministic in the pixels (seeded by their checksum)."""

    def __call__(self, pixel_values):
        outs = []
        for b in range(pixel_values.shape[0]):
            g = torch.Generator().manual_seed(_seed_from("vision", float(pixel_values[b].float().sum())))
            outs.append(torch.randn(257, 1408, generator=g))
        return torch.stack(outs).to(pixel_values.device, torch.bfloat16)


class SyntheticImageProcessor:
    """Blip2Processor stand-in: resize to 224 (bicubic), rescale, CLIP mean/std -> {'pixel_values': [1,3,224,224]}."""
    mean, std = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)

    def __call__(self, image, text=None, return_tensors="pt"):
        import numpy as np
        from PIL import Image
        arr = np.asarray(image.convert("RGB").resize((224, 224), Image.BICUBIC), dtype=np.float32) / 255.0
        t = (torch.from_numpy(arr).permute(2, 0, 1) - torch.tensor(self.mean)[:, None, None]) / torch.tensor(self.std)[:, None, None]
        return {"pixel_values": t[None]}


class SyntheticTextEncoders:
    """encode_prompt stand-in: T5 [1,max_len,4096] and CLIP pooled [1,768], deterministic in the prompt text."""

    def t5(self, prompt: str, max_sequence_length: int, device):
        g = torch.Generator().manual_seed(_seed_from("t5", prompt))
        return (0.1 * torch.randn(1, max_sequence_length, 4096, generator=g)).to(device, torch.bfloat16)

    def clip_pooled(self, prompt: str, device):
        g = torch.Generator().manual_seed(_seed_from("clip", prompt))
        return torch.randn(1, 768, generator=g).to(device, torch.bfloat16)


def load_vision(run_cfg, device):
    lw = run_cfg.get("local_weights", None) or {}
    if lw.get("blip2", None):
        from transformers import Blip2Processor
        from .vision_towers import HipBlip2VisionModel
        return Blip2Processor.from_pretrained(lw["blip2"], local_files_only=True), HipBlip2VisionModel.from_pretrained(lw["blip2"], device=device)
    if run_cfg.get("synthetic", False):
        if run_cfg.get("synthetic_tiny", False):
            return SyntheticImageProcessor(), SyntheticVisionTower()
        from .vision_towers import HipBlip2VisionModel
        return SyntheticImageProcessor(), HipBlip2VisionModel.from_random(seed=run_cfg.get("seed", 0), device=device)
    raise FileNotFoundError("no vision tower: set run.local_weights.blip2 to a local checkpoint directory or run.synthetic: true")


def load_text_encoders(run_cfg, pipe, device):
    """Attach CLIP-L / T5-XXL encoders (+ tokenizers) to `pipe` when it has none.  Returns the cheap stand-in object
    for `synthetic_tiny` runs, else None (encode_prompt then runs the HIP encoders)."""
    if pipe.text_encoder is not None and pipe.text_encoder_2 is not None:
        return None
    if run_cfg.get("synthetic", False) and not run_cfg.get("synthetic_tiny", False):
        from .text_encoders import HashTokenizer, HipCLIPTextEncoder, HipT5Encoder
        seed = run_cfg.get("seed", 0)
        pipe.text_encoder, pipe.tokenizer = HipCLIPTextEncoder.from_random(seed=seed + 11, device=device), HashTokenizer(49408)
        pipe.text_encoder_2, pipe.tokenizer_2 = HipT5Encoder.from_random(seed=seed + 12, device=device), HashTokenizer(32128)
        return None
    return SyntheticTextEncoders()


# ---- ThinkDiff-LVLM: Qwen2-VL chat template / tokenizer / vision tower ------------------------------------------------
class SyntheticQwenChat:
    """Asset-free stand-in for the Qwen2-VL AutoProcessor + tokenizer (synthetic runs): the ChatML layout the real
    template produces (system / user turns, one <|vision_start|><|image_pad|><|vision_end|> per image, generation prompt),
    word-hash token ids below the special-token range, and the real HF image processor for the pixels."""
    IM_START, IM_END, VISION_START, VISION_END, IMAGE_PAD = 151644, 151645, 151652, 151653, 151655
    SPECIAL = {"<|im_start|>": IM_START, "<|im_end|>": IM_END, "<|vision_start|>": VISION_START,
               "<|vision_end|>": VISION_END, "<|image_pad|>": IMAGE_PAD}

    def __init__(self, vocab_size: int = 151643, min_pixels: int = 56 * 56, max_pixels: int = 28 * 28 * 1280):
        self.vocab_size = min(vocab_size, 151643)
        self._ip_args = dict(min_pixels=min_pixels, max_pixels=max_pixels)
        self._ip = None

    @property
    def image_processor(self):
        if self._ip is None:
            from transformers import Qwen2VLImageProcessor
            self._ip = Qwen2VLImageProcessor(**self._ip_args)
        return self._ip

    def apply_chat_template(self, conversations, tokenize=False, add_generation_prompt=True, add_vision_id=False):
        """A list of conversations -> list of prompts; ONE conversation (a list of turns) -> one prompt, as the HF processor
        does.  add_vision_id: the Qwen2-VL template's "Picture N: " label in front of every image (the reference's
        multi-image drivers set it, scripts/test/test_mllama_t5_decoder_flux_multi_image.py:217-220)."""
        single = bool(conversations) and isinstance(conversations[0], dict)
        out = []
        for conv in ([conversations] if single else conversations):
            text, n_img = "", 0
            for turn in conv:
                body = turn["content"]
                if not isinstance(body, str):
                    parts = []
                    for part in body:
                        if part["type"] == "image":
                            n_img += 1
                            parts.append((f"Picture {n_img}: " if add_vision_id else "") + "<|vision_start|><|image_pad|><|vision_end|>")
                        else:
                            parts.append(part["text"])
                    body = "".join(parts)
                text += f"<|im_start|>{turn['role']}\n{body}<|im_end|>\n"
            out.append(text + ("<|im_start|>assistant\n" if add_generation_prompt else ""))
        return out[0] if single else out

    def encode(self, text, add_special_tokens=False):
        import re
        import zlib
        ids = []
        for piece in re.split(r"(<\|[a-z_]+\|>)", text):
            if piece in self.SPECIAL:
                ids.append(self.SPECIAL[piece])
            else:
                ids.extend(zlib.crc32(w.encode()) % self.vocab_size for w in piece.split())
        return ids

    def decode(self, ids, **_kw):
        return " ".join(f"<{int(i)}>" for i in ids)


def load_lvlm_frontend(run_cfg, model, device):
    """Attach tokenizer / chat processor / image processor / vision tower to the ThinkDiff-LVLM model.
    `run.local_weights.qwen2_vl`: a local Hugging Face Qwen2-VL directory (weights + processor); `run.synthetic: true`:
    synthetic weights drawn on the device and the asset-free chat stand-in."""
    lw = run_cfg.get("local_weights", None) or {}
    from .vision_towers import HipQwen2VisionTransformer
    if lw.get("qwen2_vl", None):
        from transformers import AutoProcessor
        proc = AutoProcessor.from_pretrained(lw["qwen2_vl"], local_files_only=True)
        model.mllama_processor, model.mllama_tokenizer, model.image_processor = proc, proc.tokenizer, proc.image_processor
        model.visual = HipQwen2VisionTransformer.from_pretrained(lw["qwen2_vl"], device=device)
        model.mllama.load_pretrained(lw["qwen2_vl"])
        return model
    if run_cfg.get("synthetic", False):
        chat = SyntheticQwenChat(vocab_size=model.mllama.config.vocab_size,
                                 max_pixels=28 * 28 * int(run_cfg.get("synthetic_max_image_tokens", 1280)))
        model.mllama_processor = model.mllama_tokenizer = chat
        model.image_processor = chat.image_processor
        seed = run_cfg.get("seed", 0)
        if model.visual is None:
            tiny = run_cfg.get("synthetic_tiny", False)
            kw = dict(embed_dim=320, depth=2, num_heads=4, mlp_ratio=2) if tiny else {}
            model.visual = HipQwen2VisionTransformer.from_random(out_hidden=model.mllama.config.hidden_size, seed=seed + 21, device=device, **kw)
        model.mllama.init_random(seed + 22)
        return model
    raise FileNotFoundError("no Qwen2-VL assets: set run.local_weights.qwen2_vl to a local checkpoint directory or run.synthetic: true")

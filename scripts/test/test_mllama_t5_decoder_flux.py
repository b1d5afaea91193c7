"""ThinkDiff-LVLM image + instruction -> FLUX driver on the MI355X path.

Mirror of the reference driver scripts/test/test_mllama_t5_decoder_flux.py:77-196 (run by
runs/test_thinkdiff_lvlm_ccsbu_image_text.sh with configs/test_thinkdiff_lvlm_ccsbu_image_text.yaml): the LVLM reads the
image and the instruction, generates 128 tokens, the hidden states at `model.norm` of the selected tokens go through the
aligner and condition FLUX in place of the T5 embeddings; the pooled vector is CLIP("").  Same config keys
(`model.embedding_type`, `model.vllm_config.*`, `run.output_dir`, `run.seed`), same default image / instruction, same output
name `<image>_output_embed_flux_<i>.png` (PNG compress_level=1).  Stages run on libthinkdiff_hip.so: Qwen2-VL ViT + decoder
engine, aligner, CLIP text encoder, FLUX engine, VAE.

    python -m scripts.test.test_mllama_t5_decoder_flux --cfg-path configs/test_thinkdiff_lvlm_ccsbu_image_text.yaml \
        [--options run.synthetic=true run.img_urls=[a.jpg] run.answers=["..."] run.flux_height=512 ...]
"""
import argparse
import os
import random
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))

import thinkdiff.models  # noqa: E402,F401  (registers the archs)
from thinkdiff import tasks  # noqa: E402
from thinkdiff.common.config import Config  # noqa: E402
from thinkdiff.common.dist_utils import get_rank, init_distributed_mode  # noqa: E402
from thinkdiff.runners import dp_inference as dp  # noqa: E402
from thinkdiff.models import providers  # noqa: E402
from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt  # noqa: E402
from thinkdiff.models.flux_transformer import FluxTransformerConfig  # noqa: E402

DEFAULT_URLS = ["assets/dot_image.jpeg"]                                                             # reference :118
DEFAULT_ANSWERS = ["Create an diffusion prompt for a dog in the style of this picture. Do not use 'similar to the image'"]   # :122


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="ThinkDiff-LVLM + FLUX inference")
    p.add_argument("--cfg-path", required=True)
    p.add_argument("--options", nargs="+", help="override settings: key=value ...")
    return p.parse_args(argv)


def setup_seeds(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class LvlmFluxDriver:
    def __init__(self, cfg):
        self.cfg, run = cfg, cfg.run_cfg
        self.device = run.get("device", "cuda")
        self.model = tasks.setup_task(cfg).build_model(cfg).eval()
        providers.load_lvlm_frontend(run, self.model, self.device)
        lw = run.get("local_weights", None) or {}
        if lw.get("flux", None):
            self.pipe = FluxPipelineRewritePrompt.from_pretrained(lw["flux"], torch_dtype=torch.bfloat16).to(self.device)
        elif run.get("synthetic", False):
            fc = FluxTransformerConfig(num_layers=1, num_single_layers=1, num_attention_heads=4) if run.get("synthetic_tiny", False) else None
            self.pipe = FluxPipelineRewritePrompt.from_random(fc, seed=run.seed, max_txt_tokens=512)
        else:
            raise FileNotFoundError("no FLUX weights: set run.local_weights.flux to a local diffusers directory or run.synthetic: true")
        self.text = providers.load_text_encoders(run, self.pipe, self.device)
        self.pipe.transformer.set_precision(run.get("flux_precision", "bf16"), fp8_gemms=(list(run.get("flux_fp8_gemms")) if run.get("flux_fp8_gemms", None) else None),
                                            act_scales=run.get("flux_act_scales", "dynamic"), smoothing=bool(run.get("flux_smoothing", False)))
        self.pipe.transformer.set_attention(run.get("flux_attention", "bf16"))      # "fp8": QK^T / P.V on the e4m3 MFMA (8-bit modes; td_flux_set_attention)
        self.pipe.images_in_flight = max(1, int(run.get("images_in_flight", 2)))
        self.pipe.set_progress_bar_config(disable=False)
        from concurrent.futures import ThreadPoolExecutor
        self._saver, self._pending_saves = ThreadPoolExecutor(max_workers=2), []

    def pooled_empty_prompt(self):
        """encode_prompt(prompt="", prompt_embeds=...) computes only the CLIP pooled vector (reference :173-178)."""
        if self.text is not None:
            return self.text.clip_pooled("", self.device)
        return self.pipe.encode_prompt(prompt="", prompt_2=None, prompt_embeds=torch.zeros(1, 1, 1))[1]

    def _render(self, members, language_model_inputs, pooled, names, out_dir, gens=None):
        """One pipeline call for requests with equal token counts (they advance together on the engine's contexts)."""
        run = self.cfg.run_cfg
        h, w = run.get("flux_height", 1024), run.get("flux_width", 1024)
        steps = run.get("flux_num_inference_steps", 28)
        written = []
        with torch.no_grad():
            lat = torch.cat([self.pipe.prepare_latents(1, h, w, generator=None if gens is None else gens[k])[0] for k in range(len(members))])
            pe = torch.stack([language_model_inputs[i].to(torch.bfloat16) for i in members])
            outs = self.pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled.expand(len(members), -1).contiguous(), num_images_per_prompt=1,
                             height=h, width=w, num_inference_steps=steps, guidance_scale=run.get("guidance_scale", 3.5), latents=lat).images
        for i, image in zip(members, outs):
            path = f"{out_dir}/{names[i]}_output_embed_flux_0.png"
            self._pending_saves.append(self._saver.submit(self._save_png, image, path))   # PNG encoding overlaps the next group's GPU work
            written.append(path)
        return written

    @staticmethod
    def _save_png(image, path):
        image.save(path, format="PNG", compress_level=1)
        print(f"Saved image to {path}")

    def _drain_saves(self):
        for f in self._pending_saves:
            f.result()
        self._pending_saves.clear()

    def run(self):
        run = self.cfg.run_cfg
        out_dir = run["output_dir"]
        os.makedirs(out_dir, exist_ok=True)
        urls = list(run.get("img_urls", None) or DEFAULT_URLS)
        answers = list(run.get("answers", None) or DEFAULT_ANSWERS)
        names = [u.split("/")[-1].split(".")[0] for u in urls]
        embedding_type = self.cfg.model_cfg.get("embedding_type", "output_embed")
        pooled = self.pooled_empty_prompt().to(torch.bfloat16)
        if run.get("shard_prompts", False):
            # rank 0 broadcasts [(request index, seed)], every rank takes work[rank::world]; each request is
            # sampled and rendered under its own seed (seed + index), so the images do not depend on the world size
            work = dp.shard(dp.broadcast_work_list([(i, run.seed + i) for i in range(len(urls))] if get_rank() == 0 else None))
            written = []
            for i, seed in work:
                setup_seeds(seed)
                sample = {"images": [[Image.open(urls[i]).convert("RGB")]], "answers": [answers[i]]}
                with torch.no_grad():
                    lm_in, generated = self.model.get_embed(sample, embedding_type=embedding_type, max_new_tokens=128)
                print(lm_in[0].shape, answers[i], generated[0], sep="\n")
                gen = torch.Generator(device=self.pipe._execution_device).manual_seed(int(seed))
                written += self._render([i], {i: lm_in[0]}, pooled, names, out_dir, gens=[gen])
            self._drain_saves()
            every = dp.gather_results(written)
            return every if every is not None else written
        images = [[Image.open(u).convert("RGB")] for u in urls]
        sample = {"images": images, "answers": answers}
        with torch.no_grad():
            language_model_inputs, generated = self.model.get_embed(sample, embedding_type=embedding_type, max_new_tokens=128)
        for i, text in enumerate(generated):
            print(language_model_inputs[i].shape)
            print(answers[i])
            print(text)
        print(urls)
        written = []
        G = self.pipe.images_in_flight
        for g0 in range(0, len(urls), G):            # requests with equal token counts advance together
            groups = {}
            for i in range(g0, min(g0 + G, len(urls))):
                groups.setdefault(language_model_inputs[i].shape[0], []).append(i)
            for members in groups.values():
                written += self._render(members, language_model_inputs, pooled, names, out_dir)
        self._drain_saves()
        return written


def main(argv=None):
    args = parse_args(argv)
    cfg = Config(args)
    init_distributed_mode(cfg.run_cfg)
    setup_seeds(cfg.run_cfg.seed + get_rank())
    cfg.pretty_print()
    return LvlmFluxDriver(cfg).run()


if __name__ == "__main__":
    main()

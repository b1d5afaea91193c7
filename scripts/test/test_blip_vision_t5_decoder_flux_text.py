"""ThinkDiff-CLIP image(+image)+text -> FLUX driver on the MI355X path.

Same command line, config keys, prompt/concat order, output naming and skip rules as the reference driver
(scripts/test/test_blip_vision_t5_decoder_flux_text.py:84-324 and the two-image variant
scripts/test/test_blip_vision_t5_decoder_flux.py:156-234); the stages run on libthinkdiff_hip.so.

    python -m scripts.test.test_blip_vision_t5_decoder_flux_text --cfg-path configs/test_thinkdiff_clip_image_text.yaml \
        [--options run.synthetic=true run.flux_height=256 run.flux_width=256 run.flux_num_inference_steps=4]
"""
import argparse
import json
import os
import random
import re
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
from thinkdiff.models import providers  # noqa: E402
from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt  # noqa: E402
from thinkdiff.models.flux_transformer import FluxTransformerConfig  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="ThinkDiff-CLIP + FLUX inference")
    p.add_argument("--cfg-path", required=True)
    p.add_argument("--options", nargs="+", help="override settings: key=value ...")
    return p.parse_args(argv)


def setup_seeds(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class ClipFluxDriver:
    def __init__(self, cfg):
        self.cfg, run = cfg, cfg.run_cfg
        self.seed = run.seed + get_rank()
        self.device = run.get("device", "cuda")
        self.model = tasks.setup_task(cfg).build_model(cfg).eval().to(self.device, torch.bfloat16)
        self.processor, self.model.vision_model = providers.load_vision(run, self.device)
        lw = run.get("local_weights", None) or {}
        if lw.get("flux", None):
            self.pipe = FluxPipelineRewritePrompt.from_pretrained(lw["flux"], torch_dtype=torch.bfloat16).to(self.device)
        elif run.get("synthetic", False):
            tiny = run.get("synthetic_tiny", False)
            fc = FluxTransformerConfig(num_layers=1, num_single_layers=1, num_attention_heads=4) if tiny else None
            self.pipe = FluxPipelineRewritePrompt.from_random(fc, seed=run.seed, max_txt_tokens=512)
        else:
            raise FileNotFoundError("no FLUX weights: set run.local_weights.flux to a local diffusers directory or run.synthetic: true")
        self.text = providers.load_text_encoders(run, self.pipe, self.device)
        self.pipe.set_progress_bar_config(disable=True)
        from concurrent.futures import ThreadPoolExecutor
        self._saver, self._pending_saves = ThreadPoolExecutor(max_workers=2), []
        self.pipe.transformer.set_precision(run.get("flux_precision", "bf16"))   # "fp8": BASELINE config 5's e4m3 block GEMMs

    # ---- config surface (reference :117-161) ---------------------------------------------------------------
    def resolve_inputs(self):
        run = self.cfg.run_cfg
        if run.get("img_folder", None):
            urls = sorted(os.path.join(run.img_folder, n) for n in os.listdir(run.img_folder))
            urls = [u for u in urls if os.path.isfile(u) and u.endswith((".png", ".jpg"))]
        elif run.get("img_json", None):
            with open(run.img_json) as fh:
                urls = json.load(fh)
        else:
            urls = run["img_urls"]
        if run.get("img_urls_len", None):
            urls = urls[: run["img_urls_len"]]
        stem = lambda u: u.split("/")[-1].split(".")[0]
        names = run.get("image_names", None) or ["_".join(stem(s) for s in u) if type(u) == list else stem(u) for u in urls]
        if run.get("prompt_json", None):
            with open(run.prompt_json) as fh:
                return urls, names, json.load(fh), None
        return urls, names, run["questions"], run["questions_names"]

    def output_path(self, out_dir, image_name, prompt, prompt_name=None):
        run = self.cfg.run_cfg
        if prompt_name is not None:
            return f"{out_dir}/{image_name}_{prompt_name}_clip_t5_flux_seed_{self.seed}.png"
        if run.get("use_image_name_as_output_name", False):
            return f"{out_dir}/{image_name}.png"
        if run.get("use_image_name_and_prompt_as_output_name", False):
            p = re.sub(r"\s+", "_", re.sub(r"[^\w\s-]", "", prompt))
            return f"{out_dir}/{image_name}_{p}.png"
        return f"{out_dir}/{image_name}_clip_t5_flux_seed_{self.seed}.png"

    # ---- stages ------------------------------------------------------------------------------------------------
    def aligner_tokens(self, path, prompt):
        inputs = self.processor(Image.open(path), prompt, return_tensors="pt")
        return self.model.forward_encoder(pixel_values=inputs["pixel_values"].to(self.device, torch.bfloat16),
                                          input_ids=inputs.get("input_ids", None))

    def text_tokens(self, prompt, max_len):
        if self.text is not None:
            return self.text.t5(prompt, max_len, self.device), self.text.clip_pooled(prompt, self.device)
        pe, pooled, _ = self.pipe.encode_prompt(prompt=prompt, prompt_2=None, max_sequence_length=max_len)
        return pe, pooled

    def condition(self, img_url, prompt):
        """[aligner(img1), aligner(img2) ..., T5(prompt)] along the token axis + the CLIP pooled vector (reference :221-233)."""
        run = self.cfg.run_cfg
        urls = img_url if type(img_url) == list else [img_url]
        vis = [self.aligner_tokens(u, prompt) for u in urls]                       # [1,65,4096] each
        t5, pooled = self.text_tokens(prompt, run["flux_max_sequence_length"])
        return torch.cat(vis + [t5], dim=1).to(torch.bfloat16), pooled.to(torch.bfloat16)   # visual tokens first, then T5

    def render_group(self, jobs):
        """jobs: [(img_url, prompt, out_path)] with equal token counts.  One pipeline call; the images advance concurrently
        on the engine's forked contexts.  Latents are drawn job by job from the global generator, i.e. exactly the draws
        the reference's one-call-per-image loop makes, so grouping does not change any image."""
        run = self.cfg.run_cfg
        with torch.no_grad():
            conds = [self.condition(u, p) for u, p, _ in jobs]
            lat = torch.cat([self.pipe.prepare_latents(1, run["flux_height"], run["flux_width"])[0] for _ in jobs])
            images = self.pipe(prompt_embeds=torch.cat([c[0] for c in conds]), pooled_prompt_embeds=torch.cat([c[1] for c in conds]),
                               num_images_per_prompt=1, height=run["flux_height"], width=run["flux_width"], latents=lat,
                               num_inference_steps=run["flux_num_inference_steps"], guidance_scale=run["guidance_scale"]).images
        for img, (_, _, out_path) in zip(images, jobs):
            self._pending_saves.append(self._saver.submit(self._save_png, img, out_path))   # PNG encoding overlaps the next group's GPU work

    @staticmethod
    def _save_png(img, out_path):
        img.save(out_path, format="PNG", compress_level=1)        # reference :247 / :322
        print(f"Image saved to {out_path}")

    def render(self, img_url, prompt, out_path):
        self.render_group([(img_url, prompt, out_path)])
        self._drain_saves()

    def _drain_saves(self):
        for f in self._pending_saves:
            f.result()
        self._pending_saves.clear()

    def run(self):
        run = self.cfg.run_cfg
        out_dir = run["output_dir"]
        os.makedirs(out_dir, exist_ok=True)
        urls, names, questions, q_names = self.resolve_inputs()
        G = max(1, int(run.get("images_in_flight", 3)))      # images rendered per pipeline call (MI355X: fills kernel tails)
        self.pipe.images_in_flight = G
        written, pending = [], []

        def flush():
            if pending:
                self.render_group(list(pending))
                written.extend(j[2] for j in pending)
                pending.clear()

        for i, url in enumerate(urls):
            jobs = [(questions[names[i]], None)] if q_names is None else list(zip(questions, q_names))
            for prompt, pname in jobs:
                path = self.output_path(out_dir, names[i], prompt, pname)
                if os.path.exists(path) or any(path == j[2] for j in pending):
                    print(f"Image already exists at {path}")
                    continue
                n_img = len(url) if type(url) == list else 1
                if pending and (len(pending[0][0]) if type(pending[0][0]) == list else 1) != n_img:
                    flush()                                    # a group shares one token count
                pending.append((url, prompt, path))
                if len(pending) == G:
                    flush()
        flush()
        self._drain_saves()
        return written


def main(argv=None):
    args = parse_args(argv)
    cfg = Config(args)
    init_distributed_mode(cfg.run_cfg)
    setup_seeds(cfg.run_cfg.seed + get_rank())
    cfg.pretty_print()
    return ClipFluxDriver(cfg).run()


if __name__ == "__main__":
    main()

"""ThinkDiff-CLIP image(+image)+text -> FLUX driver on the MI355X path.

Same command line, config keys, prompt/concat order, output naming and skip rules as the reference driver
(scripts/test/test_blip_vision_t5_decoder_flux_text.py:84-324); the two-image driver
(scripts/test/test_blip_vision_t5_decoder_flux.py:156-234) is the `TWO_IMAGE_DRIVER` flavour of the same class and has
its own entry module beside this one.  The stages run on libthinkdiff_hip.so.

    python -m scripts.test.test_blip_vision_t5_decoder_flux_text --cfg-path configs/test_thinkdiff_clip_image_text.yaml \
        [--options run.synthetic=true run.flux_height=256 run.flux_width=256 run.flux_num_inference_steps=4]

Multi-GPU (one process per GPU under torchrun).  Default = the reference's semantics: every rank renders the whole job
list with seed + rank (scripts/test/test_mllama_t5_decoder_flux.py:57-65).  `run.shard_prompts=true` = SURVEY.md 8(e):
rank 0 plans the job list with one seed per job (seed + job index, so the images do not depend on the world size),
broadcasts it (RCCL one-to-all), every rank renders `jobs[rank::world]`, and the written paths are gathered on rank 0.
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
from thinkdiff.common.dist_utils import get_rank, get_world_size, init_distributed_mode  # noqa: E402
from thinkdiff.runners import dp_inference as dp  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="ThinkDiff-CLIP + FLUX inference")
    p.add_argument("--cfg-path", required=True)
    p.add_argument("--options", nargs="+", help="override settings: key=value ...")
    return p.parse_args(argv)


def setup_seeds(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


# ---- host logic of the drivers (no device work; covered by tests/test_driver_logic_cpu.py) ------------------------------
def resolve_inputs(run, two_image_driver=False):
    """-> (img_urls, image_names, questions, questions_names | None).  Reference ..._flux_text.py:120-161: image list from
    `img_folder` / `img_json` / `img_urls` (+ `img_urls_len`), names from the file stems (joined with "_" for a list of
    images), prompts from `prompt_json` (dict keyed by image name) or `questions` x `questions_names`.  The two-image
    driver reads `questions` only (..._flux.py:145-146)."""
    if run.get("img_folder", None):
        urls = sorted(os.path.join(run["img_folder"], n) for n in os.listdir(run["img_folder"]))   # listdir order is arbitrary: fixed here
        urls = [u for u in urls if os.path.isfile(u) and (u.endswith(".png") or u.endswith(".jpg"))]
    elif run.get("img_json", None):
        with open(run["img_json"]) as fh:
            urls = json.load(fh)
    else:
        urls = run["img_urls"]
    if run.get("img_urls_len", None):
        urls = urls[: run["img_urls_len"]]
    stem = lambda u: u.split("/")[-1].split(".")[0]
    if run.get("image_names", None) is None:
        names = ["_".join(stem(s) for s in u) if type(u) == list else stem(u) for u in urls]
    else:
        names = run["image_names"]
    if not two_image_driver and run.get("prompt_json", None):
        with open(run["prompt_json"]) as fh:
            return urls, names, json.load(fh), None
    return urls, names, run["questions"], run["questions_names"]


def output_path(run, out_dir, image_name, prompt, seed, prompt_name=None, two_image_driver=False):
    """The reference's naming rules, string for string.
    prompt_json mode (..._flux_text.py:171-178): `{image}.png` | `{image}_{sanitised prompt}.png` | `{image}_clip_t5_flux_seed_{seed}.png`;
    questions mode (..._flux_text.py:254): `{image}_clip_t5_flux_{name}_seed_{seed}.png` -- the text driver ignores
    `use_image_name_as_output_name` there, the two-image driver honours it (..._flux.py:161-164)."""
    if prompt_name is not None:
        if two_image_driver and run.get("use_image_name_as_output_name", False):
            return f"{out_dir}/{image_name}.png"
        return f"{out_dir}/{image_name}_clip_t5_flux_{prompt_name}_seed_{seed}.png"
    if run.get("use_image_name_as_output_name", False):
        return f"{out_dir}/{image_name}.png"
    if run.get("use_image_name_and_prompt_as_output_name", False):
        prompt_name = re.sub(r"[^\w\s-]", "", prompt)
        prompt_name = re.sub(r"\s+", "_", prompt_name)
        return f"{out_dir}/{image_name}_{prompt_name}.png"
    return f"{out_dir}/{image_name}_clip_t5_flux_seed_{seed}.png"


def plan_jobs(run, base_seed, two_image_driver=False, per_job_seeds=False, exists=os.path.exists):
    """The (image, prompt) loop of the reference as a list of jobs in its iteration order:
    [{"url", "prompt", "path", "seed", "index"}], existing outputs skipped (reference :180-182, :256-258).
    per_job_seeds (sharded mode): job k of the full loop gets seed base_seed + k, whether or not earlier jobs were skipped."""
    out_dir = run["output_dir"]
    urls, names, questions, q_names = resolve_inputs(run, two_image_driver)
    jobs, k = [], 0
    for i, url in enumerate(urls):
        pairs = [(questions[names[i]], None)] if q_names is None else list(zip(questions, q_names))
        for prompt, pname in pairs:
            seed = base_seed + k if per_job_seeds else base_seed
            path = output_path(run, out_dir, names[i], prompt, seed, pname, two_image_driver)
            if exists(path) or any(path == j["path"] for j in jobs):
                print(f"Image already exists at {path}")
            else:
                jobs.append({"url": url, "prompt": prompt, "path": path, "seed": seed, "index": k})
            k += 1
    return jobs


def n_images(job):
    return len(job["url"]) if type(job["url"]) == list else 1


class ClipFluxDriver:
    TWO_IMAGE_DRIVER = False      # True: scripts/test/test_blip_vision_t5_decoder_flux.py (questions only, default PNG compression)
    PNG_SAVE_KW = dict(format="PNG", compress_level=1)      # reference ..._flux_text.py:247 / :322

    def __init__(self, cfg):
        from thinkdiff.models import providers
        from thinkdiff.models.flux_prompt import FluxPipelineRewritePrompt
        from thinkdiff.models.flux_transformer import FluxTransformerConfig
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
        self.pipe.transformer.set_precision(run.get("flux_precision", "bf16"), fp8_gemms=(list(run.get("flux_fp8_gemms")) if run.get("flux_fp8_gemms", None) else None),
                                            act_scales=run.get("flux_act_scales", "dynamic"), smoothing=bool(run.get("flux_smoothing", False)))   # "fp8": BASELINE config 5's e4m3 block GEMMs
        self.pipe.transformer.set_attention(run.get("flux_attention", "bf16"))      # "fp8": QK^T / P.V on the e4m3 MFMA (8-bit modes; td_flux_set_attention)
        self._init_savers()

    def _init_savers(self):
        from concurrent.futures import ThreadPoolExecutor
        self._saver, self._pending_saves = ThreadPoolExecutor(max_workers=2), []

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

    def render_group(self, jobs, per_job_seeds=False):
        """jobs with equal token counts.  One pipeline call; the images advance concurrently on the engine's forked
        contexts.  Replica mode: latents are drawn job by job from the global generator, i.e. exactly the draws the
        reference's one-call-per-image loop makes, so grouping does not change any image.  Sharded mode: each job draws
        from its own generator seeded with the job's seed."""
        run = self.cfg.run_cfg
        with torch.no_grad():
            conds = [self.condition(j["url"], j["prompt"]) for j in jobs]
            gens = [torch.Generator(device=self.pipe._execution_device).manual_seed(int(j["seed"])) if per_job_seeds else None for j in jobs]
            lat = torch.cat([self.pipe.prepare_latents(1, run["flux_height"], run["flux_width"], generator=g)[0] for g in gens])
            images = self.pipe(prompt_embeds=torch.cat([c[0] for c in conds]), pooled_prompt_embeds=torch.cat([c[1] for c in conds]),
                               num_images_per_prompt=1, height=run["flux_height"], width=run["flux_width"], latents=lat,
                               num_inference_steps=run["flux_num_inference_steps"], guidance_scale=run["guidance_scale"]).images
        for img, j in zip(images, jobs):
            self._pending_saves.append(self._saver.submit(self._save_png, img, j["path"]))   # PNG encoding overlaps the next group's GPU work

    def _save_png(self, img, out_path):
        img.save(out_path, **self.PNG_SAVE_KW)
        print(f"Image saved to {out_path}")

    def _drain_saves(self):
        for f in self._pending_saves:
            f.result()
        self._pending_saves.clear()

    def run(self):
        run = self.cfg.run_cfg
        os.makedirs(run["output_dir"], exist_ok=True)
        sharded = bool(run.get("shard_prompts", False))
        if sharded:
            # rank 0 plans (one stat per output on the shared filesystem), everyone receives the same list
            jobs = plan_jobs(run, run.seed, self.TWO_IMAGE_DRIVER, per_job_seeds=True) if get_rank() == 0 else None
            jobs = dp.shard(dp.broadcast_work_list(jobs))
        else:
            jobs = plan_jobs(run, self.seed, self.TWO_IMAGE_DRIVER)
        G = max(1, int(run.get("images_in_flight", 2)))      # images rendered per pipeline call (MI355X: fills kernel tails)
        if hasattr(self, "pipe"):
            self.pipe.images_in_flight = G
        written, pending = [], []

        def flush():
            if pending:
                self.render_group(list(pending), per_job_seeds=sharded)
                written.extend(j["path"] for j in pending)
                pending.clear()

        for j in jobs:
            if pending and n_images(pending[0]) != n_images(j):
                flush()                                    # a group shares one token count
            pending.append(j)
            if len(pending) == G:
                flush()
        flush()
        self._drain_saves()
        if sharded:
            every = dp.gather_results(written)             # work order, on rank 0
            return every if every is not None else written
        return written


def main(argv=None, driver_cls=ClipFluxDriver):
    args = parse_args(argv)
    cfg = Config(args)
    init_distributed_mode(cfg.run_cfg)
    setup_seeds(cfg.run_cfg.seed + get_rank())
    cfg.pretty_print()
    return driver_cls(cfg).run()


if __name__ == "__main__":
    main()

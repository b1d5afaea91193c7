"""ThinkDiff-LVLM embedding export: a folder of images -> aligner outputs on disk (no FLUX call), on the MI355X path.

Mirror of the reference driver scripts/test/test_mllama_t5_decoder_flux_embed.py:130-212: every `.png` / `.jpg` in
`run.image_folder` goes through `get_embed({"images": [[img]], "answers": [run.prompt]}, max_new_tokens=128)`; the first
request's aligner output is written as `{output_dir}/{image_name}.pth` (the bytes `torch.save` produces for the CPU tensor,
:195-200) and `{image_name}.json` = the input's side-car json (`<image path up to the first '.'>.json`, :191-193) plus
`generated_text` and `prompt`, `indent=4` (:203-206).  An image is skipped when `{output_dir}/{image_name}.png` exists (:147-150
-- the reference tests the .png name although it writes .pth; kept, so a directory of rendered images masks its inputs the
same way).  The FLUX pipeline the reference loads and never calls is not loaded here.

    python -m scripts.test.test_mllama_t5_decoder_flux_embed --cfg-path <lvlm yaml> \
        --options run.image_folder=<dir> run.prompt="..." [run.synthetic=true]
"""
import io
import json
import os
import sys

import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))

import thinkdiff.models  # noqa: E402,F401  (registers the archs)
from scripts.test.test_mllama_t5_decoder_flux import parse_args, setup_seeds  # noqa: E402
from thinkdiff import tasks  # noqa: E402
from thinkdiff.common.config import Config  # noqa: E402
from thinkdiff.common.dist_utils import get_rank, init_distributed_mode  # noqa: E402
from thinkdiff.models import providers  # noqa: E402
from thinkdiff.runners import dp_inference as dp  # noqa: E402


def list_inputs(folder, suffixes):
    """Files of `folder` ending in one of `suffixes`, in os.listdir order (reference :133-135)."""
    urls = [os.path.join(folder, name) for name in os.listdir(folder)]
    return [u for u in urls if os.path.isfile(u) and u.endswith(tuple(suffixes))]


def stem(url):
    """The reference's image_name (:145): last path component up to its first '.'."""
    return url.split("/")[-1].split(".")[0]


def sidecar_json(url):
    """Reference :191: the whole path cut at its FIRST '.', + '.json' (a dotted directory name cuts there too)."""
    return url.split(".")[0] + ".json"


def save_embed(out_dir, name, embed, json_dict, generated_text, prompt):
    """Reference :188-206.  Returns the two paths."""
    embed_path, json_path = f"{out_dir}/{name}.pth", f"{out_dir}/{name}.json"
    buffer = io.BytesIO()
    torch.save(embed.cpu(), buffer)
    with open(embed_path, "wb") as f:
        f.write(buffer.getvalue())
    json_dict["generated_text"] = generated_text
    json_dict["prompt"] = prompt
    with open(json_path, "w") as f:
        json.dump(json_dict, f, indent=4)
    return embed_path, json_path


class LvlmEmbedExportDriver:
    INPUT_SUFFIXES = (".png", ".jpg")
    SKIP_SUFFIX = ".png"                       # reference :146-147

    def __init__(self, cfg):
        self.cfg, run = cfg, cfg.run_cfg
        self.device = run.get("device", "cuda")
        self.model = tasks.setup_task(cfg).build_model(cfg).eval()
        providers.load_lvlm_frontend(run, self.model, self.device)

    def pending(self, urls, out_dir):
        todo = []
        for url in urls:
            done = f"{out_dir}/{stem(url)}{self.SKIP_SUFFIX}"
            if os.path.exists(done):
                print(f"Image already exists at {done}")
                continue
            todo.append(url)
        return todo

    def request(self, url):
        """-> (sample for get_embed, need_process, the json the output json starts from)."""
        with open(sidecar_json(url), "r") as f:
            json_dict = json.load(f)
        return {"images": [[Image.open(url).convert("RGB")]], "answers": [self.cfg.run_cfg["prompt"]]}, True, json_dict

    def run(self):
        run = self.cfg.run_cfg
        out_dir = run["output_dir"]
        os.makedirs(out_dir, exist_ok=True)
        embedding_type = self.cfg.model_cfg.get("embedding_type", "output_embed")
        written = []
        todo = self.pending(list_inputs(run["image_folder"], self.INPUT_SUFFIXES), out_dir)
        sharded = bool(run.get("shard_prompts", False))
        if sharded:
            # rank 0 plans the pending list (one seed per job, so an embedding does not depend on the world size),
            # broadcast (RCCL one-to-all), every rank takes jobs[rank::world]; the written paths are gathered on rank 0 in job order.
            # Default (false) = the reference's replicas: every rank walks the whole folder with seed + rank.
            todo = dp.shard(dp.broadcast_work_list([(u, run.seed + i) for i, u in enumerate(todo)] if get_rank() == 0 else None))
        for job in todo:
            url = job[0] if sharded else job
            if sharded:
                setup_seeds(job[1])
            sample, need_process, json_dict = self.request(url)
            with torch.no_grad():
                lm_in, generated = self.model.get_embed(sample, embedding_type=embedding_type, max_new_tokens=128, need_process=need_process)
            for i, text in enumerate(generated):
                print(lm_in[i].shape)
                print(text)
            paths = save_embed(out_dir, stem(url), lm_in[0], json_dict, generated[0], run["prompt"])
            print(f"Saved embed to {paths[0]}")
            written += paths
        if sharded:
            every = dp.gather_results([written])
            return [p for w in every for p in w] if every is not None else written
        return written


def main(argv=None, driver_cls=LvlmEmbedExportDriver):
    args = parse_args(argv)
    cfg = Config(args)
    init_distributed_mode(cfg.run_cfg)
    setup_seeds(cfg.run_cfg.seed + get_rank())
    cfg.pretty_print()
    return driver_cls(cfg).run()


if __name__ == "__main__":
    main()

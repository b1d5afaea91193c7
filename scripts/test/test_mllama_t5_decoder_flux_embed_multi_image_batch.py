"""Batched form of the interleaved word / picture embedding export, on the MI355X path.

Mirror of the reference driver scripts/test/test_mllama_t5_decoder_flux_embed_multi_image_batch.py:143-268: the task jsons
are taken `run.batch_size` at a time (tasks whose `.pth` exists drop out of their batch, :154-157), image paths are optionally
re-rooted under `run.image_path_prefix` (:171-177), one `get_embed(list of requests, need_process=False)` call per batch
(:239-241) -- here the requests of a batch decode together on the Qwen2-VL engine's batch slots -- and one `.pth` / `.json` pair
per request (:244-268).  No FLUX pipeline is involved (the reference comments its load out, :124-130).

    python -m scripts.test.test_mllama_t5_decoder_flux_embed_multi_image_batch --cfg-path <lvlm yaml> \
        --options run.image_folder=<dir> run.prompt="..." run.batch_size=4 [run.image_path_prefix=<root>]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))

from scripts.test.test_mllama_t5_decoder_flux_embed import list_inputs, main as _main, save_embed, stem  # noqa: E402
from scripts.test.test_mllama_t5_decoder_flux import setup_seeds  # noqa: E402
from scripts.test.test_mllama_t5_decoder_flux_embed_multi_image import LvlmMultiImageEmbedExportDriver  # noqa: E402
from thinkdiff.common.dist_utils import get_rank  # noqa: E402
from thinkdiff.runners import dp_inference as dp  # noqa: E402


def batches(urls, batch_size, exists):
    """Reference :145-157: fixed windows over the listing; finished tasks leave their window (windows do not refill)."""
    out = []
    for b0 in range(0, len(urls), batch_size):
        out.append([u for u in urls[b0:b0 + batch_size] if not exists(u)])
    return out


class LvlmMultiImageEmbedBatchExportDriver(LvlmMultiImageEmbedExportDriver):
    IMAGE_PATH_PREFIX_KEY = "image_path_prefix"

    def run(self):
        run = self.cfg.run_cfg
        out_dir = run["output_dir"]
        os.makedirs(out_dir, exist_ok=True)
        embedding_type = self.cfg.model_cfg.get("embedding_type", "output_embed")
        urls = list_inputs(run["image_folder"], self.INPUT_SUFFIXES)
        sharded = bool(run.get("shard_prompts", False))
        if sharded:     # rank 0's task list, broadcast; rank r takes tasks[r::world] and batches its own share
            urls = dp.shard(dp.broadcast_work_list(urls if get_rank() == 0 else None))
            setup_seeds(run.seed)                               # the same sampling stream on every rank: a task's text depends on its batch only
        written = []
        for group in batches(urls, run["batch_size"], lambda u: os.path.exists(f"{out_dir}/{stem(u)}.pth")):
            if not group:
                continue                                       # the reference would hand vLLM an empty list here
            reqs = [self.request(u) for u in group]
            with torch.no_grad():
                lm_in, generated = self.model.get_embed([r[0] for r in reqs], embedding_type=embedding_type, max_new_tokens=128, need_process=False)
            for i, url in enumerate(group):
                print(generated[i])
                print(lm_in[i].shape)
                paths = save_embed(out_dir, stem(url), lm_in[i], reqs[i][2], generated[i], run["prompt"])
                print(f"Saved embed to {paths[0]}")
                print(f"Saved json to {paths[1]}")
                written += paths
        if sharded:
            every = dp.gather_results([written])
            return [p for w in every for p in w] if every is not None else written
        return written


def main(argv=None, driver_cls=LvlmMultiImageEmbedBatchExportDriver):
    return _main(argv, driver_cls)


if __name__ == "__main__":
    main()

"""Embedding-precompute task (reference thinkdiff/tasks/image_text_process_data.py:17-119): iterate the loader, run the
model, write one WebDataset sample per input: {__key__, jpg, json (+generated_text, input_prompt,
input_prompt_token_ids, output_text, output_token_ids), "<layer>.output_embed.pth", "<layer>.input_embed.pth"}."""
import io
import os

import torch

from ..common.registry import registry
from ..datasets.wds_io import ShardWriter
from . import BaseTask


def flatten_dict(d, parent_key="", sep="."):
    items = {}
    for k, v in d.items():
        nk = f"{parent_key}{sep}{k}" if parent_key else k
        if isinstance(v, dict):
            items.update(flatten_dict(v, nk, sep=sep))
        else:
            items[nk] = v
    return items


@registry.register_task("image_text_process_data")
class ImageTextProcessDataTask(BaseTask):
    def build_datasets(self, cfg):
        """`datasets.cc_sbu_mllama_vllm_process_wids.build_info.storage` = the wids shard index (reference builder
        thinkdiff/datasets/builders/image_text_pair_builder.py + datasets/datasets/cc_sbu_dataset_mllama_vllm_process_wids.py:36-63);
        the shard list is split by rank here (one process per GPU, no data-path collective)."""
        from ..common.dist_utils import get_rank, get_world_size
        from ..datasets.cc_sbu_process import CCSBUMllamaVllmProcessDatasetWids
        out = {}
        for name in cfg.datasets_cfg:
            dc = cfg.datasets_cfg[name]
            if name == "cc_sbu_mllama_vllm_process_wids":
                out[name] = CCSBUMllamaVllmProcessDatasetWids(dc["build_info"]["storage"], rank=get_rank(), world=get_world_size())
        return out

    def _train_inner_loop(self, epoch, iters_per_epoch, model, data_loader, optimizer=None, lr_scheduler=None, scaler=None,
                          start_iters=None, log_freq=50, cuda_enabled=False, accum_grad_iters=1, amp_dtype=torch.bfloat16,
                          use_clip_grad_norm=False, max_grad_norm=1.0, output_shard_path=None, maxsize=(10 ** 8) * 5):
        assert output_shard_path
        os.makedirs(output_shard_path[0], exist_ok=True)
        pattern = os.path.join(output_shard_path[0], output_shard_path[1])
        n_written = 0
        with ShardWriter(pattern, maxsize=maxsize, start_shard=output_shard_path[2]) as writer:
            for samples in data_loader:
                batch = len(samples["images"])
                output = model(samples)
                embeds = flatten_dict(output["generated_embed"]) if output.get("generated_embed") is not None else None
                tok = output["generated_token"]
                for i in range(batch):
                    js = samples["jsons"][i]
                    js["generated_text"] = output["generated_text"][i]
                    js["input_prompt"] = tok["input_prompt"][i]
                    js["input_prompt_token_ids"] = tok["input_prompt_token_ids"][i]
                    js["output_text"] = tok["output_text"][i]
                    js["output_token_ids"] = list(tok["output_token_ids"][i])
                    rec = {"__key__": samples["filenames"][i], "jpg": samples["images"][i][0], "json": js}
                    if embeds is not None:
                        for k, v in embeds.items():
                            buf = io.BytesIO()
                            torch.save(v[i].cpu().clone(), buf)
                            rec[f"{k}.pth"] = buf.getvalue()
                    writer.write(rec)
                    n_written += 1
            shards = None
        return {"samples": n_written, "shards": writer.shards}

    def train_epoch(self, epoch, model, data_loader, output_shard_path=None, **kw):
        return self._train_inner_loop(epoch, len(data_loader), model, data_loader, output_shard_path=output_shard_path, **kw)

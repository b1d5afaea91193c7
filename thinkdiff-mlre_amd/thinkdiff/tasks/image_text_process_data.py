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


class _CompactTensor:
    """A row range of a batch-wide host tensor, serialised on a writer thread: torch.save of a view would write the whole
    storage, so the clone (the reference's `.cpu().clone()`, thinkdiff/tasks/image_text_process_data.py:108-112) happens there."""

    def __init__(self, view):
        self.view = view

    def save(self, buf):
        torch.save(self.view.clone(), buf)


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
        from concurrent.futures import ThreadPoolExecutor
        from ..datasets.wds_io import encode_sample
        n_written = 0
        workers = max(1, int(os.environ.get("TD_PRECOMPUTE_WRITER_THREADS", min(16, os.cpu_count() or 1))))
        with ShardWriter(pattern, maxsize=maxsize, start_shard=output_shard_path[2]) as writer, \
                ThreadPoolExecutor(max_workers=workers) as encoders, ThreadPoolExecutor(max_workers=1) as tar_thread:

            def write_batch(records):
                # JPEG / torch.save / json encoding in parallel, the tar appended in sample order by this one thread
                for enc in encoders.map(encode_sample, records):
                    writer.write(enc)

            pending = None
            import time
            timing = os.environ.get("TD_PRECOMPUTE_TIMING") is not None      # where a batch's wall time goes: loader wait | model | host copies | writer wait
            t_load = t_model = t_host = t_wait = 0.0
            t_mark = time.perf_counter()
            for samples in data_loader:
                t0 = time.perf_counter(); t_load += t0 - t_mark
                batch = len(samples["images"])
                output = model(samples)
                t1 = time.perf_counter(); t_model += t1 - t0
                embeds = flatten_dict(output["generated_embed"]) if output.get("generated_embed") is not None else None
                host = {}
                if embeds is not None:
                    for k, v in embeds.items():     # ONE device->host copy per embedding kind, split back into per-sample views
                        rows = [int(t.shape[0]) for t in v]
                        host[k] = torch.split(torch.cat(list(v)).cpu(), rows) if rows else []
                tok = output["generated_token"]
                records = []
                for i in range(batch):
                    js = samples["jsons"][i]
                    js["generated_text"] = output["generated_text"][i]
                    js["input_prompt"] = tok["input_prompt"][i]
                    js["input_prompt_token_ids"] = tok["input_prompt_token_ids"][i]
                    js["output_text"] = tok["output_text"][i]
                    js["output_token_ids"] = list(tok["output_token_ids"][i])
                    rec = {"__key__": samples["filenames"][i], "jpg": samples["images"][i][0], "json": js}
                    for k, v in host.items():
                        rec[f"{k}.pth"] = _CompactTensor(v[i])
                    records.append(rec)
                t2 = time.perf_counter(); t_host += t2 - t1
                if pending is not None:
                    pending.result()                 # at most one batch of records in flight behind the model
                pending = tar_thread.submit(write_batch, records)
                n_written += batch
                t_mark = time.perf_counter(); t_wait += t_mark - t2
            t2 = time.perf_counter()
            if pending is not None:
                pending.result()
            t_wait += time.perf_counter() - t2
            if timing:
                print(f"[precompute timing] {n_written} samples: loader wait {t_load:.2f} s, model {t_model:.2f} s, device->host + records {t_host:.2f} s, "
                      f"writer wait {t_wait:.2f} s", flush=True)
        return {"samples": n_written, "shards": writer.shards}

    def train_epoch(self, epoch, model, data_loader, output_shard_path=None, **kw):
        return self._train_inner_loop(epoch, len(data_loader), model, data_loader, output_shard_path=output_shard_path, **kw)

"""Precompute job entry point (BASELINE config 4): Qwen2-VL generated text + `model.norm` hidden states for every sample of
a WebDataset shard list, written back as WebDataset shards.

Mirror of the reference scripts/generate_embedding_webdataset.py:66-94 (run by runs/run_qwen2_vl_embed_ccsbu.sh with
configs/qwen2_vl_embed_ccsbu.yaml): Config -> init_distributed_mode -> seeds -> task.build_datasets / build_model ->
runner_process_data.train().  One process per GPU; the shard list is split by rank and every rank writes its own shard-number
range (thinkdiff/runners/runner_process_data.py), no data-path collective.

    python -m scripts.generate_embedding_webdataset --cfg-path configs/qwen2_vl_embed_ccsbu.yaml [--options run.synthetic=true ...]
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))

import thinkdiff.models  # noqa: E402,F401  (registers the archs)
import thinkdiff.runners  # noqa: E402,F401
from thinkdiff import tasks  # noqa: E402
from thinkdiff.common.config import Config  # noqa: E402
from thinkdiff.common.dist_utils import get_rank, init_distributed_mode  # noqa: E402
from thinkdiff.common.registry import registry  # noqa: E402
from thinkdiff.models import providers  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Embedding precompute")
    p.add_argument("--cfg-path", required=True, help="path to configuration file.")
    p.add_argument("--options", nargs="+", help="override settings: key=value ...")
    return p.parse_args(argv)


def setup_seeds(config):
    seed = config.run_cfg.seed + get_rank()
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def main(argv=None):
    import datetime
    job_id = datetime.datetime.now().strftime("%Y%m%d%H%M")[:-1]      # reference thinkdiff/common/utils.py:35-38
    cfg = Config(parse_args(argv))
    init_distributed_mode(cfg.run_cfg)
    setup_seeds(cfg)
    cfg.pretty_print()
    task = tasks.setup_task(cfg)
    datasets = task.build_datasets(cfg)
    model = task.build_model(cfg)
    providers.load_lvlm_frontend(cfg.run_cfg, model, cfg.run_cfg.get("device", "cuda"))
    runner = registry.get_runner_class(cfg.run_cfg.get("runner", "runner_process_data"))(cfg=cfg, job_id=job_id, task=task, model=model, datasets=datasets)
    return runner.train()


if __name__ == "__main__":
    main()

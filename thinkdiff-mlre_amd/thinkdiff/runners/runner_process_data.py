"""Precompute runner (reference thinkdiff/runners/runner_process_data.py:38-175): no DDP wrap, chunk-ordered loader,
one epoch.  New relative to the reference (whose job is single-process): with world > 1 the wids SHARD
LIST is partitioned by rank (`shardlist[rank::world]`) and every rank writes its own disjoint shard numbers
(`start_shard = base + rank * shard_stride`), so the 8 GPUs of a node run without any data-path collective; rank 0
gathers the per-rank counts at the end."""
from typing import List

from ..common.dist_utils import get_rank, get_world_size
from ..common.registry import registry
from ..datasets.wds_io import chunked_order
from . import RunnerBase, dp_inference


class _Loader:
    """Batches of a map-style dataset in ChunkedSampler order, collated by the dataset's collater."""

    def __init__(self, dataset, batch_size: int, chunksize: int = 1000, shuffle: bool = True, seed: int = 0):
        self.dataset, self.batch_size = dataset, batch_size
        self.order = chunked_order(len(dataset), chunksize, shuffle, seed)

    def __len__(self):
        return (len(self.order) + self.batch_size - 1) // self.batch_size

    def _batch(self, s):
        return self.dataset.collater([self.dataset[i] for i in self.order[s:s + self.batch_size]])

    def __iter__(self):
        """The next batch is read and decoded on a background thread while the caller works on the current one (tar reads and
        JPEG decoding release the GIL)."""
        from concurrent.futures import ThreadPoolExecutor
        starts = list(range(0, len(self.order), self.batch_size))
        if not starts:
            return
        with ThreadPoolExecutor(max_workers=1) as pool:
            nxt = pool.submit(self._batch, starts[0])
            for k in range(len(starts)):
                cur = nxt.result()
                if k + 1 < len(starts):
                    nxt = pool.submit(self._batch, starts[k + 1])
                yield cur


@registry.register_runner("runner_process_data")
class RunnerProcessData(RunnerBase):
    SHARD_STRIDE = 100000   # shard-number range reserved per rank

    @property
    def model(self):
        return self._model

    @property
    def output_shard_path(self) -> List:
        osp = self.config.run_cfg.get("output_shard_path", None)
        if osp is None:
            return None
        rank = get_rank()
        return [osp[0], osp[1], int(osp[2]) + rank * self.SHARD_STRIDE]

    def train_loader(self, dataset, batch_size: int):
        return _Loader(dataset, batch_size, seed=self.config.run_cfg.get("seed", 0) + get_rank())

    def train(self):
        name, dataset = next(iter(self.datasets.items()))
        bs = self.config.datasets_cfg[name].get("batch_size", 8192) if name in self.config.datasets_cfg else 8192
        stats = self.task.train_epoch(0, self.model, self.train_loader(dataset, bs), output_shard_path=self.output_shard_path)
        gathered = dp_inference.gather_results([{"rank": get_rank(), "samples": stats["samples"], "shards": stats["shards"]}])
        return gathered if gathered is not None else stats

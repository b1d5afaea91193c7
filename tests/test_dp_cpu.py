"""World-size-2 gloo test of the data-parallel inference plumbing (broadcast work list, shard, gather)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from thinkdiff.runners import dp_inference as dp
    work = dp.broadcast_work_list([(i, 42 + i) for i in range(7)] if rank == 0 else None)
    mine = dp.shard(work)
    shared = dp.broadcast_tensor(torch.arange(6.0).reshape(2, 3) if rank == 0 else None, (2, 3), torch.float32, "cpu")
    local = [f"img_{p}_seed_{s}_rank{rank}_{float(shared.sum()):.0f}" for p, s in mine]
    res = dp.gather_results(local)
    imgs = dp.gather_images(torch.full((1, 3, 4, 4), rank, dtype=torch.uint8))
    dist.barrier()
    if rank == 0:
        q.put((work, res, [int(t.flatten()[0]) for t in imgs]))
    dist.destroy_process_group()


def test_two_rank_sharding_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    work, res, imgs = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert work == [(i, 42 + i) for i in range(7)]
    # results come back in work order, each unit rendered by rank i % 2, shared tensor identical everywhere
    assert res == [f"img_{i}_seed_{42 + i}_rank{i % 2}_15" for i in range(7)]
    assert imgs == [0, 1]


def test_shard_balance():
    sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
    from thinkdiff.runners import dp_inference as dp
    work = list(range(64))
    parts = [dp.shard(work, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == work and {len(p) for p in parts} == {8}
    parts = [dp.shard(list(range(13)), r, 8) for r in range(8)]
    assert max(map(len, parts)) - min(map(len, parts)) == 1


def _precompute_worker(rank, world, port, root, q):
    sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_precompute_cpu import _StubModel
    from thinkdiff.common.config import Node
    from thinkdiff.datasets.cc_sbu_process import CCSBUMllamaVllmProcessDatasetWids
    from thinkdiff.runners import RunnerProcessData
    from thinkdiff.tasks.image_text_process_data import ImageTextProcessDataTask
    cfg = type("C", (), {})()
    cfg.run_cfg = Node({"output_shard_path": [os.path.join(root, "out"), "%06d.tar", 3], "seed": 1, "device": "cpu"})
    cfg.datasets_cfg = Node({"cc": {"batch_size": 4}})
    ds = CCSBUMllamaVllmProcessDatasetWids(os.path.join(root, "wids_shards.json"), rank=rank, world=world)
    runner = RunnerProcessData(cfg, ImageTextProcessDataTask(), _StubModel(), {"cc": ds}, "job")
    res = runner.train()
    dist.barrier()
    if rank == 0:
        q.put(res)
    dist.destroy_process_group()


def test_two_rank_precompute_writes_disjoint_shards(tmp_path):
    """BASELINE config 4's DP: the shard list is split by rank, each rank writes its own shard-number range."""
    sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_precompute_cpu import _make_input_shards
    from thinkdiff.datasets import wds_io
    _, n = _make_input_shards(str(tmp_path), n_shards=4, per_shard=3)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_precompute_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r["rank"] for r in res) == [0, 1] and sum(r["samples"] for r in res) == n
    names = [os.path.basename(s["url"]) for r in res for s in r["shards"]]
    assert "000003.tar" in names and "100003.tar" in names and len(set(names)) == len(names)
    keys = [s["__key__"] for r in res for sh in r["shards"] for s in wds_io.read_tar_samples(sh["url"], decode=False)]
    assert sorted(keys) == [f"sample{k:06d}" for k in range(n)]

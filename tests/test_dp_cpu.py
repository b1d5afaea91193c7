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

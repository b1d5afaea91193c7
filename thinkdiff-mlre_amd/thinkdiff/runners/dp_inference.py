"""Data-parallel sharding of the inference work list over the GPUs of one node.

The reference runs N identical replicas (every rank renders the whole prompt list with seed+rank, no
communication: scripts/test/test_mllama_t5_decoder_flux.py:57-65).  Here the independent units --
(prompt index, seed) images, or precompute shards -- are partitioned instead:
rank 0 broadcasts the work list (and any shared small tensors), every rank takes `work[rank::world]`,
results (paths / timings / optionally uint8 images) are gathered on rank 0.  Over RCCL these are
one-to-all / all-to-one transfers on the direct xGMI links; nothing on the data path needs an
all-reduce.  With the gloo backend the same code runs on CPU (tests/test_dp_cpu.py).
"""
from typing import Any, List, Optional, Sequence

import torch
import torch.distributed as dist

from ..common.dist_utils import get_rank, get_world_size, is_dist_avail_and_initialized


def shard(work: Sequence[Any], rank: Optional[int] = None, world: Optional[int] = None) -> List[Any]:
    """Round-robin partition: unit i goes to rank i % world (balanced to within one unit)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    return list(work[rank::world])


def broadcast_work_list(work: Optional[Sequence[Any]], src: int = 0) -> List[Any]:
    """Rank `src` supplies the list (e.g. [(prompt_idx, seed), ...]); everyone returns the same list."""
    if not is_dist_avail_and_initialized():
        return list(work)
    box = [list(work) if get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def broadcast_tensor(t: Optional[torch.Tensor], shape, dtype, device, src: int = 0) -> torch.Tensor:
    """Shared conditioning (e.g. T5/pooled text embeddings computed once on rank 0, <= ~2 MB)."""
    if not is_dist_avail_and_initialized():
        return t
    if get_rank() != src:
        t = torch.empty(shape, dtype=dtype, device=device)
    dist.broadcast(t, src=src)
    return t


def gather_results(local: List[Any], dst: int = 0) -> Optional[List[Any]]:
    """All-to-one gather of per-rank result lists; returns them re-interleaved in work order on `dst`."""
    if not is_dist_avail_and_initialized():
        return list(local)
    world = get_world_size()
    box = [None] * world if get_rank() == dst else None
    dist.gather_object(list(local), box, dst=dst)
    if get_rank() != dst:
        return None
    out, i = [], 0
    while any(i < len(b) for b in box):
        for b in box:
            if i < len(b):
                out.append(b[i])
        i += 1
    return out


def gather_images(img: torch.Tensor, dst: int = 0) -> Optional[List[torch.Tensor]]:
    """Gather equal-shaped uint8 image batches [n,3,H,W] on `dst` (3 MiB per 1024^2 image)."""
    if not is_dist_avail_and_initialized():
        return [img]
    world = get_world_size()
    bufs = [torch.empty_like(img) for _ in range(world)] if get_rank() == dst else None
    dist.gather(img, bufs, dst=dst)
    return bufs

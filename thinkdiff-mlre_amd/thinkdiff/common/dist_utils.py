"""Process-group bootstrap with the reference's surface (thinkdiff/common/dist_utils.py:17-140).

One process per GPU; backend "nccl" is RCCL on ROCm (xGMI inside a node).  The reference's only
collectives are barriers and a 16-byte all_reduce; the inference path here adds a
broadcast of the work list and a gather of results (thinkdiff/runners/dp_inference.py).
"""
import datetime
import functools
import os

import torch
import torch.distributed as dist


def setup_for_distributed(is_master):
    """Mute print on non-master ranks unless force=True (dist_utils.py:17-30)."""
    import builtins as __builtin__
    builtin_print = __builtin__.print

    def print(*args, **kwargs):
        force = kwargs.pop("force", False)
        if is_master or force:
            builtin_print(*args, **kwargs)

    __builtin__.print = print


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def init_distributed_mode(args):
    """Mutates args.rank / world_size / gpu / distributed / dist_backend from RANK, WORLD_SIZE, LOCAL_RANK."""
    if not args.get("distributed", True) and "RANK" not in os.environ:
        args.distributed = False
        return
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        print("Not using distributed mode")
        args.distributed = False
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available() and str(args.get("device", "cuda")).startswith("cuda")
    args.dist_backend = "nccl" if use_gpu else "gloo"
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    print(f"| distributed init (rank {args.rank}, world {args.world_size}): {args.get('dist_url', 'env://')}", flush=True)
    dist.init_process_group(backend=args.dist_backend, init_method=args.get("dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank,
                            timeout=datetime.timedelta(days=365))
    dist.barrier()
    setup_for_distributed(args.rank == 0)


def get_dist_info():
    return get_rank(), get_world_size()


def main_process(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        if get_rank() == 0:
            return func(*args, **kwargs)
    return wrapper

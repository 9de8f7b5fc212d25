"""Data-parallel plumbing shared by the trainer and the entry points: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference wraps the model in DistributedDataParallel(find_unused_parameters=True) (train_ddp.py:79), which
all-reduces every parameter in 25 MB buckets - including ~170 M that never receive a gradient.  Here the live
gradients are one contiguous fp32 buffer and the exchange is ONE all-reduce(SUM) per step; each rank's loss is
already divided by the GLOBAL bag count, so the sum is DDP's mean-over-ranks gradient."""
import math
import os
from typing import List

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str, init_method: str = "env://", world_size: int = 1, rank: int = 0, device=None):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if device is not None and backend == "nccl":
        kw["device_id"] = device
    dist.init_process_group(backend=backend, init_method=init_method, world_size=world_size, rank=rank, **kw)


def shard_indices(n: int, world: int, rank: int, epoch: int = 0, shuffle: bool = True, seed: int = 0) -> List[int]:
    """The indices torch's DistributedSampler gives rank `rank` (train_ddp.py:191 + set_epoch :201): permutation
    seeded with seed + epoch, wrap-around padding to a multiple of `world`, then every world-th index."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    total = math.ceil(n / world) * world
    while len(order) < total:
        order += order[: total - len(order)]
    return order[rank:total:world]


def broadcast_flat(buf: torch.Tensor, src: int = 0):
    """DDP broadcasts rank 0's parameters when it wraps the model; same here for the flat buffer."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(buf, src=src)


def allreduce_flat(buf: torch.Tensor):
    """The step's single data-path collective."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("MIL_FORCE_COLLECTIVES") == "1"):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)

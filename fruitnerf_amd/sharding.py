"""Ray-range sharding for the passes that need no parameter exchange: full-image evaluation is split into
contiguous row blocks, the volume export into contiguous runs of ray batches; every rank works on its block with the
same kernels as a single process and the results are concatenated in rank order, which reproduces the single-process
order exactly (rays are independent given the parameters).  The only collectives are all-gathers of the outputs
(RCCL on GPUs, gloo in the CPU tests); nothing here touches the training exchange (training.py)."""
from __future__ import annotations

from typing import List, Tuple

import torch
from torch import Tensor


def shard_range(n: int, rank: int, world_size: int, granule: int = 1) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of range(n) for `rank`: blocks tile range(n) in rank order, sizes differ by at most
    one granule, boundaries are multiples of `granule` (whole image rows / whole ray batches) except the last."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    units = -(-n // granule)
    base, extra = divmod(units, world_size)
    lo_u = rank * base + min(rank, extra)
    hi_u = lo_u + base + (1 if rank < extra else 0)
    return min(lo_u * granule, n), min(hi_u * granule, n)


def all_gather_rows(local: Tensor, world_size: int) -> Tensor:
    """Concatenate per-rank tensors that differ in their first dimension, in rank order, on every rank.
    (sizes all-gathered first, payload padded to the longest block: two collectives, no host staging)."""
    if world_size <= 1:
        return local
    import torch.distributed as dist
    local = local.contiguous()
    if local.is_cuda and dist.get_backend() == "gloo":
        # gloo has no all_gather for device tensors (the one-GPU test rig: two ranks on one device, RCCL refuses that):
        # stage through the host.  On real peers the backend is RCCL and the payload never leaves the devices.
        return all_gather_rows(local.cpu(), world_size).to(local.device)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world_size)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    if cap == 0:
        return local
    padded = local
    if local.shape[0] < cap:
        padded = torch.zeros((cap,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    blocks = [torch.empty_like(padded) for _ in range(world_size)]
    dist.all_gather(blocks, padded)
    return torch.cat([b[:k] for b, k in zip(blocks, sizes)], dim=0)


def gather_chunks(chunks: List[Tensor], world_size: int, like: Tensor) -> Tensor:
    """cat(chunks) of this rank (possibly empty: `like` gives dtype / trailing shape / device), gathered over ranks."""
    local = torch.cat(chunks, dim=0) if chunks else like.new_zeros((0,) + tuple(like.shape[1:]))
    return all_gather_rows(local, world_size)

"""Data-parallel launcher helpers: one process per GPU, images / clips sharded across ranks, weights replicated.

The hot path is embarrassingly parallel over images and whole clips: every op of
ControlNet + Ctrl-Adapter is per-sample except the frame-mixing ops of ONE clip, so the sharded path needs NO
data-path collective.  The only communication is the throughput bookkeeping below (a barrier and a MAX all-reduce
of the elapsed time), which runs over RCCL ("nccl" backend) on the GPUs and over gloo in the CPU tests.
"""
import os
import time

import torch


def init(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torchrun contract)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if torch.cuda.is_available() else "gloo"), rank=rank, world_size=world)
    return rank, world


def shard(total_units, rank, world):
    """Contiguous [begin, end) range of the units (images or whole clips) owned by `rank`; sizes differ by at most 1."""
    base, rem = divmod(total_units, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def barrier(sync_cuda=True):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
    if sync_cuda and torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(seconds, device=None):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def timed_region(fn, steps, device=None):
    """barrier+sync, run fn() `steps` times, barrier+sync; returns the MAX elapsed seconds over ranks."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    barrier()
    return max_over_ranks(time.perf_counter() - t0, device)


def aggregate_throughput(units_per_rank_step, steps, elapsed_max, world):
    """whole-job units/s: every rank processed units_per_rank_step * steps units within elapsed_max"""
    return world * units_per_rank_step * steps / elapsed_max

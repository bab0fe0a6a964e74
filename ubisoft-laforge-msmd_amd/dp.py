"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm).

The forward / sampling path shards by independent clips: no data-path collective.
The only exchanges are the timing reduction of the benchmark and (training) the gradient all-reduce.
"""
from __future__ import annotations

import os
import time

import torch


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device=None):
    import torch.distributed as td
    rank, _, world = env_rank()
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        td.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_clips(n_clips: int, rank: int, world: int):
    """Contiguous, balanced partition of clip indices across ranks (independent units, no exchange)."""
    base, rem = divmod(n_clips, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def timed_steps(step_fn, steps: int, warmup: int, sync=None, device=None):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both sides;
    returns the MAX elapsed seconds over ranks (the bench contract)."""
    import torch.distributed as td
    dist = td.is_available() and td.is_initialized() and td.get_world_size() > 1
    sync = sync or (lambda: None)
    for _ in range(warmup):
        step_fn()
    sync()
    if dist:
        td.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    if dist:
        td.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        td.all_reduce(tt, op=td.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed

"""One-process-per-GPU launch helpers for the replica-parallel denoise path.

The CTSD denoise step has no exchange step between samples (the reference only
ever runs independent samples per GPU, src/dwm/train.py:116-122, ctsd.py:1904-1911), so N GPUs
= N replicas, each denoising its own samples; the only collectives are the barrier around
the timed region and a MAX reduction of the elapsed time.  backend "nccl" is RCCL on ROCm;
"gloo" is used by the CPU tests."""
from __future__ import annotations

import os
import time
from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def env_ranks() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str, device: torch.device = None) -> Tuple[int, int, int]:
    rank, local_rank, world = env_ranks()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_samples(n_samples: int, rank: int, world: int) -> List[int]:
    """Sample indices this rank denoises: contiguous, disjoint, covering range(n_samples)."""
    per, rem = divmod(n_samples, world)
    start = rank * per + min(rank, rem)
    return list(range(start, start + per + (1 if rank < rem else 0)))


def sync(device: torch.device = None) -> None:
    """torch.cuda.synchronize() + barrier + synchronize (both sides of a timed region)."""
    on_gpu = device is not None and device.type == "cuda"
    if on_gpu:
        torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
        if on_gpu:
            torch.cuda.synchronize(device)


def max_over_ranks(value: float, device: torch.device = None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None and device.type == "cuda" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def timed_steps(step: Callable[[int], None], steps: int, warmup: int, device: torch.device = None) -> float:
    """W untimed steps, then exactly K steps bracketed by sync(); returns the MAX-over-ranks seconds."""
    for i in range(warmup):
        step(i)
    sync(device)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    sync(device)
    return max_over_ranks(time.perf_counter() - t0, device)


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()

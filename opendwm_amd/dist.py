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


def init(backend: str, device: torch.device = None, force: bool = False) -> Tuple[int, int, int]:
    """process group of a torch.distributed.run launch; a single process needs none - unless `force` (bench.py --preflight on
    one GPU: a one-rank RCCL communicator, so that the collectives' code path runs on the hardware at hand)"""
    rank, local_rank, world = env_ranks()
    if (world > 1 or force) and not dist.is_initialized():
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


def preflight(device: torch.device = None, mbytes: int = 64) -> dict:
    """Run BEFORE any timed region of a multi-rank job: every rank reports its device, all ranks all-reduce a known `mbytes`
    buffer (rank r contributes r + 1 everywhere) and check the sum - a wrong transport set-up (IPC mode, visible devices, a rank
    on the wrong GPU) fails loudly here instead of inside the measurement.  Returns the facts rank 0 puts into its JSON line;
    raises RuntimeError on any mismatch.  Works on every backend (gloo in the CPU tests, nccl = RCCL on the GPUs)."""
    rank, local_rank, world = env_ranks()
    on_gpu = device is not None and device.type == "cuda"
    info = dict(world_size=world, backend=dist.get_backend() if dist.is_initialized() else None,
                hsa_enable_ipc_mode_legacy=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
    if on_gpu:
        info.update(device=torch.cuda.get_device_name(device), device_index=device.index, visible_devices=torch.cuda.device_count())
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            info["rccl_version"] = None
        if world > 1 and torch.cuda.device_count() < 1:
            raise RuntimeError("preflight: no visible GPU on this rank")
    if not dist.is_initialized():
        return info
    dev = device if on_gpu else torch.device("cpu")
    n = mbytes * (1 << 20) // 4
    buf = torch.full((n,), float(rank + 1), dtype=torch.float32, device=dev)
    sync(device)
    t0 = time.perf_counter()
    dist.all_reduce(buf)
    sync(device)
    dt = time.perf_counter() - t0
    want = world * (world + 1) / 2
    lo, hi = buf.min().item(), buf.max().item()
    if lo != want or hi != want:
        raise RuntimeError(f"preflight: all_reduce over {world} ranks gave [{lo}, {hi}], expected {want} on rank {rank}")
    # every rank on its own device: the (host, device index) pairs must be distinct - per HOST, so that a 2 x 8 launch (the same 8
    # device indices on two nodes) passes and two ranks of one node on one GPU do not; the device index is the one the rank
    # actually selected (`device.index`), not LOCAL_RANK: a rank whose set_device went to the wrong GPU is what this catches.
    # (A tensor all-gather, not all_gather_object: no pickling through the backend's device path.)
    import hashlib
    import socket
    host = int.from_bytes(hashlib.sha256(socket.gethostname().encode()).digest()[:7], "little")
    index = device.index if on_gpu and device.index is not None else (torch.cuda.current_device() if on_gpu else -1 - rank)
    visible = torch.cuda.device_count() if on_gpu else 0
    mine = torch.tensor([host, device_identity(device) if on_gpu else -1 - rank, index, os.getpid(), visible], dtype=torch.int64, device=dev)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine)
    rows = [tuple(int(v) for v in t.tolist()) for t in got]
    ids = [r[:4] for r in rows]
    checked = check_distinct_devices(ids, [r[4] for r in rows]) if on_gpu else False
    # what the first multi-GPU run needs to diagnose itself: who answered, from where, on which physical device
    info.update(allreduce_mbytes=mbytes, allreduce_ms=1e3 * max_over_ranks(dt, device), allreduce_sum_ok=True,
                distinct_devices=(True if checked else "unchecked") if on_gpu else False,
                hosts=len({i[0] for i in ids}), ranks_seen=len(rows),
                ranks=[dict(rank=r, host_hash=f"{i[0]:014x}", pci=pci_string(i[1]), device_index=i[2], pid=i[3], visible_devices=v[4])
                       for r, (i, v) in enumerate(zip(ids, rows))])
    return info


def local_device(local_rank: int) -> torch.device:
    """the GPU of this rank: index LOCAL_RANK when the launcher leaves all GPUs of the node visible to every rank (torchrun's
    default), index LOCAL_RANK mod the visible count when it isolates them (one visible device per rank: index 0)"""
    n = torch.cuda.device_count()
    if n < 1:
        raise RuntimeError("no visible GPU on this rank")
    idx = local_rank if local_rank < n else local_rank % n
    torch.cuda.set_device(idx)
    return torch.device("cuda", idx)


def device_identity(device: torch.device) -> int:
    """PCI address (domain, bus, device) of a visible GPU as one integer: the same physical GPU has the same value whatever
    index a rank sees it under (-1 if this PyTorch build does not report it)"""
    try:
        p = torch.cuda.get_device_properties(device)
        return (int(p.pci_domain_id) << 16) | (int(p.pci_bus_id) << 8) | int(p.pci_device_id)
    except Exception:
        return -1


def pci_string(identity: int):
    """device_identity() as "dddd:bb:dd" (None if the build does not report it)"""
    if identity is None or identity < 0:
        return None
    return f"{identity >> 16:04x}:{(identity >> 8) & 0xff:02x}:{identity & 0xff:02x}"


def check_distinct_devices(ids, visible=None) -> bool:
    """ids: one (host hash, PCI identity, device index, pid) per rank - or (host hash, device index, pid); visible: the number of
    GPUs each rank sees (optional).  Raises if two ranks of one host selected the same device: the same PCI address AND the same
    index (ranks that each see only their own GPU all report index 0 with different addresses; partitions of one GPU share the
    address and differ in the index).  Returns True if the check was made, False if it had to be skipped."""
    key = [t[:-1] for t in ids]
    if any(len(t) == 4 and t[1] < 0 for t in ids):
        # this PyTorch build does not report PCI addresses.  Ranks that see ALL GPUs of their node (torchrun's default: more than
        # one visible device) are still told apart by (host, index): a duplicate there is two ranks on one GPU.  Ranks that each see
        # only their own GPU all report (host, -1, 0) and cannot be told apart - a correct launch must not be aborted on that.
        if visible is not None and len(visible) == len(ids) and all(v > 1 for v in visible):
            hk = [(t[0], t[2]) for t in ids]
            if len(set(hk)) != len(hk):
                raise RuntimeError(f"preflight: ranks share a GPU: (host hash, PCI identity, device index, pid) = {list(ids)}")
            return True
        import warnings
        warnings.warn("preflight: the device PCI identity is not available in this PyTorch build and every rank sees one device; "
                      "the distinct-device check is skipped")
        return False
    if len(set(key)) != len(key):
        raise RuntimeError(f"preflight: ranks share a GPU: (host hash, [PCI identity,] device index, pid) = {list(ids)}")
    return True


def measure_allreduce(nbytes: int, device: torch.device = None, dtype: torch.dtype = torch.float32, repeat: int = 3) -> float:
    """milliseconds of ONE all-reduce of `nbytes` (MAX over ranks, best of `repeat`): the gradient exchange of a DDP step on its
    own, to set beside the step time with and without gradient synchronisation (bench.py --train)"""
    if not dist.is_initialized():
        return 0.0
    dev = device if device is not None and device.type == "cuda" else torch.device("cpu")
    buf = torch.zeros(max(1, nbytes // torch.empty((), dtype=dtype).element_size()), dtype=dtype, device=dev)
    best = None
    for _ in range(repeat):
        sync(device)
        t0 = time.perf_counter()
        dist.all_reduce(buf)
        sync(device)
        dt = max_over_ranks(time.perf_counter() - t0, device)
        best = dt if best is None else min(best, dt)
    return 1e3 * best


def shutdown() -> None:
    if dist.is_initialized():
        dist.destroy_process_group()

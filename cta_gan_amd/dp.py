"""Data parallelism for the G+D step: one process per GPU, gradients all-reduced over RCCL/xGMI.

The reference is single-GPU (train.py:14).  Samples are independent on this path (InstanceNorm statistics
are per sample, every loss is a batch mean), so with equal per-rank batches the averaged gradient IS the
global-batch gradient.  Exchange = ONE flat all-reduce(sum)/world per optimiser group:
{G || Reg} (53.7 MB fp32) after the G-step backward, {D} (11 MB) after the D-step backward.  On a
fully connected 8-GPU xGMI node RCCL picks its own algorithm for these sizes; both messages are far
below the step's compute time, so they are issued as single large buckets rather than many small ones.
"""
from __future__ import annotations

import os
from typing import Iterable, List

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the default process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_grads(params: Iterable[torch.nn.Parameter]) -> None:
    """Average `.grad` of `params` across ranks with one flat all-reduce; no-op for world size 1.

    After the call each `.grad` is a view into the reduced flat bucket (no copy back)."""
    world = world_size()
    if world == 1:
        return
    ps: List[torch.nn.Parameter] = [p for p in params if p.grad is not None]
    if not ps:
        return
    flat = torch.cat([p.grad.reshape(-1).float() for p in ps])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    off = 0
    for p in ps:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p)
        off += n


def broadcast_params(*modules) -> None:
    """Make every rank start from rank 0's weights (replicas of a data-parallel job must be identical; the reference,
    being single-GPU, never needed this).  One flat broadcast per call; no-op for world size 1."""
    if world_size() == 1:
        return
    ps = [p for m in modules for p in m.parameters()]
    if not ps:
        return
    with torch.no_grad():
        flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
        dist.broadcast(flat, src=0)
        off = 0
        for p in ps:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            p._ctg_version = getattr(p, "_ctg_version", 0) + 1
            off += n


def barrier():
    if world_size() > 1:
        dist.barrier()

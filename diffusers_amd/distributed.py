"""Batch-parallel sampling: independent prompts are sharded over the GPUs of one node, one process per GPU.

This is the reference's data-parallel recipe (docs/source/en/training/distributed_inference.md:29-108: one replica per
rank, each rank takes its slice of the prompts, no collective in the loop) made explicit: the model is replicated, a
prompt's CFG pair stays on one GPU, and the ONLY traffic is one broadcast of the text embeddings from rank 0 before the
loop (RCCL over xGMI; ~0.64 MB per SDXL prompt) and an optional gather of the finished images.  The denoising path has
no exchange step, so there is no all-reduce / all-gather anywhere in it.
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's env; initialises the default process group when world_size > 1 -- and
    also for ONE rank when the process was started by a launcher (RANK and WORLD_SIZE both exported, as
    ``python -m torch.distributed.run --nproc-per-node 1`` does): a single-GPU box then runs the same RCCL initialisation,
    broadcast, barrier and reductions the 8-GPU job runs, instead of skipping them.  backend defaults to "nccl" (= RCCL on
    ROCm) when a HIP device is present, else "gloo"."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Prompt i -> rank i % world (round-robin keeps ranks balanced for any n_items)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("invalid rank / world size")
    return list(range(rank, n_items, world))


def broadcast_tensors(tensors: Dict[str, torch.Tensor], src: int = 0) -> Dict[str, torch.Tensor]:
    """Broadcast a dict of pre-allocated, identically shaped tensors from ``src`` (one collective per tensor)."""
    if not dist.is_initialized():     # (a one-rank process group still runs the collective: same RCCL path as N ranks)
        return tensors
    for k in sorted(tensors):
        dist.broadcast(tensors[k], src=src)
    return tensors


def select_shard(tensors: Dict[str, torch.Tensor], idx: Sequence[int]) -> Dict[str, torch.Tensor]:
    """Slice the leading (prompt) dimension of every tensor to this rank's prompts."""
    ii = torch.as_tensor(list(idx), dtype=torch.long)
    return {k: v.index_select(0, ii.to(v.device)) for k, v in tensors.items()}


def gather_images(local: torch.Tensor, n_items: int, dst: int = 0):
    """Gather per-rank image batches [n_local, ...] to ``dst`` and restore prompt order (round-robin sharding).
    Ranks may hold different counts; every rank pads to ceil(n_items / world)."""
    if not dist.is_initialized():
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_items + world - 1) // world
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = torch.empty((n_items,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard_indices(n_items, r, world)
        if idx:
            out[torch.as_tensor(idx, device=out.device)] = bufs[r][: len(idx)]
    return out


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def backend_name() -> str | None:
    """"nccl" (RCCL) / "gloo" of the default process group, None without one."""
    return dist.get_backend() if dist.is_initialized() else None

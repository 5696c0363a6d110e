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


def _free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def init_from_env(backend: str | None = None, bind_cpus: bool = True) -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's env; initialises the default process group when world_size > 1 -- and
    also for ONE rank when the process was started by a launcher (RANK and WORLD_SIZE both exported, as
    ``python -m torch.distributed.run --nproc-per-node 1`` does): a single-GPU box then runs the same RCCL initialisation,
    broadcast, barrier and reductions the 8-GPU job runs, instead of skipping them.  backend defaults to "nccl" (= RCCL on
    ROCm) when a HIP device is present, else "gloo".

    A one-rank group has no peer to meet, so when the environment names no MASTER_PORT (SLURM / k8s images export RANK and
    WORLD_SIZE = 1 without one) it rendezvouses on a free port of its own instead of the shared default 29500 -- two independent
    single-rank jobs on one host do not collide.  With ``bind_cpus`` every rank also pins itself to its share of the host's cores
    (:func:`bind_rank_to_cpus`; ``DIFFUSERS_AMD_BIND=0`` turns that off)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port()) if world == 1 else "29500"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if bind_cpus and world > 1 and os.environ.get("DIFFUSERS_AMD_BIND", "1") != "0":
        try:
            bind_rank_to_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        except Exception:       # a host without sysfs / sched_setaffinity: placement is a speed matter only
            pass
    return rank, world, local


# ---- host-side placement: one process per GPU, each on the cores next to its GPU -----------------------------------
def _parse_cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def _gpu_numa_node(local_rank: int) -> int:
    """NUMA node of the GPU this rank drives (sysfs `numa_node` of its PCI function), -1 when the host does not say."""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            return int(f.read())
    except Exception:
        return -1


def cpu_share(local_rank: int, local_world: int, allowed: Sequence[int], gpu_nodes: Sequence[int],
              node_cpus: Dict[int, Sequence[int]]) -> List[int]:
    """The cores rank ``local_rank`` of ``local_world`` pins itself to (pure function of the host description, so every rank
    computes the same partition without talking to the others).

    Policy: a rank takes cores of the NUMA node its GPU hangs off (the host thread that replays the step graphs and feeds RCCL
    then touches memory next to that GPU's root port); the ranks that share a node split its allowed cores into equal contiguous
    runs in rank order; a rank whose GPU's node is unknown (-1), or whose node has no allowed core, takes its equal contiguous
    run of ALL allowed cores instead.  Never empty: a host with fewer cores than ranks lets ranks share."""
    allowed = sorted(allowed)
    if not allowed or local_world <= 0 or not (0 <= local_rank < local_world):
        raise ValueError("cpu_share: empty affinity set or invalid rank")

    def run(cores, i, n):
        if len(cores) < n:
            return [cores[i % len(cores)]]
        per = len(cores) // n
        return list(cores[i * per:(i + 1) * per])
    node = gpu_nodes[local_rank] if local_rank < len(gpu_nodes) else -1
    mine = [c for c in node_cpus.get(node, ()) if c in set(allowed)] if node >= 0 else []
    if not mine:
        return run(allowed, local_rank, local_world)
    peers = [r for r in range(local_world) if r < len(gpu_nodes) and gpu_nodes[r] == node]
    return run(sorted(mine), peers.index(local_rank), len(peers))


def bind_rank_to_cpus(local_rank: int, local_world: int) -> List[int]:
    """Pin this process to :func:`cpu_share`'s cores (`sched_setaffinity`) and cap torch's intra-op pool to them.  Returns the
    cores.  The eight replicas of one node otherwise float over all host cores: the thread that replays 50 step graphs per image
    migrates between sockets and RCCL's proxy threads compete with the other ranks' launch threads."""
    allowed = sorted(os.sched_getaffinity(0))
    nodes: Dict[int, List[int]] = {}
    base = "/sys/devices/system/node"
    if os.path.isdir(base):
        for d in os.listdir(base):
            if d.startswith("node") and d[4:].isdigit():
                try:
                    with open(f"{base}/{d}/cpulist") as f:
                        nodes[int(d[4:])] = _parse_cpulist(f.read())
                except OSError:
                    pass
    gpu_nodes = [_gpu_numa_node(r) for r in range(local_world)] if torch.cuda.is_available() else [-1] * local_world
    cores = cpu_share(local_rank, local_world, allowed, gpu_nodes, nodes)
    os.sched_setaffinity(0, cores)
    torch.set_num_threads(max(1, min(len(cores), torch.get_num_threads())))
    return cores


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

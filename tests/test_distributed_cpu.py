"""CPU / gloo, world_size 2: the batch-parallel sharding of diffusers_amd.distributed (the N > 1 path of bench.py).

The data path has no collective: prompts are dealt round-robin, rank 0 broadcasts the text embeddings once, every rank
works on its shard, images are gathered back in prompt order.  Mirrors the reference's recipe
(docs/source/en/training/distributed_inference.md:29-108), which has no library code or tests of its own."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_prompts, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from diffusers_amd import distributed as D
    try:
        r, w, _ = D.init_from_env(backend="gloo")
        assert (r, w) == (rank, world)
        g = torch.Generator().manual_seed(1234)
        full = {"prompt_embeds": torch.randn((n_prompts, 7, 16), generator=g),
                "pooled": torch.randn((n_prompts, 8), generator=g)}
        # only rank 0 owns the prompts; the others receive them through ONE broadcast per tensor
        inputs = {k: (v.clone() if rank == 0 else torch.zeros_like(v)) for k, v in full.items()}
        D.broadcast_tensors(inputs, src=0)
        for k in full:
            assert torch.equal(inputs[k], full[k]), f"rank {rank}: broadcast of {k} differs"
        idx = D.shard_indices(n_prompts, rank, world)
        mine = D.select_shard(inputs, idx)
        assert mine["prompt_embeds"].shape[0] == len(idx)
        # stand-in for the per-prompt pipeline: something that depends on the prompt only
        local_img = mine["prompt_embeds"].sum(dim=(1, 2)).view(-1, 1, 1, 1) + torch.zeros((len(idx), 3, 4, 4))
        out = D.gather_images(local_img, n_prompts, dst=0)
        t = D.max_over_ranks(float(rank + 1), torch.device("cpu"))
        assert t == float(world)
        if rank == 0:
            want = full["prompt_embeds"].sum(dim=(1, 2)).view(-1, 1, 1, 1) + torch.zeros((n_prompts, 3, 4, 4))
            assert torch.equal(out, want), "gathered images are not in prompt order"
        else:
            assert out is None
        dist.barrier()
        q.put((rank, "ok", idx))
    except Exception as e:  # surfaced in the parent
        q.put((rank, f"fail: {e!r}", None))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("n_prompts", [2, 5, 8])
def test_two_rank_sharding_broadcast_gather(n_prompts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_prompts, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, status, idx in res:
        assert status == "ok", f"rank {rank}: {status}"
    shards = {rank: idx for rank, status, idx in res}
    assert sorted(shards[0] + shards[1]) == list(range(n_prompts))  # every prompt exactly once
    assert abs(len(shards[0]) - len(shards[1])) <= 1                # balanced


def test_shard_indices_properties():
    from diffusers_amd import distributed as D
    for n in (0, 1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            parts = [D.shard_indices(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
    with pytest.raises(ValueError):
        D.shard_indices(4, 2, 2)


def test_single_process_paths_are_identity():
    from diffusers_amd import distributed as D
    t = {"a": torch.arange(6.0).view(3, 2)}
    assert D.broadcast_tensors(t) is t
    assert D.gather_images(t["a"], 3) is t["a"]
    assert D.max_over_ranks(3.5, torch.device("cpu")) == 3.5

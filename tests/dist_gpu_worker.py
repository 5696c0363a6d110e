"""Worker of tests/test_distributed_gpu.py: ONE rank under torch.distributed.run on the GPU box -- `init_from_env()` picks
"nccl" (= RCCL), and the collectives of diffusers_amd.distributed run on HIP tensors through the communicator exactly as they
do for 8 ranks (docs/source/en/training/distributed_inference.md:29-108 is the reference recipe).  Prints one JSON line."""
import json
import sys
from pathlib import Path

import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from diffusers_amd import distributed as D  # noqa: E402

rank, world, local = D.init_from_env()
assert dist.is_initialized() and D.backend_name() == "nccl", D.backend_name()
dev = torch.device("cuda", local)
g = torch.Generator().manual_seed(7)
full = {"prompt_embeds": torch.randn((3, 77, 64), generator=g).to(torch.bfloat16).to(dev), "pooled": torch.randn((3, 32), generator=g).to(dev)}
got = D.broadcast_tensors({k: v.clone() for k, v in full.items()}, src=0)
ok_b = all(torch.equal(got[k], full[k]) for k in full)
idx = D.shard_indices(3, rank, world)
mine = D.select_shard(got, idx)
img = mine["prompt_embeds"].float().sum(dim=(1, 2)).view(-1, 1, 1, 1) + torch.zeros((len(idx), 3, 8, 8), device=dev)
out = D.gather_images(img, 3, dst=0)
ok_g = out is not None and out.shape == (3, 3, 8, 8) and torch.equal(out, img)
m = D.max_over_ranks(1.25, dev)
dist.barrier()
torch.cuda.synchronize()
print(json.dumps({"backend": D.backend_name(), "world": dist.get_world_size(), "broadcast_ok": ok_b, "gather_ok": bool(ok_g), "max": m,
                  "device": torch.cuda.get_device_name(local)}), flush=True)
dist.destroy_process_group()

#!/usr/bin/env python3
"""Developer tool: exercises sharding.RootGatherer on the nccl (RCCL) backend.
On a multi-GPU node:  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/nccl_gather_check.py
On a single-GPU box the same command maps every rank to cuda:0, which RCCL may refuse (duplicate GPU)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev_index = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
torch.cuda.set_device(dev_index)
dist.init_process_group("nccl")
entry.load_package()
from dmz_amd import sharding

dev = torch.device("cuda", dev_index)
g = sharding.RootGatherer(world)
bufs = [torch.full((1000, 1024), 10 * rank + k, dtype=torch.uint8, device=dev) for k in range(2)]
outs = []
for step in range(4):
    k = step % 2
    g.wait(slots=(k,))
    bufs[k].fill_(10 * rank + step)
    outs.append(g.submit(bufs[k], slot=k))
g.wait()
torch.cuda.synchronize()
if rank == 0:
    last = outs[-1].view(world, 1000, 1024)
    ok = all(int(last[r, 0, 0]) == 10 * r + 3 for r in range(world))
    print("nccl gather check:", "ok" if ok else "MISMATCH", [int(last[r, 0, 0]) for r in range(world)])
dist.barrier()
dist.destroy_process_group()

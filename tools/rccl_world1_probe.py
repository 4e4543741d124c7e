#!/usr/bin/env python3
"""RCCL on the one GPU of a box: a process group of ONE rank over the "nccl" backend and the collectives the
multi-GPU paths call (scoring: all_gather of the scores, barrier, all_reduce MAX of the elapsed time; training:
all_reduce of the flat gradient, reduce_scatter_tensor / all_gather_into_tensor of ShardedAdam).  With one rank every
collective is the identity, so this only shows that the library loads, a communicator comes up on the device and the
calls with these argument forms run; two ranks need two GPUs.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_world1_probe.py"""
import os

import torch
import torch.distributed as dist

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
x = torch.arange(4096, device=dev, dtype=torch.float32)
y = x.clone()
dist.all_reduce(y)
assert torch.equal(x, y)
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 1.5
out = [torch.empty_like(x)]
dist.all_gather(out, x)
assert torch.equal(out[0], x)
shard = torch.empty(4096, device=dev, dtype=torch.bfloat16)
w = dist.reduce_scatter_tensor(shard, x.to(torch.bfloat16), op=dist.ReduceOp.SUM, async_op=True)
w.wait()
assert torch.equal(shard, x.to(torch.bfloat16))
full = torch.empty(4096, device=dev)
w = dist.all_gather_into_tensor(full, x, async_op=True)
w.wait()
assert torch.equal(full, x)
dist.barrier()
torch.cuda.synchronize(dev)
print("rccl world-1 probe ok:", torch.cuda.get_device_name(dev), "nccl", ".".join(map(str, torch.cuda.nccl.version())))
dist.destroy_process_group()

"""Probe: does RCCL accept two ranks on ONE GPU (the builder's box has one)?  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dbg/rccl_two_ranks_one_gpu.py"""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://")
x = torch.full((1 << 20,), float(rank + 1), device="cuda")
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank", rank, "sum", float(x[0]), flush=True)
dist.destroy_process_group()

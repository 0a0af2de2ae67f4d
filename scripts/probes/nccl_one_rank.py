"""RCCL sanity on one GPU: the calls bench.py's multi-rank context makes (init with device_id, barrier, all_reduce MAX / MIN,
all_gather), in a world of one.  The send / recv pairs of the halo hand-over need two GPUs and are not covered."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([3.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
u = torch.tensor([1.0], dtype=torch.float64, device="cuda:0")
dist.all_reduce(u, op=dist.ReduceOp.MIN)
parts = [torch.empty_like(t)]
dist.all_gather(parts, t)
torch.cuda.synchronize()
print("nccl world of one ok:", t.item(), u.item(), parts[0].item(), "backend", dist.get_backend())
dist.destroy_process_group()

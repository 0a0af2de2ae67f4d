"""Probe: does the RCCL ("nccl") backend come up on this box and run bench.py's collectives with one rank?
(The N > 1 nccl branch of bench.py has only ever run as gloo rehearsals; two ranks cannot share one GPU under RCCL.)"""
import os
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([3.5], dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
parts = [torch.empty_like(t)]
dist.all_gather(parts, t)
x = torch.randn(1024, dtype=torch.complex64, device="cuda:0")
v = torch.view_as_real(x[-255:]).contiguous()
# no neighbour at world 1: the ring hand-over has no ops; batch_isend_irecv of an empty list must not be called
print("nccl one-rank ok: world", dist.get_world_size(), "max", float(t.item()), "gathered", float(parts[0].item()), "tail", tuple(v.shape))
dist.destroy_process_group()

"""One-rank RCCL rehearsal: the code path bench.py / FFLearner take for N > 1 (init with device_id, barrier,
all-reduce of the flat gradient buffer, max-over-ranks of the timing) on backend "nccl" with world_size 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from mava_amd import parallel

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
flat = torch.arange(76810, dtype=torch.float32, device=dev)
ref = flat.clone()
dist.barrier(); dist.all_reduce(flat, op=dist.ReduceOp.SUM); dist.broadcast(flat, src=0)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
assert torch.equal(flat, ref) and t.item() == 1.5
print("nccl(RCCL) one-rank rehearsal ok:", dist.get_backend(), torch.cuda.get_device_name(0))
dist.destroy_process_group()

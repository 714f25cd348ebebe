"""RCCL sanity on a one-GPU box: the collectives bench.py uses at N > 1 (init with device_id, async
all_gather_into_tensor of uint8, list all_gather, all_reduce MAX, barrier), world size 1."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.arange(1 << 20, device=dev, dtype=torch.int64).to(torch.uint8)
out = torch.empty_like(x)
w = dist.all_gather_into_tensor(out, x, async_op=True)
w.wait()
torch.cuda.synchronize()
assert torch.equal(out, x)
d = [torch.empty(32, dtype=torch.uint8, device=dev)]
dist.all_gather(d, torch.ones(32, dtype=torch.uint8, device=dev))
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("rccl ok: backend", dist.get_backend(), "world", dist.get_world_size(), "max", float(t.item()))
dist.destroy_process_group()

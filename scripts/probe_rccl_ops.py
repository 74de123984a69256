"""GPU probe: the RCCL operations psfm_dist.TorchComm issues, on a 1-rank nccl group (dtype / op support, not scaling):
all_reduce(MAX) on uint8, all_gather_into_tensor on float64 / uint8, all_gather_object, barrier."""
import os
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a = (torch.arange(518401, device="cuda") % 251).to(torch.uint8)
b = a.clone()
dist.all_reduce(b, op=dist.ReduceOp.MAX)
assert torch.equal(a, b)
x = torch.arange(8 * 13, dtype=torch.float64, device="cuda")
out = torch.empty(x.numel(), dtype=torch.float64, device="cuda")
dist.all_gather_into_tensor(out, x)
assert torch.equal(out, x)
p = torch.randint(0, 255, (10 * 259200,), dtype=torch.uint8, device="cuda")
o2 = torch.empty_like(p)
dist.all_gather_into_tensor(o2, p)
assert torch.equal(o2, p)
objs = [None]
dist.all_gather_object(objs, {"n": 3})
t = torch.tensor([1.5, 2.0], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
print("rccl ops ok:", dist.get_backend(), objs)
dist.destroy_process_group()

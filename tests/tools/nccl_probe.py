"""Probe of the torch.distributed calls summersph_amd/dist.py relies on, backend nccl (= RCCL), world_size 1:
the one-GPU test box cannot host two RCCL ranks, but it can show that the collectives exist and accept the
dtypes / shapes used (all_gather_into_tensor on float64 and int64, all_reduce MAX, barrier)."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
mine = torch.arange(193, dtype=torch.float64, device=dev)
out = torch.empty((1, 193), dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(out.view(-1), mine)
assert torch.equal(out[0], mine)
cnt = torch.tensor([3, 4], dtype=torch.int64).to(dev)
o2 = torch.empty((1, 2), dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(o2.view(-1), cnt)
assert o2.cpu().tolist() == [[3, 4]]
red = torch.tensor([1.0, 2.0, 3.0], dtype=torch.float64, device=dev)
dist.all_reduce(red, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
print("nccl probe ok", torch.cuda.get_device_name(0))
dist.destroy_process_group()

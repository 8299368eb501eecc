"""Can two RCCL ranks share the one GPU of the test box?  (If so the N > 1 path can be rehearsed over RCCL there.)
Run under torchrun --nproc-per-node 2."""
import os
import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    mine = torch.full((4,), float(rank), dtype=torch.float64, device=dev)
    out = torch.empty((2, 4), dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out.view(-1), mine)
    peer = 1 - rank
    r = torch.empty(3, dtype=torch.float64, device=dev)
    s = torch.full((3,), 10.0 + rank, dtype=torch.float64, device=dev)
    for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, s, peer), dist.P2POp(dist.irecv, r, peer)]):
        w.wait()
    torch.cuda.synchronize()
    print("rank", rank, "all_gather", out[:, 0].tolist(), "p2p got", r.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("rank", rank, "RCCL with two ranks on one GPU failed:", type(e).__name__, str(e)[:300], flush=True)

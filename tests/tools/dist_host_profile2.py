"""cProfile of rank 0 of a 2-rank gloo run sharing one GPU: host time of the orchestrator per step, by function."""
import cProfile, os, pstats, socket, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
import numpy as np
import torch.multiprocessing as mp


def worker(rank, world, port):
    import torch.distributed as dist
    from summersph_amd import ic
    from summersph_amd.dist import DistSim, HipBackend, slab_bounds
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    gas, sinks = ic.split_rows(ic.keplerian_disc(1000000 * world, seed=202, nngb=85.0))
    bounds = slab_bounds(gas["x"], world)
    sel = np.searchsorted(bounds, gas["x"], side="right") == rank
    mine = {k: v[sel] for k, v in gas.items()}
    mine["gid"] = np.nonzero(sel)[0]
    sim = DistSim(HipBackend(0), mine, sinks, bounds, comm_device="cpu")
    d = sim.run(3, 1e-2)
    pr = cProfile.Profile()
    if rank == 0:
        pr.enable()
    d = sim.run(20, d); sim.be.synchronize()
    if rank == 0:
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
    dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)

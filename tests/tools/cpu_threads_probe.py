"""How many host threads pay on this box?  (bench.py's cpu_baseline uses usable_cores())"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import orc
from summersph_amd import ic
import bench
print("affinity", len(os.sched_getaffinity(0)), "usable_cores", bench.usable_cores())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError as e:
        print(f, "-", e.__class__.__name__)
gas, sinks = ic.split_rows(ic.keplerian_disc(200000, seed=3))
for t in (1, 8, 16, 32, 64, 128):
    o = orc.Oracle(gas, sinks, nthreads=t)
    t0 = time.perf_counter(); o.step(1e-2); t1 = time.perf_counter()
    print(t, "threads:", round(200000 / (t1 - t0)), "particle-steps/s", flush=True)

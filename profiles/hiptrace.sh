#!/bin/bash
# HIP-API traces: how often the host waits for the device (--hip-trace on its own: no counter collection in these passes).
#   fixed : python3 bench.py --steps 40 --warmup 5 --no-cpu --no-variable --no-extras     (the headline path)
#   full  : tests/tools/step_time.py through SPH_STEP_FLAGS=48 (self-gravity + accretion + cull): simulate() as the reference runs it
# usage on the GPU box:  bash profiles/hiptrace.sh <round> <tag>        -> profiles/<round>_hiptrace_summary.txt
set -e
round=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --hip-trace --stats -d $R/gpurun_out/${tag}_hip -o p --output-format csv -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-variable --no-extras > $R/gpurun_out/${tag}_hip.log 2>&1
timeout -k 10 200 rocprofv3 --hip-trace --stats -d $R/gpurun_out/${tag}_hipfull -o p --output-format csv -- python3 $R/tests/tools/full_loop_steps.py 1000000 60 > $R/gpurun_out/${tag}_hipfull.log 2>&1
cd $R && python3 - <<PY > profiles/${round}_hiptrace_summary.txt
import csv, glob
def table(tag, title, notes):
    rows = []
    for f in glob.glob("gpurun_out/%s/**/*hip_api_stats.csv" % tag, recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    print("== rocprofv3 --hip-trace --stats of: " + title + " ==")
    for n in notes: print(n)
    print(f"{'HIP API':44s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s}")
    for r in rows[:14]:
        print(f"{r['Name']:44s} {int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f}")
    print()
table("${tag}_hip", "python3 bench.py --steps 40 --warmup 5 --no-cpu --no-variable --no-extras",
      ["(1 context, upload, 5 warm-up + 40 timed steps of the fixed-h path, 3 restarts of the same 45 steps, 10 steps of the kernel",
       " breakdown, each bracketed by synchronisations; statistics read-back, stream-copy measurement)"])
table("${tag}_hipfull", "python3 tests/tools/full_loop_steps.py 1000000 60",
      ["(1 context with SPH_FLAG_SELF_GRAVITY | SPH_FLAG_ACCRETE_CULL, upload, ONE sph_run of 60 steps: per step one wait for the exact",
       " bounding box -- the octree root -- and one for the survivor count of the accretion / cull pass)"])
PY
cp profiles/${round}_hiptrace_summary.txt gpurun_out/
rm -rf gpurun_out/${tag}_hip gpurun_out/${tag}_hipfull

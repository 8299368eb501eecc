#!/bin/bash
# HIP-API trace of the fixed-h bench run: how often the host waits for the device during the timed steps.
# (--hip-trace on its own: no counter collection in this pass.)
# usage on the GPU box:  bash profiles/r02_hiptrace.sh <tag>
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --hip-trace --stats -d $R/gpurun_out/${tag}_hip -o p --output-format csv -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu --no-variable > $R/gpurun_out/${tag}_hip.log 2>&1
cd $R && python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("gpurun_out/${tag}_hip/**/*hip_api_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("== rocprofv3 --hip-trace --stats of: python3 bench.py --steps 40 --warmup 5 --no-cpu --no-variable ==")
print("(1 context created, upload, 5 warm-up + 40 timed steps of the fixed-h path, 3 x 40 steps of the spread measurement, 10 steps of the")
print(" kernel breakdown, each bracketed by synchronisations; statistics read-back, stream-copy measurement: 175 steps in all)")
print(f"{'HIP API':44s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s}")
for r in rows[:18]:
    print(f"{r['Name']:44s} {int(r['Calls']):8d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.1f}")
PY

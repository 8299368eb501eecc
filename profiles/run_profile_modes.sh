#!/bin/bash
# kernel-trace summaries of the variable-h step and of the full simulate() step (self-gravity + accretion + cull)
# usage on the GPU box: bash profiles/run_profile_modes.sh <tag>
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_var -o kt --output-format csv -- python3 $R/bench.py --mode variable --steps 5 --warmup 1 --no-cpu > $R/gpurun_out/${tag}_var.log 2>&1
echo "variable done"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_full -o kt --output-format csv -- python3 $R/profiles/full_profile.py > $R/gpurun_out/${tag}_full.log 2>&1
echo "full done"
cd $R && python3 profiles/summarize.py gpurun_out/${tag}_var > gpurun_out/${tag}_var_summary.txt && python3 profiles/summarize.py gpurun_out/${tag}_full > gpurun_out/${tag}_full_summary.txt

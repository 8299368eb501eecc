#!/bin/bash
# Profile passes behind profiles/rNN_*: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate PMC runs
# (never combined with a trace domain). Usage on the GPU box:  bash profiles/run_profile.sh <tag> [bench args...]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
args="--steps 5 --warmup 1 --no-cpu $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_kt -o kt --output-format csv -- python3 $R/bench.py $args > $R/gpurun_out/${tag}_kt.log 2>&1
echo "kt done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/${tag}_fetch -o f --output-format csv -- python3 $R/bench.py $args > $R/gpurun_out/${tag}_fetch.log 2>&1
echo "fetch done"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/${tag}_write -o w --output-format csv -- python3 $R/bench.py $args > $R/gpurun_out/${tag}_write.log 2>&1
echo "write done"
cd $R && python3 profiles/summarize.py gpurun_out/${tag}_kt gpurun_out/${tag}_fetch gpurun_out/${tag}_write > gpurun_out/${tag}_summary.txt

#!/bin/bash
# Soak runs of the committed kernels on the GPU box (round 3): protocol and stability evidence, kept as profiles/<round>_soak.txt
#   bash profiles/soak.sh r03
set -e
round=$1
cd $GRAFT_REPO_ROOT
{
echo "== full loop (self-gravity, accretion, cull), 100k disc, 300 steps: tests/tools/long_run_probe.py =="
python tests/tools/long_run_probe.py
echo "== native multi-GPU loop, 3 ranks as threads on one GPU, 300k disc, 600 steps vs one context: tests/tools/halo_long_run.py =="
python tests/tools/halo_long_run.py 3 300000 600
echo "== native multi-GPU loop, 2 ranks, 1e6 disc, 300 steps =="
python tests/tools/halo_long_run.py 2 1000000 300
echo "== variable h, 200k disc, 300 steps with and without the re-flag pass: tests/tools/reflag_long_run.py =="
python tests/tools/reflag_long_run.py 200000 300
} 2>&1 | grep -v amdgpu.ids > gpurun_out/${round}_soak.txt
tail -3 gpurun_out/${round}_soak.txt

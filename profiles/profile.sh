#!/bin/bash
# Profile passes of bench.py's three workloads on the code in the tree.  Per workload five rocprofv3 runs of the same command:
#   kt     --kernel-trace --stats                      (kernel durations)
#   fetch  --pmc FETCH_SIZE                            (HBM reads; its own pass, as MI355X_MICROARCH.md prescribes)
#   write  --pmc WRITE_SIZE                            (HBM writes; its own pass)
#   sq     --pmc SQ_* (8 counters)                     (vector issue, LDS, wave cycles)
#   ta     --pmc TA / TCP / LDS conflicts + SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE
# Counter passes carry --kernel-trace only (never sys / hip / hsa trace domains).
# usage on the GPU box:  bash profiles/profile.sh <round> <tag> [workloads...]      round: r03 ...; workloads: fixed variable full
# Output: profiles/<round>_<workload>_summary.txt and profiles/<round>_limiters.json (which records the sha256 of the kernel
# sources it was taken on: bench.py marks its quotes "stale" once summersph_amd/csrc differs), copied to gpurun_out/ as well.
set -e
round=$1; shift
tag=$1; shift
wl=${@:-fixed variable full}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in $wl; do
  case $w in
    fixed) args="--steps 5 --warmup 2 --no-cpu --no-variable --no-extras" ;;
    variable) args="--steps 6 --warmup 14 --no-cpu --mode variable" ;;   # past the first steps of the relaxing IC, where h jumps
    full) args="--steps 4 --warmup 2 --no-cpu --no-variable --full-simulate" ;;
  esac
  o=$R/gpurun_out/${tag}_$w
  timeout -k 10 150 rocprofv3 --kernel-trace --stats -d ${o}_kt -o p --output-format csv -- python3 $R/bench.py $args > ${o}_kt.log 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d ${o}_fetch -o p --output-format csv -- python3 $R/bench.py $args > ${o}_fetch.log 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d ${o}_write -o p --output-format csv -- python3 $R/bench.py $args > ${o}_write.log 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE -d ${o}_sq -o p --output-format csv -- python3 $R/bench.py $args > ${o}_sq.log 2>&1
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d ${o}_ta -o p --output-format csv -- python3 $R/bench.py $args > ${o}_ta.log 2>&1
  echo "$w done"
done
cd $R && python3 profiles/make_limiters.py ${round} gpurun_out/${tag} $wl
# what is kept: the summaries (copied next to the logs so that they come back from the GPU box); the raw passes are tens of MB
cp profiles/${round}_*_summary.txt profiles/${round}_limiters.json gpurun_out/
for w in $wl; do rm -rf gpurun_out/${tag}_${w}_kt gpurun_out/${tag}_${w}_fetch gpurun_out/${tag}_${w}_write gpurun_out/${tag}_${w}_sq gpurun_out/${tag}_${w}_ta; done

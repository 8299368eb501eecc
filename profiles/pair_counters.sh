#!/bin/bash
# SQ / TA / LDS counters of the pair kernels on the bench disc (counter passes only, no trace domains)
# usage on the GPU box: bash profiles/pair_counters.sh <tag>
set -e
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/${tag}_sq1 -o c --output-format csv -- python3 $R/tests/tools/wt_vs_gather.py > $R/gpurun_out/${tag}_sq1.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM -d $R/gpurun_out/${tag}_sq2 -o c --output-format csv -- python3 $R/tests/tools/wt_vs_gather.py > $R/gpurun_out/${tag}_sq2.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum -d $R/gpurun_out/${tag}_ta -o c --output-format csv -- python3 $R/tests/tools/wt_vs_gather.py > $R/gpurun_out/${tag}_ta.log 2>&1
cd $R && python3 profiles/summarize.py gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2 gpurun_out/${tag}_ta > gpurun_out/${tag}_summary.txt

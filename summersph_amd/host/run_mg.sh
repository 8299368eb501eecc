#!/bin/bash
# run_mg.sh -- the ranks of ONE node for run_sph_hip_mg: rank r on GPU r.
#   run_mg.sh <nranks> <ic.txt> <max_steps> [final_snapshot.txt] [saves] [tend=<t>]
# Rank 0's output goes to the terminal, the others' to mg_rank<r>.log in the working directory.
set -u
here="$(cd "$(dirname "$0")" && pwd)"
n="$1"; shift
idf="$(mktemp -u "${TMPDIR:-/tmp}/sph_halo_id.XXXXXX")"
export HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = 1; r < n; r++)); do
  "$here/run_sph_hip_mg" "$r" "$n" "$idf" "$@" > "mg_rank$r.log" 2>&1 &
  pids+=("$!")
done
"$here/run_sph_hip_mg" 0 "$n" "$idf" "$@"
rc=$?
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
rm -f "$idf"
exit $rc

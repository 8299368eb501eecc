set -e
cd $GRAFT_REPO_ROOT
python bench.py --ic ring --particles 4000000 --no-cpu --no-variable --no-extras > gpurun_out/r03_bench_ring4m.json 2> gpurun_out/r03_ring.err
echo ring done
python bench.py --particles 20000000 --no-cpu --no-variable --no-extras > gpurun_out/r03_bench_disc2e7.json 2> gpurun_out/r03_2e7.err
echo 2e7 done
python bench.py --full-simulate --particles 10000000 --no-cpu --no-variable --no-extras > gpurun_out/r03_bench_full1e7.json 2> gpurun_out/r03_f1e7.err
echo full1e7 done
python bench.py --particles 100000000 --steps 3 --warmup 1 --no-cpu --no-variable --no-extras > gpurun_out/r03_bench_disc1e8.json 2> gpurun_out/r03_1e8.err
echo 1e8 done
bash profiles/hiptrace.sh r03 h3 > gpurun_out/hiptrace_h3.log 2>&1
cp profiles/r03_hiptrace_summary.txt gpurun_out/
echo hiptrace done
{ python tests/tools/halo_rehearsal.py 2 1000000 10 disc; python tests/tools/halo_rehearsal.py 2 500000 10 ring; python tests/tools/halo_rehearsal.py 2 500000 6 disc full; python tests/tools/halo_rehearsal.py 4 250000 6 disc full; SPH_HALO_REPLICATED=1 python tests/tools/halo_rehearsal.py 4 250000 6 disc full; } > gpurun_out/r03_multi_gpu_rehearsal.txt 2>&1
echo rehearsal done

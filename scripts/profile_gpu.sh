#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence bench.py's roofline object refers to,
# in BOTH regimes of the headline kernel.  Kernel timing and PMC counters are collected in SEPARATE runs (PMC passes serialise
# kernels and run at a lower clock; never mix a profiled arm with an un-profiled one).
#   gpurun_out/prof_stats/      --kernel-trace --stats of the DEFAULT bench command (1e7 IVPs, 160 MB working set: Infinity-Cache
#                               resident; the run also contains the 6.4e7-IVP `beyond_infinity_cache` leg: the non-temporal
#                               <..., 4, 1> instantiation of the same kernel shows up as its own row)
#   gpurun_out/prof_fetch/      --pmc FETCH_SIZE   (TCC: 3 of 4 slots -> own pass), 1e7 IVPs
#   gpurun_out/prof_write/      --pmc WRITE_SIZE
#   gpurun_out/prof_big_stats/  --kernel-trace --stats, 6.4e7 IVPs per launch (1 GB of ping-pong state: real HBM traffic)
#   gpurun_out/prof_big_fetch/, prof_big_write/   the two PMC passes at 6.4e7 IVPs
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
BENCH_ARGS="${BENCH_ARGS:---steps 3 --warmup 1 --no-cpu-baseline}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o bench -- python bench.py $BENCH_ARGS > gpurun_out/prof_stats.log 2>&1
grep '^{' gpurun_out/prof_stats.log | tail -1 > gpurun_out/prof_stats_bench.json
PMC_ARGS="${PMC_ARGS:---steps 1 --warmup 0 --rk4-steps 20 --no-cpu-baseline --no-fused}"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_fetch -o bench -- python bench.py $PMC_ARGS > gpurun_out/prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_write -o bench -- python bench.py $PMC_ARGS > gpurun_out/prof_write.log 2>&1
BIG_ARGS="--n-ivp 6.4e7 --steps 2 --warmup 1 --rk4-steps 100 --no-cpu-baseline --no-fused"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_big_stats -o bench -- python bench.py $BIG_ARGS > gpurun_out/prof_big_stats.log 2>&1
grep '^{' gpurun_out/prof_big_stats.log | tail -1 > gpurun_out/prof_big_stats_bench.json
BIG_PMC="--n-ivp 6.4e7 --steps 1 --warmup 0 --rk4-steps 20 --no-cpu-baseline --no-fused"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_big_fetch -o bench -- python bench.py $BIG_PMC > gpurun_out/prof_big_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_big_write -o bench -- python bench.py $BIG_PMC > gpurun_out/prof_big_write.log 2>&1
ls gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_big_stats gpurun_out/prof_big_fetch gpurun_out/prof_big_write

#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: collects the rocprofv3 evidence bench.py's roofline
# object refers to.  Kernel timing and PMC counters are collected in SEPARATE runs (PMC passes serialise
# kernels and run at a lower clock; never mix a profiled arm with an un-profiled one).
#   gpurun_out/prof_stats/  --kernel-trace --stats of the default bench command
#   gpurun_out/prof_fetch/  --pmc FETCH_SIZE   (TCC: 3 of 4 slots -> own pass)
#   gpurun_out/prof_write/  --pmc WRITE_SIZE
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
ls -la gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write

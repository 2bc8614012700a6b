#!/bin/bash
# Round 3: the measurements DESIGN.md section 6 quotes, in one gpurun call:  gpurun --timeout 2400 -- 'bash scripts/run_r03_gpu.sh'
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python bench.py > gpurun_out/bench_default.log 2>&1; grep '^{' gpurun_out/bench_default.log | tail -1 > gpurun_out/r03_bench_default.json
bash scripts/profile_gpu.sh > gpurun_out/profile.log 2>&1
timeout 300 python scripts/bench_configs.py > gpurun_out/r03_bench_configs.json 2> gpurun_out/bench_configs.err
timeout 600 python scripts/bench_adaptive_stream.py > gpurun_out/r03_bench_adaptive_stream.json 2> gpurun_out/bench_adaptive_stream.err
timeout 300 python scripts/ab_host_pinned.py > gpurun_out/r03_host_path.json 2> gpurun_out/host_path.err
timeout 300 python scripts/bench_extra.py > gpurun_out/r03_bench_extra.json 2> gpurun_out/bench_extra.err
bash scripts/profile_configs.sh > gpurun_out/profile_cfg.log 2>&1
tools/bin/bench_c5 --gpus 1 --steps 10 --warmup 2 --verify 2>/dev/null | grep "^{" > gpurun_out/r03_bench_c5_cpp.json
tail -1 gpurun_out/r03_bench_default.json | cut -c1-400

#!/bin/bash
# Round 4: the measurements DESIGN.md section 6 quotes, in one gpurun call:  gpurun --timeout 2700 -- 'bash scripts/run_r04_gpu.sh'
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 300 python bench.py > gpurun_out/bench_default.log 2>&1; grep '^{' gpurun_out/bench_default.log | tail -1 > gpurun_out/r04_bench_default.json
bash scripts/profile_gpu.sh > gpurun_out/profile.log 2>&1
timeout 300 python scripts/bench_configs.py > gpurun_out/bench_configs.log 2> gpurun_out/bench_configs.err
timeout 600 python scripts/bench_adaptive_stream.py > gpurun_out/bench_adaptive_stream.log 2> gpurun_out/bench_adaptive_stream.err
timeout 300 python scripts/bench_extra.py > gpurun_out/bench_extra.log 2> gpurun_out/bench_extra.err
timeout 300 python scripts/bench_divergence.py > gpurun_out/r04_bench_divergence.json 2> gpurun_out/bench_divergence.err
bash scripts/profile_configs.sh > gpurun_out/profile_cfg.log 2>&1
g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include tests/cpp/bench_c5.cpp -L numericalnim_amd/csrc -lnnhip_ode -L /opt/rocm/lib -lamdhip64 \
    -Wl,-rpath,$PWD/numericalnim_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/bench_c5 && /tmp/bench_c5 --gpus 1 --steps 10 --warmup 2 --verify 2>/dev/null | grep "^{" > gpurun_out/r04_bench_c5_cpp.json
g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include tests/cpp/bench_multithread_launch.cpp -L numericalnim_amd/csrc -lnnhip_ode -L /opt/rocm/lib -lamdhip64 -lpthread \
    -Wl,-rpath,$PWD/numericalnim_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/bml && timeout 600 /tmp/bml > gpurun_out/r04_multithread_launch.json
tail -1 gpurun_out/r04_bench_default.json | cut -c1-600

#!/bin/bash
# Everything that needs the MI355X, in one gpurun call:  gpurun --timeout 1800 -- 'bash scripts/run_all_gpu.sh'
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log | cut -c1-300
bash scripts/profile_gpu.sh > gpurun_out/profile.log 2>&1
bash scripts/profile_configs.sh > gpurun_out/profile_cfg.log 2>&1
timeout 300 python scripts/bench_configs.py > gpurun_out/bench_configs.log 2>&1
timeout 300 python scripts/bench_extra.py > gpurun_out/bench_extra.log 2>&1
timeout 300 python scripts/bench_adaptive_stream.py > gpurun_out/bench_adaptive_stream.log 2>&1
timeout 300 python scripts/bench_cumquad.py > gpurun_out/bench_cumquad.log 2>&1
timeout 300 python scripts/bench_wide.py > gpurun_out/bench_wide.log 2>&1
timeout 300 python scripts/bench_divergence.py > gpurun_out/bench_divergence.log 2>&1
ADV_BENCH_ONLY=C3_lorenz_N1e+07 bash scripts/profile_adv.sh > gpurun_out/profile_adv.log 2>&1
echo "then, in the build container: python scripts/summarize_profiles.py --round N; python scripts/summarize_configs_pmc.py --round N   (summaries into profiles/)"

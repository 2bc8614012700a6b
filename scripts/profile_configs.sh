#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 evidence for the fused adaptive kernels (BASELINE configs C3 / C4) —
# kernel-trace stats in one run, SQ VALU counters in a separate --pmc run.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg_stats -o cfg -- python scripts/bench_configs.py > gpurun_out/prof_cfg_stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_cfg_pmc -o cfg -- python scripts/bench_configs.py > gpurun_out/prof_cfg_pmc.log 2>&1
tail -3 gpurun_out/prof_cfg_pmc.log
ls -la gpurun_out/prof_cfg_stats gpurun_out/prof_cfg_pmc

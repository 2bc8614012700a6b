#!/bin/bash
# A/B on the GPU box: cap the VGPRs of the fused lanes-per-system kernels (amdgpu_waves_per_eu) and time C4.
# gpurun --timeout 1500 -- 'bash scripts/ab_lps_occupancy.sh'
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
C=numericalnim_amd/csrc
run() { timeout 300 python scripts/bench_configs.py 2>/dev/null | python -c "
import json,sys
s=sys.stdin.read(); d=json.loads(s[s.index('{'):])
print('$1', {k: round(v['ms'],3) for k,v in d.items() if k.startswith('C4') or k.startswith('C3')})"; }
run baseline
for w in 3 4; do
  rm -f $C/ode_tu_m_tsit54.o $C/ode_tu_m_dopri54.o
  make -C $C -j16 HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function -DNNHIP_LPS_WPE=$w" > gpurun_out/ab_lps_build_$w.log 2>&1
  grep -E "scratch|spill" gpurun_out/ab_lps_build_$w.log | head -3
  run wpe$w
done

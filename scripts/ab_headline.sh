#!/bin/bash
# Run ON THE GPU BOX: the headline bench line (no CPU baseline) for the default library and the variants named in VARIANTS.
cd "${GRAFT_REPO_ROOT:-.}"
for v in ${VARIANTS:-default}; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$v', 'launch_us', round(r['avg_launch_us'],3), 'frac', round(r['frac'],4), 'hbm_only', round(r['frac_hbm_only'],4), 'fused_ms', round(d['fused_solve']['ms_per_solve'],3))"
  done
done

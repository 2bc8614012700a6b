#!/bin/bash
# Run ON THE GPU BOX: A/B of build variants over ALL configs of scripts/bench_adaptive_stream.py (C3 1e6, C3 1e7, C4).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in ${VARIANTS:-default}; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  echo "=== $v"
  timeout 300 python scripts/bench_adaptive_stream.py > gpurun_out/ab_all_$v.json 2> gpurun_out/ab_all_$v.err || tail -3 gpurun_out/ab_all_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_all_$v.json"))
for k,x in d.items():
    if k.endswith("_graph"): print(k, round(x["us_per_iteration"],1), "us/iter", round(x["GBps"]), "GB/s", x["equal_to_fused"])
PY
done

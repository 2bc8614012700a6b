#!/bin/bash
# Run ON THE GPU BOX: A/B of build variants of the streamed C4 kernel (advance_lps_kernel); variants from scripts/build_variant.sh.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in ${VARIANTS:-default cpl4 wpe5 wpe6 default}; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  echo "=== $v"
  ADV_BENCH_ONLY=C4 timeout 300 python scripts/bench_adaptive_stream.py > gpurun_out/ab_c4_$v.json 2> gpurun_out/ab_c4_$v.err || tail -3 gpurun_out/ab_c4_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_c4_$v.json"))
for k,x in d.items():
    if k.endswith("_graph"): print(k, round(x["us_per_iteration"],1), "us/iter", round(x["GBps"]), "GB/s", x["equal_to_fused"])
PY
done

#!/bin/bash
# Run ON THE GPU BOX: A/B of where pow's tables live (LDS copy = default build, global memory, no tables = correctly rounded
# nth_root) on the kernels that call the controller.  Variants built by scripts/build_variant.sh.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for v in ${VARIANTS:-default wpe fastroot}; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  echo "=== $v"
  python scripts/bench_adaptive_stream.py > gpurun_out/ab_pow_${v}_stream.json 2>gpurun_out/ab_pow_${v}.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_pow_${v}_stream.json"))
for k,v in d.items():
    if "_graph" in k: print(k, round(v["us_per_iteration"],1), "us/iter", round(v["GBps"]), "GB/s", "fused", round(v["fused_ms"],2), "ms", v["equal_to_fused"])
PY
done

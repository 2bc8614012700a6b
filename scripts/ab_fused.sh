#!/bin/bash
# Run ON THE GPU BOX: fused C3 / C4 solves (scripts/bench_configs.py) for the default library and the variants named in VARIANTS.
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
for v in ${VARIANTS:-default}; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  python scripts/bench_configs.py > gpurun_out/ab_fused_$v.json 2>gpurun_out/ab_fused_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_fused_$v.json"))
print("=== $v", {k: round(x.get("ms", x.get("us", 0)), 3) for k, x in d.items() if k.startswith(("C3", "C4", "step"))})
PY
done

#!/bin/bash
# Run ON THE GPU BOX: LDS slot padding of the lanes-per-system kernels (default build: NNHIP_LPS_PAD=2; variant pad0).
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in default pad0 default pad0; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  echo "=== $v"
  ADV_BENCH_ONLY=C4 timeout 300 python scripts/bench_adaptive_stream.py > gpurun_out/ab_pad_$v.json 2> gpurun_out/ab_pad_$v.err || tail -3 gpurun_out/ab_pad_$v.err
  timeout 300 python scripts/bench_configs.py > gpurun_out/ab_padcfg_$v.json 2> gpurun_out/ab_padcfg_$v.err || tail -3 gpurun_out/ab_padcfg_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_pad_$v.json"))
for k,x in d.items():
    if k.endswith("_graph"): print(k, round(x["us_per_iteration"],1), "us/iter", round(x["GBps"]), "GB/s", x["equal_to_fused"])
d=json.load(open("gpurun_out/ab_padcfg_$v.json"))
print({k: round(x["ms"],3) for k,x in d.items() if k.startswith("C4_")})
PY
done

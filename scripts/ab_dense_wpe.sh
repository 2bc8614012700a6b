#!/bin/bash
# Run ON THE GPU BOX: dense lanes-per-system solves at 2 (default) vs 3 waves per SIMD (variant dw3 = -DNNHIP_LPS_DENSE_WAVES=3), same box, interleaved
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for rep in 1 2; do for v in default dw3; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  python scripts/bench_dense_fused.py > gpurun_out/ab_dw_${v}_$rep.json 2>/dev/null
done; done
python - <<'PY'
import json
for k in json.load(open("gpurun_out/ab_dw_default_1.json")):
    if "ring16" in k and "rk4" not in k:
        print(f"{k:34s}", {v: [round(json.load(open(f"gpurun_out/ab_dw_{v}_{r}.json"))[k], 3) for r in (1, 2)] for v in ("default", "dw3")})
PY

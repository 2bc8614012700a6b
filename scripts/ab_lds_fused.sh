cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for v in default lds default lds; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  echo "=== $v"
  timeout 300 python scripts/bench_configs.py > gpurun_out/ab_cfg_$v.json 2> gpurun_out/ab_cfg_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_cfg_$v.json"))
print({k: round(x["ms"],3) for k,x in d.items() if k.startswith(("C3_","C4_"))})
PY
done

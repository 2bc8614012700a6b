#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of the round-4 changes to the adaptive kernels (variants from scripts/build_variant.sh):
#   r03norm       LDS error norm, retry loop as in round 3        (-DNNHIP_LPS_CHAIN_MAX_L=0 -DNNHIP_PEEL_STREAM=0)
#   chain_nopeel  ordered register-chain norm, retry loop as before (-DNNHIP_PEEL_STREAM=0)
#   default       chain norm + first attempt peeled in the step-streaming kernels
#   peel_fused    ... and in the fused solves too                 (-DNNHIP_PEEL_FUSED=1)
# Box-to-box spread on this pool is 4-5 %, so every variant runs twice, interleaved.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for rep in 1 2; do
for v in ${VARIANTS:-r03norm chain_nopeel default peel_fused}; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  ADV_BENCH_MODES=default,fsal_carried ADV_BENCH_ONLY=${ADV_BENCH_ONLY:-} timeout 300 python scripts/bench_adaptive_stream.py > gpurun_out/ab_np_${v}_$rep.json 2> gpurun_out/ab_np_${v}_$rep.err || tail -3 gpurun_out/ab_np_${v}_$rep.err
  timeout 300 python scripts/bench_configs.py > gpurun_out/ab_np_cfg_${v}_$rep.json 2> gpurun_out/ab_np_cfg_${v}_$rep.err || tail -3 gpurun_out/ab_np_cfg_${v}_$rep.err
done
done
python - <<'PY'
import json, glob, collections
out = collections.defaultdict(dict)
for v in ("r03norm", "chain_nopeel", "default", "peel_fused"):
    for rep in (1, 2):
        try:
            d = json.load(open(f"gpurun_out/ab_np_{v}_{rep}.json"))
            for k, x in d.items():
                out[f"stream_us_per_iteration:{k}"].setdefault(v, []).append(round(x["us_per_iteration"], 2))
                assert x["equal_to_fused"]
            d = json.load(open(f"gpurun_out/ab_np_cfg_{v}_{rep}.json"))
            for k, x in d.items():
                if k.startswith("C3") or k.startswith("C4"):
                    out[f"fused_ms:{k}"].setdefault(v, []).append(round(x["ms"], 3))
        except Exception as e:
            print("missing", v, rep, e)
json.dump(out, open("gpurun_out/r04_norm_chain_ab.json", "w"), indent=1)
for k, x in out.items():
    print(k, {v: min(t) for v, t in x.items()})
PY

cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for rep in 1 2 3; do
for v in pinall default stag24 stag47 stag94; do
  if [ "$v" = default ]; then unset NNHIP_LIB; else export NNHIP_LIB=$PWD/numericalnim_amd/csrc/variants/libnnhip_ode_$v.so; fi
  ADV_BENCH_ONLY=C4 ADV_BENCH_MODES=default,fsal_carried timeout 300 python scripts/bench_adaptive_stream.py > gpurun_out/ab_st_${v}_$rep.json 2> gpurun_out/ab_st_${v}_$rep.err || tail -3 gpurun_out/ab_st_${v}_$rep.err
done; done
python - <<'PY'
import json, collections
out = collections.defaultdict(dict)
for v in ("pinall", "default", "stag24", "stag47", "stag94"):
    for rep in (1, 2, 3):
        try:
            d = json.load(open(f"gpurun_out/ab_st_{v}_{rep}.json"))
            for k, x in d.items():
                assert x["equal_to_fused"]
                out[k].setdefault(v, []).append(round(x["us_per_iteration"], 2))
        except Exception as e: print("missing", v, rep, e)
json.dump(out, open("gpurun_out/r04_c4_stagger_ab.json", "w"), indent=1)
for k, x in out.items(): print(k, {v: t for v, t in x.items()})
PY

#!/bin/bash
# Run ON THE GPU BOX: workgroup size of the thread-per-IVP advance kernel (tuning knob "adv_block") on the C3 streamed configs.
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out
for b in 256 128 64 256 128 64; do
  echo "=== adv_block $b"
  ADV_BLOCK=$b ADV_BENCH_ONLY=C3 timeout 300 python scripts/bench_adaptive_stream.py > gpurun_out/ab_blk_$b.json 2> gpurun_out/ab_blk_$b.err || tail -3 gpurun_out/ab_blk_$b.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_blk_$b.json"))
for k,x in d.items():
    if k.endswith("_graph"): print(k, round(x["us_per_iteration"],1), "us/iter", round(x["GBps"]), "GB/s", x["equal_to_fused"])
PY
done

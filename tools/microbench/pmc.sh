#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 PMC passes (counters only, one group per pass) over the micro-harness; per-kernel averages.
# usage: tools/microbench/pmc.sh <tag> <harness args...>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG="$1"; shift
mkdir -p gpurun_out
G=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_IFETCH"
 )
i=0
for g in "${G[@]}"; do
  timeout 300 rocprofv3 --pmc $g --output-format csv -d gpurun_out/pmc_${TAG}_$i -o p -- tools/microbench/mb_adv "$@" > gpurun_out/pmc_${TAG}_$i.log 2>&1 || tail -3 gpurun_out/pmc_${TAG}_$i.log
  i=$((i+1))
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, json, sys, re
tag = sys.argv[1]
out = collections.defaultdict(dict)
for d in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "advance_" not in k: continue
            k = re.sub(r"\(.*", "", k).replace("void nnhip::", "")
            agg[(k, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, gs, c), v in agg.items():
            v.sort()
            out[f"{k} grid={gs}"][c] = v[len(v) // 2]
json.dump(out, open(f"gpurun_out/pmc_{tag}.json", "w"), indent=1)
for k, d in out.items():
    print(k)
    print("   ", "  ".join(f"{c}={v:.4g}" for c, v in sorted(d.items())))
PY

#!/bin/bash
# Run ON THE GPU BOX: effective shader clock per kernel = GRBM_GUI_ACTIVE / duration (rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
TAG="$1"; shift
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/clk_${TAG} -o p -- tools/microbench/mb_adv "$@" > gpurun_out/clk_${TAG}.log 2>&1 || tail -3 gpurun_out/clk_${TAG}.log
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys, re
tag = sys.argv[1]
dur = {}
for f in glob.glob(f"gpurun_out/clk_{tag}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")))
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/clk_{tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur: continue
        d, k, g = dur[r["Dispatch_Id"]]
        if "advance_" not in k: continue
        k = re.sub(r"\(.*", "", k).replace("void nnhip::", "")
        agg[(k, r["Grid_Size"])].append((d, float(r["Counter_Value"])))
for (k, g), v in agg.items():
    v.sort()
    d, c = v[len(v) // 2]
    print(f"{k} grid={g}: {d/1e3:.1f} us, GRBM_GUI_ACTIVE {c:.4g} -> {c/d:.2f} counts/ns  ({c/d/8:.2f} GHz if summed over 8 XCDs)")
PY

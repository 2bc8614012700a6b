#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 evidence for the HBM-resident adaptive loop (advance kernels): kernel trace + SQ counters + TCC traffic,
# each in its own pass.  ADV_BENCH_ONLY selects the config (default C3 at 1e7 IVPs).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
export ADV_BENCH_ONLY="${ADV_BENCH_ONLY:-C3_lorenz_N1e+07}"
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_adv_stats -o adv -- python scripts/bench_adaptive_stream.py > gpurun_out/prof_adv_stats.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof_adv_sq -o adv -- python scripts/bench_adaptive_stream.py > gpurun_out/prof_adv_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_INST_CYCLES_VMEM --output-format csv -d gpurun_out/prof_adv_sq2 -o adv -- python scripts/bench_adaptive_stream.py > gpurun_out/prof_adv_sq2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_adv_fetch -o adv -- python scripts/bench_adaptive_stream.py > gpurun_out/prof_adv_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_adv_write -o adv -- python scripts/bench_adaptive_stream.py > gpurun_out/prof_adv_write.log 2>&1
python - <<'PY'
import csv, glob, collections
def load(d):
    f = glob.glob(f"gpurun_out/{d}/*counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []
for d in ("prof_adv_sq", "prof_adv_sq2", "prof_adv_fetch", "prof_adv_write"):
    rows = load(d)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if "advance" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        # working launches only: the upper half by value of the first counter
        print(d, k, {c: (sorted(v)[int(len(v) * 0.75)], len(v)) for c, v in cs.items()})
PY

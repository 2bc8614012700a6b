#!/bin/bash
# Run ON THE GPU BOX: where does the streamed C4 kernel (advance_lps_kernel, 1e6 x 16) spend its time?  rocprofv3 PMC passes
# (counters only, each group in its own pass; the per-channel TCC_EA0_* groups made rocprofv3 abort on this pool and are left out) over scripts/bench_adaptive_stream.py restricted to the C4 config.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp ADV_BENCH_ONLY="${ADV_BENCH_ONLY:-C4}"
export KFILTER="${KFILTER:-advance_lps_kernel<2,}" TAG="${TAG:-c4}"   # e.g. ADV_BENCH_ONLY=C3_lorenz_N1e+06 KFILTER="advance_tpi_kernel<2," TAG=c3
mkdir -p gpurun_out
G=(
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LEVEL_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
 "SQ_IFETCH SQ_IFETCH_LEVEL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
 "TCP_PENDING_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES"
 )
i=0
for g in "${G[@]:0:${PMC_GROUPS:-5}}"; do
  timeout 300 rocprofv3 --pmc $g --output-format csv -d gpurun_out/prof_${TAG}_$i -o c4 -- python scripts/bench_adaptive_stream.py > gpurun_out/prof_${TAG}_$i.log 2>&1 || tail -3 gpurun_out/prof_${TAG}_$i.log
  i=$((i+1))
done
python - <<'PY'
import csv, glob, collections, json, os
TAG, KF = os.environ["TAG"], os.environ["KFILTER"]
out = collections.defaultdict(dict)   # per kernel instantiation (carried / re-evaluated FSAL and K > 1 are different template arguments)
for d in sorted(glob.glob(f"gpurun_out/prof_{TAG}_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if KF in r["Kernel_Name"]:   # e.g. "advance_lps_kernel<2," = Tsit54 (kernel names are demangled)
                agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in agg.items():
            v.sort()
            out[k][c] = dict(p50=v[len(v) // 2], p75=v[int(len(v) * 0.75)], n=len(v))   # working launches: upper part of the distribution
print(json.dumps(out, indent=1))
json.dump(out, open(f"gpurun_out/prof_{TAG}_counters.json", "w"), indent=1)
PY

#!/usr/bin/env python3
"""Turn the raw rocprofv3 output under gpurun_out/ (scratch) into the committed summaries under profiles/:
  profiles/rNN_kernel_stats.csv     --kernel-trace --stats summary (our kernels + top others)
  profiles/rNN_pmc_hbm.csv          per-dispatch FETCH_SIZE / WRITE_SIZE of our kernels
  profiles/pmc_traffic.json         HBM bytes per launch, corrected as MI355X_MICROARCH.md §HBM prescribes
                                    (gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced
                                     reads -> x2; both counters are in KiB -> x1024), read by bench.py.
"""
import argparse
import csv
import json
import os
import statistics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def collect(a, tag, stats_dir, bench_json, fetch_dir, write_dir, suffix, kernel_filter):
    """One regime: kernel-trace stats row(s) of the headline kernel + FETCH/WRITE per launch.  kernel_filter(name) picks the row."""
    out = {}
    ks = os.path.join(a.src, stats_dir, "bench_kernel_stats.csv")
    if os.path.exists(ks):
        rows = list(csv.DictReader(open(ks)))
        with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats{suffix}.csv"), "w") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows[:12]:
                w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
        for r in rows:
            if a.kernel in r["Name"] and kernel_filter(r["Name"]):
                out["kernel"] = r["Name"][:140]
                out["avg_launch_ns_rocprof"] = float(r["AverageNs"])
                out["calls_rocprof"] = int(r["Calls"])
                break
    bj = os.path.join(a.src, bench_json)
    if os.path.exists(bj) and os.path.getsize(bj):
        b = json.load(open(bj))
        out["avg_launch_us_bench_hip_events_same_run"] = b["roofline"]["avg_launch_us"]
        out["ivps_per_launch"] = b["config"]["ivps_per_gpu"]
        json.dump(b, open(os.path.join(ROOT, "profiles", f"{tag}_bench_under_rocprof{suffix}.json"), "w"), indent=1)
    vals = {}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm{suffix}.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["Counter", "Kernel", "Dispatch_Id", "Grid_Size", "VGPR_Count", "Value_KiB"])
        for name, sub in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
            p = os.path.join(a.src, sub, "bench_counter_collection.csv")
            if not os.path.exists(p):
                continue
            for r in csv.DictReader(open(p)):
                if "nnhip::" in r["Kernel_Name"] and r["Counter_Name"] == name:
                    w.writerow([name, r["Kernel_Name"][:120], r["Dispatch_Id"], r["Grid_Size"], r["VGPR_Count"], r["Counter_Value"]])
                    if a.kernel in r["Kernel_Name"]:
                        vals.setdefault(name, []).append(float(r["Counter_Value"]))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        fetch = statistics.median(vals["FETCH_SIZE"]) * 1024.0 * 2.0  # gfx950 correction, MI355X_MICROARCH.md §HBM
        write = statistics.median(vals["WRITE_SIZE"]) * 1024.0
        out.update({"fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                    "fetch_size_raw_kib_median": statistics.median(vals["FETCH_SIZE"]),
                    "write_size_raw_kib_median": statistics.median(vals["WRITE_SIZE"]),
                    "dispatches_sampled": len(vals["FETCH_SIZE"]),
                    "correction": "FETCH_SIZE KiB*1024*2 (gfx950 wide-read half-count), WRITE_SIZE KiB*1024; separate --pmc passes"})
    if "avg_launch_ns_rocprof" in out and "ivps_per_launch" in out:
        out["achieved_GBps_from_rocprof"] = 16.0 * out["ivps_per_launch"] / out["avg_launch_ns_rocprof"]
        out["frac_of_8TBps_from_rocprof"] = out["achieved_GBps_from_rocprof"] / 8000.0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", type=int, default=1)
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out"))
    ap.add_argument("--kernel", default="rk4_stream_vec_kernel")
    a = ap.parse_args()
    tag = f"r{a.round:02d}"
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    # headline regime: 1e7 IVPs per launch, plain <..., 1, 0> instantiation (160 MB working set, Infinity-Cache resident)
    out = collect(a, tag, "prof_stats", "prof_stats_bench.json", "prof_fetch", "prof_write", "", lambda n: ", 1, 0>" in n)
    # HBM-only regime: 6.4e7 IVPs per launch, non-temporal <..., 4, 1> instantiation (1 GB working set)
    big = collect(a, tag, "prof_big_stats", "prof_big_stats_bench.json", "prof_big_fetch", "prof_big_write", "_6.4e7", lambda n: ", 4, 1>" in n)
    # the benchmark logs of the same gpurun call: keep the JSON each script printed (its last '{'-line)
    for name in ("bench_default", "bench_configs", "bench_extra", "bench_adaptive_stream", "bench_cumquad", "bench_wide"):
        lp = os.path.join(a.src, name + ".log")
        if not os.path.exists(lp):
            continue
        text = open(lp, errors="replace").read()
        start = None
        for line in text.splitlines():
            if line.startswith("{"):
                start = line
        if start is None:
            continue
        try:
            obj = json.loads(start)            # single-line JSON
        except ValueError:
            try:
                obj = json.loads(text[text.index("{"):])  # pretty-printed JSON to the end of the log
            except ValueError:
                continue
        json.dump(obj, open(os.path.join(ROOT, "profiles", f"{tag}_{name}.json"), "w"), indent=1)
    pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    allj = json.load(open(pj)) if os.path.exists(pj) else {}
    allj["rk4_stream"] = out
    if big:
        allj["rk4_stream_beyond_infinity_cache"] = big
    allj["round"] = a.round
    json.dump(allj, open(pj, "w"), indent=1)
    print(json.dumps({"rk4_stream": out, "rk4_stream_beyond_infinity_cache": big}, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  The -m gpu test files themselves, run WITHOUT a GPU: every test id is its own pytest process on the ISA-backed fake node
(tests/cpp/fake_hip.cpp preloaded, kernel launches executed from the library's gfx950 code objects by tools/gfx950_isa_interp.py) with tests/fake_torch standing in
for PyTorch.  The interpreter is 5-6 orders of magnitude slower than the device: tests sized for a GPU (1e5 ... 1e7 IVPs) end in the per-test timeout and are listed
as such, not as failures.  Usage:

    python scripts/run_gpu_suite_on_isa_node.py [-j 7] [--timeout 600] [--devices 1] [-o profiles/r05_gpu_suite_on_isa_node.txt] [pytest selection ...]

A record of what ran in round 5 (no GPU access during the whole round) is profiles/r05_gpu_suite_on_isa_node.txt."""
import argparse
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=max(1, (os.cpu_count() or 2) - 1))
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--devices", type=int, default=1)
    ap.add_argument("-o", default=None)
    ap.add_argument("-k", default=None)
    ap.add_argument("sel", nargs="*", default=["tests"])
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="fake_node_")
    lib = os.path.join(d, "libfakehip.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "fake_hip.cpp"), "-o", lib])
    os.symlink(lib, os.path.join(d, "librccl.so.1"))
    env = dict(os.environ, LD_PRELOAD=lib, FAKE_HIP_LIB=lib, FAKE_HIP_DEVICES=str(a.devices), LD_LIBRARY_PATH=d + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               PYTHONPATH=os.path.join(ROOT, "tests", "fake_torch") + ":" + ROOT, OMP_NUM_THREADS="1")
    col = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"] + (["-k", a.k] if a.k else []) + a.sel,
                         cwd=ROOT, capture_output=True, text=True)
    ids = [ln.strip() for ln in col.stdout.splitlines() if "::" in ln]
    print("%d test ids, %d at a time, %d s each at most, %d fake device(s)" % (len(ids), a.j, a.timeout, a.devices), flush=True)

    def run(tid):
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-p", "no:cacheprovider", "-p", "no:xdist", "-o", "addopts=", "--timeout", str(a.timeout + 60), tid],
                               cwd=ROOT, env=env, capture_output=True, text=True, timeout=a.timeout)
            out = r.stdout + r.stderr
            m = re.search(r"(\d+) (passed|failed|skipped|error)", out)
            status = {0: "passed", 5: "deselected"}.get(r.returncode, "FAILED")
            if r.returncode == 0 and m and m.group(2) == "skipped":
                status = "skipped"
            if r.returncode < 0:
                status = "CRASHED (signal %d)" % -r.returncode
            why = ""
            if status in ("FAILED",) or status.startswith("CRASHED"):
                es = [ln for ln in out.splitlines() if ln.startswith("E  ") or "isa_backed_node:" in ln or "fake_hip:" in ln]
                why = " | ".join(x.strip()[:260] for x in es[:3])
            elif status == "skipped":
                sk = re.search(r"SKIPPED.*|[Ss]kipped: .*", out)
                why = sk.group(0)[:200] if sk else ""
        except subprocess.TimeoutExpired:
            status, why = "too large for the interpreter (> %d s)" % a.timeout, ""
        return tid, status, time.time() - t0, why

    res = []
    out_path = (os.path.join(ROOT, a.o) if not os.path.isabs(a.o) else a.o) if a.o else None
    if out_path:  # results are appended as they arrive: a run that is cut short (session end, timeout of the caller) keeps what it has
        with open(out_path, "a") as f:
            f.write("# %s  -j %d --timeout %d --devices %d %s  (%d ids; lines follow as tests finish)\n" % (time.strftime("%Y-%m-%d %H:%M"), a.j, a.timeout, a.devices, " ".join(a.sel), len(ids)))
    with cf.ThreadPoolExecutor(a.j) as ex:
        futs = [ex.submit(run, t) for t in ids]
        for k, fu in enumerate(cf.as_completed(futs)):
            r = fu.result()
            res.append(r)
            print("%4d/%d  %-34s %6.0f s  %s  %s" % (k + 1, len(ids), r[1], r[2], r[0], r[3]), flush=True)
            if out_path:
                with open(out_path, "a") as f:
                    f.write("%-34s %6.0f s  %s%s\n" % (r[1], r[2], r[0], ("   " + r[3]) if r[3] else ""))
    tally = {}
    for _t, s, _d, _w in res:
        s = "too large for the interpreter" if s.startswith("too large") else s
        tally[s] = tally.get(s, 0) + 1
    summary = "  ".join("%s: %d" % kv for kv in sorted(tally.items()))
    print("SUMMARY  " + summary)
    if out_path:
        with open(out_path, "a") as f:
            f.write("# %s\n" % summary)
    return 0 if all(s in ("passed", "skipped", "deselected") or s.startswith("too large") for _t, s, _d, _w in res) else 1


if __name__ == "__main__":
    sys.exit(main())

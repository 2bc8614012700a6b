#!/usr/bin/env python3
"""How many __global__ kernels the product library carries, by kernel family and translation unit (hygiene: VERDICT r04 #8).
usage: count_instantiations.py [numericalnim_amd/csrc]"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile
import shutil

LLVM = "/opt/rocm/lib/llvm/bin"
d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "numericalnim_amd", "csrc")
fam_total, per_tu, bytes_tu = collections.Counter(), {}, {}
for o in sorted(glob.glob(os.path.join(d, "*.o"))):
    tmp = tempfile.mkdtemp(prefix="cnt_")
    try:
        shutil.copy(o, os.path.join(tmp, "t.o"))
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", "t.o"], cwd=tmp, stdout=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not co:
            continue
        syms = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "--demangle", os.path.join(tmp, co[0])], text=True)
        fams = collections.Counter()
        seen = set()  # (.dynsym and .symtab both list a kernel)
        for ln in syms.splitlines():
            if " FUNC " in ln and ("GLOBAL" in ln or "WEAK" in ln) and " PROTECTED " in ln:
                name = ln.split(None, 7)[-1]
                if name in seen:
                    continue
                seen.add(name)
                m = re.match(r"(?:void )?(?:nnhip(?:_fast)?::)?(?:\(anonymous namespace\)::)?(\w+)", name)
                fams[m.group(1) if m else name[:40]] += 1
        per_tu[os.path.basename(o)] = sum(fams.values())
        bytes_tu[os.path.basename(o)] = os.path.getsize(os.path.join(tmp, co[0]))
        fam_total.update(fams)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
print("kernels by translation unit (device code bytes):")
for k, v in sorted(per_tu.items(), key=lambda kv: -bytes_tu[kv[0]]):
    print("  %-26s %5d kernels  %8.2f MB" % (k, v, bytes_tu[k] / 1e6))
print("kernels by family:")
for k, v in fam_total.most_common():
    print("  %-32s %5d" % (k, v))
print("total: %d kernels, %.1f MB of device code" % (sum(fam_total.values()), sum(bytes_tu.values()) / 1e6))

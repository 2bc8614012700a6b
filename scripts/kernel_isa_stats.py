#!/usr/bin/env python3
"""Static instruction mix of one gfx950 kernel of a built object: extracts the device code object from the host
object (llvm-objdump --offloading), disassembles it and counts the instructions of the kernel whose mangled name matches
a regular expression, by class (FP64 VALU, other VALU by mnemonic, SALU, memory, lane / DPP operations).

A static count weighs every instruction once, whatever path a wave takes — it is the A/B instrument for "how many non-FP64
VALU instructions does the straight-line part carry", not a cycle model; the dynamic figure is SQ_ACTIVE_INST_VALU per wave
(profiles/r0*_c4_stream_pmc).

usage: kernel_isa_stats.py OBJECT 'REGEX' [--dump FILE] [--json]"""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


META = {}  # kernel name -> {vgpr_count, sgpr_count, private_segment_fixed_size (scratch), group_segment_fixed_size (LDS), ...}


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="isa_")
    try:
        o = os.path.join(tmp, "t.o")
        shutil.copy(obj, o)
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", o], cwd=tmp, stdout=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        assert co, "no device code object in " + obj
        notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co[0])], text=True)
        cur = None
        for ln in notes.splitlines():
            m = re.match(r"\s*(-?)\s*\.(\w+):\s*(\S+)\s*$", ln)
            if not m:
                continue
            dash, k, v = m.groups()
            if dash and k == "agpr_count":  # first key of a kernel's entry in amdhsa.kernels (keys are sorted)
                cur = {}
            if cur is None:
                continue
            if k in ("vgpr_count", "sgpr_count", "agpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count"):
                cur[k] = int(v)
            elif k == "name":
                META[v] = cur
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, co[0])], text=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kernels(text):
    cur, body = None, []
    for ln in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", ln)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
        elif cur and ln.strip() and not ln.startswith("Disassembly"):
            body.append(ln.strip())
    if cur:
        yield cur, body


def classify(ins):
    op = ins.split()[0]
    dpp = "dpp" in ins or "row_" in ins or "quad_perm" in ins
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane(read/writelane)"
    if op.startswith("v_"):
        if re.match(r"v_(fma|mul|add|div_scale|div_fmas|div_fixup|rcp|rsq|sqrt|min|max|trig|frexp|ldexp|fract|floor|ceil|rndne|trunc|cmp\w*|cvt)_?\w*f64", op) or op.endswith("_f64") or "_f64_" in op:
            return "valu_f64" + ("_dpp" if dpp else "")
        if dpp or op.startswith(("v_mov_b32_dpp", "v_permlane")):
            return "valu_dpp_move"
        return "valu_other:" + re.sub(r"_e(32|64)$", "", op)
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio", "s_code_end")):
            return "s_wait/nop"
        if op.startswith(("s_cbranch", "s_branch")):
            return "s_branch"
        if op.startswith("s_load") or op.startswith("s_buffer_load"):
            return "smem"
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem:" + op.split("_")[0] + ("_load" if "load" in op else "_store" if "store" in op else "")
    if op.startswith("ds_"):
        return "lds"
    return "other:" + op


def main():
    obj, rx = sys.argv[1], re.compile(sys.argv[2])
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    res = {}
    for name, body in kernels(disassemble(obj)):
        if not rx.search(name):
            continue
        ins = [b.split("//")[0].strip() for b in body]
        ins = [i for i in ins if i and not i.endswith(":") and not i.startswith("s_code_end")]
        cnt = collections.Counter(classify(i) for i in ins)
        valu = sum(v for k, v in cnt.items() if k.startswith(("valu", "lane")))
        first_branch = next((k for k, i in enumerate(ins) if i.startswith(("s_cbranch", "s_branch"))), len(ins))
        hoisted = sum(1 for i in ins[:first_branch] if i.startswith(("global_load", "flat_load", "buffer_load")))  # vector loads issued before any branch: one memory round trip
        res[name] = {"resources": META.get(name, {}), "instructions": len(ins), "valu_total": valu, "valu_f64": cnt["valu_f64"] + cnt["valu_f64_dpp"], "classes": dict(cnt.most_common()),
                     "loads_before_first_branch": hoisted}
        if dump:
            with open(dump, "w") as f:
                f.write(name + "\n" + "\n".join(body) + "\n")
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1))
        return
    for name, r in res.items():
        print(name)
        print("  resources", r["resources"])
        print("  instructions %d, VALU %d of which FP64 %d (non-FP64 VALU %d)" % (r["instructions"], r["valu_total"], r["valu_f64"], r["valu_total"] - r["valu_f64"]))
        for k, v in r["classes"].items():
            print("   %5d  %s" % (v, k))


if __name__ == "__main__":
    main()

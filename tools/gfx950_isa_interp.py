#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — an interpreter for the gfx950 (CDNA4) machine code hipcc emits for this repository's kernels, on the host.

Why: tests/cpp/hip_cpu_emu.hpp runs the kernel SOURCE on the host; it says nothing about what the device compiler made of it.  Rounds without GPU access still
cross-compile (hipcc needs no device), so the code object exists: this module loads it (llvm-objdump / llvm-readelf of the ROCm toolchain), and executes a kernel
wavefront by wavefront — 64 lanes as numpy vectors under the EXEC mask, SGPRs / VCC / SCC, DPP lane moves, LDS, s_barrier between the wavefronts of a workgroup,
global memory as host arrays — one IEEE-754 operation per machine instruction (FMA through libm's fma).  What it gives:
  * the RESULT of the compiled code (compared bit for bit with the oracle by tests/test_isa_execution.py), and
  * DYNAMIC instruction counts per wavefront by class (VALU / FP64 / SALU / VMEM / LDS) — the figure SQ_INSTS_VALU / SQ_WAVES measures on hardware.
What it is not: a timing model, a memory model, or complete — it implements the instructions the interpreted kernels contain (anything else raises
NotImplementedError naming the instruction) and the subset of their semantics IEEE arithmetic on finite operands exercises; v_rcp_f64 / v_rsq_f64 return the
correctly rounded value where hardware returns a 1-ulp approximation (the division and square-root expansions that consume them correct either to the same
result).  Nothing in the library or the package imports it.
"""
import collections
import ctypes
import math
import os
import re
import shutil
import struct
import subprocess
import tempfile

import numpy as np

LLVM = "/opt/rocm/lib/llvm/bin"
U32, U64, I32, I64, F64 = np.uint32, np.uint64, np.int32, np.int64, np.float64
LANES = 64
ALL = (1 << 64) - 1
_LANE64 = np.arange(LANES, dtype=np.uint64)

_libm = ctypes.CDLL("libm.so.6")
_libm.fma.restype = ctypes.c_double
_libm.fma.argtypes = [ctypes.c_double] * 3


def _fma(a, b, c):
    out = np.empty(LANES, dtype=F64)
    f = _libm.fma
    for i in range(LANES):
        out[i] = f(a[i], b[i], c[i])
    return out


try:  # a vectorised fma (64 lanes per call) when a C compiler is at hand; the loop above otherwise
    _d = tempfile.mkdtemp(prefix="isa_fma_")
    with open(os.path.join(_d, "f.c"), "w") as _f:
        _f.write("#include <math.h>\nvoid vfma(const double*a,const double*b,const double*c,double*o){for(int i=0;i<64;++i)o[i]=fma(a[i],b[i],c[i]);}\n")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(_d, "f.so"), os.path.join(_d, "f.c"), "-lm"], stderr=subprocess.DEVNULL)
    _vf = ctypes.CDLL(os.path.join(_d, "f.so")).vfma
    _vf.argtypes = [ctypes.c_void_p] * 4

    def _fma(a, b, c):  # noqa: F811
        a, b, c = (np.ascontiguousarray(x, dtype=F64) for x in (a, b, c))
        out = np.empty(LANES, dtype=F64)
        _vf(a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data)
        return out
except Exception:  # noqa: BLE001
    pass


# ---------------------------------------------------------------------------------------------------------------------------------------
class CodeObject:
    """The gfx950 code object of a host object file / shared library built by hipcc: text (disassembled), loadable segments, kernel descriptors."""

    def __init__(self, path=None, elf_bytes=None):
        """path: a host object file / shared library of hipcc (its one device code object is extracted); elf_bytes: a gfx950 code object itself (hiprtc's output)"""
        self.tmp = tmp = tempfile.mkdtemp(prefix="isa_co_")
        if elf_bytes is not None:
            self.co_path = os.path.join(tmp, "rtc.co")
            with open(self.co_path, "wb") as f:
                f.write(elf_bytes)
        else:
            o = os.path.join(tmp, "t.o")
            shutil.copy(path, o)
            subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", o], cwd=tmp, stdout=subprocess.DEVNULL)
            os.remove(o)
            co = [f for f in os.listdir(tmp) if "amdgcn" in f]
            assert len(co) == 1, "expected one device code object in " + path
            self.co_path = os.path.join(tmp, co[0])
        self.raw = open(self.co_path, "rb").read()
        self._meta = None
        syms = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", self.co_path], text=True)
        self.symbols = {}
        for ln in syms.splitlines():
            f = ln.split()
            if len(f) >= 8 and f[0].rstrip(":").isdigit() and f[3] in ("FUNC", "OBJECT"):
                self.symbols[f[7]] = (int(f[1], 16), int(f[2]))
        # PT_LOAD segments -> one flat image at the code object's own virtual addresses
        e_phoff, = struct.unpack_from("<Q", self.raw, 32)
        e_phentsize, e_phnum = struct.unpack_from("<HH", self.raw, 54)
        top = 0
        segs = []
        for k in range(e_phnum):
            p_type, _fl, p_off, p_va, _pa, p_fsz, p_msz, _al = struct.unpack_from("<IIQQQQQQ", self.raw, e_phoff + k * e_phentsize)
            if p_type == 1:
                segs.append((p_off, p_va, p_fsz, p_msz))
                top = max(top, p_va + p_msz)
        self.image = np.zeros(top + 64, dtype=np.uint8)
        for p_off, p_va, p_fsz, _m in segs:
            self.image[p_va:p_va + p_fsz] = np.frombuffer(self.raw[p_off:p_off + p_fsz], dtype=np.uint8)
        self._kernels = {}

    def __del__(self):
        shutil.rmtree(getattr(self, "tmp", ""), ignore_errors=True)

    def kernel_args(self, name):
        """[(offset, size, value_kind)] of a kernel's arguments, from the code object's metadata note"""
        if self._meta is None:
            import yaml
            txt = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", self.co_path], text=True)
            a = txt.index("amdhsa.kernels:")
            b = txt.index("\n...", a) if "\n..." in txt[a:] else len(txt)
            doc = yaml.safe_load(txt[a:b])
            self._meta = {k[".name"]: [(x[".offset"], x[".size"], x[".value_kind"]) for x in k.get(".args", [])] for k in doc["amdhsa.kernels"]}
        return self._meta[name]

    def disassemble(self, name):
        """the instructions of one kernel (llvm-objdump restricted to the symbol: a library translation unit holds hundreds of kernels)"""
        return subprocess.check_output([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + name, self.co_path], text=True)

    def kernel(self, pattern):
        names = [n for n in self.symbols if re.search(pattern, n) and not n.endswith(".kd") and n + ".kd" in self.symbols]
        assert len(names) == 1, (pattern, names)
        name = names[0]
        if name not in self._kernels:
            self._kernels[name] = Kernel(self, name)
        return self._kernels[name]


_MOD_RE = [
    ("quad_perm", re.compile(r"quad_perm:\[(\d),(\d),(\d),(\d)\]")), ("row_ror", re.compile(r"row_ror:(\d+)")), ("row_shr", re.compile(r"row_shr:(\d+)")),
    ("row_shl", re.compile(r"row_shl:(\d+)")), ("row_bcast", re.compile(r"row_bcast:(\d+)")), ("row_mask", re.compile(r"row_mask:(0x[0-9a-f]+)")),
    ("bank_mask", re.compile(r"bank_mask:(0x[0-9a-f]+)")), ("bound_ctrl", re.compile(r"bound_ctrl:(\d)")), ("offset", re.compile(r"offset:(-?\d+)")),
    ("bitop3", re.compile(r"bitop3:(0x[0-9a-f]+|\d+)")), ("offset0", re.compile(r"offset0:(\d+)")), ("offset1", re.compile(r"offset1:(\d+)")), ("mul", re.compile(r"\bmul:(\d)")), ("div", re.compile(r"\bdiv:(\d)")),
]
# every DPP control the decoder knows: an instruction carrying one of them goes through _dpp_fetch (round 6, second session: the four wave-wide controls were parsed as
# flags but missing here, so `v_mov_b32_dpp ... wave_shl:1` — the cross-row step of __syncthreads_or's wave reduction — ran as a plain move and the reduction lost rows 1 and 3)
_DPP_CTRL = ("quad_perm", "row_ror", "row_shr", "row_shl", "row_bcast", "row_mirror", "row_half_mirror", "wave_shr:1", "wave_shl:1", "wave_ror:1", "wave_rol:1")
_FLAGS = ("clamp", "nt", "sc0", "sc1", "glc", "slc", "row_mirror", "row_half_mirror", "wave_shr:1", "wave_shl:1", "wave_ror:1", "wave_rol:1", "gds")


class Ins:
    __slots__ = ("addr", "op", "ops", "mods", "target", "cls", "text", "fn")

    def __repr__(self):
        return "%x: %s" % (self.addr, self.text)


class Kernel:
    def __init__(self, co, name):
        self.co, self.name = co, name
        self.start, self.size = co.symbols[name]
        kd, _ = co.symbols[name + ".kd"]
        d = bytes(co.image[kd:kd + 64])
        self.lds_bytes, self.scratch_bytes, self.kernarg_size = struct.unpack_from("<III", d, 0)
        rsrc1, rsrc2 = struct.unpack_from("<II", d, 48)
        props, = struct.unpack_from("<H", d, 56)
        self.user_sgprs = (rsrc2 >> 1) & 0x1f
        self.enable_wg_id = [(rsrc2 >> 7) & 1, (rsrc2 >> 8) & 1, (rsrc2 >> 9) & 1]
        self.props = props
        self.preload_len, = struct.unpack_from("<H", d, 58)
        self.preload_len &= 0x7f
        # user SGPR layout in the order the hardware fills them
        self.sgpr_layout, k = {}, 0
        for bit, nm, n in ((0, "private_segment_buffer", 4), (1, "dispatch_ptr", 2), (2, "queue_ptr", 2), (3, "kernarg_segment_ptr", 2), (4, "dispatch_id", 2),
                           (5, "flat_scratch_init", 2), (6, "private_segment_size", 1)):
            if props >> bit & 1:
                self.sgpr_layout[nm] = k
                k += n
        self.preload_first = k
        self.ins, self.index = [], {}
        body, on = [], False
        for ln in co.disassemble(name).splitlines():
            m = re.match(r"^([0-9a-f]+) <(.*)>:$", ln)
            if m:
                on = m.group(2) == name
                continue
            if on and ln.strip() and "//" in ln:
                body.append(ln)
        for ln in body:
            code, tail = ln.split("//", 1)
            code = code.strip()
            addr = int(tail.strip().split(":")[0], 16)
            if not code or code.startswith("s_code_end"):
                continue
            i = Ins()
            i.addr, i.text = addr, code
            tgt = re.search(r"<[^>]*\+0x([0-9a-f]+)>\s*$", tail)
            i.target = self.start + int(tgt.group(1), 16) if tgt else (self.start if re.search(r"<[^+>]*>\s*$", tail) and code.startswith(("s_cbranch", "s_branch")) else None)
            parts = code.split(None, 1)
            i.op = parts[0]
            rest = parts[1] if len(parts) > 1 else ""
            i.mods = {}
            for key, rx in _MOD_RE:
                m = rx.search(rest)
                if m:
                    i.mods[key] = tuple(int(x, 0) for x in m.groups()) if key == "quad_perm" else int(m.group(1), 0)
                    rest = rest[:m.start()] + rest[m.end():]
            for key in ("dst_sel", "dst_unused", "src0_sel", "src1_sel"):
                m = re.search(key + r":(\w+)", rest)
                if m:
                    i.mods[key] = m.group(1)
                    rest = rest[:m.start()] + rest[m.end():]
            for fl in _FLAGS:
                if re.search(r"(^|\s)" + re.escape(fl) + r"(\s|$)", rest):
                    i.mods[fl] = 1
                    rest = re.sub(r"(^|\s)" + re.escape(fl) + r"(?=\s|$)", " ", rest)
            i.ops = [t.strip() for t in rest.split(",") if t.strip()]
            i.cls = classify(i.op, i.mods)
            i.fn = None
            self.index[addr] = len(self.ins)
            self.ins.append(i)


def classify(op, mods):
    dpp = any(k in mods for k in _DPP_CTRL)
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "valu_lane"
    if op.startswith("v_"):
        if "_f64" in op:
            return "valu_f64"
        return "valu_dpp" if dpp else "valu_other"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_sleep", "s_setprio")):
            return "s_wait"
        if op.startswith(("s_cbranch", "s_branch")):
            return "s_branch"
        if op.startswith(("s_load", "s_buffer_load")):
            return "smem"
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    return "other"


# ---------------------------------------------------------------------------------------------------------------------------------------
class Memory:
    """Global memory: the code object's image at its own addresses (constant tables, kernel descriptors) + host arrays at fake device addresses."""
    BASE = 1 << 40

    def __init__(self, co):
        self.regions = [(0, co.image)]
        self.next = self.BASE

    def alloc(self, arr):
        """Maps `arr` (any contiguous numpy array; written in place by stores) and returns its device address."""
        assert arr.flags["C_CONTIGUOUS"]
        b = arr.view(np.uint8).reshape(-1)
        addr = self.next
        self.regions.append((addr, b))
        self.next += (b.size + 4095 + 4096) & ~4095  # a guard page between arrays: an out-of-bounds access of up to 4 KiB is caught
        return addr

    def _find(self, addr, n):
        for base, b in self.regions:
            if base <= addr and addr + n <= base + b.size:
                return b, addr - base
        raise MemoryError("access of %d bytes at 0x%x is outside every mapped array" % (n, addr))

    def read(self, addr, n):
        b, o = self._find(addr, n)
        return b[o:o + n]

    def write(self, addr, data):
        b, o = self._find(addr, len(data))
        b[o:o + len(data)] = data


F64_INLINE = {"0.5": 0.5, "-0.5": -0.5, "1.0": 1.0, "-1.0": -1.0, "2.0": 2.0, "-2.0": -2.0, "4.0": 4.0, "-4.0": -4.0, "0.15915494": 0.15915494309189532}


class Wave:
    def __init__(self, kern, mem, lds, wg_id, wave_in_wg, block, kernarg_addr, nlanes):
        self.k, self.mem, self.lds = kern, mem, lds
        self.s = np.zeros(128, dtype=U32)
        self.v = np.zeros((512, LANES), dtype=U32)
        self.a = np.zeros((256, LANES), dtype=U32)  # accumulation VGPRs (used as spill space by register-heavy kernels)
        self.vcc, self.scc, self.m0 = 0, 0, 0
        self.gpr_idx = None   # VGPR indexing mode (s_set_gpr_idx_on): (index, set of SRC0 / SRC1 / SRC2 / DST) or None
        self.exec = (1 << nlanes) - 1 if nlanes < 64 else ALL
        self.pc = 0
        self.counts = collections.Counter()
        self.done = False
        self.scratch = None
        # initial register state (ABI of the code object version 5, gfx9): user SGPRs, then workgroup ids; v0 = packed work-item id
        lay = kern.sgpr_layout
        if "kernarg_segment_ptr" in lay:
            self.wr_s64(lay["kernarg_segment_ptr"], kernarg_addr)
        if kern.preload_len:
            ka = mem.read(kernarg_addr, 4 * kern.preload_len).view(U32)
            self.s[kern.preload_first:kern.preload_first + kern.preload_len] = ka
        k = kern.user_sgprs
        for d in range(3):
            if kern.enable_wg_id[d]:
                self.s[k] = wg_id[d]
                k += 1
        tid = wave_in_wg * LANES + np.arange(LANES, dtype=U32)
        self.v[0] = (tid % block[0]) | ((tid // block[0]) % block[1]) << 10 | (tid // (block[0] * block[1])) << 20
        self.lane = np.arange(LANES)

    # ---- registers ------------------------------------------------------------------------------------------------
    def rd_s64(self, i):
        return int(self.s[i]) | int(self.s[i + 1]) << 32

    def wr_s64(self, i, v):
        self.s[i] = v & 0xffffffff
        self.s[i + 1] = (v >> 32) & 0xffffffff

    def mask(self):
        return Wave.bits_to_mask(self.exec)

    @staticmethod
    def bits_to_mask(bits):
        if not isinstance(bits, int):
            return bits
        return ((U64(bits & ALL) >> _LANE64) & U64(1)).astype(bool)

    @staticmethod
    def mask_to_bits(m):
        out = 0
        for i in np.nonzero(m)[0]:
            out |= 1 << int(i)
        return out

    def _reg(self, tok):
        m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
        if m:
            return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
        m = re.match(r"^([vsa])(\d+)$", tok)
        if m:
            return m.group(1), int(m.group(2)), 1
        return None

    def src32(self, tok):
        """-> np.uint32[64] (vector) or python int (uniform)"""
        neg = ab = False
        if tok.startswith("-") and not re.match(r"^-[\d.]", tok):
            neg, tok = True, tok[1:]
        if tok.startswith("|") and tok.endswith("|"):
            ab, tok = True, tok[1:-1]
        r = self._reg(tok)
        if r:
            val = self.v[r[1]] if r[0] == "v" else int(self.s[r[1]])
        elif tok in ("vcc_lo", "vcc"):
            val = self.vcc & 0xffffffff
        elif tok == "vcc_hi":
            val = self.vcc >> 32
        elif tok in ("exec_lo", "exec"):
            val = self.exec & 0xffffffff
        elif tok == "exec_hi":
            val = self.exec >> 32
        elif tok == "m0":
            val = self.m0
        elif tok in ("scc", "src_scc"):
            val = self.scc
        elif tok in ("off", "null"):
            val = 0
        elif re.match(r"^-?0x[0-9a-f]+$", tok) or re.match(r"^-?\d+$", tok):
            val = int(tok, 0) & 0xffffffff
        elif re.match(r"^-?\d+\.\d+(e[-+]?\d+)?$", tok):
            val = int(np.float32(float(tok)).view(U32))
        else:
            raise NotImplementedError("operand " + tok)
        if neg or ab:
            raise NotImplementedError("float modifiers on a 32-bit operand: " + tok)
        return val

    def src64(self, tok, fp):
        """-> np.uint64[64]; fp: the operand is an f64 (inline constants and literals are interpreted accordingly)"""
        neg = ab = False
        if tok.startswith("-") and not re.match(r"^-[\d.]", tok):
            neg, tok = True, tok[1:]
        if tok.startswith("|") and tok.endswith("|"):
            ab, tok = True, tok[1:-1]
        r = self._reg(tok)
        if r:
            if r[0] == "v":
                val = self.v[r[1]].astype(U64) | (self.v[r[1] + 1].astype(U64) << U64(32))
            else:
                val = np.full(LANES, self.rd_s64(r[1]), dtype=U64)
        elif tok == "vcc":
            val = np.full(LANES, self.vcc, dtype=U64)
        elif tok == "exec":
            val = np.full(LANES, self.exec, dtype=U64)
        elif tok in F64_INLINE:  # (also the source of v_mov_b64: a float inline constant of a 64-bit operand is the f64)
            val = np.full(LANES, F64_INLINE[tok], dtype=F64).view(U64)
        elif re.match(r"^-?\d+$", tok):
            val = np.full(LANES, int(tok) & ALL, dtype=U64)  # integer inline constant, sign-extended (as an f64 operand: its bit pattern)
        elif re.match(r"^0x[0-9a-f]+$", tok):
            lit = int(tok, 16)
            val = np.full(LANES, (lit << 32) if fp else lit, dtype=U64)  # a 32-bit literal is the HIGH half of an f64 operand
        elif fp and re.match(r"^-?\d+\.\d+(e[-+]?\d+)?$", tok):
            raise NotImplementedError("f64 literal " + tok)
        else:
            raise NotImplementedError("operand " + tok)
        if ab:
            val = val & U64(0x7fffffffffffffff)
        if neg:
            val = val ^ U64(0x8000000000000000)
        return val

    def f64(self, tok):
        return self.src64(tok, True).view(F64)

    def wr_v32(self, tok, val, m=None):
        r = self._reg(tok)
        assert r and r[0] == "v", tok
        m = self.mask() if m is None else m
        self.v[r[1]][m] = (np.asarray(val, dtype=U32) if not isinstance(val, int) else np.full(LANES, val, dtype=U32))[m]

    def wr_v64(self, tok, val, m=None):
        r = self._reg(tok)
        assert r and r[0] == "v" and r[2] == 2, tok
        m = self.mask() if m is None else m
        val = np.asarray(val).view(U64)
        self.v[r[1]][m] = (val & U64(0xffffffff)).astype(U32)[m]
        self.v[r[1] + 1][m] = (val >> U64(32)).astype(U32)[m]

    def wr_s(self, tok, val):
        """scalar destination of 32 or 64 bits by the token's width"""
        if tok == "vcc":
            self.vcc = val & ALL
        elif tok == "exec":
            self.exec = val & ALL
        elif tok == "vcc_lo":
            self.vcc = (self.vcc & ~0xffffffff) | (val & 0xffffffff)
        elif tok == "vcc_hi":
            self.vcc = (self.vcc & 0xffffffff) | ((val & 0xffffffff) << 32)
        elif tok == "exec_lo":
            self.exec = (self.exec & ~0xffffffff) | (val & 0xffffffff)
        elif tok == "exec_hi":
            self.exec = (self.exec & 0xffffffff) | ((val & 0xffffffff) << 32)
        elif tok == "m0":
            self.m0 = val & 0xffffffff
        elif tok == "null":
            pass
        else:
            r = self._reg(tok)
            assert r and r[0] == "s", tok
            if r[2] == 1:
                self.s[r[1]] = val & 0xffffffff
            else:
                for j in range(r[2]):
                    self.s[r[1] + j] = (val >> (32 * j)) & 0xffffffff

    def rd_s(self, tok, bits=32):
        """scalar source: 64-bit for register pairs / vcc / exec, else 32-bit; python int.  bits = 64: an integer inline constant is sign-extended to 64 bits"""
        if bits == 64 and re.match(r"^-\d+$", tok):
            return int(tok) & ALL
        if tok == "vcc":
            return self.vcc
        if tok == "exec":
            return self.exec
        r = self._reg(tok)
        if r and r[0] == "s" and r[2] == 2:
            return self.rd_s64(r[1])
        v = self.src32(tok)
        assert isinstance(v, int), tok
        return v

    def sx(self, v, bits):
        return v - (1 << bits) if v >> (bits - 1) & 1 else v


def _vec(x):
    return x if isinstance(x, np.ndarray) else np.full(LANES, x, dtype=U32)


def _cmp_f64(kind, a, b):
    un = np.isnan(a) | np.isnan(b)
    with np.errstate(invalid="ignore"):
        lt, gt, eq = a < b, a > b, a == b
    table = {"lt": lt, "gt": gt, "eq": eq, "le": lt | eq, "ge": gt | eq, "lg": lt | gt, "neq": ~eq, "nlt": ~lt, "ngt": ~gt, "nle": ~(lt | eq), "nge": ~(gt | eq),
             "nlg": ~(lt | gt), "u": un, "o": ~un, "f": np.zeros(LANES, bool), "tru": np.ones(LANES, bool)}
    return table[kind]


def _cmp_int(kind, a, b):
    return {"lt": a < b, "gt": a > b, "eq": a == b, "le": a <= b, "ge": a >= b, "ne": a != b, "lg": a != b}[kind]


def _class_f64(x, maskbits):
    """maskbits: python int (uniform) or one mask per lane"""
    if not isinstance(maskbits, int):
        out = np.zeros(LANES, bool)
        for mb in np.unique(maskbits):
            out |= (maskbits == mb) & _class_f64(x, int(mb))
        return out
    u = x.view(U64)
    sign = (u >> U64(63)).astype(bool)
    expo = ((u >> U64(52)) & U64(0x7ff)).astype(np.int64)
    mant = u & U64(0xfffffffffffff)
    nan = (expo == 0x7ff) & (mant != 0)
    snan = nan & ((mant >> U64(51)) == 0)
    qnan = nan & ~snan
    inf = (expo == 0x7ff) & (mant == 0)
    zero = (expo == 0) & (mant == 0)
    den = (expo == 0) & (mant != 0)
    norm = (expo > 0) & (expo < 0x7ff)
    cls = [snan, qnan, inf & sign, norm & sign, den & sign, zero & sign, zero & ~sign, den & ~sign, norm & ~sign, inf & ~sign]
    out = np.zeros(LANES, bool)
    for b, c in enumerate(cls):
        if maskbits >> b & 1:
            out |= c
    return out


def _dpp_source(w, mods):
    """-> (source lane index per lane, valid per lane) of a DPP operand"""
    lane = w.lane
    row = lane & ~15
    valid = np.ones(LANES, bool)
    if "quad_perm" in mods:
        q = np.array(mods["quad_perm"])
        src = (lane & ~3) | q[lane & 3]
    elif "row_ror" in mods:
        src = row | ((lane - mods["row_ror"]) & 15)
    elif "row_shr" in mods:
        src = lane - mods["row_shr"]
        valid = (src >= row)
    elif "row_shl" in mods:
        src = lane + mods["row_shl"]
        valid = (src < row + 16)
    elif "row_bcast" in mods:
        if mods["row_bcast"] == 15:   # lane 15 of every row to all lanes of the NEXT row
            src = row - 1
            valid = lane >= 16
        else:                          # row_bcast:31 — lane 31 to all lanes of rows 2 and 3
            src = np.full(LANES, 31)
            valid = lane >= 32
    elif "row_mirror" in mods:
        src = row | (15 - (lane & 15))
    elif "row_half_mirror" in mods:
        src = (lane & ~7) | (7 - (lane & 7))
    elif "wave_shl:1" in mods:   # DPP_WF_SL1: lane i reads lane i + 1 of the WAVE (lane 63 has no source)
        src = lane + 1
        valid = src < LANES
    elif "wave_shr:1" in mods:   # DPP_WF_SR1: lane i reads lane i - 1 (lane 0 has no source)
        src = lane - 1
        valid = src >= 0
    elif "wave_rol:1" in mods:   # DPP_WF_RL1: rotate, lane i reads lane (i + 1) mod 64
        src = (lane + 1) % LANES
    elif "wave_ror:1" in mods:   # DPP_WF_RR1
        src = (lane - 1) % LANES
    else:
        raise NotImplementedError("DPP control " + str(mods))
    return np.where(valid, src, lane), valid


def _dpp_fetch(w, val, mods):
    """value of a DPP source operand per lane, and the lanes whose write is suppressed"""
    src, valid = _dpp_source(w, mods)
    em = w.mask()
    ok = valid & em[src]
    rm, bm = mods.get("row_mask", 0xf), mods.get("bank_mask", 0xf)
    enabled = (((rm >> (w.lane >> 4)) & 1) & ((bm >> ((w.lane >> 2) & 3)) & 1)).astype(bool)
    fetched = _vec(val)[src]
    if mods.get("bound_ctrl", 0):
        return np.where(ok, fetched, U32(0)), ~enabled
    return fetched, ~(enabled & ok)


class Machine:
    """Runs one kernel launch: workgroups one after the other, the wavefronts of a workgroup interleaved at s_barrier."""

    def __init__(self, co):
        self.co = co

    def launch(self, kern, grid, block, kernarg, mem, max_instructions=20_000_000):
        block = tuple(block) + (1,) * (3 - len(block))
        grid = tuple(grid) + (1,) * (3 - len(grid))
        kb = bytearray(kernarg) + bytearray(max(0, kern.kernarg_size - len(kernarg)) + 64)
        vals = {"hidden_block_count_x": grid[0], "hidden_block_count_y": grid[1], "hidden_block_count_z": grid[2], "hidden_group_size_x": block[0],
                "hidden_group_size_y": block[1], "hidden_group_size_z": block[2], "hidden_remainder_x": 0, "hidden_remainder_y": 0, "hidden_remainder_z": 0,
                "hidden_global_offset_x": 0, "hidden_global_offset_y": 0, "hidden_global_offset_z": 0,
                "hidden_grid_dims": 1 + (grid[1] * block[1] > 1) + (grid[2] * block[2] > 1)}
        for off, size, kind in kern.co.kernel_args(kern.name):  # the implicit arguments of code object v5, where the metadata puts them
            if kind in vals:
                kb[off:off + size] = int(vals[kind]).to_bytes(size, "little")
        ka = np.frombuffer(kb, dtype=np.uint8).copy()
        ka_addr = mem.alloc(ka)
        nthreads = block[0] * block[1] * block[2]
        stats = []
        for gz in range(grid[2]):
            for gy in range(grid[1]):
                for gx in range(grid[0]):
                    lds = np.zeros(max(kern.lds_bytes, 4) + 64, dtype=np.uint8)
                    waves = []
                    for wv in range((nthreads + LANES - 1) // LANES):
                        waves.append(Wave(kern, mem, lds, (gx, gy, gz), wv, block, ka_addr, min(LANES, nthreads - wv * LANES)))
                    live = list(waves)
                    while live:
                        for w in list(live):
                            self.run_to_barrier(w, max_instructions)
                            if w.done:
                                live.remove(w)
                    stats += [w.counts for w in waves]
        mem.regions = [r for r in mem.regions if r[0] != ka_addr]
        return stats

    def run_to_barrier(self, w, budget):
        try:
            return self._run_to_barrier(w, budget)
        except (MemoryError, NotImplementedError, AssertionError, IndexError, ValueError, KeyError) as e:   # say where: the instruction and its address
            i = w.k.ins[max(w.pc - 1, 0)]
            raise type(e)("%s  [at 0x%x: %s]" % (e, i.addr, i.text)) from e

    def _run_to_barrier(self, w, budget):
        k = w.k
        ins = k.ins
        while True:
            i = ins[w.pc]
            w.counts[i.cls] += 1
            w.counts["total"] += 1
            if w.counts["total"] > budget:
                raise RuntimeError("instruction budget exceeded at " + repr(i))
            fn = i.fn
            if fn is None:
                fn = i.fn = _resolve(i)
            w.pc += 1
            if w.gpr_idx is not None and i.cls.startswith("valu") and fn is not _v_mov_b32:
                raise NotImplementedError("VGPR indexing mode is modelled for v_mov_b32 only: " + i.text)
            r = fn(w, i)
            if r == "end":
                w.done = True
                return
            if r == "barrier":
                return


# ---- instruction semantics ---------------------------------------------------------------------------------------------------------------
def _resolve(i):
    op = i.op
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    fn = _OPS.get(base)
    if fn is None:
        m = re.match(r"^v_cmpx?_(\w+?)_(f64|f32|u32|i32|u64|i64)$", base)
        if m:
            return _v_cmp
        raise NotImplementedError("gfx950 instruction not implemented by the interpreter: " + i.text)
    return fn


def _jump(w, i):
    w.pc = w.k.index[i.target]


def _s_branch(w, i):
    _jump(w, i)


def _s_setpc(w, i):   # a computed jump inside the kernel (s_getpc + offset: switch tables, outlined blocks)
    target = w.rd_s(i.ops[0])
    if target not in w.k.index:
        raise NotImplementedError("s_setpc_b64 to 0x%x, which is not an instruction of this kernel" % target)
    w.pc = w.k.index[target]


def _s_cbranch(cond):
    def f(w, i):
        if cond(w):
            _jump(w, i)
    return f


def _v_cmp(w, i):
    base = re.sub(r"_(e32|e64)$", "", i.op)
    m = re.match(r"^v_cmp(x?)_(\w+?)_(f64|u32|i32|u64|i64)$", base)
    x, kind, ty = m.groups()
    ops = i.ops
    dst = "vcc" if len(ops) == 2 else ops[0]
    a, b = ops[-2], ops[-1]
    if kind == "class":
        res = _class_f64(w.f64(a), w.src32(b))  # (the class mask may sit in a VGPR written under another EXEC: per lane)
    elif ty == "f64":
        res = _cmp_f64(kind, w.f64(a), w.f64(b))
    elif ty in ("u32", "i32"):
        A, B = _vec(w.src32(a)), _vec(w.src32(b))
        if ty == "i32":
            A, B = A.view(I32), B.view(I32)
        res = _cmp_int(kind, A, B)
    else:
        A, B = w.src64(a, False), w.src64(b, False)
        if ty == "i64":
            A, B = A.view(I64), B.view(I64)
        res = _cmp_int(kind, A, B)
    bits = Wave.mask_to_bits(res & w.mask())
    w.wr_s(dst, bits)
    if x:
        w.exec = bits


def _vop_f64(fn, nsrc):
    def f(w, i):
        src = [w.f64(t) for t in i.ops[1:1 + nsrc]]
        with np.errstate(all="ignore"):
            r = fn(*src)
        if i.mods.get("mul") or i.mods.get("div") or i.mods.get("clamp"):
            raise NotImplementedError("output modifier: " + i.text)
        w.wr_v64(i.ops[0], np.asarray(r, dtype=F64))
    return f


def _v_fmac_f64(w, i):
    a, b = w.f64(i.ops[1]), w.f64(i.ops[2])
    w.wr_v64(i.ops[0], _fma(a, b, w.f64(i.ops[0])))


def _v_ldexp_f64(w, i):
    x = w.f64(i.ops[1])
    e = _vec(w.src32(i.ops[2])).view(I32).astype(np.int64)
    with np.errstate(all="ignore"):
        w.wr_v64(i.ops[0], np.ldexp(x, np.clip(e, -5000, 5000)))


def _v_rcp_f64(w, i):
    with np.errstate(all="ignore"):
        w.wr_v64(i.ops[0], 1.0 / w.f64(i.ops[1]))


def _v_rsq_f64(w, i):
    x = w.f64(i.ops[1])
    out = np.empty(LANES, dtype=F64)
    for j in range(LANES):  # correctly rounded 1 / sqrt(x) through exact rational arithmetic would be overkill: two roundings, <= 1 ulp like the hardware's
        xv = float(x[j])
        out[j] = (1.0 / math.sqrt(xv)) if xv > 0.0 and math.isfinite(xv) else (math.inf if xv == 0.0 else (0.0 if xv == math.inf else math.nan))
    w.wr_v64(i.ops[0], out)


def _v_sqrt_f64(w, i):
    with np.errstate(all="ignore"):
        w.wr_v64(i.ops[0], np.sqrt(w.f64(i.ops[1])))


def _expo(x):
    return ((x.view(U64) >> U64(52)) & U64(0x7ff)).astype(np.int64)


def _v_div_scale_f64(w, i):
    # vdst, sdst, S0, S1 (denominator), S2 (numerator): V_DIV_SCALE_F64 of the ISA manual.  The cases other than "nothing to scale" need operands within
    # 2^-970 of the ends of the exponent range; they are implemented per the manual but a hit is reported: nothing in this repository's tests reaches them.
    s0, s1, s2 = w.f64(i.ops[2]), w.f64(i.ops[3]), w.f64(i.ops[4])
    out = s0.copy()
    vcc = np.zeros(LANES, bool)
    e1, e2 = _expo(s1), _expo(s2)
    m = w.mask()
    with np.errstate(all="ignore"):
        zero = (s2 == 0.0) | (s1 == 0.0)
        big = ~zero & (e2 - e1 >= 768)
        den1 = ~zero & ~big & (e1 == 0)
        rcp_den = ~zero & ~big & ~den1 & (_expo(1.0 / s1) == 0)
        q_den = ~zero & ~big & ~den1 & (_expo(s2 / s1) == 0)
        tiny = ~zero & ~big & ~den1 & ~rcp_den & ~q_den & (e2 <= 53)
    if (m & (zero | big | den1 | rcp_den | q_den | tiny)).any():
        out = np.where(zero, np.nan, out)
        same1 = s0.view(U64) == s1.view(U64)
        same2 = s0.view(U64) == s2.view(U64)
        vcc |= big | (rcp_den & q_den) | (~rcp_den & q_den)
        out = np.where(big & same1, np.ldexp(s0, 128), out)
        out = np.where(den1, np.ldexp(s0, 128), out)
        out = np.where(rcp_den & q_den & same1, np.ldexp(s0, 128), out)
        out = np.where(rcp_den & ~q_den, np.ldexp(s0, -128), out)
        out = np.where(~rcp_den & q_den & same2, np.ldexp(s0, 128), out)
        out = np.where(tiny, np.ldexp(s0, 128), out)
        w.counts["div_scale_special"] += 1
    w.wr_v64(i.ops[0], out)
    w.wr_s(i.ops[1], Wave.mask_to_bits(vcc & m))


def _v_div_fmas_f64(w, i):
    r = _fma(w.f64(i.ops[1]), w.f64(i.ops[2]), w.f64(i.ops[3]))
    sc = Wave.bits_to_mask(w.vcc)
    if sc.any():
        with np.errstate(all="ignore"):
            r = np.where(sc, np.ldexp(r, 64), r)
    w.wr_v64(i.ops[0], r)


def _v_div_fixup_f64(w, i):
    q, den, num = w.f64(i.ops[1]), w.f64(i.ops[2]), w.f64(i.ops[3])
    sign = ((den.view(U64) ^ num.view(U64)) >> U64(63)).astype(bool)
    with np.errstate(all="ignore"):
        out = np.where(sign, -np.abs(q), np.abs(q))
        nan = np.isnan(num) | np.isnan(den) | ((den == 0) & (num == 0)) | (np.isinf(den) & np.isinf(num))
        inf = ~nan & ((den == 0) | np.isinf(num))
        zero = ~nan & ~inf & (np.isinf(den) | (num == 0))
        out = np.where(inf, np.where(sign, -np.inf, np.inf), out)
        out = np.where(zero, np.where(sign, -0.0, 0.0), out)
        out = np.where(nan, np.nan, out)
        d = _expo(num) - _expo(den)
        rest = ~nan & ~inf & ~zero
        out = np.where(rest & (d < -1075), np.where(sign, -0.0, 0.0), out)
        out = np.where(rest & (d >= 1024), np.where(sign, -np.inf, np.inf), out)
    w.wr_v64(i.ops[0], out)


def _v_mov_b32(w, i):
    if w.gpr_idx is not None:     # s_set_gpr_idx_on: M0[7:0] is added to the VGPR number of the operands the mode names
        idx, modes = w.gpr_idx
        src, dst = i.ops[1], i.ops[0]
        if "SRC0" in modes and re.match(r"^v\d+$", src):
            src = "v%d" % (int(src[1:]) + idx)
        if "DST" in modes:
            dst = "v%d" % (int(dst[1:]) + idx)
        w.wr_v32(dst, w.src32(src))
        return
    val = w.src32(i.ops[1])
    if any(k in i.mods for k in _DPP_CTRL):
        fetched, off = _dpp_fetch(w, val, i.mods)
        w.wr_v32(i.ops[0], fetched, w.mask() & ~off)
    else:
        w.wr_v32(i.ops[0], val)


def _v_mov_b64(w, i):
    w.wr_v64(i.ops[0], w.src64(i.ops[1], False))


_SEL = {"BYTE_0": (0, 0xff), "BYTE_1": (8, 0xff), "BYTE_2": (16, 0xff), "BYTE_3": (24, 0xff), "WORD_0": (0, 0xffff), "WORD_1": (16, 0xffff), "DWORD": (0, 0xffffffff)}


def _sdwa_src(val, sel):
    sh, msk = _SEL[sel]
    return (_vec(val) >> U32(sh)) & U32(msk)


def _vop2_int(fn):
    def f(w, i):
        if "src0_sel" in i.mods:  # SDWA: sub-dword source selects (zero-extended), whole-dword destination
            if i.mods.get("dst_sel", "DWORD") != "DWORD":
                raise NotImplementedError("SDWA destination select: " + i.text)
            a = _sdwa_src(w.src32(i.ops[1]), i.mods["src0_sel"])
            b = _sdwa_src(w.src32(i.ops[2]), i.mods.get("src1_sel", "DWORD"))
            with np.errstate(over="ignore"):
                w.wr_v32(i.ops[0], fn(a, b).astype(U32))
            return
        a = w.src32(i.ops[1])
        m = None
        if any(k in i.mods for k in _DPP_CTRL):
            a, off = _dpp_fetch(w, a, i.mods)
            m = w.mask() & ~off
        b = _vec(w.src32(i.ops[2]))
        with np.errstate(over="ignore"):
            w.wr_v32(i.ops[0], fn(_vec(a), b).astype(U32), m)
    return f


def _v_add_co(sub=False, rev=False, carry_in=False):
    def f(w, i):
        # vdst, vcc(out), src0, src1 [, vcc(in)]
        a, b = _vec(w.src32(i.ops[2])).astype(np.int64), _vec(w.src32(i.ops[3])).astype(np.int64)
        if rev:
            a, b = b, a
        cin = Wave.bits_to_mask(w.rd_s(i.ops[4])).astype(np.int64) if carry_in else 0
        r = a - b - cin if sub else a + b + cin
        w.wr_v32(i.ops[0], (r & 0xffffffff).astype(U32))
        carry = (r < 0) if sub else (r > 0xffffffff)
        w.wr_s(i.ops[1], Wave.mask_to_bits(carry & w.mask()))
    return f


def _v_lshl_add_u64(w, i):
    a = w.src64(i.ops[1], False)
    sh = _vec(w.src32(i.ops[2])).astype(U64) & U64(63)
    c = w.src64(i.ops[3], False)
    with np.errstate(over="ignore"):
        w.wr_v64(i.ops[0], (a << sh) + c)


def _v_bfe_u32(w, i):
    a, off, wid = _vec(w.src32(i.ops[1])), _vec(w.src32(i.ops[2])) & U32(31), _vec(w.src32(i.ops[3])) & U32(31)
    msk = ((U64(1) << wid.astype(U64)) - U64(1)).astype(U32)
    w.wr_v32(i.ops[0], (a >> off) & msk)


def _v_mad_u64_u32(w, i):
    # vdst(64), sdst(carry), s0, s1, s2(64)
    a, b = _vec(w.src32(i.ops[2])).astype(object), _vec(w.src32(i.ops[3])).astype(object)
    c = w.src64(i.ops[4], False).astype(object)
    r = a * b + c
    w.wr_v64(i.ops[0], np.array([int(x) & ALL for x in r], dtype=U64))
    w.wr_s(i.ops[1], Wave.mask_to_bits(np.array([int(x) >> 64 != 0 for x in r]) & w.mask()))


def _v_mad_u32_u24(w, i):
    a, b, c = (_vec(w.src32(t)) for t in i.ops[1:4])
    with np.errstate(over="ignore"):
        w.wr_v32(i.ops[0], ((a & U32(0xffffff)).astype(U64) * (b & U32(0xffffff)).astype(U64) + c.astype(U64)).astype(U32))


def _v_cndmask_b32(w, i):
    a, b = _vec(w.src32(i.ops[1])), _vec(w.src32(i.ops[2]))
    sel = Wave.bits_to_mask(w.rd_s(i.ops[3]) if len(i.ops) > 3 else w.vcc)
    w.wr_v32(i.ops[0], np.where(sel, b, a))


def _v_cvt_f64_u32(w, i):
    w.wr_v64(i.ops[0], _vec(w.src32(i.ops[1])).astype(F64))


def _v_cvt_f64_i32(w, i):
    w.wr_v64(i.ops[0], _vec(w.src32(i.ops[1])).view(I32).astype(F64))


# f32 operations: the opt-in contracted build's controller estimates err^0.2 in f32 (v_log_f32 / v_exp_f32: the hardware's are ~1 ulp approximations of log2 / exp2 —
# numpy's float32 log2 / exp2 stand in, so kernels that use them are compared within a tolerance, never bit for bit) and refines it in f64.
F32 = np.float32


def _f32(w, tok):
    return _vec(w.src32(tok)).view(F32)


def _v_cvt_f32_f64(w, i):
    with np.errstate(all="ignore"):
        w.wr_v32(i.ops[0], w.f64(i.ops[1]).astype(F32).view(U32))


def _v_cvt_f64_f32(w, i):
    w.wr_v64(i.ops[0], _f32(w, i.ops[1]).astype(F64))


def _f32_unary(fn):
    def f(w, i):
        with np.errstate(all="ignore"):
            w.wr_v32(i.ops[0], fn(_f32(w, i.ops[1])).astype(F32).view(U32))
    return f


def _f32_binary(fn):
    def f(w, i):
        with np.errstate(all="ignore"):
            w.wr_v32(i.ops[0], fn(_f32(w, i.ops[1]), _f32(w, i.ops[2])).astype(F32).view(U32))
    return f


def _v_mbcnt(hi):
    def f(w, i):
        msk = w.src32(i.ops[1])
        assert isinstance(msk, int)
        add = _vec(w.src32(i.ops[2]))
        lane = w.lane
        if hi:
            cnt = np.array([bin(msk & ((1 << max(int(l) - 32, 0)) - 1)).count("1") for l in lane], dtype=U32)
        else:
            cnt = np.array([bin(msk & ((1 << min(int(l), 32)) - 1)).count("1") for l in lane], dtype=U32)
        w.wr_v32(i.ops[0], cnt + add)
    return f


def _v_readlane(w, i):
    lane = w.rd_s(i.ops[2]) & 63
    r = w._reg(i.ops[1])
    w.wr_s(i.ops[0], int(w.v[r[1]][lane]))


def _v_readfirstlane(w, i):
    m = np.nonzero(w.mask())[0]
    lane = int(m[0]) if len(m) else 0
    v = w.src32(i.ops[1])
    w.wr_s(i.ops[0], int(_vec(v)[lane]))


def _v_writelane(w, i):
    lane = w.rd_s(i.ops[2]) & 63
    r = w._reg(i.ops[0])
    w.v[r[1]][lane] = w.rd_s(i.ops[1]) & 0xffffffff


def _sx24(x):   # the low 24 bits of a 32-bit lane value, sign-extended
    return ((x.astype(I64) & 0xffffff) ^ 0x800000) - 0x800000


def _vop3_int(fn):
    def f(w, i):
        a, b, c = (_vec(w.src32(t)) for t in i.ops[1:4])
        with np.errstate(over="ignore"):
            w.wr_v32(i.ops[0], np.asarray(fn(a, b, c)).astype(U32))
    return f


def _v_not_b32(w, i):
    w.wr_v32(i.ops[0], ~_vec(w.src32(i.ops[1])))


def _v_cvt_u32_f64(w, i):
    x = w.f64(i.ops[1])
    with np.errstate(all="ignore"):
        r = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x), 0.0, 4294967295.0))
    w.wr_v32(i.ops[0], r.astype(U64).astype(U32))


def _v_cvt_i32_f64(w, i):
    x = w.f64(i.ops[1])
    with np.errstate(all="ignore"):
        r = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x), -2147483648.0, 2147483647.0))
    w.wr_v32(i.ops[0], r.astype(I64).astype(I32).view(U32))


def _v_frexp_mant_f64(w, i):
    x = w.f64(i.ops[1])
    with np.errstate(all="ignore"):
        m, _e = np.frexp(x)
    w.wr_v64(i.ops[0], np.where(np.isfinite(x), m, x))


def _v_frexp_exp_i32_f64(w, i):
    x = w.f64(i.ops[1])
    with np.errstate(all="ignore"):
        _m, e = np.frexp(x)
    w.wr_v32(i.ops[0], np.where(np.isfinite(x) & (x != 0), e, 0).astype(I32).view(U32))


def _v_rounding_f64(fn):
    def f(w, i):
        with np.errstate(all="ignore"):
            w.wr_v64(i.ops[0], fn(w.f64(i.ops[1])))
    return f


def _v_shift64(kind):
    def f(w, i):
        sh = _vec(w.src32(i.ops[1])).astype(U64) & U64(63)
        a = w.src64(i.ops[2], False)
        if kind == "l":
            r = a << sh
        elif kind == "r":
            r = a >> sh
        else:
            r = (a.view(I64) >> sh.astype(I64)).view(U64)
        w.wr_v64(i.ops[0], r)
    return f


def _v_bfrev_b32(w, i):
    a = _vec(w.src32(i.ops[1]))
    w.wr_v32(i.ops[0], np.array([int("{:032b}".format(int(x))[::-1], 2) for x in a], dtype=U32))


def _v_ffbh_u32(w, i):
    a = _vec(w.src32(i.ops[1]))
    w.wr_v32(i.ops[0], np.array([(32 - int(x).bit_length()) if x else 0xffffffff for x in a], dtype=U32))


def _global_atomic(fn, n, ret_possible=True):
    """global_atomic_<op>[_x2] [vdst,] vaddr, vdata, saddr|off — lanes applied in lane order (any order is a legal execution)"""
    def f(w, i):
        ops = i.ops
        has_ret = len(ops) == 4 or (len(ops) == 3 and ops[-1] not in ("off",) and not ops[-1].startswith("s"))
        if "sc0" in i.mods and len(ops) >= 4:
            has_ret = True
        k = 1 if has_ret else 0
        addr = _addr(w, i, ops[k], ops[k + 2] if len(ops) > k + 2 else None)
        r = w._reg(ops[k + 1])
        dt = U32 if n == 1 else U64
        for lane in np.nonzero(w.mask())[0]:
            cur = w.mem.read(addr[lane], 4 * n).view(dt)
            val = np.ascontiguousarray(w.v[r[1]:r[1] + n, lane]).view(dt)[0]
            old = cur[0].copy()
            cur[0] = fn(cur[0], val)
            if has_ret:
                rd = w._reg(ops[0])
                w.v[rd[1]:rd[1] + n, lane] = np.array([old], dtype=dt).view(U32)
    return f


def _v_accvgpr_write(w, i):
    r = w._reg(i.ops[0])
    m = w.mask()
    w.a[r[1]][m] = _vec(w.src32(i.ops[1]))[m]


def _v_accvgpr_read(w, i):
    r = w._reg(i.ops[1])
    w.wr_v32(i.ops[0], w.a[r[1]])


# ---- scalar ------------------------------------------------------------------------------------------------------------------------------
def _width(tok):
    if tok in ("vcc", "exec"):
        return 64
    m = re.match(r"^s\[(\d+):(\d+)\]$", tok)
    return 32 * (int(m.group(2)) - int(m.group(1)) + 1) if m else 32


def _s_mov(w, i):
    w.wr_s(i.ops[0], w.rd_s(i.ops[1], 64 if i.op.endswith("b64") else 32))


def _s_movk_i32(w, i):
    w.wr_s(i.ops[0], w.sx(int(i.ops[1], 0) & 0xffff, 16) & 0xffffffff)


def _s_mulk_i32(w, i):   # D = D * sext(imm16), low 32 bits; SCC untouched
    k = w.sx(int(i.ops[1], 0) & 0xffff, 16)
    w.wr_s(i.ops[0], (w.sx(w.rd_s(i.ops[0]), 32) * k) & 0xffffffff)


def _s_addk_i32(w, i):   # D = D + sext(imm16); SCC = signed overflow
    k = w.sx(int(i.ops[1], 0) & 0xffff, 16)
    r = w.sx(w.rd_s(i.ops[0]), 32) + k
    w.scc = int(not -(1 << 31) <= r < (1 << 31))
    w.wr_s(i.ops[0], r & 0xffffffff)


def _s_bitop(fn, bits):
    mask = (1 << bits) - 1

    def f(w, i):
        r = fn(w.rd_s(i.ops[1], bits), w.rd_s(i.ops[2], bits)) & mask
        w.wr_s(i.ops[0], r)
        w.scc = int(r != 0)
    return f


def _s_saveexec(fn):
    def f(w, i):
        old = w.exec
        w.exec = fn(w.rd_s(i.ops[1], 64), old) & ALL
        w.wr_s(i.ops[0], old)
        w.scc = int(w.exec != 0)
    return f


def _s_add_u32(w, i):
    r = w.rd_s(i.ops[1]) + w.rd_s(i.ops[2])
    w.wr_s(i.ops[0], r & 0xffffffff)
    w.scc = r >> 32


def _s_addc_u32(w, i):
    r = w.rd_s(i.ops[1]) + w.rd_s(i.ops[2]) + w.scc
    w.wr_s(i.ops[0], r & 0xffffffff)
    w.scc = r >> 32


def _s_sub_u32(w, i):
    a, b = w.rd_s(i.ops[1]), w.rd_s(i.ops[2])
    w.wr_s(i.ops[0], (a - b) & 0xffffffff)
    w.scc = int(b > a)


def _s_subb_u32(w, i):
    a, b = w.rd_s(i.ops[1]), w.rd_s(i.ops[2]) + w.scc
    w.wr_s(i.ops[0], (a - b) & 0xffffffff)
    w.scc = int(b > a)


def _s_add_i32(w, i):
    a, b = w.sx(w.rd_s(i.ops[1]), 32), w.sx(w.rd_s(i.ops[2]), 32)
    r = a + b
    w.wr_s(i.ops[0], r & 0xffffffff)
    w.scc = int(not -(1 << 31) <= r < (1 << 31))


def _s_sub_i32(w, i):
    a, b = w.sx(w.rd_s(i.ops[1]), 32), w.sx(w.rd_s(i.ops[2]), 32)
    r = a - b
    w.wr_s(i.ops[0], r & 0xffffffff)
    w.scc = int(not -(1 << 31) <= r < (1 << 31))


def _s_mul_i32(w, i):
    w.wr_s(i.ops[0], (w.rd_s(i.ops[1]) * w.rd_s(i.ops[2])) & 0xffffffff)


def _s_mul_hi_i32(w, i):
    w.wr_s(i.ops[0], ((w.sx(w.rd_s(i.ops[1]), 32) * w.sx(w.rd_s(i.ops[2]), 32)) >> 32) & 0xffffffff)


def _s_mul_hi_u32(w, i):
    w.wr_s(i.ops[0], ((w.rd_s(i.ops[1]) * w.rd_s(i.ops[2])) >> 32) & 0xffffffff)


def _s_shift(left, bits, arith=False):
    def f(w, i):
        a, sh = w.rd_s(i.ops[1], bits), w.rd_s(i.ops[2]) & (bits - 1)
        if left:
            r = (a << sh) & ((1 << bits) - 1)
        elif arith:
            r = (w.sx(a, bits) >> sh) & ((1 << bits) - 1)
        else:
            r = a >> sh
        w.wr_s(i.ops[0], r)
        w.scc = int(r != 0)
    return f


def _s_bfe_i32(w, i):
    a, c = w.rd_s(i.ops[1]), w.rd_s(i.ops[2])
    off, wid = c & 31, (c >> 16) & 0x7f
    r = (a >> off) & ((1 << wid) - 1) if wid else 0
    r = w.sx(r, wid) & 0xffffffff if wid else 0
    w.wr_s(i.ops[0], r)
    w.scc = int(r != 0)


def _s_bfe_u32(w, i):
    a, c = w.rd_s(i.ops[1]), w.rd_s(i.ops[2])
    off, wid = c & 31, (c >> 16) & 0x7f
    r = (a >> off) & ((1 << wid) - 1) if wid else 0
    w.wr_s(i.ops[0], r)
    w.scc = int(r != 0)


def _s_brev_b32(w, i):
    w.wr_s(i.ops[0], int("{:032b}".format(w.rd_s(i.ops[1]))[::-1], 2))


def _s_cselect(w, i):
    b = 64 if i.op.endswith("b64") else 32
    w.wr_s(i.ops[0], w.rd_s(i.ops[1], b) if w.scc else w.rd_s(i.ops[2], b))


def _s_cmp(kind, signed, bits):
    def f(w, i):
        a, b = w.rd_s(i.ops[0], bits), w.rd_s(i.ops[1], bits)
        if signed:
            a, b = w.sx(a, bits), w.sx(b, bits)
        w.scc = int({"eq": a == b, "lg": a != b, "gt": a > b, "ge": a >= b, "lt": a < b, "le": a <= b}[kind])
    return f


def _s_bitcmp(one):
    def f(w, i):
        bit = w.rd_s(i.ops[0]) >> (w.rd_s(i.ops[1]) & 31) & 1
        w.scc = int(bit == one)
    return f


def _s_getpc(w, i):
    w.wr_s(i.ops[0], i.addr + 4)


def _s_load(n):
    def f(w, i):
        base = w.rd_s(i.ops[1])
        off = w.rd_s(i.ops[2]) if len(i.ops) > 2 else 0
        off += i.mods.get("offset", 0)
        data = w.mem.read(base + off, 4 * n).view(U32)
        r = w._reg(i.ops[0])
        w.s[r[1]:r[1] + n] = data
    return f


# ---- vector memory, LDS ---------------------------------------------------------------------------------------------------------------------
def _addr(w, i, vtok, stok):
    off = i.mods.get("offset", 0)
    if stok in ("off", None):
        a = w.src64(vtok, False).astype(object)
        return [int(x) + off for x in a]
    base = w.rd_s(stok)
    v = _vec(w.src32(vtok))
    return [base + int(x) + off for x in v]


def _global_load(n):
    def f(w, i):
        addr = _addr(w, i, i.ops[1], i.ops[2] if len(i.ops) > 2 else None)
        r = w._reg(i.ops[0])
        for lane in np.nonzero(w.mask())[0]:
            (w.a if r[0] == "a" else w.v)[r[1]:r[1] + n, lane] = w.mem.read(addr[lane], 4 * n).view(U32)
    return f


def _global_store(n):
    def f(w, i):
        addr = _addr(w, i, i.ops[0], i.ops[2] if len(i.ops) > 2 else None)
        r = w._reg(i.ops[1])
        for lane in np.nonzero(w.mask())[0]:
            w.mem.write(addr[lane], np.ascontiguousarray((w.a if r[0] == "a" else w.v)[r[1]:r[1] + n, lane]).view(np.uint8))
    return f


# per-lane private memory (register spills): one byte array per lane of the wave
def _scratch_addr(w, i, vtok, stok):
    off = i.mods.get("offset", 0)
    base = 0 if stok in ("off", None) else w.rd_s(stok)
    if vtok in ("off", None):
        return np.full(LANES, base + off, dtype=np.int64)
    return _vec(w.src32(vtok)).astype(np.int64) + base + off


def _scratch_store(n):
    def f(w, i):
        # scratch_store_dwordxN vaddr|off, vdata, saddr|off
        a = _scratch_addr(w, i, i.ops[0], i.ops[2] if len(i.ops) > 2 else None)
        r = w._reg(i.ops[1])
        if w.scratch is None:
            w.scratch = np.zeros((LANES, max(w.k.scratch_bytes, 4) + 64), dtype=np.uint8)
        for lane in np.nonzero(w.mask())[0]:
            w.scratch[lane, a[lane]:a[lane] + 4 * n] = np.ascontiguousarray((w.a if r[0] == "a" else w.v)[r[1]:r[1] + n, lane]).view(np.uint8)
    return f


def _scratch_load(n):
    def f(w, i):
        a = _scratch_addr(w, i, i.ops[1], i.ops[2] if len(i.ops) > 2 else None)
        r = w._reg(i.ops[0])
        if w.scratch is None:
            w.scratch = np.zeros((LANES, max(w.k.scratch_bytes, 4) + 64), dtype=np.uint8)
        for lane in np.nonzero(w.mask())[0]:
            (w.a if r[0] == "a" else w.v)[r[1]:r[1] + n, lane] = w.scratch[lane, a[lane]:a[lane] + 4 * n].view(U32)
    return f


def _ds_addr(w, i, tok):
    return _vec(w.src32(tok)).astype(np.int64) + i.mods.get("offset", 0)


def _ds_write(n):
    def f(w, i):
        a = _ds_addr(w, i, i.ops[0])
        r = w._reg(i.ops[1])
        for lane in np.nonzero(w.mask())[0]:
            w.lds[a[lane]:a[lane] + 4 * n] = np.ascontiguousarray((w.a if r[0] == "a" else w.v)[r[1]:r[1] + n, lane]).view(np.uint8)
    return f


def _ds_read(n):
    def f(w, i):
        a = _ds_addr(w, i, i.ops[1])
        r = w._reg(i.ops[0])
        for lane in np.nonzero(w.mask())[0]:
            (w.a if r[0] == "a" else w.v)[r[1]:r[1] + n, lane] = w.lds[a[lane]:a[lane] + 4 * n].view(U32)
    return f


def _ds_or_b32(w, i):
    a = _ds_addr(w, i, i.ops[0])
    r = w._reg(i.ops[1])
    for lane in np.nonzero(w.mask())[0]:
        cur = w.lds[a[lane]:a[lane] + 4].view(U32)
        cur[0] |= (w.a if r[0] == "a" else w.v)[r[1], lane]


def _ds_bpermute_b32(w, i):
    # vdst, vaddr, vdata: lane l reads vdata of lane (vaddr[l] / 4) mod 64; a source lane that is not active reads as 0
    a = (_ds_addr(w, i, i.ops[1]) >> 2) & 63
    data = _vec(w.src32(i.ops[2]))
    em = w.mask()
    w.wr_v32(i.ops[0], np.where(em[a], data[a], U32(0)))


def _ds_write2(n, stride):
    def f(w, i):
        # ds_write2[st64]_bN vaddr, vdata0, vdata1 offset0:a offset1:b  (offsets in units of the element size, x64 for the st64 forms)
        a = _vec(w.src32(i.ops[0])).astype(np.int64)
        unit = 4 * n * stride
        for tok, off in ((i.ops[1], i.mods.get("offset0", 0)), (i.ops[2], i.mods.get("offset1", 0))):
            r = w._reg(tok)
            for lane in np.nonzero(w.mask())[0]:
                p = a[lane] + off * unit
                w.lds[p:p + 4 * n] = np.ascontiguousarray((w.a if r[0] == "a" else w.v)[r[1]:r[1] + n, lane]).view(np.uint8)
    return f


def _ds_read2(n, stride):
    def f(w, i):
        a = _vec(w.src32(i.ops[1])).astype(np.int64)
        unit = 4 * n * stride
        r = w._reg(i.ops[0])
        for k, off in enumerate((i.mods.get("offset0", 0), i.mods.get("offset1", 0))):
            for lane in np.nonzero(w.mask())[0]:
                p = a[lane] + off * unit
                (w.a if r[0] == "a" else w.v)[r[1] + k * n:r[1] + (k + 1) * n, lane] = w.lds[p:p + 4 * n].view(U32)
    return f


def _ds_add_u32(ret):
    def f(w, i):
        k = 1 if ret else 0
        a = _ds_addr(w, i, i.ops[k])
        r = w._reg(i.ops[k + 1])
        rd = w._reg(i.ops[0]) if ret else None
        for lane in np.nonzero(w.mask())[0]:  # lane order: one of the legal orders of the hardware's atomics
            cur = w.lds[a[lane]:a[lane] + 4].view(U32)
            old = int(cur[0])
            cur[0] = (old + int((w.a if r[0] == "a" else w.v)[r[1], lane])) & 0xffffffff
            if ret:
                w.v[rd[1], lane] = old
    return f


def _global_load_small(nbytes, signed):
    def f(w, i):
        addr = _addr(w, i, i.ops[1], i.ops[2] if len(i.ops) > 2 else None)
        r = w._reg(i.ops[0])
        for lane in np.nonzero(w.mask())[0]:
            v = int.from_bytes(bytes(w.mem.read(addr[lane], nbytes)), "little", signed=signed)
            (w.a if r[0] == "a" else w.v)[r[1], lane] = v & 0xffffffff
    return f


def _global_store_small(nbytes):
    def f(w, i):
        addr = _addr(w, i, i.ops[0], i.ops[2] if len(i.ops) > 2 else None)
        r = w._reg(i.ops[1])
        for lane in np.nonzero(w.mask())[0]:
            w.mem.write(addr[lane], np.frombuffer(int((w.a if r[0] == "a" else w.v)[r[1], lane] & ((1 << (8 * nbytes)) - 1)).to_bytes(nbytes, "little"), dtype=np.uint8))
    return f


def _s_not(bits):
    def f(w, i):
        r = ~w.rd_s(i.ops[1], bits) & ((1 << bits) - 1)
        w.wr_s(i.ops[0], r)
        w.scc = int(r != 0)
    return f


def _s_flbit_i32_b64(w, i):
    v = w.rd_s(i.ops[1], 64)
    w.wr_s(i.ops[0], (64 - v.bit_length()) if v else 0xffffffff)


def _s_bcnt1(bits):
    def f(w, i):
        r = bin(w.rd_s(i.ops[1], bits) & ((1 << bits) - 1)).count("1")
        w.wr_s(i.ops[0], r)
        w.scc = int(r != 0)
    return f


def _s_ff1(bits):
    def f(w, i):
        v = w.rd_s(i.ops[1], bits) & ((1 << bits) - 1)
        w.wr_s(i.ops[0], ((v & -v).bit_length() - 1) if v else 0xffffffff)
    return f


def _s_flbit_i32_b32(w, i):
    v = w.rd_s(i.ops[1])
    w.wr_s(i.ops[0], (32 - v.bit_length()) if v else 0xffffffff)


def _s_minmax(fn, signed):
    def f(w, i):
        a, b = w.rd_s(i.ops[1]), w.rd_s(i.ops[2])
        sa, sb = (w.sx(a, 32), w.sx(b, 32)) if signed else (a, b)
        r = fn(sa, sb)
        w.scc = int(r == sa)
        w.wr_s(i.ops[0], r & 0xffffffff)
    return f


def _v_bitop3_b32(w, i):
    a, b, c = (_vec(w.src32(t)) for t in i.ops[1:4])
    tt = i.mods["bitop3"]
    out = np.zeros(LANES, dtype=U32)
    for idx in range(8):
        if tt >> idx & 1:
            ta = a if idx & 4 else ~a
            tb = b if idx & 2 else ~b
            tc = c if idx & 1 else ~c
            out |= ta & tb & tc
    w.wr_v32(i.ops[0], out)


def _nop(w, i):
    return None


def _v_accvgpr_mov(w, i):
    d, r = w._reg(i.ops[0]), w._reg(i.ops[1])
    m = w.mask()
    w.a[d[1]][m] = w.a[r[1]][m]


def _s_set_gpr_idx_on(w, i):
    m = re.match(r"^gpr_idx\((.*)\)$", i.ops[1])
    modes = set(m.group(1).split(",")) if m else {n for b, n in enumerate(("SRC0", "SRC1", "SRC2", "DST")) if int(i.ops[1], 0) >> b & 1}
    idx = w.rd_s(i.ops[0]) & 0xff
    w.m0 = (w.m0 & ~0xf0ff) | idx | (sum(1 << b for b, n in enumerate(("SRC0", "SRC1", "SRC2", "DST")) if n in modes) << 12)
    w.gpr_idx = (idx, modes)


def _s_set_gpr_idx_off(w, i):
    w.gpr_idx = None


_OPS = {
    "v_accvgpr_mov_b32": _v_accvgpr_mov, "s_set_gpr_idx_on": _s_set_gpr_idx_on, "s_set_gpr_idx_off": _s_set_gpr_idx_off,
    "s_nop": _nop, "s_waitcnt": _nop, "s_sleep": _nop, "s_setprio": _nop, "s_waitcnt_vscnt": _nop, "s_setreg_imm32_b32": _nop,
    "s_endpgm": lambda w, i: "end", "s_barrier": lambda w, i: "barrier",
    "s_branch": _s_branch, "s_setpc_b64": _s_setpc,
    "s_cbranch_scc0": _s_cbranch(lambda w: not w.scc), "s_cbranch_scc1": _s_cbranch(lambda w: w.scc),
    "s_cbranch_vccz": _s_cbranch(lambda w: w.vcc == 0), "s_cbranch_vccnz": _s_cbranch(lambda w: w.vcc != 0),
    "s_cbranch_execz": _s_cbranch(lambda w: w.exec == 0), "s_cbranch_execnz": _s_cbranch(lambda w: w.exec != 0),
    "s_mov_b32": _s_mov, "s_mov_b64": _s_mov, "s_movk_i32": _s_movk_i32, "s_mulk_i32": _s_mulk_i32, "s_addk_i32": _s_addk_i32,
    "s_and_b32": _s_bitop(lambda a, b: a & b, 32), "s_and_b64": _s_bitop(lambda a, b: a & b, 64),
    "s_or_b32": _s_bitop(lambda a, b: a | b, 32), "s_or_b64": _s_bitop(lambda a, b: a | b, 64),
    "s_xor_b32": _s_bitop(lambda a, b: a ^ b, 32), "s_xor_b64": _s_bitop(lambda a, b: a ^ b, 64),
    "s_andn2_b32": _s_bitop(lambda a, b: a & ~b, 32), "s_andn2_b64": _s_bitop(lambda a, b: a & ~b, 64),
    "s_nor_b32": _s_bitop(lambda a, b: ~(a | b), 32), "s_nor_b64": _s_bitop(lambda a, b: ~(a | b), 64), "s_nand_b32": _s_bitop(lambda a, b: ~(a & b), 32),
    "s_nand_b64": _s_bitop(lambda a, b: ~(a & b), 64), "s_xnor_b32": _s_bitop(lambda a, b: ~(a ^ b), 32), "s_xnor_b64": _s_bitop(lambda a, b: ~(a ^ b), 64),
    "s_orn2_b32": _s_bitop(lambda a, b: a | ~b, 32), "s_orn2_b64": _s_bitop(lambda a, b: a | ~b, 64),
    "s_and_saveexec_b64": _s_saveexec(lambda s, e: s & e), "s_or_saveexec_b64": _s_saveexec(lambda s, e: s | e),
    "s_xor_saveexec_b64": _s_saveexec(lambda s, e: s ^ e), "s_andn2_saveexec_b64": _s_saveexec(lambda s, e: s & ~e),
    "s_orn2_saveexec_b64": _s_saveexec(lambda s, e: s | ~e), "s_andn1_saveexec_b64": _s_saveexec(lambda s, e: ~s & e),
    "s_add_u32": _s_add_u32, "s_addc_u32": _s_addc_u32, "s_sub_u32": _s_sub_u32, "s_subb_u32": _s_subb_u32, "s_add_i32": _s_add_i32, "s_sub_i32": _s_sub_i32,
    "s_mul_i32": _s_mul_i32, "s_mul_hi_u32": _s_mul_hi_u32, "s_mul_hi_i32": _s_mul_hi_i32,
    "s_lshl_b32": _s_shift(True, 32), "s_lshl_b64": _s_shift(True, 64), "s_lshr_b32": _s_shift(False, 32), "s_lshr_b64": _s_shift(False, 64),
    "s_ashr_i32": _s_shift(False, 32, True), "s_ashr_i64": _s_shift(False, 64, True),
    "s_bfe_i32": _s_bfe_i32, "s_bfe_u32": _s_bfe_u32, "s_brev_b32": _s_brev_b32, "s_cselect_b32": _s_cselect, "s_cselect_b64": _s_cselect,
    "s_cmp_eq_u32": _s_cmp("eq", False, 32), "s_cmp_lg_u32": _s_cmp("lg", False, 32), "s_cmp_gt_u32": _s_cmp("gt", False, 32), "s_cmp_ge_u32": _s_cmp("ge", False, 32),
    "s_cmp_lt_u32": _s_cmp("lt", False, 32), "s_cmp_le_u32": _s_cmp("le", False, 32), "s_cmp_eq_i32": _s_cmp("eq", True, 32), "s_cmp_lg_i32": _s_cmp("lg", True, 32),
    "s_cmp_gt_i32": _s_cmp("gt", True, 32), "s_cmp_ge_i32": _s_cmp("ge", True, 32), "s_cmp_lt_i32": _s_cmp("lt", True, 32), "s_cmp_le_i32": _s_cmp("le", True, 32),
    "s_cmp_eq_u64": _s_cmp("eq", False, 64), "s_cmp_lg_u64": _s_cmp("lg", False, 64),
    "s_bitcmp1_b32": _s_bitcmp(1), "s_bitcmp0_b32": _s_bitcmp(0), "s_getpc_b64": _s_getpc,
    "s_load_dword": _s_load(1), "s_load_dwordx2": _s_load(2), "s_load_dwordx4": _s_load(4), "s_load_dwordx8": _s_load(8), "s_load_dwordx16": _s_load(16),
    "v_nop": _nop,
    "v_add_f64": _vop_f64(lambda a, b: a + b, 2), "v_mul_f64": _vop_f64(lambda a, b: a * b, 2), "v_fma_f64": _vop_f64(_fma, 3), "v_fmac_f64": _v_fmac_f64,
    "v_min_f64": _vop_f64(np.fmin, 2), "v_max_f64": _vop_f64(np.fmax, 2),
    "v_ldexp_f64": _v_ldexp_f64, "v_rcp_f64": _v_rcp_f64, "v_rsq_f64": _v_rsq_f64, "v_sqrt_f64": _v_sqrt_f64,
    "v_div_scale_f64": _v_div_scale_f64, "v_div_fmas_f64": _v_div_fmas_f64, "v_div_fixup_f64": _v_div_fixup_f64,
    "v_mov_b32": _v_mov_b32, "v_mov_b64": _v_mov_b64,
    "v_and_b32": _vop2_int(lambda a, b: a & b), "v_or_b32": _vop2_int(lambda a, b: a | b), "v_xor_b32": _vop2_int(lambda a, b: a ^ b),
    "v_add_u32": _vop2_int(lambda a, b: a + b), "v_sub_u32": _vop2_int(lambda a, b: a - b), "v_subrev_u32": _vop2_int(lambda a, b: b - a),
    "v_lshlrev_b32": _vop2_int(lambda a, b: b << (a & U32(31))), "v_lshrrev_b32": _vop2_int(lambda a, b: b >> (a & U32(31))),
    "v_ashrrev_i32": _vop2_int(lambda a, b: (b.view(I32) >> (a & U32(31)).astype(I32)).view(U32)),
    "v_mul_lo_u32": _vop2_int(lambda a, b: (a.astype(U64) * b.astype(U64)).astype(U32)),
    "v_add_co_u32": _v_add_co(), "v_sub_co_u32": _v_add_co(sub=True), "v_subrev_co_u32": _v_add_co(sub=True, rev=True),
    "v_addc_co_u32": _v_add_co(carry_in=True), "v_subb_co_u32": _v_add_co(sub=True, carry_in=True), "v_subbrev_co_u32": _v_add_co(sub=True, rev=True, carry_in=True),
    "v_add3_u32": _vop3_int(lambda a, b, c: a + b + c), "v_lshl_add_u32": _vop3_int(lambda a, b, c: (a << (b & U32(31))) + c),
    "v_add_lshl_u32": _vop3_int(lambda a, b, c: (a + b) << (c & U32(31))), "v_lshl_or_b32": _vop3_int(lambda a, b, c: (a << (b & U32(31))) | c),
    "v_mad_i32_i24": _vop3_int(lambda a, b, c: (_sx24(a) * _sx24(b) + c.view(I32).astype(I64)).astype(I64).astype(U64).astype(U32)),
    "v_mad_legacy_u16": _vop3_int(lambda a, b, c: ((a & U32(0xffff)) * (b & U32(0xffff)) + (c & U32(0xffff))) & U32(0xffff)),
    "v_mad_u16": _vop3_int(lambda a, b, c: ((a & U32(0xffff)) * (b & U32(0xffff)) + (c & U32(0xffff))) & U32(0xffff)),
    "v_med3_i32": _vop3_int(lambda a, b, c: np.sort(np.stack([a.view(I32), b.view(I32), c.view(I32)]), axis=0)[1].view(U32)),
    "v_med3_u32": _vop3_int(lambda a, b, c: np.sort(np.stack([a, b, c]), axis=0)[1]),
    "v_min3_i32": _vop3_int(lambda a, b, c: np.minimum(np.minimum(a.view(I32), b.view(I32)), c.view(I32)).view(U32)),
    "v_max3_i32": _vop3_int(lambda a, b, c: np.maximum(np.maximum(a.view(I32), b.view(I32)), c.view(I32)).view(U32)),
    "v_min3_u32": _vop3_int(lambda a, b, c: np.minimum(np.minimum(a, b), c)), "v_max3_u32": _vop3_int(lambda a, b, c: np.maximum(np.maximum(a, b), c)),
    "v_and_or_b32": _vop3_int(lambda a, b, c: (a & b) | c), "v_or3_b32": _vop3_int(lambda a, b, c: a | b | c), "v_xad_u32": _vop3_int(lambda a, b, c: (a ^ b) + c),
    "v_bfi_b32": _vop3_int(lambda a, b, c: (a & b) | (~a & c)), "v_alignbit_b32": _vop3_int(lambda a, b, c: ((a.astype(U64) << U64(32) | b.astype(U64)) >> (c & U32(31)).astype(U64)).astype(U32)),
    # 16-bit integer VALU (GFX9: the result's low half, the destination's high half zeroed)
    "v_add_u16": _vop2_int(lambda a, b: (a + b) & U32(0xffff)), "v_sub_u16": _vop2_int(lambda a, b: (a - b) & U32(0xffff)),
    "v_subrev_u16": _vop2_int(lambda a, b: (b - a) & U32(0xffff)), "v_mul_lo_u16": _vop2_int(lambda a, b: ((a & U32(0xffff)) * (b & U32(0xffff))) & U32(0xffff)),
    "v_lshlrev_b16": _vop2_int(lambda a, b: ((b & U32(0xffff)) << (a & U32(15))) & U32(0xffff)), "v_lshrrev_b16": _vop2_int(lambda a, b: (b & U32(0xffff)) >> (a & U32(15))),
    "v_max_u16": _vop2_int(lambda a, b: np.maximum(a & U32(0xffff), b & U32(0xffff))), "v_min_u16": _vop2_int(lambda a, b: np.minimum(a & U32(0xffff), b & U32(0xffff))),
    "v_mul_i32_i24": _vop2_int(lambda a, b: (_sx24(a) * _sx24(b)).astype(U64).astype(U32)),
    "v_mul_u32_u24": _vop2_int(lambda a, b: ((a & U32(0xffffff)).astype(U64) * (b & U32(0xffffff)).astype(U64)).astype(U32)),
    "v_mul_hi_u32": _vop2_int(lambda a, b: ((a.astype(U64) * b.astype(U64)) >> U64(32)).astype(U32)),
    "v_min_u32": _vop2_int(np.minimum), "v_max_u32": _vop2_int(np.maximum),
    "v_min_i32": _vop2_int(lambda a, b: np.minimum(a.view(I32), b.view(I32)).view(U32)), "v_max_i32": _vop2_int(lambda a, b: np.maximum(a.view(I32), b.view(I32)).view(U32)),
    "v_bfrev_b32": _v_bfrev_b32, "v_ffbh_u32": _v_ffbh_u32,
    "v_accvgpr_write_b32": _v_accvgpr_write, "v_accvgpr_read_b32": _v_accvgpr_read,
    "v_not_b32": _v_not_b32, "v_cvt_u32_f64": _v_cvt_u32_f64, "v_cvt_i32_f64": _v_cvt_i32_f64, "v_frexp_mant_f64": _v_frexp_mant_f64, "v_frexp_exp_i32_f64": _v_frexp_exp_i32_f64,
    "v_trunc_f64": _v_rounding_f64(np.trunc), "v_floor_f64": _v_rounding_f64(np.floor), "v_ceil_f64": _v_rounding_f64(np.ceil), "v_rndne_f64": _v_rounding_f64(np.rint),
    "v_lshlrev_b64": _v_shift64("l"), "v_lshrrev_b64": _v_shift64("r"), "v_ashrrev_i64": _v_shift64("a"),
    "v_lshl_add_u64": _v_lshl_add_u64, "v_bfe_u32": _v_bfe_u32, "v_mad_u64_u32": _v_mad_u64_u32, "v_mad_u32_u24": _v_mad_u32_u24,
    "v_cndmask_b32": _v_cndmask_b32, "v_cvt_f64_u32": _v_cvt_f64_u32, "v_cvt_f64_i32": _v_cvt_f64_i32,
    "v_cvt_f32_f64": _v_cvt_f32_f64, "v_cvt_f64_f32": _v_cvt_f64_f32, "v_log_f32": _f32_unary(np.log2), "v_exp_f32": _f32_unary(np.exp2),
    # the compiler's 32-bit integer division: an f32 reciprocal estimate, then integer correction steps that absorb the estimate's last-bit error
    "v_cvt_f32_u32": lambda w, i: w.wr_v32(i.ops[0], _vec(w.src32(i.ops[1])).astype(F32).view(U32)),
    "v_cvt_f32_i32": lambda w, i: w.wr_v32(i.ops[0], _vec(w.src32(i.ops[1])).view(I32).astype(F32).view(U32)),
    "v_cvt_u32_f32": lambda w, i: w.wr_v32(i.ops[0], np.clip(np.nan_to_num(np.trunc(_f32(w, i.ops[1]).astype(F64)), nan=0.0), 0.0, 4294967295.0).astype(U64).astype(U32)),
    "v_cvt_i32_f32": lambda w, i: w.wr_v32(i.ops[0], np.clip(np.nan_to_num(np.trunc(_f32(w, i.ops[1]).astype(F64)), nan=0.0), -2147483648.0, 2147483647.0).astype(I64).astype(I32).view(U32)),
    "v_rcp_iflag_f32": _f32_unary(lambda x: F32(1.0) / x), "v_rcp_f32": _f32_unary(lambda x: F32(1.0) / x),
    "v_mul_f32": _f32_binary(np.multiply), "v_add_f32": _f32_binary(np.add), "v_sub_f32": _f32_binary(np.subtract),
    "v_mbcnt_lo_u32_b32": _v_mbcnt(False), "v_mbcnt_hi_u32_b32": _v_mbcnt(True),
    "v_readlane_b32": _v_readlane, "v_readfirstlane_b32": _v_readfirstlane, "v_writelane_b32": _v_writelane,
    "global_load_dword": _global_load(1), "global_load_dwordx2": _global_load(2), "global_load_dwordx3": _global_load(3), "global_load_dwordx4": _global_load(4),
    "global_store_dword": _global_store(1), "global_store_dwordx2": _global_store(2), "global_store_dwordx3": _global_store(3), "global_store_dwordx4": _global_store(4),
    # flat_*: the address decides the memory; the kernels of this library only ever hand them global addresses (an LDS / scratch aperture address would fail the
    # memory lookup, loudly)
    "flat_load_dword": _global_load(1), "flat_load_dwordx2": _global_load(2), "flat_load_dwordx3": _global_load(3), "flat_load_dwordx4": _global_load(4),
    "flat_store_dword": _global_store(1), "flat_store_dwordx2": _global_store(2), "flat_store_dwordx3": _global_store(3), "flat_store_dwordx4": _global_store(4),
    "scratch_store_dword": _scratch_store(1), "scratch_store_dwordx2": _scratch_store(2), "scratch_store_dwordx4": _scratch_store(4),
    "scratch_load_dword": _scratch_load(1), "scratch_load_dwordx2": _scratch_load(2), "scratch_load_dwordx4": _scratch_load(4),
    "global_atomic_add": _global_atomic(lambda a, b: a + b, 1), "global_atomic_add_x2": _global_atomic(lambda a, b: a + b, 2),
    "global_atomic_umax": _global_atomic(max, 1), "global_atomic_umax_x2": _global_atomic(max, 2), "global_atomic_umin": _global_atomic(min, 1), "global_atomic_umin_x2": _global_atomic(min, 2),
    "global_atomic_or": _global_atomic(lambda a, b: a | b, 1), "global_atomic_or_x2": _global_atomic(lambda a, b: a | b, 2),
    "ds_write_b32": _ds_write(1), "ds_write_b64": _ds_write(2), "ds_read_b32": _ds_read(1), "ds_read_b64": _ds_read(2), "ds_or_b32": _ds_or_b32,
    "ds_bpermute_b32": _ds_bpermute_b32,
    "ds_write_b128": _ds_write(4), "ds_read_b128": _ds_read(4), "ds_write_b96": _ds_write(3), "ds_read_b96": _ds_read(3),
    "ds_write2_b32": _ds_write2(1, 1), "ds_write2_b64": _ds_write2(2, 1), "ds_write2st64_b32": _ds_write2(1, 64), "ds_write2st64_b64": _ds_write2(2, 64),
    "ds_read2_b32": _ds_read2(1, 1), "ds_read2_b64": _ds_read2(2, 1), "ds_read2st64_b32": _ds_read2(1, 64), "ds_read2st64_b64": _ds_read2(2, 64),
    "ds_add_u32": _ds_add_u32(False), "ds_add_rtn_u32": _ds_add_u32(True),
    "global_load_ushort": _global_load_small(2, False), "global_load_sshort": _global_load_small(2, True), "global_load_ubyte": _global_load_small(1, False),
    "global_load_sbyte": _global_load_small(1, True), "global_store_short": _global_store_small(2), "global_store_byte": _global_store_small(1),
    "s_bcnt1_i32_b32": _s_bcnt1(32), "s_bcnt1_i32_b64": _s_bcnt1(64), "s_ff1_i32_b32": _s_ff1(32), "s_ff1_i32_b64": _s_ff1(64), "s_flbit_i32_b32": _s_flbit_i32_b32,
    "s_not_b32": _s_not(32), "s_not_b64": _s_not(64), "s_flbit_i32_b64": _s_flbit_i32_b64,
    "s_min_u32": _s_minmax(min, False), "s_max_u32": _s_minmax(max, False), "s_min_i32": _s_minmax(min, True), "s_max_i32": _s_minmax(max, True),
    "v_bitop3_b32": _v_bitop3_b32,
}

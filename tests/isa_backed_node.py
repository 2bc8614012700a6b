"""TEST INFRASTRUCTURE — the two host-side stand-ins put together: tests/cpp/fake_hip.cpp (a fake node: HIP runtime + RCCL over tracked host memory) hands every
kernel launch to tools/gfx950_isa_interp.py, which executes the kernel's COMPILED gfx950 code — taken from the library's own object files
(numericalnim_amd/csrc/*.o) or, for run-time compiled right-hand sides, from the code object hiprtc just produced.  With it the library's C-ABI entries run end to
end in a process without a GPU: the real host logic, the real device code, results one can compare with the oracle — slowly (tens of microseconds per
wave-instruction), so on batches of tens to hundreds of IVPs.  It exists to give the code paths that need more than one GPU, or were written in a round without
one, a first execution whose RESULTS are checked.  It is not a CPU path of the product: nothing in the package or the library refers to it, it needs the ROCm LLVM
tools, the built object files and LD_PRELOAD, and it is five to six orders of magnitude slower than a GPU.

Usage (inside a process started with LD_PRELOAD=<fake_hip.so> and FAKE_HIP_LIB pointing at it):  node = isa_backed_node.attach()  — before the first launch."""
import ctypes as C
import glob
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gfx950_isa_interp as G  # noqa: E402

CSRC = os.path.join(ROOT, "numericalnim_amd", "csrc")
HOOK = C.CFUNCTYPE(C.c_int, C.c_char_p, C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_void_p))


class HostMemory(G.Memory):
    """Device pointers of the fake node ARE host addresses: accesses outside the interpreter's own regions (code object image, kernarg buffer) go to the address
    itself — after the fake runtime confirmed that the range lies inside one live allocation."""

    def __init__(self, co, fake):
        super().__init__(co)
        self.fake = fake

    def _find(self, addr, n):
        for base, b in self.regions:
            if base <= addr and addr + n <= base + b.size:
                return b, addr - base
        if not self.fake.fake_hip_owns(C.c_void_p(addr), C.c_size_t(n)):
            raise MemoryError("kernel access of %d bytes at 0x%x is outside every live allocation of the fake node" % (n, addr))
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(addr)), 0


class Node:
    def __init__(self, fake_path):
        self.F = C.CDLL(fake_path)
        self.F.fake_hip_owns.restype = C.c_int
        self.objects = {}      # object file -> CodeObject
        self.by_name = {}      # mangled kernel name -> (CodeObject, Kernel)
        self.rtc = {}          # image address -> CodeObject
        self.launches = 0
        self.instructions = 0
        self.errors = []
        self._hook = HOOK(self._launch)
        self.F.fake_hip_set_launch_hook(self._hook)

    def detach(self):
        self.F.fake_hip_set_launch_hook(None)

    # ---- kernel lookup ----
    def _compiled_in(self, name):
        if name in self.by_name:
            return self.by_name[name]
        needle = name.encode()
        for path in sorted(glob.glob(os.path.join(CSRC, "*.o"))):
            with open(path, "rb") as f:
                if needle not in f.read():
                    continue
            co = self.objects.get(path)
            if co is None:
                co = self.objects[path] = G.CodeObject(path)
            if name in co.symbols and name + ".kd" in co.symbols:
                k = self.by_name[name] = (co, G.Kernel(co, name))
                return k
        raise LookupError("no object file under numericalnim_amd/csrc holds kernel " + name)

    def _run_time_compiled(self, name, image):
        co = self.rtc.get(image)
        if co is None:
            head = C.string_at(image, 64)
            assert head[:4] == b"\x7fELF", "hipModuleLoadData image is not an ELF code object"
            e_shoff, = struct.unpack_from("<Q", head, 40)
            e_shentsize, e_shnum = struct.unpack_from("<HH", head, 58)
            co = self.rtc[image] = G.CodeObject(elf_bytes=C.string_at(image, e_shoff + e_shentsize * e_shnum))
        key = (image, name)
        if key not in self.by_name:
            self.by_name[key] = (co, G.Kernel(co, name))
        return self.by_name[key]

    # ---- the hook ----
    def _launch(self, name, image, gx, gy, gz, bx, by, bz, args):
        try:
            name = name.decode()
            co, kern = self._run_time_compiled(name, image) if image else self._compiled_in(name)
            explicit = [(o, s) for o, s, kind in co.kernel_args(kern.name) if not kind.startswith("hidden_")]
            ka = bytearray(max([o + s for o, s in explicit], default=0))
            for i, (off, size) in enumerate(explicit):
                ka[off:off + size] = C.string_at(args[i], size)
            mem = HostMemory(co, self.F)
            st = G.Machine(co).launch(kern, (gx, gy, gz), (bx, by, bz), bytes(ka), mem, max_instructions=400_000_000)
            self.launches += 1
            self.instructions += sum(c["total"] for c in st)
            return 0
        except BaseException as e:  # noqa: BLE001  (an exception must not unwind through the C caller)
            import traceback
            self.errors.append("%s: %s\n%s" % (name, e, traceback.format_exc()))
            print("isa_backed_node: launch of %s failed: %r" % (name, e), file=sys.stderr, flush=True)
            if os.environ.get("NNHIP_ISA_NODE_VERBOSE"):
                print(self.errors[-1], file=sys.stderr, flush=True)
            return 1


def attach():
    return Node(os.environ["FAKE_HIP_LIB"])

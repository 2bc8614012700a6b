"""Run by scripts/tsan_host_audit.sh (SAN=address) in a subprocess per entry: every function include/nnhip_ode.h declares, called through ctypes with hostile
arguments on the fake HIP runtime (tests/cpp/fake_hip.cpp; kernels do nothing) against the AddressSanitizer + UBSan build of the library's host code.
TEST INFRASTRUCTURE, a one-off audit tool.  The contract it checks is the boundary's error behaviour: a C caller that passes NULL where data is needed, a
negative size, an unknown enum or a non-finite time gets an error code (and nnhip_last_error says why) — never a crash, an out-of-bounds access or a hang.

Two modes per entry, seeded:
  null   every pointer argument NULL (the options block: valid or NULL), scalars from a hostile menu incl. large sizes — whatever the sizes say, nothing may be
         dereferenced
  valid  every pointer argument (every other trial: a random 70 % of them, the rest NULL) a zeroed 64 KiB tracked device allocation (pointer arrays: filled with the address of another such block), scalars from a SMALL
         hostile menu (sizes <= 5, enums -1..4, times incl. NaN / inf / reversed): the library must stay inside the blocks whatever the scalars are (the
         stand-in aborts on a copy outside the allocation it addresses)
usage: fake_hip_abi_arg_fuzz.py NAME [TRIALS]     (prints one line per call BEFORE making it, so the last line names the call that died)"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from numericalnim_amd import _lib  # noqa: E402

L = _lib.lib()
F = C.CDLL(os.environ["FAKE_HIP_LIB"])
BLOCK = 64 << 10


def block():
    p = C.c_void_p()
    assert F.hipMalloc(C.byref(p), C.c_size_t(BLOCK)) == 0
    C.memset(p.value, 0, BLOCK)
    return p.value


def main(name, trials):
    restype, argtypes = _lib.SIGNATURES[name]
    rng = random.Random(hash(name) & 0xffff)
    opt = _lib.Options()
    assert L.nnhip_ode_default_options(C.byref(opt)) == 0
    inner = block()
    ptr_array = block()
    (C.c_void_p * (BLOCK // 8)).from_address(ptr_array)[:] = [inner] * (BLOCK // 8)
    blocks = [block() for _ in range(12)]
    for t in range(trials):
        mode = "null" if t % 2 == 0 or name == "nnhip_host_free" else "valid"   # (freeing a block that nnhip_host_alloc did not hand out is the caller's error)
        args, shown = [], []
        k = 0
        for at in argtypes:
            if at is C.c_int:
                v = rng.choice([-1, 0, 1, 2, 3, 16, 100000, 2 ** 31 - 1] if mode == "null" else [-1, 0, 1, 2, 3, 4])
            elif at is C.c_int64:
                v = rng.choice([-1, 0, 1, 7, 1000, 2 ** 40] if mode == "null" else [-1, 0, 1, 2, 5])
            elif at is C.c_double:
                v = rng.choice([0.0, 1.0, -1.0, float("nan"), float("inf"), 0.5, 1e-3])
            elif at is C.c_char_p:
                v = rng.choice([None, b"", b"rk4", b"tsit54", b"no such thing", b"adv_lean", b"dy[0] = -y[0];"])
            elif at is C.POINTER(_lib.Options):
                if rng.random() < 0.85:
                    o = _lib.Options()
                    C.memmove(C.byref(o), C.byref(opt), C.sizeof(o))
                    if rng.random() < 0.3:  # a hostile options block: one field replaced
                        fld = rng.choice([f[0] for f in _lib.Options._fields_])
                        setattr(o, fld, rng.choice([0.0, -1.0, float("nan"), float("inf")]))
                    v = C.pointer(o)
                else:
                    v = None
            elif mode == "null" or (t % 4 == 3 and rng.random() < 0.3):   # (every other "valid" trial: some of the pointers NULL, the rest valid)
                v = None
            elif at in (C.POINTER(C.c_void_p), C.POINTER(C.c_char_p)):
                v = C.cast(ptr_array, at)
            else:
                v = C.cast(blocks[k % len(blocks)], at) if at is not C.c_void_p else blocks[k % len(blocks)]
                k += 1
            args.append(v)
            shown.append("opt" if at is C.POINTER(_lib.Options) and v is not None else ("ptr" if (v is not None and not isinstance(v, (int, float, bytes))) or (at is C.c_void_p and v) else repr(v)))
        print("%s[%d %s](%s)" % (name, t, mode, ", ".join(shown)), flush=True)
        rc = getattr(L, name)(*args)
        if restype is C.c_int and isinstance(rc, int) and rc < 0:
            msg = L.nnhip_last_error() or L.nnhip_multigpu_last_error()   # (the multi-GPU entries' worker threads report through their own channel)
            assert msg, "an error code without a message"
    print("DONE %s" % name, flush=True)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60)

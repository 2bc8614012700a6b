"""numericalnim_amd/csrc/glibc_pow.hpp restates glibc's table-driven pow operation for operation so that the device's
step-size controller factor min(4, max(0.125, 0.9*pow(1/error, 1/order))) (ode.nim:71,537) carries the bits of the C
library the reference links against.  Here: the host build of the same header vs the live libm, exact equality."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "test_glibc_pow")
    flags = ["-O2", "-ffp-contract=off"]
    if "fma" in open("/proc/cpuinfo").read():
        flags.append("-mfma")  # hardware FMA; without it __builtin_fma calls libm's (equally exact) fma()
    subprocess.check_call(["g++", *flags, "-o", exe, os.path.join(ROOT, "tests", "cpp", "test_glibc_pow.cpp"), "-lm"])
    return exe


def test_port_equals_libm_pow_bit_for_bit(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "300000", "20260928"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "mismatch=0" in r.stdout and int(r.stdout.split("total=")[1]) > 1e7


def test_committed_tables_are_this_libms():
    """The tables in glibc_pow_tables.inc are the ones inside the libm of the box the tests run on."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "extract_glibc_pow_tables.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_oracle_controller_uses_libm_pow(oracle):
    """The oracle's controller calls the C library's pow (as Nim's std/math pow does): its factor equals libm's bit for bit."""
    import ctypes as C
    import ctypes.util
    libm = C.CDLL(ctypes.util.find_library("m"))
    libm.pow.restype = C.c_double
    libm.pow.argtypes = [C.c_double, C.c_double]
    rng = np.random.default_rng(5)
    err = 10 ** rng.uniform(-6, 3, 2000)
    for order in (2, 3, 5, 6):
        got = oracle.controller_factor(err, order)
        ref = np.array([min(4.0, max(0.125, 0.9 * libm.pow(1.0 / e, 1.0 / order))) for e in err])
        assert np.array_equal(got, ref)


def test_clamp_early_out_thresholds_are_safe(oracle):
    """shrink_factor (ode_device.hpp) skips pow when error < ClampThresholds<order>::hi (-> 4.0) or > ::lo (-> 0.125).  With the
    C library's pow (oracle.controller_factor) the full expression gives exactly those values on and beyond the thresholds."""
    import re
    src = open(os.path.join(ROOT, "numericalnim_amd", "csrc", "ode_device.hpp")).read()
    thr = {int(o): (float(h), float(l)) for o, h, l in re.findall(r"ClampThresholds<(\d)> \{ static constexpr double hi = ([0-9.e+-]+), lo = ([0-9.e+-]+);", src)}
    assert set(thr) == {2, 3, 5, 6}
    rng = np.random.default_rng(0)
    for order, (hi, lo) in thr.items():
        assert hi <= (0.9 / 4) ** order * (1 - 0.9e-3) and lo >= (0.9 / 0.125) ** order * (1 + 0.9e-3)
        below = np.concatenate([hi * (1 - 10 ** rng.uniform(-16, 0, 200_000)), 10 ** rng.uniform(-300, np.log10(hi), 200_000), [hi, np.nextafter(hi, 0), 5e-324]])
        below = below[(below > 0) & (below <= hi)]
        assert np.all(oracle.controller_factor(below, order) == 4.0)
        above = np.concatenate([lo * (1 + 10 ** rng.uniform(-16, 0, 200_000)), 10 ** rng.uniform(np.log10(lo), 308, 200_000), [lo, np.nextafter(lo, np.inf), np.inf]])
        with np.errstate(divide="ignore"):
            assert np.all(oracle.controller_factor(above[above >= lo], order) == 0.125)

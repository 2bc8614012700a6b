"""The BODIES of the adaptive streaming kernels, executed on the host (tests/cpp/hip_cpu_emu.hpp: every lane of a workgroup a thread, DPP moves and
lane permutes as slot exchanges between the lanes of a system) — TEST INFRASTRUCTURE: nothing of it is in libnnhip_ode.so, which still has no CPU path.

Round 5 had no GPU access; this is how the round's new kernels were run at all.  tests/cpp/emu_advance.cpp drives the whole loop of
nnhip_ode_adaptive_stream_f64_dev — one launch per iteration until no workgroup reports work left — with the lean kernel (advance_lps_lean_kernel /
advance_tpi_lean_kernel) and the general one it replaces side by side, and fails on the first launch after which their states differ.  Here the final
state is compared with the oracle, bit for bit, and the launch count with the oracle's largest step count.  What this does NOT show: the gfx950 code the
device compiler emits — tests/test_gpu_adaptive_parity.py::test_lean_advance_kernels_give_the_general_kernels_bits is the same comparison on the GPU."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path_factory.mktemp("emu") / "emu_advance")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-DNNHIP_CPU_EMU", "-Wno-attributes", "-I", os.path.join(ROOT, "tests", "cpp"),
                           "-I", os.path.join(ROOT, "numericalnim_amd", "csrc"), "-pthread", os.path.join(ROOT, "tests", "cpp", "emu_advance.cpp"), "-o", exe])
    return exe


def _run(exe, case, n, method, kw, t_end):
    r = subprocess.run([exe, case, str(n), str(method)] + [repr(float(kw[k])) for k in ("absTol", "relTol", "dtMin", "dtMax")] + [repr(float(t_end))],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    lines = r.stdout.strip().splitlines()
    launches = int(lines[0].split()[1])
    y0 = np.array([float.fromhex(x) for x in lines[1].split()[1:]])
    y = np.array([float.fromhex(x) for x in lines[2].split()[1:]])
    return launches, y0, y


DEFAULT = dict(absTol=1e-4, relTol=1e-4, dtMin=1e-4, dtMax=1e-2)   # newODEoptions() (ode.nim:78-80): BASELINE's C3 / C4 options
TIGHT = dict(absTol=1e-9, relTol=1e-13, dtMin=1e-8, dtMax=0.2)       # rejections, in-step shrinks through pow, systems that finish at different launches


@pytest.mark.parametrize("case,n,dim", [("ring16", 70, 16), ("ring16", 1, 16), ("ring8", 131, 8), ("ring32", 37, 32), ("linear16", 67, 16)])
@pytest.mark.parametrize("method,name", [(1, "dopri54"), (2, "tsit54")])
@pytest.mark.parametrize("opts,t_end", [(DEFAULT, 0.25), (TIGHT, 0.7)], ids=["default", "tight"])
def test_lanes_per_system_lean_kernel_body(emu, oracle, case, n, dim, method, name, opts, t_end):
    """C4's streamed form: 4 lanes of a wavefront per system (16 components: 4 per lane; 8: 2 per lane), ring neighbours by DPP, ordered register-chain
    norm; batches that do not fill their last workgroup (70 = 64 + 6 systems; 1 system; 131 = 2 x 64 + 3).  ring32: 8 lanes per system (neighbours by
    ds_bpermute, the ordered error sum through LDS); linear16: a right-hand side that is not banded (stage vector through LDS)."""
    O = oracle
    launches, y0, y = _run(emu, case, n, method, opts, t_end)
    kind, par = (O.RHS_LINEAR, [-0.8]) if case == "linear16" else (O.RHS_RING, [0.1])
    ref = O.solve_ode_batch(kind, par, y0.reshape(n, dim), n, dim, [0.0, t_end], O.new_options(**opts), name, layout=O.LAYOUT_AOS)
    assert np.array_equal(y.reshape(n, dim), ref["y"][-1]), (case, name)
    assert launches == int(ref["steps"].max())
    if opts is TIGHT and n > 1:
        assert int(ref["rejected"].sum()) > 0 or int(ref["steps"].max()) > int(ref["steps"].min())  # the batch really exercised the retry loop / finished unevenly


@pytest.mark.parametrize("method,name", [(1, "dopri54"), (2, "tsit54")])
@pytest.mark.parametrize("opts,t_end,n", [(DEFAULT, 0.25, 150), (TIGHT, 0.5, 67), (DEFAULT, 0.05, 1)], ids=["default", "tight", "one"])
def test_thread_per_ivp_lean_kernel_body(emu, oracle, method, name, opts, t_end, n):
    """C3's streamed form: SoA planes, workgroups of 64 (the driver's choice), Lorenz."""
    O = oracle
    launches, y0, y = _run(emu, "lorenz", n, method, opts, t_end)
    ref = O.solve_ode_batch(O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0], y0.reshape(3, n), n, 3, [0.0, t_end], O.new_options(**opts), name)
    assert np.array_equal(y.reshape(3, n), ref["y"][-1]), name
    assert launches == int(ref["steps"].max())

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
# NNHIP_EMU_SANITIZE=1: the kernel bodies run under AddressSanitizer + UBSan (out-of-bounds lane accesses of partly filled workgroups, misaligned vector
# accesses, signed overflow in index arithmetic abort the run) — several times slower, so opt-in; profiles/LAB_NOTES_r05.md records a full pass
SANITIZE = ["-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-DNNHIP_EMU_THREADS"] if os.environ.get("NNHIP_EMU_SANITIZE") else []  # (lanes as OS threads there: ASan does not follow the default engine's swapcontext)


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path_factory.mktemp("emu") / "emu_advance")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-DNNHIP_CPU_EMU", *SANITIZE, "-Wno-attributes", "-I", os.path.join(ROOT, "tests", "cpp"),
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


def _fuzz_cases(n_cases=30, seed=20250929):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n_cases):
        case, dim = [("ring16", 16), ("ring8", 8), ("lorenz", 3), ("ring32", 32)][int(rng.integers(0, 4))]
        out.append(dict(case=case, dim=dim, n=int(rng.integers(1, 160)), method=int(rng.integers(1, 3)),
                        opts=dict(absTol=10.0 ** rng.uniform(-10, -3), relTol=10.0 ** rng.uniform(-12, -3), dtMin=10.0 ** rng.uniform(-9, -4), dtMax=10.0 ** rng.uniform(-2, -0.3)),
                        t_end=float(rng.uniform(0.05, 0.9)), id="%02d_%s" % (k, case)))
    return out


@pytest.mark.parametrize("c", _fuzz_cases(), ids=lambda c: c["id"])
def test_lean_kernel_bodies_fuzz(emu, oracle, c):
    """30 seeded random configurations of the streaming loop (system, batch size 1..159, DOPRI54 / Tsit54, tolerances over seven decades, dtMin, dtMax, span):
    the harness stops at the first launch after which the lean and the general kernel differ; the final states and the launch count equal the oracle's."""
    O = oracle
    name = {1: "dopri54", 2: "tsit54"}[c["method"]]
    launches, y0, y = _run(emu, c["case"], c["n"], c["method"], c["opts"], c["t_end"])
    n, d = c["n"], c["dim"]
    if c["case"] == "lorenz":
        ref = O.solve_ode_batch(O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0], y0.reshape(3, n), n, 3, [0.0, c["t_end"]], O.new_options(**c["opts"]), name)
        assert np.array_equal(y.reshape(3, n), ref["y"][-1]), c
    else:
        ref = O.solve_ode_batch(O.RHS_RING, [0.1], y0.reshape(n, d), n, d, [0.0, c["t_end"]], O.new_options(**c["opts"]), name, layout=O.LAYOUT_AOS)
        assert np.array_equal(y.reshape(n, d), ref["y"][-1]), c
    assert launches == int(ref["steps"].max())


def test_step_kernel_body_equals_the_reference_text(nn, tmp_path):
    """One IntegratorProc call per integrator through the BODY of step_tpi_kernel (what nnhip_ode_step_batch_f64_dev launches), on the host, against the 42
    single steps the reference's own text produced (tests/golden/reference_text_vectors.json: accepted steps, in-step retries through pow, the dtMin double
    hit): yNew, the FSAL slot, dtUsed and error, bit for bit — the GPU twin is tests/test_reference_text_pin.py::test_hip_single_step_matches_the_reference_text."""
    import json
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path / "emu_step")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-DNNHIP_CPU_EMU", *SANITIZE, "-Wno-attributes", "-I", os.path.join(ROOT, "tests", "cpp"),
                           "-I", os.path.join(ROOT, "numericalnim_amd", "csrc"), "-pthread", os.path.join(ROOT, "tests", "cpp", "emu_step.cpp"), "-o", exe])
    steps = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_text_vectors.json")))["steps"]
    assert len(steps) == 42 and len({s["integrator"] for s in steps}) == 14
    lines = []
    for s in steps:
        o = nn.newODEoptions(**s["options"])    # abs() of everything, as newODEoptions does (ode.nim:101-102): host logic of the library, no device needed
        lines.append(" ".join([str(nn.ode.integrator_id(s["integrator"])), s["t"], s["dt"]] + [float(v).hex() for v in (o.absTol, o.relTol, o.dtMax, o.dtMin)] + s["y"] + s["fsal"]))
    r = subprocess.run([exe], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-300:]
    out = r.stdout.strip().splitlines()
    assert len(out) == 2 * len(steps)
    hx = lambda v: float.fromhex(v).hex()  # noqa: E731  (one spelling of a hex float)
    for k, s in enumerate(steps):
        for row in (out[2 * k], out[2 * k + 1]):          # the first and the last of the five lanes that ran the same IVP
            f = row.split()
            assert [hx(v) for v in f[0:3]] == [hx(v) for v in s["yNew"]], (s["integrator"], s["input"])
            assert [hx(v) for v in f[3:6]] == [hx(v) for v in s["fsalOut"]], (s["integrator"], s["input"], "FSAL slot")
            assert hx(f[6]) == hx(s["dtUsed"]) and hx(f[7]) == hx(s["error"]), (s["integrator"], s["input"])


@pytest.fixture(scope="module")
def emu_solve(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path_factory.mktemp("emu_solve") / "emu_solve")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-DNNHIP_CPU_EMU", *SANITIZE, "-Wno-attributes", "-I", os.path.join(ROOT, "tests", "cpp"),
                           "-I", os.path.join(ROOT, "numericalnim_amd", "csrc"), "-pthread", os.path.join(ROOT, "tests", "cpp", "emu_solve.cpp"), "-o", exe])
    return exe


def _golden_cases():
    from golden_util import load_cases
    return load_cases()


@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: c["name"])
def test_fused_solve_kernel_bodies_equal_the_reference_text(nn, emu_solve, case):
    """All 101 fixtures of tests/golden/ode_golden.json (14 integrators; scalar, 2-, 3-, 4- and 16-component systems — the last on the lanes-per-system
    kernel; 2-point and dense tspans on both sides of tStart; rejections, dtMin escapes, the dropped-rows quirk) through the BODIES of the fused solve kernels,
    on the host, with the launch record the library's own planning code builds (solve_plan.hpp): the rows, the row counts and the output times equal what the
    reference's own text returned (reference_text_vectors.json), bit for bit, in both layouts; accepted / rejected step counts equal the oracle's.  The GPU twin,
    through the C ABI: tests/test_gpu_golden.py."""
    import json
    from golden_util import fh
    reftext = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "reference_text_vectors.json")))["cases"]}[case["name"]]
    dimv = max(case["dim"], 1)
    n = len(case["y0"])
    y0 = np.stack([fh(y) for y in case["y0"]])            # [n, dimv]
    o = nn.newODEoptions(**case["options"])               # abs() of everything, as newODEoptions does: host logic of the library
    hexes = lambda xs: " ".join(float(v).hex() for v in xs)  # noqa: E731
    for layout in ((0,) if case["dim"] == 0 else (0, 1)):
        flat = (y0.T if layout == 0 else y0).ravel()
        text = "\n".join([
            " ".join(str(v) for v in (nn.ode.integrator_id(case["integrator"]), case["rhs_kind"], case["dim"], layout, n, len(case["tspan"]), 0, len(case["params"]))),
            hexes([o.dt, o.dtMax, o.dtMin, o.tStart, o.absTol, o.relTol, o.scaleMax, o.scaleMin]),
            hexes(fh(case["params"])), hexes(fh(case["tspan"])), hexes(flat)]) + "\n"
        r = subprocess.run([emu_solve], input=text, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (case["name"], r.returncode, r.stderr[-300:])
        lines = r.stdout.strip().splitlines()
        hx = lambda v: float.fromhex(v).hex()  # noqa: E731
        tt = lines[0].split()
        assert [hx(v) for v in tt[2:]] == reftext["t"] and int(tt[1]) == len(reftext["t"])
        for i, exp in enumerate(case["ivps"]):
            f = lines[1 + i].split()
            ny, steps, rej = int(f[1]), int(f[2]), int(f[3])
            assert ny == exp["n_y"] == reftext["ivps"][i]["n_y"], (case["name"], i)
            assert [hx(v) for v in f[4:]] == reftext["ivps"][i]["y"], (case["name"], i, layout, "differs from the reference's text")
            assert (steps, rej) == (exp["steps"], exp["rejected"]), (case["name"], i)


# ---- the step-streaming kernels (tests/cpp/emu_stream.cpp) ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emu_stream(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path_factory.mktemp("emu_stream") / "emu_stream")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-DNNHIP_CPU_EMU", *SANITIZE, "-Wno-attributes", "-I", os.path.join(ROOT, "tests", "cpp"),
                           "-I", os.path.join(ROOT, "numericalnim_amd", "csrc"), "-pthread", os.path.join(ROOT, "tests", "cpp", "emu_stream.cpp"), "-o", exe])
    return exe


def _rows(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (args, r.returncode, r.stderr[-500:])
    out = {}
    for ln in r.stdout.strip().splitlines():
        f = ln.split()
        out[f[0]] = np.array([float.fromhex(x) for x in f[1:]])
    return out


@pytest.mark.parametrize("vec,mode,inplace,neg,n", [(1, 0, 0, 0, 1024 + 77), (2, 0, 1, 0, 2 * 1024 + 5), (4, 0, 0, 0, 2048 + 300), (2, 1, 0, 0, 1024 + 1), (4, 1, 1, 0, 4096),
                                                    (2, 2, 0, 0, 5 * 1024 + 9), (4, 3, 1, 0, 7 * 2048 + 1), (1, 2, 1, 1, 3 * 512 + 17), (2, 0, 0, 1, 1024), (1, 0, 0, 0, 3)])
def test_headline_kernel_body(emu_stream, oracle, vec, mode, inplace, neg, n):
    """rk4_stream_vec_kernel — the kernel BASELINE's metric is measured on — as nnhip_ode_fixed_stream_f64_dev drives it (one launch per RK4_step, ode.nim:180-189,
    t accumulated on the host): 16-byte lane accesses over full tiles, the scalar ragged tail tile, the persistent form with fewer workgroups than tiles, the
    non-temporal forms, in place and ping-pong, and the backward branch's g(t, y) = -f(-t, y) (ode.nim:544-584).  40 steps of dt = 2^-10 == the oracle's solveODE."""
    O = oracle
    steps, dt = 40, 2.0 ** -10
    d = _rows(emu_stream, "rk4", n, steps, vec, mode, inplace, neg)
    y0 = d["y0"]
    tspan = [-steps * dt, 0.0] if neg else [0.0, steps * dt]
    ref = O.solve_ode_batch(O.RHS_NEG_Y, [], y0, n, 0, tspan, O.new_options(dt=dt), "rk4")
    assert np.array_equal(d["y"], ref["y"][0 if neg else -1, 0]), (vec, mode, inplace, neg)
    assert int(ref["steps"].max()) == steps == int(ref["steps"].min())


_FIXED = ["heun2", "ralston2", "kutta3", "heun3", "ralston3", "ssprk3", "ralston4", "kutta4", "rk4"]


@pytest.mark.parametrize("name", _FIXED)
@pytest.mark.parametrize("aos,per_ivp,n", [(0, 0, 512 + 30), (1, 0, 512 + 1), (0, 1, 512), (1, 1, 7)])
def test_fixed_step_streaming_kernel_body(nn, emu_stream, oracle, name, aos, per_ivp, n):
    """fixed_stream_vec_kernel: any fixed-step IntegratorProc (ode.nim:107-189) over Lorenz, two IVPs per lane with 16-byte accesses (SoA: neighbouring IVPs of a
    component plane; AoS: the six doubles of two neighbouring IVPs), the bounds-checked tail tile, uniform and per-IVP (t, dt); the FSAL slot is yNew (:189).
    12 steps of dt = 2^-8 == the oracle's solveODE, bit for bit.  (SoA planes of an odd number of IVPs are not 16-byte aligned: nnhip_ode_step_batch_f64_dev
    sends those to step_tpi_kernel — the same predicate here; under NNHIP_EMU_SANITIZE=1 UBSan enforces the alignment the kernel relies on.)"""
    O = oracle
    assert aos or n % 2 == 0
    steps, dt = 12, 2.0 ** -8
    d = _rows(emu_stream, "fixed", nn.ode.integrator_id(name), n, steps, aos, per_ivp)
    y0 = d["y0"].reshape(n, 3) if aos else d["y0"].reshape(3, n)
    ref = O.solve_ode_batch(O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0], y0, n, 3, [0.0, steps * dt], O.new_options(dt=dt), name, layout=O.LAYOUT_AOS if aos else O.LAYOUT_SOA)
    got = d["y"].reshape(n, 3) if aos else d["y"].reshape(3, n)
    assert np.array_equal(got, ref["y"][-1]), name
    assert np.array_equal(d["fsal"], d["y"])


@pytest.mark.parametrize("name", ["rk4", "kutta3", "rk21", "bs32", "dopri54", "tsit54", "vern65"])
@pytest.mark.parametrize("n", [16 + 3, 1])
def test_lanes_per_system_step_kernel_body(nn, emu_stream, oracle, name, n):
    """step_lps_kernel: one IntegratorProc call (ode.nim:38) of a 16-component Vector[float] system spread over 16 lanes of a wavefront (ring neighbours by DPP
    row rotation, the error norm as the ordered sum of utils.nim:233-235) — yNew, the FSAL slot, dtUsed and error of every system == the oracle's step."""
    O = oracle
    d = _rows(emu_stream, "steplps", nn.ode.integrator_id(name), n, 0)
    y0, f0 = d["y0"].reshape(n, 16), d["f0"].reshape(n, 16)
    opt = O.new_options(absTol=1e-9, relTol=1e-9, dtMax=1.0, dtMin=1e-6)
    for i in range(n):
        assert np.array_equal(f0[i], O.rhs(O.RHS_RING, [0.1], 0.25, y0[i]))
        yn, fs, dtu, err = O.step(O.RHS_RING, [0.1], name, opt, 0.25, y0[i], f0[i], 2.0 ** -6)
        assert np.array_equal(d["y"].reshape(n, 16)[i], yn), (name, i)
        assert np.array_equal(d["fsal"].reshape(n, 16)[i], fs), (name, i, "FSAL slot")
        assert d["dt"][i] == dtu and d["err"][i] == err, (name, i)


@pytest.mark.parametrize("name", ["rk4", "kutta3"])
def test_lanes_per_system_step_kernel_body_backward(nn, emu_stream, oracle, name):
    """The same call for the backward branch g(t, y) = -f(-t, y) (ode.nim:544-584): one step of a solve towards t < tStart."""
    O = oracle
    n, dt = 5, 2.0 ** -6
    d = _rows(emu_stream, "steplps", nn.ode.integrator_id(name), n, 1)
    y0 = d["y0"].reshape(n, 16)
    ref = O.solve_ode_batch(O.RHS_RING, [0.1], y0, n, 16, [-dt, 0.0], O.new_options(dt=dt), name, layout=O.LAYOUT_AOS)
    assert np.array_equal(d["y"].reshape(n, 16), ref["y"][0]), name


@pytest.mark.parametrize("neg", [0, 1])
def test_dense_rows_kernel_body(emu_stream, oracle, neg):
    """dense_rows_kernel: the rows due at the head of an iteration of ODESolver's loop (ode.nim:512-524) between two states — f at both ends (the backward
    branch's g = -f(-t, .)) and hermiteSpline (utils.nim:273-279) per requested time == the oracle's rhs + hermite_spline, component by component."""
    O = oracle
    n = 256 + 9
    d = _rows(emu_stream, "rows", n, neg)
    ya, yb = d["ya"].reshape(3, n), d["yb"].reshape(3, n)
    tA, tB, treq = 0.125, 0.15625, [0.125, 0.140625, 0.15]
    par = [10.0, 28.0, 8.0 / 3.0]
    for i in list(range(0, n, 37)) + [n - 1]:
        da, db = O.rhs(O.RHS_LORENZ, par, -tA if neg else tA, ya[:, i]), O.rhs(O.RHS_LORENZ, par, -tB if neg else tB, yb[:, i])
        if neg:
            da, db = -da, -db
        for k, tq in enumerate(treq):
            row = d["r%d" % k].reshape(3, n)[:, i]
            exp = [O.hermite_spline(tq, tA, tB, ya[c, i], yb[c, i], da[c], db[c]) for c in range(3)]
            assert list(row) == exp, (i, k)


# ---- the output consumers (tests/cpp/emu_consumers.cpp) against an execution of the reference's text ----------------------------------
@pytest.fixture(scope="module")
def emu_consumers(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path_factory.mktemp("emu_consumers") / "emu_consumers")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-DNNHIP_CPU_EMU", *SANITIZE, "-Wno-attributes", "-I", os.path.join(ROOT, "tests", "cpp"),
                           "-I", os.path.join(ROOT, "numericalnim_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-pthread",
                           os.path.join(ROOT, "tests", "cpp", "emu_consumers.cpp"), "-o", exe])
    return exe


def _quad_vectors():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_text_quad_vectors.json")))


def _hx(xs):
    return " ".join(float(v).hex() for v in np.asarray(xs, dtype=np.float64).ravel())


def _fh(xs):
    return np.array([float.fromhex(x) for x in xs], dtype=np.float64)


def _consume(exe, requests):
    """-> one list of rows (each a float array) per request, plus the header line of `fn` requests"""
    r = subprocess.run([exe], input="\n".join(requests) + "\n", capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, (r.returncode, r.stderr[-500:])
    out, cur = [], []
    for ln in r.stdout.splitlines():
        if ln == "end":
            out.append(cur)
            cur = []
        else:
            cur.append(ln)
    assert len(out) == len(requests)
    return out


def test_function_form_quadrature_kernel_bodies_equal_the_reference_text(emu_consumers):
    """cumtrapz(f, X, ctx, dx) / cumsimpson(f, X, ctx, dx) (integrate.nim:138-175, 377-400): the library's host replay of everything X and dx decide
    (quad_plan.hpp: grid size, the grid interval and Hermite weights of every result row, the non-uniform Simpson tables, where the march may stop) and the BODIES of
    cumtrapz_fn_kernel / cumsimpson_fn_kernel on the host == all 144 function-form cases the reference's own text produced (scalar and 3-component integrands;
    sorted, unsorted and duplicated X; queries on grid points and on the last one), bit for bit, in both layouts, for the first and the last item of a batch
    that does not fill its workgroup.  GPU twin: tests/test_gpu_reference_text_quad.py::test_function_forms_equal_the_reference_text."""
    cases = _quad_vectors()["cumquad_fn"]
    assert len(cases) == 144
    n = 3
    reqs, meta = [], []
    for c in cases:
        d = max(c["dim"], 1)
        p = _fh(c["params"])
        for layout in ((0,) if d == 1 else (0, 1)):
            reqs.append("fn %d %d %d %d %s %s %d %s" % (0 if c["rule"] == "trapz" else 1, d, layout, n, c["dx"], _hx(p[:3]), len(c["X"]), " ".join(c["X"])))
            meta.append((c, d, layout))
    outs = _consume(emu_consumers, reqs)
    for (c, d, layout), o in zip(meta, outs):
        head = o[0].split()
        assert head[1] == "0" and int(head[3]) == c["rows"], c["name"]
        got = np.array([[float.fromhex(v) for v in ln.split()] for ln in o[1:]]).reshape(c["rows"], *((d, n) if layout == 0 else (n, d)))
        want = _fh(c["out"])
        for i in (0, n - 1):
            g = got[:, :, i] if layout == 0 else got[:, i, :]
            assert np.array_equal(g.ravel(), want), (c["name"], layout, i)


def test_discrete_quadrature_kernel_bodies_equal_the_reference_text(emu_consumers):
    """cumtrapz(Y, X) / cumsimpson(Y, X) (integrate.nim:120-135, 329-375): the library's sortAndTrimDataset (dataset_plan.hpp + gather_rows_kernel /
    dup_rows_differ_kernel), cumtrapz_kernel (chunks of 384 interval weights) and cumsimpson_kernel with the library's simpson_tables == the discrete cases of the
    reference's text, fed with the CALLER's X and Y — unsorted, with pure duplicates — exactly as the text was.  GPU twin: ::test_discrete_forms_equal_the_reference_text."""
    cases = _quad_vectors()["cumquad_discrete"]
    reqs, meta = [], []
    for c in cases:
        X = _fh(c["X"])
        Y = np.tile(np.stack([_fh(y) for y in c["Y"]], axis=1), (1, 90))     # [n, 270]: more than one workgroup of series
        for what in ("trapz", "simpson"):
            reqs.append("%s %d %d %s %s" % (what, len(X), Y.shape[1], _hx(X), _hx(Y)))
            meta.append((c, what))
    outs = _consume(emu_consumers, reqs)
    assert len(meta) >= 30
    for (c, what), o in zip(meta, outs):
        if what == "simpson" and isinstance(c["cumsimpson"], dict):        # fewer than 3 distinct abscissae: ValueError (integrate.nim:345-346), refused
            assert len(o) == 1 and o[0].startswith("error too few"), (c["name"], o[:1])
            continue
        got = np.array([[float.fromhex(v) for v in ln.split()] for ln in o])
        want = np.stack([_fh(v) for v in c["cumtrapz" if what == "trapz" else "cumsimpson"]], axis=1)
        assert got.shape == (want.shape[0], 270) and np.array_equal(got, np.tile(want, (1, 90))), (c["name"], what)
    assert {c["name"] for c in cases if not c["strictly_ascending"]} >= {"unsorted", "unsorted_pure_duplicates", "sorted_repeated_maximum", "sorted_repeated_inside", "descending",
                                                                        "negative_zero_pair", "all_but_two_equal"}
    # what the reference refuses (utils.nim:372): the same x with different y — in ANY series of the batch; NaN duplicates are impure too (NaN != NaN); NaN in X
    reqs = []
    for c in _quad_vectors()["impure"]:
        X, y = _fh(c["X"]), _fh(c["Y"])
        clean = np.cos(X)
        Y = np.stack([clean] * 69 + [y] + [clean] * 5, axis=1)            # one impure series among 75, in the second wave of the workgroup
        for what in ("trapz", "simpson", "slopes"):
            assert c[{"trapz": "cumtrapz", "simpson": "cumsimpson", "slopes": "newHermiteSpline"}[what]] == {"raises": "ValueError"}
            reqs.append("%s %d %d %s %s" % (what, len(X), Y.shape[1], _hx(X), _hx(Y)))
    reqs.append("trapz 3 2 %s %s" % (_hx([0.0, float("nan"), 1.0]), _hx(np.ones((3, 2)))))
    outs = _consume(emu_consumers, reqs)
    assert all(len(o) == 1 and o[0].startswith("error impure") for o in outs[:-1]), outs
    assert outs[-1][0].startswith("error X[1] is NaN")


def test_sort_and_trim_plan_agrees_with_the_oracle_on_random_datasets(emu_consumers, oracle):
    """The product's sortAndTrimDataset (dataset_plan.hpp on the host, gather_rows_kernel / dup_rows_differ_kernel bodies) against the oracle's restatement of
    utils.nim:360-413 on 60 random datasets: shuffled abscissae, runs of pure duplicates, -0.0 / 0.0 pairs, sorted inputs with repeated maxima, lengths 3 ... 40 —
    cumtrapz rows (one per distinct abscissa), cumsimpson rows (at the caller's abscissae, hermiteInterpolate's two branches) and the slopes of the sorted knots."""
    rng = np.random.default_rng(2026)
    reqs, meta = [], []
    for case in range(60):
        n = int(rng.integers(3, 41))
        X = np.round(rng.uniform(-2.0, 3.0, n), 1 if case % 3 else 3)          # one decimal: many duplicates
        if case % 5 == 0:
            X = np.sort(X)                                                      # the sorted branch, with whatever repeats the rounding produced
            X[-1] = X[-2] if case % 10 == 0 else X[-1]
        if case % 7 == 0:
            X[rng.integers(0, n)] = 0.0
            X[rng.integers(0, n)] = -0.0
        Y = np.stack([np.cos(X) * 2.0, (0.75 * X - 1.25) * X + 0.5, np.exp(-X) - 0.3 * X], axis=1)     # functions of x: every duplicate is pure
        for what in ("trapz", "simpson", "slopes"):
            reqs.append("%s %d %d %s %s" % (what, n, 3, _hx(X), _hx(Y)))
            meta.append((what, X, Y))
    outs = _consume(emu_consumers, reqs)
    seen = {"refused": 0, "trimmed": 0, "fewer_simpson_rows": 0}
    for (what, X, Y), o in zip(meta, outs):
        xs, _ = oracle.sort_and_trim(X, Y[:, 0])
        seen["trimmed"] += len(xs) < len(X)
        if o and o[0].startswith("error"):
            assert (what == "simpson" and len(xs) < 3) or (what == "slopes" and len(xs) < 2), (what, X, o)
            seen["refused"] += 1
            continue
        got = np.array([[float.fromhex(v) for v in ln.split()] for ln in o])
        for m in range(3):
            if what == "trapz":
                want = oracle.cumtrapz(Y[:, m], X)
            elif what == "simpson":
                want = oracle.cumsimpson(Y[:, m], X)
                seen["fewer_simpson_rows"] += len(want) < len(X)
            else:
                want = oracle.hermite_slopes(X, Y[:, m])
            assert got.shape[0] == len(want) and np.array_equal(got[:, m], want), (what, m, X)
    assert seen["trimmed"] > 60 and seen["fewer_simpson_rows"] > 0, seen


def test_long_cumtrapz_crosses_weight_chunks(emu_consumers, oracle):
    """1000 points: three launches of cumtrapz_kernel (384 weights each), every one resuming from the stored running integral == the oracle."""
    rng = np.random.default_rng(5)
    X = np.cumsum(rng.uniform(0.01, 0.2, 1000))
    Y = rng.standard_normal((1000, 3))
    o = _consume(emu_consumers, ["trapz 1000 3 %s %s" % (_hx(X), _hx(Y))])[0]
    got = np.array([[float.fromhex(v) for v in ln.split()] for ln in o])
    for m in range(3):
        assert np.array_equal(got[:, m], oracle.cumtrapz(Y[:, m], X))


def test_hermite_spline_kernel_bodies_equal_the_reference_text(emu_consumers):
    """newHermiteSpline(X, Y[, dY]) + eval / derivEval with every ExtrapolateKind (interpolate.nim:186-253, 299-390): hermite_slopes_kernel, herm_chunk_fill (the
    library's findInterval / basis weights / extrapolation branch per query) and hermite_interp_kernel == the reference's own text, bit for bit.  GPU twin:
    ::test_hermite_spline_equals_the_reference_text."""
    cases = _quad_vectors()["hermite"]
    EX = {"Constant": 0, "Edge": 1, "Linear": 2, "Native": 3}
    M = 70
    reqs, meta = [], []
    for c in cases:
        X, Y, dY, xq = (_fh(c[k]) for k in ("X", "Y", "dY", "xq"))     # knots in the CALLER's order (two cases unsorted, one of them with pure duplicates)
        Yb = np.tile(Y[:, None], (1, M))
        reqs.append("slopes %d %d %s %s" % (len(X), M, _hx(X), _hx(Yb)))
        meta.append((c, "slopes", None, None, None))
        Xs = _fh(c["X_sorted_trimmed"])                                # newHermiteSpline(X, Y): the slopes belong to the sorted, trimmed knots (interpolate.nim:244-251)
        Ys = Y[[list(X).index(x) for x in Xs]]
        for key, (Xk, Yk, slopes) in (("with_dY", (X, Y, dY)), ("estimated_slopes", (Xs, Ys, _fh(c["slopes_from_text"])))):
            Ykb, dYb = np.tile(Yk[:, None], (1, M)), np.tile(slopes[:, None], (1, M))
            for ex, rec in c[key].items():
                if ex not in EX:
                    continue
                for deriv in (0, 1):
                    reqs.append("eval %d %d %d %d %d %s %s %s %s %s" % (len(Xk), M, len(xq), deriv, EX[ex], c["extrap_value"], _hx(Xk), _hx(Ykb), _hx(dYb), _hx(xq)))
                    meta.append((c, "eval", key, ex, deriv))
    outs = _consume(emu_consumers, reqs)
    n_eval = 0
    for (c, what, key, ex, deriv), o in zip(meta, outs):
        got = np.array([[float.fromhex(v) for v in ln.split()] for ln in o])
        if what == "slopes":
            assert np.array_equal(got, np.tile(_fh(c["slopes_from_text"])[:, None], (1, M))), c["name"]
        else:
            want = _fh(c[key][ex]["derivEval" if deriv else "eval"])
            assert np.array_equal(got, np.tile(want[:, None], (1, M))), (c["name"], key, ex, deriv)
            n_eval += 1
    assert n_eval == len(cases) * 2 * 4 * 2 and len(cases) == 6


# ---- the adaptive streaming driver with dense output (tests/cpp/emu_dense.cpp) ---------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def emu_dense(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    exe = str(tmp_path_factory.mktemp("emu_dense") / "emu_dense")
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-ffp-contract=off", "-DNNHIP_CPU_EMU", *SANITIZE, "-Wno-attributes", "-I", os.path.join(ROOT, "tests", "cpp"),
                           "-I", os.path.join(ROOT, "numericalnim_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-pthread",
                           os.path.join(ROOT, "tests", "cpp", "emu_dense.cpp"), "-o", exe])
    return exe


_ADAPTIVE = ("rk21", "bs32", "dopri54", "tsit54", "vern65")


@pytest.mark.parametrize("case", [c for c in _golden_cases() if c["integrator"].lower() in _ADAPTIVE], ids=lambda c: c["name"])
def test_dense_streaming_kernel_bodies_equal_the_reference_text(nn, emu_dense, case):
    """Every adaptive fixture of tests/golden/ode_golden.json through the step-streaming form of the whole ODESolver (ode.nim:471-586) — state, (t, dt), denseIndex
    and rows in memory, ONE loop iteration per launch (advance_dense_tpi_kernel / advance_dense_lps_kernel), the emission block of the next iteration's head
    (:512-524) inside the launch that produced the step, forward direction first, then advance_dense_finalize_kernel's six modes (final rows, the t0 row, the
    dropped-rows quirk, NaN fill) — on the host: output times, row counts and rows equal what the reference's own text returned, bit for bit, in both layouts.
    The GPU twins go through nnhip_ode_adaptive_stream_dense_f64_dev (tests/test_gpu_golden.py, test_gpu_adaptive_parity.py)."""
    import json
    from golden_util import fh
    reftext = {c["name"]: c for c in json.load(open(os.path.join(ROOT, "tests", "golden", "reference_text_vectors.json")))["cases"]}[case["name"]]
    n = len(case["y0"])
    y0 = np.stack([fh(y) for y in case["y0"]])
    o = nn.newODEoptions(**case["options"])
    hexes = lambda xs: " ".join(float(v).hex() for v in xs)  # noqa: E731
    for layout in ((0,) if case["dim"] == 0 else (0, 1)):
        flat = (y0.T if layout == 0 else y0).ravel()
        text = "\n".join([
            " ".join(str(v) for v in (nn.ode.integrator_id(case["integrator"]), case["rhs_kind"], case["dim"], layout, n, len(case["tspan"]), 0, len(case["params"]))),
            hexes([o.dt, o.dtMax, o.dtMin, o.tStart, o.absTol, o.relTol, o.scaleMax, o.scaleMin]),
            hexes(fh(case["params"])), hexes(fh(case["tspan"])), hexes(flat)]) + "\n"
        r = subprocess.run([emu_dense], input=text, capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, (case["name"], r.returncode, r.stderr[-300:])
        lines = r.stdout.strip().splitlines()
        hx = lambda v: float.fromhex(v).hex()  # noqa: E731
        tt = lines[0].split()
        assert [hx(v) for v in tt[2:]] == reftext["t"] and int(tt[1]) == len(reftext["t"])
        for i, exp in enumerate(case["ivps"]):
            f = lines[1 + i].split()
            assert int(f[1]) == exp["n_y"] == reftext["ivps"][i]["n_y"], (case["name"], i)
            assert [hx(v) for v in f[4:]] == reftext["ivps"][i]["y"], (case["name"], i, layout, "differs from the reference's text")

"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/*.h
declares, and its host-only logic (option normalisation, integrator dispatch, output time grid) matches the
reference semantics (cross-checked with the oracle).  No compute entry is called without a GPU — except to
assert that it FAILS loudly (there is no CPU fallback)."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(nnhip_[a-z0-9_]+)\s*\(", src):
            names.add(m.group(1))
    return names


def test_library_exports_every_declared_symbol(nn):
    from numericalnim_amd import _lib
    lib = C.CDLL(_lib.SO_PATH)
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/nnhip_ode.h but not exported"
    assert declared == set(_lib.SIGNATURES), "python binding table out of sync with the header"
    assert _lib.lib().nnhip_abi_version() == 1


def test_nim_bindings_are_generated_from_the_header():
    """nim/nnhip_ode_bindings.nim (the raw importc procs a numericalnim maintainer links against) is regenerated from
    include/nnhip_ode.h and must be up to date: one proc per declared entry, same argument order."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("gen_nim_bindings", os.path.join(ROOT, "scripts", "gen_nim_bindings.py"))
    g = importlib.util.module_from_spec(spec)
    dont = sys.dont_write_bytecode
    sys.dont_write_bytecode = True  # keep scripts/ free of __pycache__
    try:
        spec.loader.exec_module(g)
    finally:
        sys.dont_write_bytecode = dont
    header = open(os.path.join(ROOT, "include", "nnhip_ode.h")).read()
    text = g.generate(header)
    assert open(os.path.join(ROOT, "nim", "nnhip_ode_bindings.nim")).read() == text, "run python scripts/gen_nim_bindings.py"
    from numericalnim_amd import _lib
    assert {name for _, name, _ in g.prototypes(header)} == set(_lib.SIGNATURES)
    shim = open(os.path.join(ROOT, "nim", "numericalnim_hip.nim")).read()
    assert "import ./nnhip_ode_bindings" in shim and "{.importc, cdecl.}" not in shim  # no hand-written duplicates of the raw procs


def test_no_oracle_or_cpu_fallback_in_product():
    """The product tree must not reference the oracle (parity would be void)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "numericalnim_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f in (), f"{f} mentions the oracle"
    for f in glob.glob(os.path.join(ROOT, "include", "*.h*")) + glob.glob(os.path.join(ROOT, "nim", "*.nim")):
        assert "oracle" not in open(f).read().lower()
    # dev tooling and examples outside tests/ must not use it either; bench.py only in its cpu_baseline leg
    for f in glob.glob(os.path.join(ROOT, "scripts", "*")) + glob.glob(os.path.join(ROOT, "examples", "*")):
        if os.path.isdir(f):
            continue
        assert "import oracle" not in open(f).read() and "from oracle" not in open(f).read(), f
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("from oracle import") == 1 and bench.index("from oracle import") > bench.index("CPU baseline: the oracle")


def test_new_options_semantics(nn):
    o = nn.newODEoptions()  # DEFAULT_ODEoptions (ode.nim:78-79,104)
    assert (o.dt, o.absTol, o.relTol, o.dtMax, o.dtMin, o.scaleMax, o.scaleMin, o.tStart) == (1e-4, 1e-4, 1e-4, 1e-2, 1e-4, 4.0, 0.1, 0.0)
    o = nn.newODEoptions(dt=-1e-3, absTol=-2e-5, relTol=-3e-5, dtMax=-1.0, dtMin=-1e-3, scaleMax=-5.0, scaleMin=-0.5, tStart=-2.0)
    assert (o.dt, o.absTol, o.relTol, o.dtMax, o.dtMin, o.scaleMax, o.scaleMin, o.tStart) == (1e-3, 2e-5, 3e-5, 1.0, 1e-3, 5.0, 0.5, -2.0)
    for kw in (dict(dtMax=1e-5, dtMin=1e-4), dict(scaleMax=0.5), dict(scaleMin=2.0)):
        with pytest.raises(ValueError):
            nn.newODEoptions(**kw)


def test_options_match_oracle(nn, oracle):
    a = nn.newODEoptions(dt=-0.5, relTol=1e-9, tStart=3.0)
    b = oracle.new_options(dt=-0.5, relTol=1e-9, tStart=3.0)
    assert bytes(a) == bytes(b)


def test_integrator_dispatch(nn):
    L = nn._lib.lib()
    # names / (useFSAL, order, adaptive) as solveODE passes them (ode.nim:607-649)
    want = {"dopri54": (1, 5.0, 1), "rk21": (0, 2.0, 1), "bs32": (1, 3.0, 1), "rk4": (0, 4.0, 0), "heun2": (0, 2.0, 0),
            "ralston2": (0, 2.0, 0), "kutta3": (0, 3.0, 0), "heun3": (0, 3.0, 0), "ralston3": (0, 3.0, 0), "ssprk3": (0, 3.0, 0),
            "ralston4": (0, 4.0, 0), "kutta4": (0, 4.0, 0), "vern65": (1, 6.0, 1), "tsit54": (1, 5.0, 1)}
    assert sorted(want) == sorted(nn.allODE)
    for name, (fs, order, ad) in want.items():
        i = L.nnhip_ode_integrator_id(name.upper().encode())  # toLower (ode.nim:607)
        assert i >= 0 and L.nnhip_ode_integrator_name(i).decode() == name
        f, o, a = C.c_int(), C.c_double(), C.c_int()
        assert L.nnhip_ode_integrator_traits(i, C.byref(f), C.byref(o), C.byref(a)) == 0
        assert (f.value, o.value, a.value) == (fs, order, ad)
    assert L.nnhip_ode_integrator_id(b"rk5") == -2
    with pytest.raises(ValueError, match="not a valid integrator"):
        nn.solveODE(nn.Rhs.neg_y(), np.ones(3), [0.0, 1.0], integrator="rk5")


@pytest.mark.parametrize("tspan,tstart", [([0.0, 1.0], 0.0), ([1.0, -1.0, 0.0], 0.0), ([3.0, 1.0, 2.0], 0.0), ([-3.0, -1.0], 0.0),
                                          ([0.5, 1.0, 1.5, 2.5], 1.5), ([0.0, 0.0, 1.0], 0.0), (list(np.linspace(-10, 10, 100)), 0.0),
                                          ([2.0], 0.0), ([], 0.0)])
def test_time_grid_matches_oracle(nn, oracle, tspan, tstart):
    L = nn._lib.lib()
    o = nn.newODEoptions(dt=0.25, tStart=tstart)
    ts = np.asarray(tspan, dtype=np.float64)
    out = np.empty(max(len(ts), 1))
    n = C.c_int()
    assert L.nnhip_ode_time_grid(C.byref(o), ts.ctypes.data_as(C.POINTER(C.c_double)), len(ts), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)) == 0
    if len(ts):
        t_ref, _, _ = oracle.solve_ode(oracle.RHS_NEG_Y, [], 1.0, ts, oracle.new_options(dt=0.25, tStart=tstart), "rk4")
        assert np.array_equal(out[:n.value], t_ref)
    else:
        assert n.value == 0


def test_supported_matrix(nn):
    L = nn._lib.lib()
    R = nn.Rhs
    for integ in (0, 1, 2):
        assert L.nnhip_ode_supported(integ, R.NEG_Y, 1, 0, 0) == 1
        assert L.nnhip_ode_supported(integ, R.LINEAR, 3, 1, 0) == 1
        assert L.nnhip_ode_supported(integ, R.LORENZ, 3, 0, 0) == 1
        assert L.nnhip_ode_supported(integ, R.LORENZ, 2, 0, 0) == 0
        assert L.nnhip_ode_supported(integ, R.LORENZ, 3, 0, 1) == 1
    assert L.nnhip_ode_supported(99, 0, 1, 0, 0) == 0
    # size-generic kinds run at any dim in 1..256 (run-time instantiation); beyond that, and for fixed-size systems, no
    for kind, dim, ok in ((R.NEG_Y, 5, 1), (R.LINEAR, 13, 1), (R.AFFINE_T, 9, 1), (R.RING, 6, 1), (R.RING, 32, 1), (R.NEG_Y, 32, 1),
                          (R.NEG_Y, 17, 1), (R.RING, 64, 1), (R.LINEAR, 256, 1), (R.RING, 257, 0), (R.LORENZ, 5, 0), (R.VANDERPOL, 3, 0)):
        assert L.nnhip_ode_supported(1, kind, dim, 0, 0) == ok and L.nnhip_ode_supported(1, kind, dim, 0, 1) == ok, (kind, dim)


def test_compute_fails_loudly_without_gpu(nn):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(nn.NnhipError, match="(no CPU fallback|hipGetDeviceCount)"):
        nn.solveODE(nn.Rhs.neg_y(), np.ones(4), [0.0, 1.0], nn.newODEoptions(dt=0.25), integrator="rk4")
    with pytest.raises(ValueError):
        nn.solveODE(nn.Rhs.neg_y(), torch.ones(4, dtype=torch.float64), [0.0, 1.0], integrator="rk4")  # CPU tensor refused
    # the quadrature entries do their host-side planning first (row count is known without a device), then fail on the device part
    import ctypes as C
    from numericalnim_amd import _lib
    X = np.array([0.0, 0.5, 1.0])
    p = np.array([0.0, 1.0])
    rows = C.c_int(-1)
    dp = C.POINTER(C.c_double)
    for entry in (_lib.lib().nnhip_cumtrapz_fn_batch_f64_dev, _lib.lib().nnhip_cumsimpson_fn_batch_f64_dev):
        rc = entry(nn.Rhs.AFFINE_T, p.ctypes.data_as(dp), 2, None, 0, 4, 1, 0, X.ctypes.data_as(dp), 3, 0.01, C.c_void_p(16), C.byref(rows), None)
        assert rc == _lib.NNHIP_EHIP and rows.value == 3, (rc, rows.value, _lib.last_error())


def test_argument_validation(nn):
    with pytest.raises(KeyError):
        nn.solveODE(nn.Rhs.linear(), np.ones(3), [0.0, 1.0], integrator="rk4")      # ctx.fValues lacks "a"
    ctx = nn.newNumContext()
    ctx.setF("a", -0.1)
    assert ctx.getF("a") == -0.1 and nn.Rhs.linear().params(ctx) == [-0.1]
    ctx["k"] = 3
    assert ctx["k"] == 3


def test_per_call_wrappers_validate_before_touching_the_device(nn):
    """The host forms of the per-call solves (one ODEoptions object / one tspan per IVP) check their shapes on the host; with valid
    arguments and no GPU they fail loudly like every compute entry (no CPU fallback)."""
    import torch
    y0 = np.ones((3, 5))
    opts = [nn.newODEoptions() for _ in range(5)]
    with pytest.raises(ValueError):
        nn.solveODECalls(nn.Rhs.lorenz(), y0, np.ones(4), opts)            # t_end: one value per IVP
    with pytest.raises(ValueError):
        nn.solveODECalls(nn.Rhs.lorenz(), y0, np.ones(5), opts[:3])        # options: one object, or one per IVP
    with pytest.raises(ValueError):
        nn.solveODECallsTspan(nn.Rhs.lorenz(), y0, np.ones((4, 3)), opts)  # tspans: one row per IVP
    with pytest.raises(ValueError):
        nn.solveODECallsTspan(nn.Rhs.lorenz(), y0, np.ones((5, 3)), opts, sweep=np.ones((2, 4)))
    L = nn._lib.lib()
    assert L.nnhip_ode_solve_tspans_workspace_bytes(1000, 7) >= 1000 * 7 * 8 + 1000 * 12
    assert L.nnhip_tune_set(b"adv_block", 96) == -1 and L.nnhip_tune_set(b"adv_block", 0) == 0
    if not torch.cuda.is_available():
        with pytest.raises(nn.NnhipError, match="(no CPU fallback|hipGetDeviceCount|HIP)"):
            nn.solveODECalls(nn.Rhs.lorenz(), y0, np.ones(5), opts)
        with pytest.raises(nn.NnhipError, match="(no CPU fallback|hipGetDeviceCount|HIP)"):
            nn.solveODECallsTspan(nn.Rhs.lorenz(), y0, np.ones((5, 3)), opts)


def test_non_finite_times_are_refused(nn):
    """inf / NaN in tspan or tStart would make the reference loop forever; the ABI refuses them before touching the device."""
    for ts in ([0.0, float("inf")], [float("nan"), 1.0]):
        with pytest.raises(ValueError, match="not finite"):
            nn.solveODE(nn.Rhs.neg_y(), np.ones(3), ts, nn.newODEoptions(dt=0.1), integrator="rk4")
    with pytest.raises(ValueError, match="not finite"):
        nn.solveODE(nn.Rhs.neg_y(), np.ones(3), [0.0, 1.0], nn.newODEoptions(dt=0.1, tStart=float("inf")), integrator="rk4")


def test_header_is_plain_c_and_host_entries_work_from_c(nn, tmp_path):
    """include/nnhip_ode.h compiles as C99 (-pedantic) and the host-only entries behave from a C caller."""
    import subprocess
    from numericalnim_amd import _lib
    exe = str(tmp_path / "abi_is_c")
    libdir = os.path.dirname(_lib.SO_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "abi_is_c.c"), "-L", libdir, "-lnnhip_ode", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "abi 1 ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_tune_set_validation(nn):
    L = nn._lib.lib()
    assert L.nnhip_tune_set(b"no_such_knob", 1) == -1
    for key, bad in ((b"rk4_stream_vec", 3), (b"rk4_stream_mode", 7), (b"rk4_stream_blocks_per_cu", 0), (b"dim16_variant", 9), (b"host_chunks", 100), (b"stream_graph", 3),
                     (b"adv_recompute_fsal", 2), (b"adv_recompute_fsal", -2), (b"adv_lean", 2), (b"adv_steps_per_launch", 0), (b"adv_block", 32)):
        assert L.nnhip_tune_set(key, bad) == -1, key
    for key, good, reset in ((b"rk4_stream_vec", 2, None), (b"rk4_stream_mode", 1, None), (b"rk4_stream_auto", 1, 1), (b"stream_graph", 1, 2),
                             (b"dim16_variant", 1, 0), (b"fp_contract", 1, 0), (b"host_chunks", 4, 0), (b"host_register", 1, 0),
                             (b"adv_recompute_fsal", 0, -1), (b"adv_recompute_fsal", 1, -1), (b"adv_steps_per_launch", 5, 1), (b"adv_block", 128, 0), (b"adv_nontemporal", 1, -1), (b"adv_lean", 1, 0)):
        assert L.nnhip_tune_set(key, good) == 0, key
        if reset is not None:
            assert L.nnhip_tune_set(key, reset) == 0
    assert L.nnhip_tune_set(b"rk4_stream_auto", 1) == 0  # back to automatic variant selection
    # what the library itself reports afterwards: the recorded configuration (this test once "reset" adv_lean to 1 — round 5's default — for the rest of the process)
    for key, default in (("stream_graph", 2), ("dim16_variant", 0), ("fp_contract", 0), ("host_chunks", 0), ("host_register", 0), ("adv_recompute_fsal", -1),
                         ("adv_steps_per_launch", 1), ("adv_block", 0), ("adv_nontemporal", -1), ("adv_lean", 0), ("adv_auto_poll", 0), ("rk4_stream_auto", 1)):
        assert nn.tuneGet(key) == default, key


def test_rtc_compiler_choice(nn):
    """nnhip_rtc_compiler(): which libhiprtc builds right-hand sides given as source.  In a process that carries its own copy (PyTorch's
    wheels do) the ROCm the library was built with is loaded into a link namespace of its own, unless NNHIP_HIPRTC=process; both
    compile the same source to kernels with the same results (the GPU suite runs on the default)."""
    import subprocess
    import sys
    who = nn._lib.lib().nnhip_rtc_compiler().decode()
    assert "libhiprtc" in who
    code = ("import torch, numericalnim_amd as nn; L = nn._lib.lib(); print(L.nnhip_rtc_compiler().decode()); "
            "f = nn.Rhs.custom(2, 'dy[0] = y[1]; dy[1] = -p[0] * y[0];', keys=('k',), defaults={'k': 2.0}, name='osc_rtc_choice'); print('kind', f.kind)")
    for env_val, expect in (("process", "the process's libhiprtc"), (None, "libhiprtc")):
        env = dict(os.environ)
        env.pop("NNHIP_HIPRTC", None)
        if env_val:
            env["NNHIP_HIPRTC"] = env_val
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and expect in r.stdout and "kind" in r.stdout, (env_val, r.stdout[-400:], r.stderr[-400:])


def test_the_profiled_headline_variant_is_the_one_the_tuner_picks(nn):
    """profiles/pmc_traffic.json's counters belong to one instantiation of rk4_stream_vec_kernel; bench.py attaches them to the line only
    if nnhip_ode_rk4_stream_variant names the same (VEC, MODE) for the same batch — for both committed regimes it must."""
    import ctypes as C
    import json
    import re
    L = nn._lib.lib()
    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for key in ("rk4_stream", "rk4_stream_beyond_infinity_cache"):
        vec, mode = C.c_int(-1), C.c_int(-1)
        assert L.nnhip_ode_rk4_stream_variant(int(pm[key]["ivps_per_launch"]), 0, C.byref(vec), C.byref(mode)) == 0
        m = re.search(r"rk4_stream_vec_kernel<.*?, (?:false|true), (\d+), (\d+)>", pm[key]["kernel"])
        assert (int(m.group(1)), int(m.group(2))) == (vec.value, mode.value), (key, pm[key]["kernel"], vec.value, mode.value)
    assert L.nnhip_ode_rk4_stream_variant(-1, 0, C.byref(vec), C.byref(mode)) == -1
    # an explicit knob pins the variant; "rk4_stream_auto" hands the choice back
    try:
        assert L.nnhip_tune_set(b"rk4_stream_vec", 8) == 0
        assert L.nnhip_ode_rk4_stream_variant(10, 0, C.byref(vec), C.byref(mode)) == 0 and vec.value == 8
    finally:
        assert L.nnhip_tune_set(b"rk4_stream_vec", 4) == 0 and L.nnhip_tune_set(b"rk4_stream_auto", 1) == 0
    assert L.nnhip_ode_rk4_stream_variant(10, 0, C.byref(vec), C.byref(mode)) == 0 and (vec.value, mode.value) == (1, 0)


def test_no_emulation_symbol_in_the_product_library():
    """NNHIP_CPU_EMU (tests/cpp/hip_cpu_emu.hpp) is a compile-time hook of the TEST harnesses: the product build never defines it, and nothing of the host emulation
    is in libnnhip_ode.so — no `hipemu` symbol or string, no NNHIP_CPU_EMU in the library's Makefile or in anything the package imports."""
    import re
    so = os.path.join(ROOT, "numericalnim_amd", "csrc", "libnnhip_ode.so")
    if not os.path.exists(so):
        pytest.skip("libnnhip_ode.so is not built")
    import shutil
    import subprocess
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    syms = subprocess.check_output([nm, "-a", so], text=True) + subprocess.check_output([nm, "-D", so], text=True)
    assert "hipemu" not in syms and "hip_cpu_emu" not in syms        # (the #ifdef lines themselves travel as TEXT inside the library: the device headers embedded for hiprtc)
    mk = open(os.path.join(ROOT, "numericalnim_amd", "csrc", "Makefile")).read()
    assert "NNHIP_CPU_EMU" not in mk
    for dirpath, _d, files in os.walk(os.path.join(ROOT, "numericalnim_amd")):
        for f in files:
            if f.endswith((".py", ".hip")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"NNHIP_CPU_EMU|hip_cpu_emu", txt), os.path.join(dirpath, f)   # (.hpp device headers carry the #ifdef; nothing selects it)


def test_tuning_blocks_nest_and_restore_what_was_set_before():
    """nn.tuning(...) (the Python side of nnhip_tune_set / nnhip_tune_get): an inner block hands the knob back as the OUTER block set it, a block hands back what a
    direct nnhip_tune_set call had set — what the LIBRARY reports, not a default table; a refused value undoes the block's earlier knobs."""
    import numericalnim_amd as nn
    L = nn._lib.lib()
    g = nn.tuneGet
    assert (g("adv_auto_poll"), g("adv_lean"), g("fp_contract"), g("stream_graph"), g("adv_steps_per_launch")) == (0, 0, 0, 2, 1)   # the recorded configuration
    with nn.tuning(adv_auto_poll=1, adv_lean=1):
        assert g("adv_auto_poll") == 1 and g("adv_lean") == 1
        with nn.tuning(adv_auto_poll=0):
            assert g("adv_auto_poll") == 0 and g("adv_lean") == 1
        assert g("adv_auto_poll") == 1
    assert g("adv_auto_poll") == 0 and g("adv_lean") == 0
    assert L.nnhip_tune_set(b"calls_bin", 0) == 0            # set behind the context manager's back ...
    with nn.tuning(calls_bin=1, sort_min_spread_permille=125):
        assert g("calls_bin") == 1 and g("sort_min_spread_permille") == 125
    assert g("calls_bin") == 0 and g("sort_min_spread_permille") == 50   # ... and handed back as it was
    assert L.nnhip_tune_set(b"calls_bin", 1) == 0
    with pytest.raises(ValueError, match="unknown tuning key"):
        with nn.tuning(adv_lean=1, no_such_knob=1):
            pass
    assert g("adv_lean") == 0                                # the knob set before the refused one came back
    with pytest.raises(ValueError, match="adv_block must be"):
        with nn.tuning(adv_lean=1, adv_block=96):
            pass
    assert g("adv_lean") == 0 and g("adv_block") == 0
    # the headline kernel's variant: pinning (vec, mode) switches the automatic choice off; the block restores all three
    with nn.tuning(rk4_stream_auto=0, rk4_stream_vec=8, rk4_stream_mode=1):
        assert (g("rk4_stream_auto"), g("rk4_stream_vec"), g("rk4_stream_mode")) == (0, 8, 1)
    assert g("rk4_stream_auto") == 1
    v = C.c_int(7)
    assert L.nnhip_tune_get(None, C.byref(v)) == -1 and L.nnhip_tune_get(b"adv_lean", None) == -1 and L.nnhip_tune_get(b"nope", C.byref(v)) == -1 and v.value == 7


def test_hostile_arguments_get_an_error_code_before_anything_is_touched(nn):
    """Findings of the argument fuzz under AddressSanitizer (scripts/tsan_host_audit.sh SAN=address, tests/fake_hip_abi_arg_fuzz.py): a NULL where data is
    needed, or sizes no device could hold, are refused with NNHIP_EVALUE and a message BEFORE a byte is allocated, copied or dereferenced — these return
    without a HIP call, so they are checkable here."""
    _lib = nn._lib
    L = _lib.lib()
    ev = _lib.NNHIP_EVALUE
    dp = C.POINTER(C.c_double)
    opt = nn.newODEoptions()
    ts = np.array([0.0, 1.0])
    one = np.ones(8)
    p = lambda a: a.ctypes.data_as(dp)  # noqa: E731
    # the right-hand side alone: rhs_params NULL with n_params > 0 was dereferenced; y / dy NULL went to the launch; the layout was not checked
    assert L.nnhip_ode_rhs_batch_f64_dev(nn.Rhs.LORENZ, None, 3, 4, 3, 0, 0.0, C.c_void_p(16), C.c_void_p(16), None) == ev and b"rhs_params" in L.nnhip_last_error()
    assert L.nnhip_ode_rhs_batch_f64_dev(nn.Rhs.NEG_Y, None, 0, 4, 1, 0, 0.0, None, C.c_void_p(16), None) == ev
    assert L.nnhip_ode_rhs_batch_f64_dev(nn.Rhs.NEG_Y, None, 0, 4, 1, 7, 0.0, C.c_void_p(16), C.c_void_p(16), None) == ev and b"layout" in L.nnhip_last_error()
    # host solves: y0 / y_out NULL with N > 0 (the copies would have started from address 0)
    assert L.nnhip_ode_solve_batch_f64(C.byref(opt), 0, nn.Rhs.NEG_Y, None, 0, None, 4, 1, 0, p(ts), 2, p(one), one.ctypes.data, None, None, None, 0, None, 0) == ev
    assert b"y0" in L.nnhip_last_error()
    assert L.nnhip_ode_solve_batch_f64(C.byref(opt), 0, nn.Rhs.NEG_Y, None, 0, one.ctypes.data, 4, 1, 0, p(ts), 2, p(one), None, None, None, None, 0, None, 0) == ev
    assert L.nnhip_ode_solve_batch_sorted_f64(C.byref(opt), 1, nn.Rhs.LORENZ, None, 0, None, 0, None, 1 << 40, 2 ** 31 - 1, 0, p(ts), 2, p(one), None, None, None, None, 0,
                                              None, 8, 0) == ev
    assert L.nnhip_ode_solve_batch_sorted_f64(C.byref(opt), 1, nn.Rhs.LORENZ, None, 0, None, 0, None, 5, 3, 0, p(ts), 2, p(one), one.ctypes.data, None, None, None, 0,
                                              None, 8, 0) == ev
    # the discrete consumers' host forms: X / Y / out NULL used to be noticed after 2 x n x M x 8 bytes had been allocated on the device
    assert L.nnhip_cumtrapz_batch_f64(None, 2 ** 31 - 1, None, 1, None, 0) == ev and L.nnhip_cumsimpson_batch_f64(p(one), 8, None, 5, p(one), 0) == ev
    assert L.nnhip_hermite_spline_eval_batch_f64(p(one), 8, None, None, 3, p(one), 2, 0, 0, 0.0, p(one), 0) == ev
    rows = C.c_int(-1)
    assert L.nnhip_cumtrapz_fn_batch_f64(nn.Rhs.AFFINE_T, None, 2, None, 0, 4, 1, 0, p(one), 3, 0.01, p(one), C.byref(rows), 0) == ev
    assert L.nnhip_cumsimpson_fn_batch_f64(nn.Rhs.AFFINE_T, p(one), 2, None, 0, 1 << 40, 2 ** 31 - 1, 0, p(one), 100000, 0.01, None, C.byref(rows), 0) == ev
    # workspace sizes of batches no device could hold: 0 ("invalid", like N < 0), not an overflowed product
    assert L.nnhip_ode_adaptive_stream_workspace_bytes(1 << 40, 2 ** 31 - 1) == 0 and L.nnhip_ode_adaptive_stream_dense_workspace_bytes(1 << 40, 2 ** 31 - 1, 1) == 0
    assert L.nnhip_ode_fixed_stream_dense_workspace_bytes(1 << 40, 2 ** 31 - 1) == 0 and L.nnhip_ode_solve_tspans_workspace_bytes(1 << 40, 2 ** 31 - 1) == 0
    assert L.nnhip_ode_adaptive_stream_workspace_bytes(10 ** 6, 16) == 8 * (16 * 10 ** 6 + 3 * 10 ** 6) + 4 * 64 or L.nnhip_ode_adaptive_stream_workspace_bytes(10 ** 6, 16) > 8 * 19 * 10 ** 6
    # the RCCL reassembly: its refusals carry a message on the multi-GPU entries' own channel
    assert L.nnhip_allgather_states_f64_dev(2, None, None, 3, 0, None, None) == ev and b"allgather_states" in L.nnhip_multigpu_last_error()


def test_per_step_seams_refuse_tensors_they_would_misread(nn):
    """The per-step seams take device pointers: a CPU tensor, a float32 tensor or a numpy array handed to them would be MISREAD by the kernels (8 bytes per value
    from whatever address), not converted.  The Python mirror refuses them before the library is called (checkable without a GPU: the refusal comes first)."""
    import torch
    f = nn.Rhs.neg_y()
    for bad in (torch.ones(8, dtype=torch.float64), torch.ones(8, dtype=torch.float32), np.ones(8)):
        for call in (lambda y: nn.fixedStream(f, y, 0.0, 1.0), lambda y: nn.adaptiveStream(f, y, 0.0, 1.0), lambda y: nn.integratorStep(f, 0.0, y, None, 0.1),
                     lambda y: nn.fixedStreamSolve(f, y, [0.0, 1.0]), lambda y: nn.adaptiveStreamSolve(f, y, [0.0, 1.0]),
                     lambda y: nn.solveODEPerIvpEnd(f, y, y), lambda y: nn.solveODEPerIvpTspan(f, y, y)):
            with pytest.raises(ValueError, match="float64 tensor on a CUDA/HIP device"):
                call(bad)


def test_consumers_refuse_device_series_they_would_misread(nn):
    """The same for the consumers of the path (cumtrapz / cumsimpson / newHermiteSpline / sortAndTrimDataset over device series): a float32 or CPU tensor is refused,
    numpy series keep going to the host-pointer entries (which fail loudly without a GPU: no CPU fallback)."""
    import torch
    from numericalnim_amd import interpolate as ni
    X = np.linspace(0.0, 1.0, 5)
    for bad in (torch.ones((5, 4), dtype=torch.float64), torch.ones((5, 4), dtype=torch.float32)):
        for call in (lambda Y: ni.cumtrapz(Y, X), lambda Y: ni.cumsimpson(Y, X), lambda Y: ni.HermiteSpline(X, Y), lambda Y: ni.sortAndTrimDataset(X[::-1].copy(), Y),
                     lambda Y: ni.rhsBatch(nn.Rhs.neg_y(), 0.0, Y[0])):
            with pytest.raises(ValueError, match="float64 tensor on a CUDA/HIP device"):
                call(bad)

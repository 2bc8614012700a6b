"""The COMPILED gfx950 code of the hot kernels — the instructions hipcc put into the library's own object files (numericalnim_amd/csrc/*.o) — executed on the host
by tools/gfx950_isa_interp.py (64 lanes as numpy vectors under EXEC, SGPRs / VCC / SCC, DPP moves, LDS, s_barrier, one IEEE operation per instruction) and compared
bit for bit with the oracle.  tests/test_kernel_bodies_on_cpu.py runs the kernel SOURCE on the host and says nothing about what the device compiler made of it;
this closes that gap as far as a host can (no timing, no memory model): register allocation, EXEC-mask handling of the nested accept / reject branches, DPP
controls, the expansions of FP64 division and square root, 32-bit lane offsets against SGPR bases — all as emitted.

It also yields DYNAMIC instruction counts per wavefront — what SQ_INSTS_VALU / SQ_WAVES measures on hardware.  Cross-check against the one such figure this
repository holds from a GPU: round 4's PMC set of the general streamed C4 kernel, 587 VALU per wave (profiles/r04_c4_stream_pmc/): the interpreter counts 591 on
the same kernel.  The lean kernels of round 5, which have never met a GPU: 501 (streamed C4, was 591), 348 (streamed C3, was 397).

TEST INFRASTRUCTURE: nothing in the library or the package imports the interpreter; the product has no CPU path."""
import json
import math
import os
import shutil
import struct
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "numericalnim_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))

DEFAULT = dict(absTol=1e-4, relTol=1e-4, dtMin=1e-4, dtMax=1e-2)     # newODEoptions(): BASELINE's C3 / C4 options
TIGHT = dict(absTol=1e-9, relTol=1e-13, dtMin=1e-8, dtMax=0.2)         # rejections, in-step shrinks through pow, uneven finishing


@pytest.fixture(scope="module")
def G():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or shutil.which("g++") is None:
        pytest.skip("needs the ROCm LLVM tools and g++")
    import gfx950_isa_interp
    return gfx950_isa_interp


@pytest.fixture(scope="module")
def helpers(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa_helpers")
    out = {}
    for name in ("struct_layout", "kernarg_solve"):
        exe = str(d / name)
        subprocess.check_call(["g++", "-std=c++20", "-O1", "-DNNHIP_CPU_EMU", "-Wno-attributes", "-Wno-invalid-offsetof", "-I", os.path.join(ROOT, "tests", "cpp"), "-I", CSRC,
                               "-I", os.path.join(ROOT, "include"), "-pthread", os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe])
        out[name] = exe
    out["layout"] = json.loads(subprocess.check_output([out["struct_layout"]]))
    return out


_CO = {}


def _code_object(G, obj):
    p = os.path.join(CSRC, obj)
    if not os.path.exists(p):
        pytest.skip(obj + " is not built (python -c 'import __graft_entry__ as g; g.build()')")
    if obj not in _CO:
        _CO[obj] = G.CodeObject(p)
    return _CO[obj]


def _valu(c):
    return c["valu_f64"] + c["valu_other"] + c["valu_dpp"] + c["valu_lane"]


def _stream_loop(G, helpers, co, kernel_rx, lean, y, layout_aos, dim, par, opts, t_end, block, per_block):
    """nnhip_ode_adaptive_stream_f64_dev's loop over the interpreted kernel: t = 0, dt = sqrt(dtMax dtMin), one launch per iteration until no workgroup reports
    work left.  y is advanced in place.  -> (launches, per-wave counters of the second launch — an unpolled one; ["polled_launch"]: those of the third, a polled one)"""
    L = helpers["layout"]
    n = y.shape[0] if layout_aos else y.shape[1]
    mem = G.Memory(co)
    td = np.zeros((n, 2))
    td[:, 1] = math.sqrt(opts["dtMax"] * opts["dtMin"])
    fsal = np.zeros_like(y)
    active = np.zeros(L["kAggSlots"], dtype=np.uint32)
    ay, atd, afs, aact = mem.alloc(y), mem.alloc(td), mem.alloc(fsal), mem.alloc(active)
    k = co.kernel(kernel_rx)
    if lean:
        A = L["AdvLeanArgs"]
        ka = bytearray(A["sizeof"])
        struct.pack_into("<QQqd", ka, 0, ay, atd, n, t_end)
        struct.pack_into("<Q", ka, A["active"], aact)
    else:
        A = L["StepArgs"]
        ka = bytearray(A["sizeof"])
        struct.pack_into("<qqq", ka, 0, n, dim if layout_aos else 1, 1 if layout_aos else n)
        for f, v in (("y_in", ay), ("y_out", ay), ("fsal_in", afs), ("fsal_out", afs), ("t_io", atd), ("active", aact)):
            struct.pack_into("<Q", ka, A[f], v)
        struct.pack_into("<d", ka, A["tEnd"], t_end)
        struct.pack_into("<i", ka, A["stepsPerLaunch"], 1)
        struct.pack_into("<i", ka, A["recomputeFsal"], 1)
        struct.pack_into("<q", ka, A["perIvpStride"], n)
    struct.pack_into("<dddd", ka, A["ctl"], opts["absTol"], opts["relTol"], opts["dtMax"], opts["dtMin"])
    struct.pack_into("<%dd" % len(par), ka, A["P"], *par)
    M = G.Machine(co)
    grid = (n + per_block - 1) // per_block
    launches, second, third = 0, None, None
    while True:
        if launches == 1:  # the counted launch is an UNPOLLED one (active = nullptr), like 100 of BASELINE's 104: nobody finishes in the second iteration
            kb = bytearray(ka)
            struct.pack_into("<Q", kb, A["active"], 0)
            second = M.launch(k, (grid,), (block,), bytes(kb), mem)
            launches += 1
            continue
        st = M.launch(k, (grid,), (block,), bytes(ka), mem)
        launches += 1
        if launches == 3:
            third = st  # a POLLED launch: the workgroup's "anyone left?" reduction and store included
        if not active.any():
            second[0]["polled_launch"] = dict(third[0]) if third else None
            return launches, second
        active[:] = 0
        assert launches < 5000


def _ring_y0(n, d):
    return (1.0 + np.arange(d)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]).copy()


def _lorenz_y0(n):
    return np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)]).copy()


RESULTS = {}


@pytest.mark.parametrize("opts,t_end,n", [(DEFAULT, 0.25, 70), (TIGHT, 0.4, 9)], ids=["default", "tight"])
def test_streamed_c4_kernels_as_compiled(G, helpers, oracle, opts, t_end, n):
    """advance_lps_lean_kernel<Tsit54, RhsRing<16>, 4> and the general advance_lps_kernel it stands in for, from the library's ode_tu_m_tsit54.o: 16-component ring,
    four lanes per system (ring neighbours and the ordered norm chain by DPP quad_perm), a batch that leaves its last workgroup partly filled; default options
    (first attempt accepted, clamp early-out) and tight ones (rejections, the pow path, systems finishing at different launches)."""
    O = oracle
    co = _code_object(G, "ode_tu_m_tsit54.o")
    ref = None
    got = {}
    for lean, rx in ((True, r"advance_lps_lean_kernelILi2ENS_7RhsRingILi16EEELi4EE"), (False, r"advance_lps_kernelILi2ENS_7RhsRingILi16EEELi4ELb0EE")):
        y = _ring_y0(n, 16)
        y0 = y.copy()
        launches, second = _stream_loop(G, helpers, co, rx, lean, y, True, 16, [0.1], opts, t_end, 256, 64)
        if ref is None:
            ref = O.solve_ode_batch(O.RHS_RING, [0.1], y0, n, 16, [0.0, t_end], O.new_options(**opts), "tsit54", layout=O.LAYOUT_AOS)
        assert np.array_equal(y, ref["y"][-1]), ("lean" if lean else "general", "differs from the oracle")
        assert launches == int(ref["steps"].max())
        got[lean] = second[0]
    if opts is TIGHT:
        assert int(ref["steps"].max()) >= 10       # (the ring is too benign to reject a step; the controller's pow runs at every one of these steps — Lorenz below rejects)
        return
    polled = {k: got[k].pop("polled_launch") for k in got}
    lean_valu, gen_valu = _valu(got[True]), _valu(got[False])
    RESULTS["streamed_c4"] = {"lean": dict(got[True]), "general": dict(got[False]), "valu_per_wave": {"lean": lean_valu, "general": gen_valu},
                              "valu_per_wave_polled_launch": {"lean": _valu(polled[True]), "general": _valu(polled[False])}}
    assert got[True]["valu_f64"] == got[False]["valu_f64"]          # the same arithmetic ...
    gen_polled = _valu(polled[False])
    assert abs(gen_polled - 587) <= 0.03 * 587, gen_polled          # ... the interpreter's count of the general kernel's polled launch = round 4's PMC figure (587 per wave, p75) within 3 %
    assert lean_valu <= gen_valu - 50, (lean_valu, gen_valu)        # ... and the lean kernel issues at least 50 fewer VALU instructions per wave (unpolled launches, round 6: 481 vs 565)


REJECTING = dict(absTol=1e-6, relTol=1e-6, dtMin=1e-9, dtMax=1.0)   # Lorenz from spread-out initial states: 170 rejected attempts, IVPs finish 28 .. 35 launches in


@pytest.mark.parametrize("opts,t_end,n", [(DEFAULT, 0.25, 70), (REJECTING, 0.4, 70)], ids=["default", "rejecting"])
def test_streamed_c3_kernels_as_compiled(G, helpers, oracle, opts, t_end, n):
    """advance_tpi_lean_kernel<DOPRI54, RhsLorenz> and the general advance_tpi_kernel, from ode_tu_m_dopri54.o: SoA planes, one-wave workgroups (the driver's choice),
    a partly filled second workgroup."""
    O = oracle
    co = _code_object(G, "ode_tu_m_dopri54.o")
    par = [10.0, 28.0, 8.0 / 3.0]
    ref, got = None, {}
    for lean, rx in ((True, r"advance_tpi_lean_kernelILi1ENS_9RhsLorenzEEE"), (False, r"advance_tpi_kernelILi1ENS_9RhsLorenzELb0ELb0EE")):
        y = _lorenz_y0(n)
        if opts is REJECTING:
            y[0] *= np.linspace(1.0, 30.0, n)
        y0 = y.copy()
        launches, second = _stream_loop(G, helpers, co, rx, lean, y, False, 3, par, opts, t_end, 64, 64)
        if ref is None:
            ref = O.solve_ode_batch(O.RHS_LORENZ, par, y0, n, 3, [0.0, t_end], O.new_options(**opts), "dopri54")
        assert np.array_equal(y, ref["y"][-1]), ("lean" if lean else "general", "differs from the oracle")
        assert launches == int(ref["steps"].max())
        got[lean] = second[0]
    if opts is REJECTING:
        assert int(ref["rejected"].sum()) > 0 and int(ref["steps"].max()) > int(ref["steps"].min())   # the retry loop ran, IVPs finished at different launches
    polled = {k: got[k].pop("polled_launch") for k in got}
    if opts is DEFAULT:
        RESULTS["streamed_c3"] = {"lean": dict(got[True]), "general": dict(got[False]), "valu_per_wave": {"lean": _valu(got[True]), "general": _valu(got[False])},
                                  "valu_per_wave_polled_launch": {"lean": _valu(polled[True]), "general": _valu(polled[False])}}
        assert got[True]["valu_f64"] == got[False]["valu_f64"]
        assert _valu(got[True]) <= _valu(got[False]) - 25, (got[True], got[False])


def test_contracted_lean_kernels_as_compiled(G, helpers, oracle):
    """The opt-in FMA-contracted lean kernels (knob "fp_contract", ode_tu_lean_fast.o; round 6): streamed C4 (Tsit54, ring 16) and streamed C3 (DOPRI54, Lorenz) at
    BASELINE's default options — inside north_star's tolerance for adaptive methods (1e-6 absolute per component), the same launch count as the bit-exact loop, and
    the FP64 instructions a wavefront issues per launch: the figure that puts the streamed C4 kernel's issue floor below its HBM floor."""
    O = oracle
    co = _code_object(G, "ode_tu_lean_fast.o")
    n, t_end = 70, 0.25
    y = _ring_y0(n, 16)
    y0 = y.copy()
    launches, second = _stream_loop(G, helpers, co, r"advance_lps_lean_kernelILi2ENS_7RhsRingILi16EEELi4EE", True, y, True, 16, [0.1], DEFAULT, t_end, 256, 64)
    ref = O.solve_ode_batch(O.RHS_RING, [0.1], y0, n, 16, [0.0, t_end], O.new_options(**DEFAULT), "tsit54", layout=O.LAYOUT_AOS)
    dev4 = float(np.abs(y - ref["y"][-1]).max())
    assert dev4 <= 1e-6 and launches == int(ref["steps"].max()), (dev4, launches)
    c4 = second[0]
    c4.pop("polled_launch")
    y = _lorenz_y0(n)
    y0 = y.copy()
    par = [10.0, 28.0, 8.0 / 3.0]
    launches, second = _stream_loop(G, helpers, co, r"advance_tpi_lean_kernelILi1ENS_9RhsLorenzEEE", True, y, False, 3, par, DEFAULT, t_end, 64, 64)
    ref = O.solve_ode_batch(O.RHS_LORENZ, par, y0, n, 3, [0.0, t_end], O.new_options(**DEFAULT), "dopri54")
    dev3 = float(np.abs(y - ref["y"][-1]).max())
    assert dev3 <= 1e-6 and launches == int(ref["steps"].max()), (dev3, launches)
    c3 = second[0]
    c3.pop("polled_launch")
    RESULTS["streamed_contracted"] = {"c4": dict(c4), "c3": dict(c3), "valu_per_wave": {"c4": _valu(c4), "c3": _valu(c3)},
                                      "max_abs_deviation_from_oracle": {"c4": dev4, "c3": dev3}}
    if "streamed_c4" in RESULTS:
        assert c4["valu_f64"] <= 0.75 * RESULTS["streamed_c4"]["lean"]["valu_f64"], c4   # a*b+c fused: at least a quarter of the FP64 instructions gone


@pytest.mark.parametrize("vec,mode", [(1, 0), (4, 1), (2, 2)])
def test_headline_kernel_as_compiled(G, helpers, oracle, vec, mode):
    """rk4_stream_vec_kernel<RhsNegY<1>, false, VEC, MODE> from ode_tu_rk4_stream.o — compiled with kernarg preloading: the scalar arguments arrive in SGPRs with the
    wave (the kernel descriptor says how many; the interpreter fills them as the command processor does).  Full tiles + the ragged tail tile, 12 steps."""
    O = oracle
    co = _code_object(G, "ode_tu_rk4_stream.o")
    k = co.kernel(r"rk4_stream_vec_kernelINS_7RhsNegYILi1EEELb0ELi%dELi%dEE" % (vec, mode))
    tile = 256 * 2 * vec
    n = 2 * tile + 77
    steps, dt = 12, 2.0 ** -10
    a = (1.0 + (np.arange(n) % 1024) * 2.0 ** -10).copy()
    b = np.full(n, -7.0)
    y0 = a.copy()
    mem = G.Memory(co)
    aa, ab = mem.alloc(a), mem.alloc(b)
    M = G.Machine(co)
    grid = (n + tile - 1) // tile
    if mode & 2:
        grid = min(grid, 2)
    t, cur, nxt, st = 0.0, (aa, a), (ab, b), None
    for _ in range(steps):
        ka = struct.pack("<QQqdddd", cur[0], nxt[0], n, t, dt, 0.5 * dt, dt / 6.0) + bytes(helpers["layout"]["Params"]["sizeof"])
        st = M.launch(k, (grid,), (256,), ka, mem)
        t += dt
        cur, nxt = nxt, cur
    ref = O.solve_ode_batch(O.RHS_NEG_Y, [], y0, n, 0, [0.0, steps * dt], O.new_options(dt=dt), "rk4")
    assert np.array_equal(cur[1], ref["y"][-1, 0])
    c = st[0]
    RESULTS["headline_vec%d_mode%d" % (vec, mode)] = dict(c)
    assert c["vmem"] >= 2 * vec and c["lds"] == 0


def _fused(G, helpers, co, kernel_rx, adaptive, y0, dim, layout, par, opt_fields, tspan, block, per_block):
    n = y0.shape[0] if (layout == 1 and dim > 1) else (y0.shape[-1] if dim > 1 else y0.size)
    mem = G.Memory(co)
    yout = np.full((2,) + y0.shape, -7.0)
    ny = np.full(n, -1, dtype=np.int32)
    steps = np.full(n, -1, dtype=np.int64)
    rej = np.full(n, -1, dtype=np.int64)
    addrs = [mem.alloc(x) for x in (y0, yout, ny, steps, rej)]
    cmd = [helpers["kernarg_solve"], str(int(adaptive)), str(n), str(dim), str(layout), "0", repr(tspan[0]), repr(tspan[1])] + [repr(float(v)) for v in opt_fields] + \
          [str(len(par))] + [repr(float(p)) for p in par] + [str(a) for a in addrs]
    size, hexbytes = subprocess.check_output(cmd, text=True).split()
    ka = bytes.fromhex(hexbytes)
    assert len(ka) == int(size)
    k = co.kernel(kernel_rx)
    st = G.Machine(co).launch(k, ((n + per_block - 1) // per_block,), (block,), ka, mem, max_instructions=60_000_000)
    return yout, ny, steps, rej, st


@pytest.mark.slow
def test_fused_solves_as_compiled(G, helpers, oracle):
    """The fused solve kernels (ODESolver per IVP, ode.nim:471-586, state in VGPRs for the whole solve) with the launch record the library's own planning code builds:
    scalar RK4 (C1 / C2's fused form; host-replayed step schedule), DOPRI54 Lorenz (C3; 48 SGPRs spilled through lanes), Tsit54 on the 16-component ring, four lanes
    per system (C4; 26 VGPRs in scratch around the direction loops) — rows, row counts and accepted / rejected counts equal the oracle's."""
    O = oracle
    # C1-shaped: dy = -y, 300 IVPs (a partly filled second workgroup), 200 steps of 2^-10
    n, steps, dt = 300, 200, 2.0 ** -10
    y0 = (1.0 + np.arange(n) * 2.0 ** -10).copy()
    o = O.new_options(dt=dt)
    yout, ny, st_, rej, st = _fused(G, helpers, _code_object(G, "ode_tu_m_rk4.o"), r"solve_tpi_kernelILi0ENS_7RhsNegYILi1EEELi0EE", False, y0, 1, 0, [],
                                    [o.dt, o.dtMax, o.dtMin, o.tStart, o.absTol, o.relTol, o.scaleMax, o.scaleMin], [0.0, steps * dt], 256, 256)
    ref = O.solve_ode_batch(O.RHS_NEG_Y, [], y0, n, 0, [0.0, steps * dt], o, "rk4")
    assert np.array_equal(yout[1], ref["y"][-1, 0]) and np.array_equal(yout[0], y0) and (ny == 2).all() and (st_ == steps).all()
    RESULTS["fused_rk4_200_steps"] = dict(st[0])
    # C3-shaped: Lorenz, DOPRI54, default options
    n = 70
    y0 = _lorenz_y0(n)
    o = O.new_options(**DEFAULT)
    fields = [o.dt, o.dtMax, o.dtMin, o.tStart, o.absTol, o.relTol, o.scaleMax, o.scaleMin]
    par = [10.0, 28.0, 8.0 / 3.0]
    yout, ny, st_, rej, st = _fused(G, helpers, _code_object(G, "ode_tu_m_dopri54.o"), r"solve_tpi_kernelILi1ENS_9RhsLorenzELi0EE", True, y0, 3, 0, par, fields, [0.0, 0.25], 256, 256)
    ref = O.solve_ode_batch(O.RHS_LORENZ, par, y0, n, 3, [0.0, 0.25], o, "dopri54")
    assert np.array_equal(yout[1], ref["y"][-1]) and np.array_equal(st_, ref["steps"]) and np.array_equal(rej, ref["rejected"]) and (ny == 2).all()
    # C4-shaped: 16-component ring, Tsit54, 4 lanes per system
    n = 37
    y0 = _ring_y0(n, 16)
    yout, ny, st_, rej, st = _fused(G, helpers, _code_object(G, "ode_tu_m_tsit54.o"), r"solve_lps_kernelILi2ENS_7RhsRingILi16EEELi4ELb0ELi0EE", True, y0, 16, 1, [0.1], fields,
                                    [0.0, 0.25], 256, 64)
    ref = O.solve_ode_batch(O.RHS_RING, [0.1], y0, n, 16, [0.0, 0.25], o, "tsit54", layout=O.LAYOUT_AOS)
    assert np.array_equal(yout[1], ref["y"][-1]) and np.array_equal(st_, ref["steps"]) and np.array_equal(rej, ref["rejected"]) and (ny == 2).all()


MIXED = dict(absTol=1e-7, relTol=1e-7, dtMin=1e-9, dtMax=0.25)


_RHS_TAGS = {"7RhsNegY": ("RHS_NEG_Y", []), "9RhsLinear": ("RHS_LINEAR", [-0.7]), "10RhsAffineT": ("RHS_AFFINE_T", [-0.5, 0.3]), "7RhsRing": ("RHS_RING", [0.1]),
             "12RhsVanDerPol": ("RHS_VANDERPOL", [3.0]), "9RhsLorenz": ("RHS_LORENZ", [10.0, 28.0, 8.0 / 3.0])}


def _lean_symbols():
    """every advance_*_lean_kernel instantiation of the two method objects that have them: (object, mangled name, lanes-per-system?, method, rhs tag, dim)"""
    import re
    out = []
    llvm = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    for method, obj in ((1, "ode_tu_m_dopri54.o"), (2, "ode_tu_m_tsit54.o")):
        p = os.path.join(CSRC, obj)
        if not (os.path.exists(p) and os.path.exists(llvm)):
            continue
        try:
            import gfx950_isa_interp as G
            co = G.CodeObject(p)
        except Exception:  # noqa: BLE001
            continue
        for name in sorted(co.symbols):
            m = re.match(r"^_ZN5nnhip23advance_(lps|tpi)_lean_kernelILi(\d)ENS_(\d+Rhs[A-Za-z]+?)(?:ILi(\d+)EE)?E(?:Li(\d)E)?EEvNS_11AdvLeanArgsE$", name)
            if m and not name.endswith(".kd"):
                out.append((obj, name, m.group(1) == "lps", int(m.group(2)), m.group(3), int(m.group(4) or {"12RhsVanDerPol": 2, "9RhsLorenz": 3}[m.group(3)]), int(m.group(5) or 0)))
        _CO[obj] = co
    return out


@pytest.mark.slow
@pytest.mark.parametrize("sym", _lean_symbols(), ids=lambda t: "%s-%s%d-m%d" % ("lps" if t[2] else "tpi", t[4].lstrip("0123456789"), t[5], t[3]))
def test_every_lean_instantiation_as_compiled(G, helpers, oracle, sym):
    """EVERY instantiation of the lean kernels in the library (40: both methods x the compiled-in right-hand sides — scalars to the 32-component ring on 8 lanes per
    system with the LDS error sum, t-dependent right-hand sides, Van der Pol, Lorenz), each from its object file, at tolerances where the controller's pow runs at
    every step: bits and launch count == the oracle.  (A miscompiled instantiation would show here; round 5 had no other way to look.)"""
    O = oracle
    obj, name, lps, method, tag, d, cpl = sym
    co = _code_object(G, obj)
    kind, par = _RHS_TAGS[tag]
    kind = getattr(O, kind)
    integ = {1: "dopri54", 2: "tsit54"}[method]
    n, t_end = 37, 0.6
    rng = np.random.default_rng(3)
    rx = "^" + name.replace("$", "\\$") + "$"
    if lps:
        y = (0.5 + rng.random((n, d))).copy()
        y0 = y.copy()
        launches, _ = _stream_loop(G, helpers, co, rx, True, y, True, d, par, MIXED, t_end, 256, 256 // (d // cpl))
        ref = O.solve_ode_batch(kind, par, y0, n, d, [0.0, t_end], O.new_options(**MIXED), integ, layout=O.LAYOUT_AOS)
        want = ref["y"][-1]
    else:
        y = (0.5 + rng.random((d, n))).copy()
        y0 = y.copy()
        launches, _ = _stream_loop(G, helpers, co, rx, True, y, False, d, par, MIXED, t_end, 64, 64)
        ref = O.solve_ode_batch(kind, par, y0 if d > 1 else y0.reshape(-1), n, d if d > 1 else 0, [0.0, t_end], O.new_options(**MIXED), integ)
        want = ref["y"][-1] if d > 1 else ref["y"][-1].reshape(1, n)
    assert np.array_equal(y, want), name
    assert launches == int(ref["steps"].max())


_ALL = ["heun2", "ralston2", "kutta3", "heun3", "ralston3", "ssprk3", "ralston4", "kutta4", "rk4", "rk21", "bs32", "dopri54", "tsit54", "vern65"]


@pytest.mark.slow
@pytest.mark.parametrize("name", _ALL)
def test_fused_lorenz_solve_of_every_integrator_as_compiled(nn, G, helpers, oracle, name):
    """All 14 integrators (ode.nim:107-468): the fused thread-per-IVP Lorenz solve from each method's object file — the tableau, the stage sums and (for the five
    adaptive ones) the controller as the device compiler emitted them — final states, row counts and accepted / rejected counters == the oracle."""
    O = oracle
    mid = nn.ode.integrator_id(name)
    adaptive = name in ("rk21", "bs32", "dopri54", "tsit54", "vern65")
    co = _code_object(G, "ode_tu_m_%s.o" % name)
    n = 70
    y0 = _lorenz_y0(n)
    y0[0] *= np.linspace(1.0, 4.0, n)
    tol = 1e-3 if name == "rk21" else 1e-6            # (a second-order method at 1e-6 takes thousands of steps: minutes in the interpreter)
    o = O.new_options(**(dict(absTol=tol, relTol=tol, dtMin=1e-9, dtMax=1.0) if adaptive else dict(dt=2.0 ** -7)))
    fields = [o.dt, o.dtMax, o.dtMin, o.tStart, o.absTol, o.relTol, o.scaleMax, o.scaleMin]
    par = [10.0, 28.0, 8.0 / 3.0]
    t_end = 0.3 if adaptive else 0.25
    yout, ny, st_, rej, st = _fused(G, helpers, co, r"solve_tpi_kernelILi%dENS_9RhsLorenzELi0EE" % mid, adaptive, y0, 3, 0, par, fields, [0.0, t_end], 256, 256)
    ref = O.solve_ode_batch(O.RHS_LORENZ, par, y0, n, 3, [0.0, t_end], o, name)
    assert np.array_equal(yout[1], ref["y"][-1]) and np.array_equal(yout[0], y0) and (ny == 2).all(), name
    assert np.array_equal(st_, ref["steps"]) and np.array_equal(rej, ref["rejected"])
    if adaptive and name != "rk21":
        assert int(ref["rejected"].sum()) > 0 or int(ref["steps"].max()) > int(ref["steps"].min())


def _ordered_img(v):
    b = np.asarray(v, dtype=np.float64).view(np.uint64)
    return np.where(b >> np.uint64(63), ~b, b | np.uint64(0x8000000000000000))


@pytest.mark.parametrize("shape", ["uniform_with_a_zero", "both_signs", "one_sign_six_decades", "with_nan_and_inf"])
def test_bin_order_kernels_as_compiled(G, shape):
    """bin_count_kernel / bin_place_kernel from ode_sort.o (round 5: bins linear in value when the keys' range touches or straddles zero) — 1024 threads per
    workgroup = 16 wavefronts meeting at s_barrier, LDS histograms with ds_add_u32 / ds_add_rtn_u32, the DPP / ds_bpermute scan, global atomics, 16-bit loads and
    stores: the bins equal a numpy restatement of sort_kernels.hpp's arithmetic, the histogram the bin counts, and the order is a permutation sorted by bin."""
    co = _code_object(G, "ode_sort.o")
    rng = np.random.default_rng(11)
    n = 4096 + 900                                       # two workgroups, the second partly filled
    keys = {"uniform_with_a_zero": np.concatenate([[0.0], rng.uniform(0.0, 10.0, n - 1)]), "both_signs": rng.uniform(-5.0, 10.0, n),
            "one_sign_six_decades": -10.0 ** rng.uniform(-6.0, 0.0, n), "with_nan_and_inf": rng.uniform(-1.0, 1.0, n)}[shape].copy()
    if shape == "with_nan_and_inf":
        keys[::97] = np.nan
        keys[5::313] = np.inf
    rng.shuffle(keys)
    fin = np.isfinite(keys)
    img = _ordered_img(keys)
    rng_img = np.array([img[fin].min(), img[fin].max()], dtype=np.uint64)
    bins = np.zeros(n, dtype=np.uint16)
    hist = np.zeros(4096, dtype=np.uint32)
    cursor = np.zeros(4096, dtype=np.uint32)
    perm = np.full(n, 0xffffffff, dtype=np.uint32)
    mem = G.Memory(co)
    ak, ar, ab, ah, ac, ap = (mem.alloc(x) for x in (keys, rng_img, bins, hist, cursor, perm))
    M = G.Machine(co)
    blocks = (n + 4095) // 4096
    M.launch(co.kernel(r"bin_count_kernel"), (blocks,), (1024,), struct.pack("<QQQQq", ak, ar, ab, ah, n), mem)
    M.launch(co.kernel(r"bin_place_kernel"), (blocks,), (1024,), struct.pack("<QQQQqQd", ab, ah, ac, ap, n, ar, 0.0), mem)
    # sort_kernels.hpp restated
    mn, mx = float(keys[fin].min()), float(keys[fin].max())
    by_image = (mn > 0 and mx > 0) or (mn < 0 and mx < 0)
    want = np.full(n, 4095, dtype=np.int64)
    nan_or_pinf = np.isnan(keys) | (keys == np.inf)
    if by_image:
        span = int(rng_img[1]) - int(rng_img[0])
        shift = max(span.bit_length() - 12, 0)
        if (span >> shift) > 4094:
            shift += 1
        d = np.array([(int(o) - int(rng_img[0])) >> shift if int(o) > int(rng_img[0]) else 0 for o in img], dtype=np.int64)
        q = np.minimum(d, 4094)
    else:
        mn_half = mn * 0.5
        half_span = mx * 0.5 - mn_half
        per_unit = 4094.0 / half_span if half_span > 0 else 0.0
        with np.errstate(invalid="ignore"):
            r = (keys * 0.5 - mn_half) * per_unit
            q = np.where(r >= 4094.0, 4094, np.where(r > 0.0, np.nan_to_num(r, nan=0.0, posinf=0.0, neginf=0.0).astype(np.int64), 0))
    want[~nan_or_pinf] = q[~nan_or_pinf]
    assert np.array_equal(bins.astype(np.int64), want), shape
    assert np.array_equal(hist, np.bincount(want, minlength=4096).astype(np.uint32))
    assert np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))
    assert (np.diff(want[perm].astype(np.int64)) >= 0).all()


def test_write_the_dynamic_counts(G):
    """(runs last in this module) profiles/r06_isa_dynamic_counts.json is refreshed when NNHIP_WRITE_PROFILES=1; otherwise the committed file must agree."""
    if "streamed_c4" not in RESULTS or "streamed_c3" not in RESULTS:
        pytest.skip("the streamed cases did not run")
    path = os.path.join(ROOT, "profiles", "r06_isa_dynamic_counts.json")
    doc = {"what": "instructions one wavefront executes per launch, by class, counted by tools/gfx950_isa_interp.py on the library's own code objects "
                   "(second launch of the default-option loop — an unpolled one, like 100 of BASELINE's 104 —, wave 0 of workgroup 0: first attempt accepted, clamp early-out; valu_per_wave_polled_launch: the third launch, "
                   "which answers 'anyone left?'); VALU = valu_f64 + valu_other + valu_dpp + valu_lane",
           "reference_point": "round 4 measured 587 VALU per wave on the general streamed C4 kernel (SQ_INSTS_VALU / SQ_WAVES, profiles/r04_c4_stream_pmc/)",
           "kernels": RESULTS}
    if os.environ.get("NNHIP_WRITE_PROFILES"):
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    committed = json.load(open(path))
    for key in ("streamed_c4", "streamed_c3"):
        assert committed["kernels"][key]["valu_per_wave"] == RESULTS[key]["valu_per_wave"], (key, "the committed counts are stale: rerun with NNHIP_WRITE_PROFILES=1")


def test_workgroup_or_reduction_sees_every_row_of_the_wave(G, tmp_path):
    """__syncthreads_or — the "anyone still integrating?" of every polled launch — as hipcc compiles it for gfx950: row_shl scans, then ONE `v_mov_b32_dpp ... wave_shl:1`
    carries each row's result into its left neighbour, then row_mirror.  The interpreter once ran that wave-wide control as a plain move: a wave whose only active lane
    sat in row 1 or 3 (lanes 16-31, 48-63) answered "nobody", and a heterogeneous batch on the ISA-backed node ended its polling loop while its slowest IVP was still
    integrating (found through test_adaptive_dense_output_through_the_step_streaming_seam; the device code was right, round 4 passed it on hardware).  Every lane alone, one-wave
    and four-wave workgroups, must be seen."""
    if shutil.which("hipcc") is None:
        pytest.skip("needs hipcc")
    src = tmp_path / "wgor.hip"
    src.write_text('#include <hip/hip_runtime.h>\nextern "C" __global__ void wg_or(unsigned int* out, int k) { const int any = __syncthreads_or(k >= 0 && (int)threadIdx.x == k);\n'
                   '  if (threadIdx.x == 0) out[blockIdx.x] = any ? 1u : 0u; }\n')
    obj = str(tmp_path / "wgor.o")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-c", str(src), "-o", obj])
    co = G.CodeObject(obj)
    assert any("wave_shl:1" in i.mods for i in co.kernel("wg_or").ins), "the compiler no longer uses the wave-wide shift: this test no longer covers it"
    k = co.kernel("wg_or")
    M = G.Machine(co)
    for block in (64, 256):
        for lane in list(range(0, block, 5)) + [15, 16, 31, 32, 47, 48, 63, block - 1, -1]:
            mem = G.Memory(co)
            out = np.full(1, 7, dtype=np.uint32)
            a = mem.alloc(out)
            M.launch(k, (1,), (block,), struct.pack("<Qi", a, lane), mem)
            assert out[0] == (1 if lane >= 0 else 0), (block, lane, int(out[0]))

"""CPU half of the interpreted Nim shim (tests/nimrun.py, tests/test_gpu_nim_shim_interpreted.py): the whole of nim/numericalnim_hip.nim and the
generated bindings load into the interpreter, the host-only entries answer through the ctypes bridge, and a solve gets as far as the library —
which, without a GPU, refuses it with its own message, raised by the shim's `check` as the reference-style exception."""
import pytest


@pytest.fixture(scope="module")
def it():
    import nimrun
    return nimrun.load()


def test_the_whole_shim_loads(it):
    procs = {k for k, v in it.globals.vars.items() if isinstance(v, list)}
    assert {"solveode", "rhsfromsource", "rhsfromsourcepercomponent", "rhsfromsourcectx", "bindctx", "readaux", "cumtrapz", "cumsimpson", "newhermitespline",
            "eval", "deriveval", "toc", "check"} <= procs
    assert len(it.globals.vars["solveode"]) == 3 and len(it.globals.vars["cumtrapz"]) == 2
    assert {"Odebatch", "Rhsspec", "Rhskind", "Batchlayout", "Ctxvector", "Batchhermitespline", "Nnhipoptions", "Nnhipstats"} <= set(it.types)
    assert it.expr("rhsVanDerPol") == 5 and it.expr("layoutAoS") == 1                       # enum nnhip_rhs_kind / the layouts of include/nnhip_ode.h
    ffi = [v for v in it.globals.vars.values() if type(v).__name__ == "FFIProc"]
    assert len(ffi) >= 60


def test_host_only_entries_through_the_bridge(it):
    import numericalnim_amd as nn
    assert it.call("nnhip_abi_version") == 1
    for name in ("rk4", "DOPRI54", "Tsit54", "vern65", "bs32", "rk21", "heun2", "ralston4"):
        assert it.call("nnhip_ode_integrator_id", name) == nn.ode.integrator_id(name)
    assert it.call("nnhip_ode_integrator_id", "rk5") < 0
    # toC: ODEoptions -> the C struct, field for field; nnhip_ode_time_grid through `addr opt`, `addr ts[0]`, `addr tOut[0]`, `addr nt`
    got = it.expr("""newODEoptions(dt = 1e-3, absTol = -1e-7, tStart = 0.5).toC""")
    assert got.tname == "Nnhipoptions" and got.get("dt") == 1e-3 and got.get("abstol") == 1e-7 and got.get("tstart") == 0.5 and got.get("dtmax") == 1e-2
    it.exec_toplevel('''
proc timeGrid(tspan: openArray[float], tStart: float): seq[float] =
  var opt = newODEoptions(tStart = tStart).toC
  var ts = @tspan
  var tOut = newSeq[cdouble](ts.len + 1)
  var nt: cint
  check nnhip_ode_time_grid(addr opt, addr ts[0], ts.len.cint, addr tOut[0], addr nt)
  result = tOut[0 ..< nt.int]
''')
    assert it.call("timeGrid", [3.0, -1.0, 0.5, 2.0, 0.5], 0.5) == [-1.0, 0.5, 2.0, 3.0]      # ode.nim:476-487, :585: sorted, tStart once
    assert it.call("timeGrid", [3.0, -1.0, 2.0], 0.5) == [-1.0, 2.0, 3.0]


def test_a_solve_reaches_the_library(it):
    import torch
    spec = it.expr('RhsSpec(kind: rhsLinear, keys: @["a"])')
    ctx = it.call("newNumContext")
    ctx.get("fvalues")["a"] = -0.1
    batch = it.expr("OdeBatch(n: 3, dim: 1, layout: layoutSoA, data: @[1.0, 2.0, 3.0])")
    with pytest.raises(Exception, match="not a valid integrator"):
        it.call("solveODE", spec, batch, [0.0, 1.0], ctx=ctx, integrator="rk5")
    n0 = len(it.ffi_log)
    if torch.cuda.is_available():
        t, ys = it.call("solveODE", spec, batch, [0.0, 1.0], ctx=ctx, integrator="rk4")
        assert t == [0.0, 1.0] and len(ys) == 2 and ys[0].get("data") == [1.0, 2.0, 3.0]
    else:
        with pytest.raises(Exception, match="nnhip error|IOError|device"):
            it.call("solveODE", spec, batch, [0.0, 1.0], ctx=ctx, integrator="rk4")
    assert it.ffi_log[n0:n0 + 2] == ["nnhip_ode_integrator_id", "nnhip_ode_solve_batch_sweep_f64"]


# ---- nim/rhs_macro.nim: the translating procs of the `deviceRhs` macro, run on the tree of a Nim body ------------------------------------------
LORENZ = '''
dy[0] = ctx.fValues["sigma"] * (y[1] - y[0])
dy[1] = y[0] * (ctx.fValues["rho"] - y[2]) - y[1]
dy[2] = y[0] * y[1] - ctx.fValues["beta"] * y[2]
'''


@pytest.fixture(scope="module")
def mac():
    import nimrun
    return nimrun.load(macros=True), nimrun


def test_the_macro_emits_what_its_own_self_test_expects(mac):
    import os
    import re
    it, nimrun = mac
    text = open(os.path.join(nimrun.NIM_DIR, "rhs_macro.nim")).read()
    expected = re.search(r'doAssert src == "(.*)"', text).group(1).replace("\\n", "\n")          # the literal of the file's `when isMainModule` block
    assert it.call("deviceRhsSource", ["sigma", "rho", "beta"], nimrun.nim_ast(LORENZ)) == expected
    # every Nim infix node is one parenthesised C operation, integer literals and loop variables become doubles where Nim would convert them
    src = it.call("deviceRhsSource", ["c"], nimrun.nim_ast('''
for i in 0 ..< 16:
  let w = float(i + 1) / 16.0
  dy[i] = -w * y[i] + ctx.fValues["c"] * y[(i + 1) mod 16]
'''))
    assert src == ("for (int i = 0; i < 16; ++i) {\n  const double w = ((double)(((double)i + 1.0)) / 16.0);\n"
                   "  dy[i] = (((-w) * y[i]) + (p[0] * y[(((i + 1)) % 16)]));\n}\n")
    assert it.call("deviceRhsSource", ["a"], nimrun.nim_ast("dy[0] = sqrt(abs(y[0])) * ctx.fValues[\"a\"] + min(t, 2) - 1e-3")) == \
        "dy[0] = (((sqrt(fabs(y[0])) * p[0]) + nnhip::nmin(t, 2.0)) - 0.001);\n"


@pytest.mark.parametrize("body,why", [
    ("dy[0] = y[0] ^ 2", "operator"), ("dy[0] = ctx.fValues[\"nope\"]", "not in `keys`"), ("y[0] = 1.0", "assignment"), ("dy[0] = foo(y[0])", "call of"),
    ("dy[k] = 1.0", "index variable"), ("dy[0] = ctx.tValues[\"w\"][0]", "use deviceRhsCtx"), ("if t > 0.0: dy[0] = 1.0", "statement of kind")])
def test_the_macro_refuses_what_it_cannot_translate(mac, body, why):
    it, nimrun = mac
    with pytest.raises(Exception, match=why):
        it.call("deviceRhsSource", ["a"], nimrun.nim_ast(body))


def test_ctx_vectors_get_their_layout_late(mac):
    """deviceRhsCtxSource: (text with the vector accesses still bracket-neutral, the vectors in order of first use)"""
    it, nimrun = mac
    src, vectors = it.call("deviceRhsCtxSource", ["a"], nimrun.nim_ast('dy[0] = ctx.fValues["a"] * y[0] * ctx.tValues["w"][1] + ctx.tValues["g"][0] * ctx.tValues["w"][0]'))
    assert vectors == ["w", "g"]
    assert src.replace(chr(3), "").replace(chr(1), "<").replace(chr(2), ">") == "dy[0] = (((p[0] * y[0]) * w<1>) + (g<0> * w<0>));\n"
    # a vector whose name ends another's ("b" / "ab") is still told apart: every access carries a marker in front of its name
    src, vectors = it.call("deviceRhsCtxSource", ["a"], nimrun.nim_ast('dy[0] = ctx.tValues["ab"][1] + ctx.tValues["b"][0]'))
    assert vectors == ["ab", "b"] and src == "dy[0] = (%sab%s1%s + %sb%s0%s);\n" % (chr(3), chr(1), chr(2), chr(3), chr(1), chr(2))


def test_arguments_are_checked_against_the_declared_parameter_types(it):
    """a run-time stand-in for the compiler's check: a proc of the shim called with an argument of the wrong kind is refused, not coerced"""
    spec = it.expr('RhsSpec(kind: rhsLinear, keys: @["a"])')
    batch = it.expr("OdeBatch(n: 1, dim: 1, layout: layoutSoA, data: @[1.0])")
    with pytest.raises(Exception, match="no overload|type mismatch"):
        it.call("solveODE", spec, [1.0], [0.0, 1.0])                                  # a seq where an OdeBatch is declared
    with pytest.raises(Exception, match="type mismatch"):
        it.call("solveODE", spec, batch, [0.0, 1.0], integrator=5)                   # integrator: string
    with pytest.raises(Exception, match="no overload|type mismatch"):
        it.call("bindCtx", spec, [1.0], [], [], "one", 1)                            # nAux: int
    with pytest.raises(Exception, match="type mismatch"):
        it.exec_toplevel("proc badDecl(): int =\n  var n: int = @[1.0]\n  result = n\n") or it.call("badDecl")

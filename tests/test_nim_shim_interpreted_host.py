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

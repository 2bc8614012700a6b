"""The reference-side binding (nim/numericalnim_hip.nim + the generated raw bindings nim/nnhip_ode_bindings.nim) against the built
library — where a Nim toolchain exists.  The build image and the GPU boxes of this project have none, so these
tests probe at run time and SKIP LOUDLY instead of pretending: the C ABI itself is exercised from C and
C++ by tests/test_gpu_cpp_host.py."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "numericalnim_amd", "csrc")


def _nim():
    nim = shutil.which("nim")
    if nim is None:
        pytest.skip("no Nim toolchain on this box (`nim` is not on PATH): nim/numericalnim_hip.nim has never met a compiler here; "
                    "the C ABI it binds is covered by tests/cpp/*.c* (tests/test_gpu_cpp_host.py)")
    return nim


def test_raw_bindings_compile_link_and_call(tmp_path):
    """nim/nnhip_ode_bindings.nim (generated from include/nnhip_ode.h): a Nim program that links libnnhip_ode.so, checks the ABI
    version, builds ODEoptions through the C constructor and resolves the integrator names the reference dispatches on."""
    nim = _nim()
    prog = tmp_path / "smoke.nim"
    prog.write_text(f'''
import "{os.path.join(ROOT, "nim", "nnhip_ode_bindings")}"
doAssert nnhip_abi_version() == 1
var o: NnhipOptions
doAssert nnhip_ode_default_options(addr o) == 0
doAssert o.dt == 1e-4 and o.dtMax == 1e-2 and o.absTol == 1e-4
for name in ["rk4", "DOPRI54", "Tsit54", "vern65", "bs32", "rk21"]:
  doAssert nnhip_ode_integrator_id(name.cstring) >= 0
doAssert nnhip_ode_integrator_id("no_such".cstring) < 0
echo "nim raw bindings ok, devices: ", nnhip_device_count()
''')
    r = subprocess.run([nim, "c", "-r", "--hints:off", f"--passL:-L{LIBDIR} -lnnhip_ode -Wl,-rpath,{LIBDIR}", f"--nimcache:{tmp_path}/cache", str(prog)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "nim raw bindings ok" in r.stdout


def test_shim_solveode_matches_reference_solveode(tmp_path):
    """nim/numericalnim_hip.nim: the batched `solveODE` overload next to the reference's own scalar `solveODE` on the reference's
    harness (tests/test_ode.nim:5-46: f = -0.1 y, linspace(-10, 10, 100)) — needs the numericalnim package itself."""
    nim = _nim()
    probe = subprocess.run([nim, "c", "--hints:off", "--eval:import numericalnim"], capture_output=True, text=True, timeout=300)
    if probe.returncode != 0:
        pytest.skip("Nim is here but the numericalnim package is not installed (`import numericalnim` fails): the shim imports the "
                    "reference's ODEoptions / NumContext and cannot be compiled without it")
    prog = tmp_path / "harness.nim"
    prog.write_text(f'''
import std/[math, tables]
import numericalnim
import "{os.path.join(ROOT, "nim", "numericalnim_hip")}"
let tspan = linspace(-10.0, 10.0, 100)
proc f(t: float, y: float, ctx: NumContext[float, float]): float = -0.1 * y
for integ in ["rk4", "dopri54", "tsit54"]:
  let (tRef, yRef) = solveODE(f, 1.0, tspan, integrator = integ)
  var ctx = newNumContext[OdeBatch, float]()
  ctx.fValues["a"] = -0.1
  let batch = OdeBatch(n: 4, dim: 1, layout: layoutSoA, data: @[1.0, 1.0, 1.0, 1.0])
  let (t, y) = solveODE(RhsSpec(kind: rhsLinear, keys: @["a"]), batch, tspan, ctx = ctx, integrator = integ)
  doAssert t == tRef
  for j in 0 ..< t.len:
    doAssert y[j].data[0] == yRef[j], integ & ": row " & $j   # bit-identical to the reference's own CPU path
echo "nim shim ok"
''')
    r = subprocess.run([nim, "c", "-r", "--hints:off", "-d:release", f"--passL:-L{LIBDIR} -lnnhip_ode -Wl,-rpath,{LIBDIR}", f"--nimcache:{tmp_path}/cache",
                        str(prog)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "nim shim ok" in r.stdout


def test_rhs_macro_translates_a_nim_body(tmp_path):
    """nim/rhs_macro.nim: `deviceRhs` turns a restricted Nim body (the user's f(t, y, ctx), ode.nim:36) into the HIP source the backend compiles;
    its self-test compares the emitted text for Lorenz with the expected one, and a Lorenz batch solved through the macro must equal the
    compiled-in kind bit for bit.  Needs Nim AND the numericalnim package (the shim imports it)."""
    nim = _nim()
    probe = subprocess.run([nim, "c", "--hints:off", "--eval:import numericalnim"], capture_output=True, text=True, timeout=300)
    if probe.returncode != 0:
        pytest.skip("Nim is here but the numericalnim package is not installed: nim/rhs_macro.nim imports the shim, which imports it")
    r = subprocess.run([nim, "c", "-r", "--hints:off", f"--passL:-L{LIBDIR} -lnnhip_ode -Wl,-rpath,{LIBDIR}", f"--nimcache:{tmp_path}/cache0",
                        os.path.join(ROOT, "nim", "rhs_macro.nim")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    prog = tmp_path / "macro_lorenz.nim"
    prog.write_text(f'''
import std/tables
import numericalnim
import "{os.path.join(ROOT, "nim", "numericalnim_hip")}"
import "{os.path.join(ROOT, "nim", "rhs_macro")}"
let f = deviceRhs(3, ["sigma", "rho", "beta"]):
  dy[0] = ctx.fValues["sigma"] * (y[1] - y[0])
  dy[1] = y[0] * (ctx.fValues["rho"] - y[2]) - y[1]
  dy[2] = y[0] * y[1] - ctx.fValues["beta"] * y[2]
var ctx = newNumContext[OdeBatch, float]()
ctx.fValues["sigma"] = 10.0; ctx.fValues["rho"] = 28.0; ctx.fValues["beta"] = 8.0 / 3.0
let batch = OdeBatch(n: 2, dim: 3, layout: layoutSoA, data: @[1.0, 1.5, 1.0, 1.0, 1.0, 1.0])
let (t1, y1) = solveODE(f, batch, @[0.0, 0.5, 1.0], ctx = ctx, integrator = "tsit54")
let (t2, y2) = solveODE(RhsSpec(kind: rhsLorenz, keys: @["sigma", "rho", "beta"]), batch, @[0.0, 0.5, 1.0], ctx = ctx, integrator = "tsit54")
doAssert t1 == t2
for j in 0 ..< t1.len: doAssert y1[j].data == y2[j].data
echo "nim rhs macro ok"
''')
    r = subprocess.run([nim, "c", "-r", "--hints:off", "-d:release", f"--passL:-L{LIBDIR} -lnnhip_ode -Wl,-rpath,{LIBDIR}", f"--nimcache:{tmp_path}/cache",
                        str(prog)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "nim rhs macro ok" in r.stdout

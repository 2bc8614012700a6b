"""The pin of the oracle (and of the device's constant tables) to the REFERENCE'S OWN TEXT.

The reference is Nim and cannot be compiled in this image, so `oracle/_ref` does not exist.  Instead oracle/nim_subset.py interprets the
subset of Nim that ode.nim's solver path is written in and executes the reference's source text directly:
  * tests/golden/reference_text_vectors.json holds what solveODE / the 14 *_step procs of the reference's text return on the inputs of
    every fixture of tests/golden/ode_golden.json (generated in the build container by tests/golden/make_reference_text_vectors.py);
  * the tests below require the oracle to reproduce those vectors BIT FOR BIT (runs on any host), re-run the interpreter on a sample and on
    all single steps when /root/reference is present (build container), and compare the `const` sections of DOPRI54_step / TSIT54_step /
    VERN65_step (ode.nim:240-282, 310-352, 380-443), evaluated from the text, with the tables the oracle and the HIP kernels are compiled
    with (oracle_tableau / nnhip_ode_tableau_f64).
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

from golden_util import fh, load_cases

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
VEC = json.load(open(os.path.join(HERE, "golden", "reference_text_vectors.json")))
LORENZ_P = [10.0, 28.0, 8.0 / 3.0]


def _ref():
    from oracle import nim_subset as N
    if not N.reference_available():
        pytest.skip("/root/reference is not present on this host (the reference does not travel); the committed vectors stand in")
    return N, N.load_reference_ode()


def test_vectors_cover_every_fixture_and_agree_with_the_oracle_generated_goldens():
    """Data against data: the oracle-generated fixtures and the reference-text vectors hold the same bits for all 101 cases / 180 IVPs."""
    gold = {c["name"]: c for c in load_cases()}
    assert sorted(gold) == sorted(c["name"] for c in VEC["cases"])
    n = 0
    for c in VEC["cases"]:
        g = gold[c["name"]]
        assert c["t"] == g["t"], c["name"]
        assert len(c["ivps"]) == len(g["ivps"])
        for a, b in zip(c["ivps"], g["ivps"]):
            assert a["n_y"] == b["n_y"] and a["y"] == b["y"], c["name"]
            n += 1
    assert n >= 180 and {c["integrator"] for c in gold.values()} == set(__import__("oracle.oracle", fromlist=["x"]).ALL_ODE)


@pytest.mark.parametrize("case", load_cases(), ids=lambda c: c["name"])
def test_oracle_reproduces_the_reference_text_vectors(oracle, case):
    """solveODE of the oracle (live) == solveODE of the reference's text (committed), bit for bit, rows and time grid included."""
    O = oracle
    want = next(c for c in VEC["cases"] if c["name"] == case["name"])
    opt = O.new_options(**case["options"])
    for y0, w in zip(case["y0"], want["ivps"]):
        y0v = fh(y0)
        t, y, st = O.solve_ode(case["rhs_kind"], fh(case["params"]), float(y0v[0]) if case["dim"] == 0 else y0v, fh(case["tspan"]), opt, case["integrator"])
        assert [float(v).hex() for v in t] == want["t"]
        assert st.n_y == w["n_y"]
        assert [float(v).hex() for v in np.asarray(y).ravel()] == w["y"]


@pytest.mark.parametrize("s", VEC["steps"], ids=lambda s: f"{s['integrator']}-{s['input']}")
def test_oracle_single_step_matches_the_reference_text(oracle, s):
    """One IntegratorProc call (ode.nim:38): (yNew, FSAL, dt, error) of every *_step proc on a NONLINEAR right-hand side — accepted at once,
    through in-step retries with pow (ode.nim:58-76), and through the dtMin double hit.  Every tableau coefficient and every literal of
    ode.nim:107-234 takes part in these numbers."""
    O = oracle
    yN, fs, dtU, err = O.step(O.RHS_LORENZ, LORENZ_P, s["integrator"], O.new_options(**s["options"]), float.fromhex(s["t"]), fh(s["y"]), fh(s["fsal"]),
                              float.fromhex(s["dt"]))
    assert [float(v).hex() for v in yN] == s["yNew"]
    assert float(dtU).hex() == s["dtUsed"] and float(err).hex() == s["error"]
    if s["integrator"] in ("dopri54", "tsit54", "vern65", "bs32"):  # the FSAL slot is the last stage; the other methods return yNew there (unused: useFSAL = false)
        assert [float(v).hex() for v in fs] == s["fsalOut"]


def test_retry_inputs_really_retry():
    """The single-step inputs exercise what they claim: input 1 shrinks dt inside the step, input 2 ends on dtMin for every adaptive method."""
    for s in VEC["steps"]:
        if s["integrator"] in ("rk21", "bs32", "dopri54", "tsit54", "vern65"):
            if s["input"] == 1:
                assert float.fromhex(s["dtUsed"]) < float.fromhex(s["dt"]), s["integrator"]
            if s["input"] == 2:
                assert float.fromhex(s["dtUsed"]) == 0.1 and float.fromhex(s["error"]) > 1.0, s["integrator"]
        else:
            assert s["dtUsed"] == s["dt"] and float.fromhex(s["error"]) == 0.0


def test_linspace_of_the_reference_text(oracle):
    assert [float(v).hex() for v in oracle.linspace(-10.0, 10.0, 100)] == VEC["linspace_m10_10_100"]


# ---- build container only: the interpreter re-run against the committed vectors, and the tableaux ------------------------------------

def test_interpreter_rerun_matches_committed_vectors():
    """The committed vectors are what the reference's text produces TODAY.  Re-run from /root/reference: all 42 single steps and the Vector-state
    fixtures that are cheap with utils.nim's Vector operators INTERPRETED as well (every `+`, `*`, `/.`, `abs`, `sum` on a state is the
    reference's own proc); a wider sample — every 7th fixture, the reference-quirk and dtMin ones — with a stand-in class for the Vector type (20 x
    faster; the two agree bit for bit, which the first part shows).  The full set is regenerated by tests/golden/make_reference_text_vectors.py."""
    N, it = _ref()
    import make_reference_text_vectors as M
    assert it.interpret_vector
    assert M.step_cases(it) == VEC["steps"]
    gold = load_cases()
    want = {v["name"]: v for v in VEC["cases"]}
    cheap = [c for c in gold if c["dim"] > 0 and (c["name"].startswith(("lorenz_tight", "ring16_tight", "rejecting_lorenz", "vdp_rk4", "ring4_")) or c["name"] in ("lorenz_default_vern65", "ring16_bs32"))]
    assert len(cheap) >= 8
    for c in cheap:
        assert M.solve_case(it, c) == want[c["name"]], c["name"]
    fast = N.load_reference_ode(interpret_vector=False)
    sample = [c for k, c in enumerate(gold) if k % 7 == 0 or c["name"].startswith("quirk") or c["name"].startswith("dtmin_escape")]
    for c in sample:
        assert M.solve_case(fast, c) == want[c["name"]], c["name"]
    assert [float(v).hex() for v in it.call("linspace", -10.0, 10.0, 100)] == VEC["linspace_m10_10_100"]


def test_vector_operators_of_the_reference_text(oracle):
    """utils.nim's Vector procs, interpreted, against the oracle's restatement of them (oracle_vector_op) — the operators the path uses, incl. the
    size check that raises (utils.nim:22-26) and `sum` = left-to-right from 0.0 (utils.nim:243-250 -> :233-235)."""
    N, it = _ref()
    rng = np.random.default_rng(11)
    a, b = rng.normal(size=7), rng.normal(size=7)
    va, vb = N.vector(it, a), N.vector(it, b)
    for op, got in (("+", it.user_op("+", [va, vb])), ("-", it.user_op("-", [va, vb])), ("*.", it.user_op("*.", [va, vb])), ("/.", it.user_op("/.", [va, vb]))):
        assert [float(x).hex() for x in N.components(got)] == [float(x).hex() for x in oracle.vector_op(op, a, b)], op
    assert [float(x).hex() for x in N.components(it.user_op("*", [0.3, va]))] == [float(x).hex() for x in oracle.vector_op("s*", a, d=0.3)]
    assert [float(x).hex() for x in N.components(it.user_op("+.", [0.3, va]))] == [float(x).hex() for x in oracle.vector_op("+.", a, d=0.3)]
    assert [float(x).hex() for x in N.components(it.call("abs", va))] == [float(x).hex() for x in oracle.vector_op("abs", a)]
    assert float(it.call("sum", va)).hex() == float(oracle.vector_op("sum", a)[0]).hex()
    with pytest.raises(N.NimError):
        it.user_op("+", [va, N.vector(it, [1.0, 2.0])])


def _device_tableau(nn, integrator, device=-1):
    L = nn._lib.lib()
    out = np.full(128, np.nan)
    k = L.nnhip_ode_tableau_f64(nn.ode.integrator_id(integrator), device, out.ctypes.data_as(C.POINTER(C.c_double)), 128)
    assert k > 0, nn._lib.last_error()
    S, NB = int(out[0]), int(out[1])
    named, p = {}, 2
    for s in range(1, S + 1):
        named[f"c{s}"] = out[p]; p += 1
    for s in range(2, S + 1):
        for j in range(1, s):
            named[f"a{s}{j}"] = out[p]; p += 1
    for j in range(1, NB + 1):
        named[f"b{j}"] = out[p]; p += 1
    for j in range(1, S + 1):
        named[f"bhat{j}"] = out[p]; p += 1
    assert p == k
    return named


@pytest.mark.parametrize("integrator,proc,n_consts", [("dopri54", "DOPRI54_step", 40), ("tsit54", "TSIT54_step", 40), ("vern65", "VERN65_step", 61)])
def test_tableaux_equal_the_reference_const_sections_bit_for_bit(oracle, nn, integrator, proc, n_consts):
    """ode.nim:240-282 / 310-352 / 380-443 evaluated from the text (Nim constant folding = IEEE double arithmetic on the literals) against
    (a) the constants the oracle's step procs are compiled with and (b) the constexpr tables behind Tableau<M>, which the HIP steppers read."""
    N, it = _ref()
    ref = it.consts_of(proc)
    assert len(ref) == n_consts
    orc = oracle.tableau(integrator)
    assert [N.norm_ident(n) for n, _ in orc] == list(ref)  # same constants, same declaration order
    for n, v in orc:
        assert float(v).hex() == float(ref[N.norm_ident(n)]).hex(), (integrator, n)
    dev = _device_tableau(nn, integrator)
    assert dev.pop("c1") == 0.0  # the first stage is at t (k1 = FSAL, ode.nim:293); the reference has no c1
    assert sorted(dev) == sorted(ref), "the device table holds exactly the reference's constants"
    for n, v in dev.items():
        assert float(v).hex() == float(ref[n]).hex(), (integrator, n)


@pytest.mark.gpu
@pytest.mark.parametrize("integrator", ["dopri54", "tsit54", "vern65"])
def test_tableau_seen_by_a_kernel_is_the_host_table(nn, dev, integrator):
    """The same accessors evaluated inside a kernel on the GPU give the bits the host reads (which the CPU test above pins to the reference's text)."""
    host, gpu = _device_tableau(nn, integrator, -1), _device_tableau(nn, integrator, 0)
    assert host.keys() == gpu.keys()
    for n in host:
        assert float(host[n]).hex() == float(gpu[n]).hex(), n


@pytest.mark.gpu
@pytest.mark.parametrize("s", VEC["steps"], ids=lambda s: f"{s['integrator']}-{s['input']}")
def test_hip_single_step_matches_the_reference_text(nn, dev, s):
    """nnhip_ode_step_batch_f64_dev (the IntegratorProc seam) against the reference's text on the GPU box: no oracle in between."""
    import torch
    n = 5  # the same IVP in several lanes
    y = torch.tensor(fh(s["y"]), dtype=torch.float64, device=dev).reshape(3, 1).repeat(1, n).contiguous()
    fsal = torch.tensor(fh(s["fsal"]), dtype=torch.float64, device=dev).reshape(3, 1).repeat(1, n).contiguous()
    yN, fs, dtU, err = nn.integratorStep(nn.Rhs.lorenz(), float.fromhex(s["t"]), y, fsal, float.fromhex(s["dt"]), nn.newODEoptions(**s["options"]),
                                         integrator=s["integrator"])
    for i in (0, n - 1):
        assert [float(v).hex() for v in yN[:, i].cpu().numpy()] == s["yNew"]
        if dtU is not None:  # adaptive methods; the fixed-step ones return their input dt and error 0.0 (ode.nim:113 ... 189)
            assert float(dtU[i]).hex() == s["dtUsed"] and float(err[i]).hex() == s["error"]
        if s["integrator"] in ("dopri54", "tsit54", "vern65", "bs32"):
            assert [float(v).hex() for v in fs[:, i].cpu().numpy()] == s["fsalOut"]


# ---- the Butcher tableau of EVERY step proc, extracted from the reference's text by symbolic execution, against the order conditions ----
class _Lin:
    """A state that is a linear combination of symbols (y, k1, k2, ...): running a *_step proc of the reference on it, with an f that hands out
    a fresh symbol per call and records its arguments, yields the method's Butcher tableau (c from the time arguments with t = 0, dt = 1)."""
    def __init__(self, terms):
        self.terms = {k: v for k, v in terms.items() if v != 0.0}

    def binop(self, op, other, swapped):
        if isinstance(other, _Lin):
            sign = {"+": 1.0, "-": -1.0}[op]
            a, b = (other, self) if swapped else (self, other)
            out = dict(a.terms)
            for k, v in b.terms.items():
                out[k] = out.get(k, 0.0) + sign * v
            return _Lin(out)
        if op == "*":
            return _Lin({k: v * other for k, v in self.terms.items()})
        raise AssertionError(f"state {op} scalar does not occur in a step proc")

    def neg(self):
        return _Lin({k: -v for k, v in self.terms.items()})


def _extract_tableau(it, proc):
    calls = []

    def f(t, y, ctx):
        calls.append((t, y))
        return _Lin({f"k{len(calls)}": 1.0})
    y = _Lin({"y": 1.0})
    # (the five adaptive procs need abs / sum of the state inside commonAdaptiveMethodCode: they are covered by their exported tableaux, by
    # test_third_party_pins.py's order conditions and by the single-step vectors instead)
    res = it.call(proc, f, 0.0, y, _Lin({"fsal": 1.0}), 1.0, it.call("newODEoptions"), None)
    S = len(calls)
    A, c = np.zeros((S, S)), np.zeros(S)
    for i, (t, arg) in enumerate(calls):
        c[i] = t
        assert arg.terms.get("y", 0.0) == 1.0
        for k, v in arg.terms.items():
            if k != "y":
                assert int(k[1:]) <= i, "explicit method"
                A[i, int(k[1:]) - 1] = v
    b = np.array([res[0].terms.get(f"k{j + 1}", 0.0) for j in range(S)])
    assert res[0].terms.get("y") == 1.0
    return A, b, c


@pytest.mark.parametrize("integrator,order", [("heun2", 2), ("ralston2", 2), ("kutta3", 3), ("heun3", 3), ("ralston3", 3), ("ssprk3", 3), ("ralston4", 4),
                                              ("kutta4", 4), ("rk4", 4)])
def test_fixed_step_procs_of_the_reference_have_their_nominal_order(integrator, order):
    """ode.nim:107-189 as tableaux: row sums = c, all rooted-tree conditions up to `order` (ode.nim:608-649) hold, the next order's do not.
    (Ralston4's literals are printed to 8 digits, ode.nim:163-167: its conditions hold to 1e-8.)"""
    from test_third_party_pins import _trees, _gamma, _phi
    N, it = _ref()
    A, b, c = _extract_tableau(it, N.STEP_PROCS[integrator])
    tol = 2e-8 if integrator == "ralston4" else 1e-15
    assert np.abs(A.sum(axis=1) - c).max() <= tol
    for k in range(1, order + 1):
        for t in _trees(k):
            assert abs(b @ _phi(t, A) - 1.0 / _gamma(t)) <= tol, (integrator, k, t)
    assert max(abs(b @ _phi(t, A) - 1.0 / _gamma(t)) for t in _trees(order + 1)) > 1e-3

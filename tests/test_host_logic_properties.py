"""Property tests (hypothesis) of the host-side logic of the boundary against the oracle: option normalisation and the
output time grid (tspan.sorted(), split around tStart, reversed-negative ++ zero ++ positive; ode.nim:476-487, 585)."""
import ctypes as C

import numpy as np
from hypothesis import given, settings, strategies as st

finite = st.floats(min_value=-1e6, max_value=1e6, allow_nan=False, allow_infinity=False)
pos = st.floats(min_value=1e-12, max_value=1e3, allow_nan=False)


@settings(max_examples=200, deadline=None)
@given(tspan=st.lists(st.one_of(finite, st.sampled_from([0.0, 0.5, -0.5, 1.0])), min_size=0, max_size=12), tstart=st.sampled_from([0.0, 0.5, -0.5, 2.0]))
def test_time_grid_property(nn, oracle, tspan, tstart):
    L = nn._lib.lib()
    o = nn.newODEoptions(dt=1e3, tStart=tstart)   # huge dt: the oracle takes one step per branch
    ts = np.asarray(tspan, dtype=np.float64)
    out = np.empty(max(len(ts), 1))
    n = C.c_int()
    assert L.nnhip_ode_time_grid(C.byref(o), ts.ctypes.data_as(C.POINTER(C.c_double)), len(ts), out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)) == 0
    got = out[:n.value]
    if len(ts):
        t_ref, _, _ = oracle.solve_ode(oracle.RHS_NEG_Y, [], 1.0, ts, oracle.new_options(dt=1e3, tStart=tstart), "rk4")
        assert np.array_equal(got, t_ref)
    # structure: sorted, at most one copy of tStart, everything else preserved with multiplicity
    assert np.all(np.diff(got) >= 0)
    assert (got == tstart).sum() == (1 if (ts == tstart).any() else 0)
    assert sorted(got[got != tstart]) == sorted(ts[ts != tstart])


@settings(max_examples=200, deadline=None)
@given(dt=finite, absTol=finite, relTol=finite, dtMax=finite, dtMin=finite, scaleMax=finite, scaleMin=finite, tStart=finite)
def test_new_options_property(nn, oracle, dt, absTol, relTol, dtMax, dtMin, scaleMax, scaleMin, tStart):
    kw = dict(dt=dt, absTol=absTol, relTol=relTol, dtMax=dtMax, dtMin=dtMin, scaleMax=scaleMax, scaleMin=scaleMin, tStart=tStart)
    try:
        ref = oracle.new_options(**kw)
    except ValueError:
        ref = None
    try:
        got = nn.newODEoptions(**kw)
    except ValueError:
        got = None
    assert (ref is None) == (got is None)
    if ref is not None:
        assert bytes(ref) == bytes(got)

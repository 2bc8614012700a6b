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


# cumtrapz(f, X, ctx, dx) / cumsimpson(f, X, ctx, dx): the host replays the reference's sampling grid and hermiteInterpolate's
# control flow (utils.nim:282-312) to decide how many rows come back and whether a ValueError is raised.  That part runs
# before any device work, so it can be checked against the oracle without a GPU (with one, the call simply goes on to succeed).
xq = st.one_of(st.floats(min_value=-2.0, max_value=2.0, allow_nan=False), st.sampled_from([0.0, 0.25, 0.5, 1.0, -1.0]))


@settings(max_examples=300, deadline=None)
@given(X=st.lists(xq, min_size=1, max_size=8), dx=st.sampled_from([0.01, 0.1, 0.25, 0.3, 1.0, 1.5, 7.0]), rule=st.sampled_from(["trapz", "simpson"]),
       sort=st.booleans())
def test_cumquad_fn_row_plan_property(nn, oracle, X, dx, rule, sort):
    import torch
    L = nn._lib.lib()
    Xa = np.asarray(sorted(X) if sort else X, dtype=np.float64)
    p = np.array([0.0, 1.0])
    dp = C.POINTER(C.c_double)
    rows = C.c_int(-1)
    entry = L.nnhip_cumtrapz_fn_batch_f64_dev if rule == "trapz" else L.nnhip_cumsimpson_fn_batch_f64_dev
    gpu = torch.cuda.is_available()
    out = torch.empty((len(Xa) + 1) * 4, dtype=torch.float64, device="cuda") if gpu else None
    rc = entry(nn.Rhs.AFFINE_T, p.ctypes.data_as(dp), 2, None, 0, 4, 1, 0, Xa.ctypes.data_as(dp), len(Xa), dx,
               C.c_void_p(out.data_ptr() if gpu else 16), C.byref(rows), None)
    try:
        ref = oracle.cumquad_fn(rule, oracle.RHS_AFFINE_T, p, 0, Xa, dx)
    except ValueError:
        ref = None
    if rule == "simpson" and rc == nn._lib.NNHIP_EUNSUPPORTED:
        # max(X) == min(X) with a grid of >= 3 coincident points: the reference would deduplicate the grid; refused here
        assert Xa.max() == Xa.min()
        return
    if ref is None:
        assert rc == nn._lib.NNHIP_EVALUE, (rc, nn._lib.last_error())
    else:
        assert rows.value == len(ref), (Xa, dx, rule, rows.value, len(ref))
        assert rc == (nn._lib.NNHIP_OK if (gpu or len(ref) == 0) else nn._lib.NNHIP_EHIP), (rc, nn._lib.last_error())

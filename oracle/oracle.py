"""ctypes loader for the CPU oracle (oracle/ode_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (numericalnim_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_ode.so")

# RHS kinds — must match include/nnhip_ode.h (enum nnhip_rhs_kind)
RHS_NEG_Y, RHS_LINEAR, RHS_LORENZ, RHS_RING, RHS_AFFINE_T, RHS_VANDERPOL, RHS_DUFFING, RHS_COS_T, RHS_POLY_T, RHS_HEAT, RHS_MATVEC, RHS_LORENZ_ZCROSS = range(12)  # DUFFING.. : oracle-only
LAYOUT_SOA, LAYOUT_AOS = 0, 1

ALL_ODE = ["heun2", "ralston2", "kutta3", "heun3", "ralston3", "ssprk3", "ralston4", "kutta4", "rk4",
           "rk21", "bs32", "dopri54", "tsit54", "vern65"]  # ode.nim:40-42


class Options(C.Structure):  # ODEoptions field order, ode.nim:26-34
    _fields_ = [(n, C.c_double) for n in ("dt", "dtMax", "dtMin", "tStart", "absTol", "relTol", "scaleMax", "scaleMin")]


class Stats(C.Structure):
    _fields_ = [("rhs_evals", C.c_int64), ("steps", C.c_int64), ("rejected", C.c_int64),
                ("n_t", C.c_int32), ("n_y", C.c_int32), ("nan_abort", C.c_int32), ("_pad", C.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, "ode_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_ode.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        _lib.oracle_new_options.argtypes = [C.POINTER(Options)] + [C.c_double] * 8
        _lib.oracle_integrator_id.argtypes = [C.c_char_p]
        _lib.oracle_solve_ode.argtypes = [C.c_int, dp, C.c_int, C.c_int, dp, dp, C.c_int, C.POINTER(Options), C.c_char_p,
                                          dp, dp, C.POINTER(Stats)]
        _lib.oracle_solve_ode_batch.argtypes = [C.c_int, dp, C.c_int, C.c_int, C.c_int, dp, C.c_int64, dp, C.c_int,
                                                C.POINTER(Options), C.c_char_p, dp, dp, C.POINTER(C.c_int32),
                                                C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
        _lib.oracle_step.argtypes = [C.c_int, dp, C.c_int, C.c_int, C.c_char_p, C.POINTER(Options), C.c_double, dp, dp,
                                     C.c_double, dp, dp, dp, dp]
        _lib.oracle_rhs.argtypes = [C.c_int, dp, C.c_int, C.c_int, C.c_double, dp, dp]
        _lib.oracle_hermite_spline.argtypes = [C.c_double] * 7
        _lib.oracle_hermite_spline.restype = C.c_double
        _lib.oracle_linspace.argtypes = [C.c_double, C.c_double, C.c_int, dp]
        _lib.oracle_hermite_interp.argtypes = [dp, C.c_int, dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_double, dp]
        _lib.oracle_cumtrapz.argtypes = [dp, C.c_int, dp, dp]
        _lib.oracle_sort_and_trim.argtypes = [dp, C.c_int, C.POINTER(dp), C.c_int, dp, C.POINTER(dp)]
        _lib.oracle_cumsimpson.argtypes = [dp, C.c_int, dp, dp]
        _lib.oracle_vector_op.argtypes = [C.c_int, dp, C.c_int, dp, C.c_int, C.c_double, dp]
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def new_options(dt=1e-4, absTol=1e-4, relTol=1e-4, dtMax=1e-2, dtMin=1e-4, scaleMax=4.0, scaleMin=0.1, tStart=0.0):
    """newODEoptions (ode.nim:78-102). Raises ValueError like the reference."""
    o = Options()
    rc = lib().oracle_new_options(C.byref(o), dt, absTol, relTol, dtMax, dtMin, scaleMax, scaleMin, tStart)
    if rc != 0:
        raise ValueError("invalid ODEoptions (dtMin<=dtMax, scaleMax>=1, scaleMin<=1 required)")
    return o


def controller_factor(error, order):
    """min(4, max(0.125, 0.9*pow(1/error, 1/order))) (ode.nim:71,537) with libm's pow, element-wise."""
    e = np.ascontiguousarray(error, dtype=np.float64)
    out = np.empty_like(e)
    lib().oracle_controller_factor.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int, C.POINTER(C.c_double)]
    lib().oracle_controller_factor.restype = None
    lib().oracle_controller_factor(_dp(e), e.size, int(order), _dp(out))
    return out


def linspace(x1, x2, n):
    """utils.nim:498-507 (NOT numpy.linspace: x1 + dx*i with the endpoint appended verbatim)."""
    if n <= 0:
        raise ValueError("Number of samples must be greater then 0")
    out = np.empty(n + 1, dtype=np.float64)
    k = lib().oracle_linspace(x1, x2, n, _dp(out))
    return out[:k].copy()


def solve_ode(rhs_kind, params, y0, tspan, options=None, integrator="dopri54"):
    """solveODE (ode.nim:589-651) for one IVP. y0 float → scalar path; y0 sequence → Vector path.
    Returns (t, y, stats) where y has stats.n_y rows (may be < len(t): reference quirk)."""
    options = options or new_options()
    params = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
    tspan = np.ascontiguousarray(np.asarray(tspan, dtype=np.float64))
    scalar = np.isscalar(y0)
    y0a = np.ascontiguousarray(np.atleast_1d(np.asarray(y0, dtype=np.float64)))
    dim = 0 if scalar else len(y0a)
    dimv = max(dim, 1)
    n_t = len(tspan)
    t_out = np.empty(n_t, dtype=np.float64)
    y_out = np.full((n_t, dimv), np.nan, dtype=np.float64)
    st = Stats()
    rc = lib().oracle_solve_ode(rhs_kind, _dp(params), len(params), dim, _dp(y0a), _dp(tspan), n_t, C.byref(options),
                                integrator.encode(), _dp(t_out), _dp(y_out), C.byref(st))
    if rc == -2:
        raise ValueError(f"{integrator} is not a valid integrator")
    if rc != 0:
        raise ValueError("oracle error %d" % rc)
    y = y_out[:st.n_y]
    return t_out[:st.n_t].copy(), (y[:, 0].copy() if scalar else y.copy()), st


def solve_ode_batch(rhs_kind, params, y0, N, dim, tspan, options=None, integrator="dopri54", layout=LAYOUT_SOA, n_threads=1):
    """The reference called once per IVP. y0 flat in `layout`. Returns dict(t, y, ny, steps, rejected)."""
    options = options or new_options()
    params = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
    tspan = np.ascontiguousarray(np.asarray(tspan, dtype=np.float64))
    y0 = np.ascontiguousarray(np.asarray(y0, dtype=np.float64).ravel())
    dimv = max(dim, 1)
    n_t = len(tspan)
    assert y0.size == N * dimv
    t_out = np.empty(n_t, dtype=np.float64)
    y_out = np.empty(n_t * dimv * N, dtype=np.float64)
    ny = np.empty(N, dtype=np.int32)
    steps = np.empty(N, dtype=np.int64)
    rej = np.empty(N, dtype=np.int64)
    rc = lib().oracle_solve_ode_batch(rhs_kind, _dp(params), len(params), dim, layout, _dp(y0), N, _dp(tspan), n_t,
                                      C.byref(options), integrator.encode(), _dp(t_out), _dp(y_out),
                                      ny.ctypes.data_as(C.POINTER(C.c_int32)), steps.ctypes.data_as(C.POINTER(C.c_int64)),
                                      rej.ctypes.data_as(C.POINTER(C.c_int64)), n_threads)
    if rc == -2:
        raise ValueError(f"{integrator} is not a valid integrator")
    if rc != 0:
        raise ValueError("oracle error %d" % rc)
    shape = (n_t, dimv, N) if layout == LAYOUT_SOA else (n_t, N, dimv)
    return dict(t=t_out, y=y_out.reshape(shape), ny=ny, steps=steps, rejected=rej)


def solve_ode_batch_ctx(rhs_kind, shared, per_ivp, aux, y0, N, dim, tspan, options=None, integrator="dopri54", layout=LAYOUT_SOA, n_threads=1):
    """N reference calls whose closures each capture their own ctx: env_i = [shared..., per_ivp[:, i]..., aux[:, i]...]; `aux`
    ([n_aux, N] float64, may be None) is the part the closure mutates and is updated in place.  Returns dict(t, y, ny, steps, rejected, aux)."""
    options = options or new_options()
    shared = np.ascontiguousarray(np.atleast_1d(np.asarray(shared, dtype=np.float64)))
    per_ivp = np.zeros((0, N)) if per_ivp is None else np.ascontiguousarray(np.asarray(per_ivp, dtype=np.float64).reshape(-1, N))
    auxa = np.zeros((0, N)) if aux is None else np.ascontiguousarray(np.asarray(aux, dtype=np.float64).reshape(-1, N))
    tspan = np.ascontiguousarray(np.asarray(tspan, dtype=np.float64))
    y0 = np.ascontiguousarray(np.asarray(y0, dtype=np.float64).ravel())
    n_t = len(tspan)
    assert y0.size == N * dim and dim >= 1
    t_out = np.empty(n_t, dtype=np.float64)
    y_out = np.empty(n_t * dim * N, dtype=np.float64)
    ny = np.empty(N, dtype=np.int32)
    steps = np.empty(N, dtype=np.int64)
    rej = np.empty(N, dtype=np.int64)
    f = lib().oracle_solve_ode_batch_ctx
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int,
                  C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_double),
                  C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
    rc = f(rhs_kind, _dp(shared), len(shared), _dp(per_ivp), per_ivp.shape[0], _dp(auxa), auxa.shape[0], dim, layout, _dp(y0), N, _dp(tspan), n_t,
           C.cast(C.byref(options), C.c_void_p), integrator.encode(), _dp(t_out), _dp(y_out), ny.ctypes.data_as(C.POINTER(C.c_int32)),
           steps.ctypes.data_as(C.POINTER(C.c_int64)), rej.ctypes.data_as(C.POINTER(C.c_int64)), n_threads)
    if rc != 0:
        raise ValueError("oracle error %d" % rc)
    shape = (n_t, dim, N) if layout == LAYOUT_SOA else (n_t, N, dim)
    return dict(t=t_out, y=y_out.reshape(shape), ny=ny, steps=steps, rejected=rej, aux=auxa)


def step(rhs_kind, params, integrator, options, t, y, fsal, dt):
    """One IntegratorProc call (ode.nim:38): returns (yNew, FSAL, dtUsed, error)."""
    params = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
    scalar = np.isscalar(y)
    ya = np.ascontiguousarray(np.atleast_1d(np.asarray(y, dtype=np.float64)))
    fa = np.ascontiguousarray(np.atleast_1d(np.asarray(fsal, dtype=np.float64)))
    dim = 0 if scalar else len(ya)
    yn = np.empty_like(ya)
    fn = np.empty_like(ya)
    dtu = C.c_double()
    err = C.c_double()
    rc = lib().oracle_step(rhs_kind, _dp(params), len(params), dim, integrator.encode(), C.byref(options), t, _dp(ya),
                           _dp(fa), dt, _dp(yn), _dp(fn), C.byref(dtu), C.byref(err))
    if rc:
        raise ValueError("oracle_step error %d" % rc)
    if scalar:
        return yn[0], fn[0], dtu.value, err.value
    return yn, fn, dtu.value, err.value


def rhs(rhs_kind, params, t, y):
    params = np.ascontiguousarray(np.atleast_1d(np.asarray(params, dtype=np.float64)))
    scalar = np.isscalar(y)
    ya = np.ascontiguousarray(np.atleast_1d(np.asarray(y, dtype=np.float64)))
    out = np.empty_like(ya)
    lib().oracle_rhs(rhs_kind, _dp(params), len(params), 0 if scalar else len(ya), t, _dp(ya), _dp(out))
    return out[0] if scalar else out


def hermite_spline(x, x1, x2, y1, y2, dy1, dy2):
    return lib().oracle_hermite_spline(x, x1, x2, y1, y2, dy1, dy2)


EXTRAP = {"Constant": 0, "Edge": 1, "Linear": 2, "Native": 3, "Error": 4}  # interpolate.nim:89-90


def hermite_interp(X, Y, dY, xq, deriv=False, extrap="Native", extrap_value=0.0):
    """newHermiteSpline(X, Y, dY).eval / .derivEval (interpolate.nim:186-240, 299-390) for one scalar series."""
    X, Y, dY, xq = (np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (X, Y, dY, xq))
    if len(X) > 1 and not np.all(X[1:] > X[:-1]):   # `let sortedDataset = sortAndTrimDataset(@X, @[@Y, @dY])` (interpolate.nim:231)
        X, (Y, dY) = sort_and_trim(X, Y, dY)
    out = np.empty(len(xq), dtype=np.float64)
    rc = lib().oracle_hermite_interp(_dp(X), len(X), _dp(Y), _dp(dY), _dp(xq), len(xq), int(deriv), EXTRAP[extrap], extrap_value, _dp(out))
    if rc:
        raise ValueError("x isn't in the interval")
    return out


def hermite_slopes(X, Y):
    """The derivative estimates of newHermiteSpline(X, Y) (interpolate.nim:241-253) for one scalar series."""
    X, Y = (np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (X, Y))
    if len(X) > 1 and not np.all(X[1:] > X[:-1]):   # `let (xSorted, ySorted) = sortAndTrimDataset(@X, @Y)` (interpolate.nim:244): slopes of the sorted, trimmed data
        X, (Y,) = sort_and_trim(X, Y)
    out = np.empty(len(X), dtype=np.float64)
    if lib().oracle_hermite_slopes(_dp(X), len(X), _dp(Y), _dp(out)):
        raise ValueError("need at least 2 points")
    return out


def _raise_dataset(k):
    if k == -2:
        raise ArithmeticError("NaN in X: sortAndTrimDataset has no defined order for it (nothing to restate)")
    if k < 0:
        raise ValueError("ValueError in the reference (impure y-duplicates / too few distinct abscissae)")


def sort_and_trim(X, *Ys):
    """sortAndTrimDataset(x, @[y_0, ...]) (utils.nim:404-407) on scalar series -> (x, [y_0, ...]); ValueError for impure duplicates."""
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    Ys = [np.ascontiguousarray(np.asarray(y, dtype=np.float64)) for y in Ys]
    xo = np.empty_like(X)
    outs = [np.empty_like(X) for _ in Ys]
    pp = C.POINTER(C.c_double) * max(len(Ys), 1)
    k = lib().oracle_sort_and_trim(_dp(X), len(X), pp(*[_dp(y) for y in Ys]), len(Ys), _dp(xo), pp(*[_dp(o) for o in outs]))
    _raise_dataset(k)
    return xo[:k].copy(), [o[:k].copy() for o in outs]


def cumtrapz(Y, X):
    """cumtrapz(Y, X) (integrate.nim:120-135) for one scalar series; X in any order (sorted and trimmed first, :130)."""
    X, Y = (np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (X, Y))
    out = np.empty(len(X), dtype=np.float64)
    k = lib().oracle_cumtrapz(_dp(X), len(X), _dp(Y), _dp(out))
    _raise_dataset(k)
    return out[:k].copy()


def cumsimpson(Y, X):
    """cumsimpson(Y, X) (integrate.nim:329-375) for one scalar series; X in any order (3 distinct abscissae or more); the result at the caller's abscissae."""
    X, Y = (np.ascontiguousarray(np.asarray(a, dtype=np.float64)) for a in (X, Y))
    out = np.empty(len(X), dtype=np.float64)
    k = lib().oracle_cumsimpson(_dp(X), len(X), _dp(Y), _dp(out))
    _raise_dataset(k)
    return out[:k].copy()


def cumquad_fn(rule, rhs_kind, params, dim, X, dx=1e-5):
    """cumtrapz(f, X, ctx, dx) (rule "trapz", integrate.nim:138-175) / cumsimpson(f, X, ctx, dx) (rule "simpson", :377-400) with
    f(x) := rhs(x, y=0, params).  dim == 0: T = float -> [rows]; dim >= 1: T = Vector[float] -> [rows][dim].  The reference can
    return fewer rows than len(X) (hermiteInterpolate quirks); ValueError as the reference raises it."""
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    p = np.ascontiguousarray(np.asarray(params, dtype=np.float64))
    out = np.empty((len(X) + 1, max(dim, 1)), dtype=np.float64)
    lib().oracle_cumquad_fn.restype = C.c_int
    lib().oracle_cumquad_fn.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                        C.c_int, C.c_double, C.c_void_p]
    k = lib().oracle_cumquad_fn({"trapz": 0, "simpson": 1}[rule], rhs_kind, p.ctypes.data, len(p), dim, X.ctypes.data, len(X), float(dx),
                                out.ctypes.data)
    if k < 0:
        raise ValueError("cumulative quadrature: ValueError in the reference")
    return out[:k, 0].copy() if dim == 0 else out[:k].copy()


def vector_op(op, a, b=None, d=0.0):
    ops = {"+": 0, "-": 1, "s*": 2, "abs": 3, "*.": 4, "/.": 5, "+.": 6, "sum": 7}
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    out = np.empty(max(len(a), 1), dtype=np.float64)
    if b is not None:
        b = np.ascontiguousarray(np.asarray(b, dtype=np.float64))
        n = lib().oracle_vector_op(ops[op], _dp(a), len(a), _dp(b), len(b), d, _dp(out))
    else:
        n = lib().oracle_vector_op(ops[op], _dp(a), len(a), None, 0, d, _dp(out))
    if n == -1:
        raise ValueError("Vectors must have the same size.")
    return out[:n].copy()


def tableau(integrator):
    """The constants the oracle's DOPRI54 / TSIT54 / VERN65 step procs are compiled with, in declaration order: [(name, value)]."""
    f = lib().oracle_tableau
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.c_int]
    names = C.create_string_buffer(4096)
    vals = np.empty(128, dtype=np.float64)
    n = f(integrator.encode(), names, 4096, _dp(vals), 128)
    if n < 0:
        raise ValueError(f"{integrator} has no tableau")
    return list(zip(names.value.decode().split("\n"), vals[:n].tolist()))

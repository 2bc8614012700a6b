"""Output consumer of the ODE path: the batched form of the reference's
`newHermiteSpline(t, y, dy)` + `eval` / `derivEval` (src/numericalnim/interpolate.nim:186-240, 299-390),
the consumer the reference's README advertises for `(t, y, dy)` from `solveODE` (README.md:147)."""
import ctypes as C

import numpy as np

from . import _lib
from .ode import _check, _params_array, _require_state, _shape_info, LAYOUT_SOA

ExtrapolateKind = {"Constant": 0, "Edge": 1, "Linear": 2, "Native": 3, "Error": 4}  # interpolate.nim:89-90


def rhsBatch(f, t, y, ctx=None, layout=LAYOUT_SOA):
    """dy = f(t, y) over a device batch (one RHS evaluation per IVP)."""
    import torch
    _require_state("y", y)
    p, pp = _params_array(f, ctx)
    N, dim, _ = _shape_info(y, layout)
    yc = y.contiguous()
    out = torch.empty_like(yc)
    with torch.cuda.device(yc.device):
        rc = _lib.lib().nnhip_ode_rhs_batch_f64_dev(f.kind, pp, int(p.size), N, dim, layout, float(t), yc.data_ptr(), out.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream)
    if rc:
        raise ValueError("rhs batch evaluation failed (rc=%d)" % rc)
    return out


def _dataset_rows(Xa):
    """(rows of sortAndTrimDataset's result, rows of cumsimpson(Y, X)'s) for the caller's X — decided on the host (nnhip_dataset_rows_f64)."""
    ns, nr = C.c_int(0), C.c_int(0)
    _check(_lib.lib().nnhip_dataset_rows_f64(Xa.ctypes.data_as(C.POINTER(C.c_double)), len(Xa), C.byref(ns), C.byref(nr)))
    return ns.value, nr.value


def sortAndTrimDataset(X, *Ys):
    """sortAndTrimDataset(x, @[y_0, ...]) (utils.nim:404-413) over batched series: X host, every Y an [n, ...] CUDA tensor (each trailing element its
    own series).  Returns (X_sorted_trimmed, [Y_sorted_trimmed ...]); impure duplicates raise ValueError as in the reference."""
    import torch
    Xa = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    for k, y in enumerate(Ys):
        _require_state("series %d" % k, y, Ys[0] if k else None)
    Ys = [y.contiguous() for y in Ys]
    if any(len(Xa) != y.shape[0] for y in Ys):
        raise ValueError("X and Y must have the same length")
    M = int(Ys[0][0].numel()) if Ys else 0
    outs = [torch.empty_like(y) for y in Ys]
    Xo = np.empty_like(Xa)
    no = C.c_int(0)
    arr = (C.c_void_p * max(len(Ys), 1))
    dev = Ys[0].device if Ys else None
    import contextlib
    with (torch.cuda.device(dev) if dev is not None else contextlib.nullcontext()):
        _check(_lib.lib().nnhip_sort_and_trim_dataset_f64_dev(Xa.ctypes.data_as(C.POINTER(C.c_double)), len(Xa), arr(*[y.data_ptr() for y in Ys]), len(Ys), M,
                                                              Xo.ctypes.data_as(C.POINTER(C.c_double)), arr(*[o.data_ptr() for o in outs]), C.byref(no),
                                                              torch.cuda.current_stream().cuda_stream if dev is not None else None))
    return Xo[:no.value].copy(), [o[:no.value] for o in outs]


class HermiteSpline:
    """newHermiteSpline(X, Y, dY) for a whole batch: Y, dY are [n_knots, ...] CUDA tensors (every trailing element its
    own series).  X in any order: the constructor sorts and trims the data as the reference's does (interpolate.nim:231, 244); the solver's output
    grid is strictly ascending already, and then nothing moves."""

    def __init__(self, X, Y, dY=None):
        self.X = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
        self.host = isinstance(Y, np.ndarray)  # numpy series: host-pointer entry (staged through the device per call, which sorts and trims there)
        if len(self.X) != Y.shape[0] or (dY is not None and len(self.X) != dY.shape[0]):
            raise ValueError("X and Y and dY must have the same length.")  # interpolate.nim:229-230
        if self.host:
            self.Y = np.ascontiguousarray(Y, dtype=np.float64)
            self.dY = None if dY is None else np.ascontiguousarray(dY, dtype=np.float64)
            self.M = int(self.Y[0].size)
            return
        import torch
        _require_state("Y", Y)
        if dY is not None:
            _require_state("dY", dY, Y)
        if len(self.X) > 1 and not bool(np.all(self.X[1:] > self.X[:-1])):   # sortAndTrimDataset(@X, @[@Y, @dY]) / (@X, @Y), once, here
            self.X, sorted_ = sortAndTrimDataset(self.X, *([Y] if dY is None else [Y, dY]))
            Y = sorted_[0]
            dY = None if dY is None else sorted_[1]
        self.Y = Y.contiguous()
        self.M = int(self.Y[0].numel())
        if dY is None:  # newHermiteSpline(X, Y): three-point difference slopes (interpolate.nim:241-253)
            self.dY = torch.empty_like(self.Y)
            with torch.cuda.device(self.Y.device):
                _check(_lib.lib().nnhip_hermite_spline_slopes_f64_dev(self.X.ctypes.data_as(C.POINTER(C.c_double)), len(self.X), self.Y.data_ptr(),
                                                                      self.M, self.dY.data_ptr(), torch.cuda.current_stream().cuda_stream))
        else:
            self.dY = dY.contiguous()

    def _run(self, x, deriv, extrap, extrapValue):
        if extrap == "Constant" and extrapValue is None:
            # the reference asserts (AssertionDefect) when extrapValue is missing
            raise AssertionError("When using `extrap = Constant`, a value `extrapValue` must be supplied!")  # interpolate.nim:312,359
        xq = np.ascontiguousarray(np.atleast_1d(np.asarray(x, dtype=np.float64)))
        dp = C.POINTER(C.c_double)
        if self.host:
            out = np.empty((len(xq),) + self.Y.shape[1:], dtype=np.float64)
            _check(_lib.lib().nnhip_hermite_spline_eval_batch_f64(
                self.X.ctypes.data_as(dp), len(self.X), self.Y.ctypes.data_as(dp), None if self.dY is None else self.dY.ctypes.data_as(dp), self.M,
                xq.ctypes.data_as(dp), len(xq), int(deriv), ExtrapolateKind[extrap], float(extrapValue or 0.0), out.ctypes.data_as(dp), 0))
            return out if np.ndim(x) else out[0]
        import torch
        out = torch.empty((len(xq),) + tuple(self.Y.shape[1:]), dtype=torch.float64, device=self.Y.device)
        with torch.cuda.device(self.Y.device):
            _check(_lib.lib().nnhip_hermite_spline_eval_batch_f64_dev(
                self.X.ctypes.data_as(C.POINTER(C.c_double)), len(self.X), self.Y.data_ptr(), self.dY.data_ptr(), self.M,
                xq.ctypes.data_as(C.POINTER(C.c_double)), len(xq), int(deriv), ExtrapolateKind[extrap], float(extrapValue or 0.0),
                out.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return out if np.ndim(x) else out[0]

    def eval(self, x, extrap="Native", extrapValue=None):       # interpolate.nim:299-345, 392-404
        return self._run(x, 0, extrap, extrapValue)

    def derivEval(self, x, extrap="Native", extrapValue=None):  # interpolate.nim:346-390, 406-418
        return self._run(x, 1, extrap, extrapValue)


def newHermiteSpline(X, Y, dY=None):
    """newHermiteSpline(X, Y, dY) (interpolate.nim:216-239) or, without dY, newHermiteSpline(X, Y) (:241-257)."""
    return HermiteSpline(X, Y, dY)


def _cumquad_fn(rule, f, X, ctx, dx, sweep, n, dim, device, layout):
    """The function-argument forms cumtrapz(f, X, ctx, dx) / cumsimpson(f, X, ctx, dx) (integrate.nim:138-175, 377-400):
    f is an Rhs whose value at (x, y=0) is the integrand; the batch axis is a parameter sweep (`sweep` [k, N] CUDA tensor of per-item
    values for the first k parameters) or N identical items.  Returns [rows, dim, N] (SoA) / [rows, N, dim] (AoS) with
    rows <= len(X) exactly as the reference's hermiteInterpolate produces them; dim == 1 results are squeezed to [rows, N]."""
    Xa = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    if Xa.ndim != 1 or len(Xa) < 1:
        raise ValueError("X must be a non-empty 1-d sequence")
    p, pp = _params_array(f, ctx)
    dim = int(getattr(f, "dim", dim))  # run-time compiled integrands know their own size
    if isinstance(sweep, np.ndarray) or (sweep is None and device == "host"):  # host arrays in, host array out
        dp = C.POINTER(C.c_double)
        sw = None if sweep is None else np.ascontiguousarray(sweep, dtype=np.float64)
        if sw is not None and sw.ndim != 2:
            raise ValueError("sweep must have shape [k, N]")
        N, k = (int(n), 0) if sw is None else (int(sw.shape[1]), int(sw.shape[0]))
        shape = (len(Xa), dim, N) if layout == LAYOUT_SOA else (len(Xa), N, dim)
        out = np.full(shape, np.nan)
        rows = C.c_int(0)
        _check(getattr(_lib.lib(), f"nnhip_{rule}_fn_batch_f64")(f.kind, pp, int(p.size), None if sw is None else sw.ctypes.data_as(dp), k, N, dim, layout, Xa.ctypes.data_as(dp), len(Xa),
                          float(dx), out.ctypes.data_as(dp), C.byref(rows), 0))
        out = out[:rows.value]
        return out.reshape(rows.value, N) if dim == 1 else out
    import torch
    if sweep is not None:
        sw = sweep.contiguous()
        if not sw.is_cuda or sw.dtype != torch.float64 or sw.ndim != 2:
            raise ValueError("sweep must be a CUDA float64 tensor of shape [k, N]")
        N, k, swp, device = int(sw.shape[1]), int(sw.shape[0]), sw.data_ptr(), sw.device
    else:
        N, k, swp = int(n), 0, None
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    shape = (len(Xa), dim, N) if layout == LAYOUT_SOA else (len(Xa), N, dim)
    out = torch.full(shape, float("nan"), dtype=torch.float64, device=device)
    rows = C.c_int(0)
    with torch.cuda.device(device):
        _check(getattr(_lib.lib(), f"nnhip_{rule}_fn_batch_f64_dev")(f.kind, pp, int(p.size), swp, k, N, dim, layout, Xa.ctypes.data_as(C.POINTER(C.c_double)), len(Xa), float(dx),
                     out.data_ptr(), C.byref(rows), torch.cuda.current_stream().cuda_stream))
    out = out[:rows.value]
    if dim == 1:
        out = out.reshape(rows.value, N)
    return out


def cumtrapz(Y, X, ctx=None, dx=1e-5, sweep=None, n=1, dim=1, device=None, layout=LAYOUT_SOA):
    """cumtrapz(Y, X) for discrete points (integrate.nim:120-135), batched: Y is an [n, ...] CUDA tensor, every trailing
    element its own series; X in any order (sorted and trimmed first, as the reference does: the result has one row per distinct abscissa, ascending).
    Returns the cumulative integrals.
    With an Rhs as first argument: cumtrapz(f, X, ctx, dx) (integrate.nim:138-175), see _cumquad_fn."""
    import torch
    from .ode import Rhs
    if isinstance(Y, Rhs):
        return _cumquad_fn("cumtrapz", Y, X, ctx, dx, sweep, n, dim, device, layout)
    Xa = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    if len(Xa) != Y.shape[0]:
        raise ValueError("X and Y must have the same length")  # utils.nim:423-424
    if isinstance(Y, np.ndarray):  # host series: host-pointer entry
        Yh = np.ascontiguousarray(Y, dtype=np.float64)
        outh = np.empty_like(Yh)
        dp = C.POINTER(C.c_double)
        _check(_lib.lib().nnhip_cumtrapz_batch_f64(Xa.ctypes.data_as(dp), len(Xa), Yh.ctypes.data_as(dp), int(Yh[0].size), outh.ctypes.data_as(dp), 0))
        return outh[:_dataset_rows(Xa)[0]]
    _require_state("Y", Y)
    Yc = Y.contiguous()
    out = torch.empty_like(Yc)
    with torch.cuda.device(Yc.device):
        _check(_lib.lib().nnhip_cumtrapz_batch_f64_dev(Xa.ctypes.data_as(C.POINTER(C.c_double)), len(Xa), Yc.data_ptr(), int(Yc[0].numel()),
                                                       out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return out[:_dataset_rows(Xa)[0]]


def cumsimpson(Y, X, ctx=None, dx=1e-5, sweep=None, n=1, dim=1, device=None, layout=LAYOUT_SOA):
    """cumsimpson(Y, X) for discrete points (integrate.nim:329-375), batched like cumtrapz; needs 3 distinct abscissae.  X in any order: the result
    is returned at the caller's abscissae, in the caller's order (hermiteInterpolate, :375).
    With an Rhs as first argument: cumsimpson(f, X, ctx, dx) (integrate.nim:377-400), see _cumquad_fn."""
    import torch
    from .ode import Rhs
    if isinstance(Y, Rhs):
        return _cumquad_fn("cumsimpson", Y, X, ctx, dx, sweep, n, dim, device, layout)
    Xa = np.ascontiguousarray(np.asarray(X, dtype=np.float64))
    if len(Xa) != Y.shape[0]:
        raise ValueError("X and Y must have the same length")
    if len(Xa) < 3:
        raise ValueError("X and Y must have at least 3 elements to perform Simpson, use cumtrapz instead")  # integrate.nim:345-346
    if isinstance(Y, np.ndarray):
        Yh = np.ascontiguousarray(Y, dtype=np.float64)
        outh = np.empty_like(Yh)
        dp = C.POINTER(C.c_double)
        _check(_lib.lib().nnhip_cumsimpson_batch_f64(Xa.ctypes.data_as(dp), len(Xa), Yh.ctypes.data_as(dp), int(Yh[0].size), outh.ctypes.data_as(dp), 0))
        return outh[:_dataset_rows(Xa)[1]]
    _require_state("Y", Y)
    Yc = Y.contiguous()
    out = torch.empty_like(Yc)
    with torch.cuda.device(Yc.device):
        _check(_lib.lib().nnhip_cumsimpson_batch_f64_dev(Xa.ctypes.data_as(C.POINTER(C.c_double)), len(Xa), Yc.data_ptr(), int(Yc[0].numel()),
                                                         out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return out[:_dataset_rows(Xa)[1]]


def trapz(Y, X):
    """trapz(Y, X) (integrate.nim:104-117): the last cumulative value."""
    return cumtrapz(Y, X)[-1]

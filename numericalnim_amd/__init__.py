"""numericalnim_amd — MI355X (gfx950) native batched ODE backend for numericalnim.

The product is the C-ABI library `csrc/libnnhip_ode.so` (include/nnhip_ode.h).  This package is the thin
host-side mirror of the reference's `solveODE` / `ODEoptions` / `NumContext` interface
(/root/reference/src/numericalnim/ode.nim:26-34,78-104,589-651; common/commonTypes.nim:4-39) on top of
it, used by the tests and bench.py.  There is no CPU fallback: without the built HIP library every
call raises.
"""
from .ode import (  # noqa: F401
    ODEoptions, newODEoptions, NumContext, newNumContext, Rhs, solveODE, integratorStep, fixedStream, fixedStreamSolve, adaptiveStream, adaptiveStreamSolve, solveODEPerIvpEnd, solveODEPerIvpTspan, solveODECalls, solveODECallsTspan,
    fixedODE, adaptiveODE, allODE, implementedODE, LAYOUT_SOA, LAYOUT_AOS, NnhipError, hostLibmMatchesDevicePow, tuning, tuneGet,
)
from .interpolate import newHermiteSpline, HermiteSpline, rhsBatch, cumtrapz, cumsimpson, trapz, sortAndTrimDataset  # noqa: F401
from . import _lib  # noqa: F401


def __getattr__(name):
    if name == "DEFAULT_ODEoptions":
        from . import ode
        return ode._default_options()
    raise AttributeError(name)

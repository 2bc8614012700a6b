"""Host-side mirror of the reference's ODE interface over the C ABI.

Reference (Nim)                                        here
  newODEoptions(dt, absTol, ...)  ode.nim:78-102        newODEoptions(...)        -> ValueError on bad input
  DEFAULT_ODEoptions              ode.nim:104           DEFAULT_ODEoptions
  NumContext[T, float]            commonTypes.nim:4-39  NumContext (fValues / tValues, [] / setF / getF)
  solveODE(f, y0, tspan, options, ctx, integrator)      solveODE(f, y0, tspan, options, ctx, integrator)
                                  ode.nim:589-651          f: Rhs (compiled-in device RHS + names of its ctx.fValues)
                                                           y0: batch of initial states (torch CUDA tensor or numpy)
  IntegratorProc call             ode.nim:38,531        integratorStep(...)       one step over a device batch
  fixedODE / adaptiveODE / allODE ode.nim:40-42         same names

A batch y0 is [N] (scalar `float` states), [dim, N] (layout SoA) or [N, dim] (layout AoS).
torch CUDA tensors stay on the device (device-pointer entry points, torch's current stream);
numpy arrays go through the host-pointer entry point.  Nothing here computes on the CPU.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib
from ._lib import Options as ODEoptions, Stats

LAYOUT_SOA, LAYOUT_AOS = 0, 1

fixedODE = ["heun2", "ralston2", "kutta3", "heun3", "ralston3", "ssprk3", "ralston4", "kutta4", "rk4"]  # ode.nim:40
adaptiveODE = ["rk21", "bs32", "dopri54", "tsit54", "vern65"]  # ode.nim:41
allODE = fixedODE + adaptiveODE  # ode.nim:42
implementedODE = list(allODE)  # every integrator of the reference has HIP kernels


class NnhipError(RuntimeError):
    pass


def _check(rc):
    if rc == _lib.NNHIP_OK:
        return
    msg = _lib.last_error()
    if rc in (_lib.NNHIP_EVALUE, _lib.NNHIP_EINTEGRATOR):
        raise ValueError(msg)  # the reference raises ValueError (ode.nim:95-100,651; utils.nim:26)
    if rc == _lib.NNHIP_EUNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == _lib.NNHIP_ENOMEM:
        raise MemoryError(msg)
    raise NnhipError(f"nnhip error {rc}: {msg}")


def newODEoptions(dt=1e-4, absTol=1e-4, relTol=1e-4, dtMax=1e-2, dtMin=1e-4, scaleMax=4.0, scaleMin=0.1, tStart=0.0):
    """ode.nim:78-102 (same argument order and defaults)."""
    o = ODEoptions()
    _check(_lib.lib().nnhip_ode_new_options(C.byref(o), dt, absTol, relTol, dtMax, dtMin, scaleMax, scaleMin, tStart))
    return o


_DEFAULT = None


def _default_options():
    """DEFAULT_ODEoptions (ode.nim:104), built on first use (needs the C library)."""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = newODEoptions()
    return _DEFAULT


def __getattr__(name):  # module attribute DEFAULT_ODEoptions, evaluated lazily
    if name == "DEFAULT_ODEoptions":
        return _default_options()
    raise AttributeError(name)


class NumContext:
    """commonTypes.nim:4-39: two string-keyed tables; enum keys are stringified."""

    def __init__(self, fValues=None, tValues=None):
        self.fValues = dict(fValues or {})
        self.tValues = dict(tValues or {})

    def __getitem__(self, key):
        return self.tValues[str(getattr(key, "name", key))]

    def __setitem__(self, key, val):
        self.tValues[str(getattr(key, "name", key))] = val

    def getF(self, key):
        return self.fValues[str(getattr(key, "name", key))]

    def setF(self, key, val):
        self.fValues[str(getattr(key, "name", key))] = val


def newNumContext(fValues=None, tValues=None):
    return NumContext(fValues, tValues)


class Rhs:
    """A compiled-in device RHS standing in for the user closure `f(t, y, ctx)` (ODEProc[T], ode.nim:36).

    kind: nnhip_rhs_kind; keys: names looked up in ctx.fValues (in this order) to build rhs_params;
    defaults: values used for keys absent from ctx."""
    NEG_Y, LINEAR, LORENZ, RING, AFFINE_T, VANDERPOL = range(6)
    _compiled = {}  # (dim, body, n_params, per_component) -> rhs_kind of Rhs.custom

    def __init__(self, kind, keys=(), defaults=None):
        self.kind = kind
        self.keys = tuple(keys)
        self.defaults = dict(defaults or {})

    def params(self, ctx):
        if len(self.keys) > 8 and getattr(self, "ctx_layout", None) is not None:
            return []  # more scalars than travel as kernel arguments: they lead the shared block of the context (bind)
        out = []
        for k in self.keys:
            if ctx is not None and k in ctx.fValues:
                out.append(float(ctx.fValues[k]))
            elif k in self.defaults:
                out.append(float(self.defaults[k]))
            else:
                raise KeyError(f"ctx.fValues has no '{k}' for this RHS")  # Nim: KeyError from Table lookup
        return out

    @staticmethod
    def custom(dim, body, keys=(), defaults=None, name="user", per_component=False, tvalues=None, per_ivp=(), n_aux=0, aux_key="aux", halo=None):
        """A user right-hand side from HIP C++ source (compiled at run time with hiprtc; include/nnhip_ode.h,
        nnhip_ode_rhs_compile).  `body` sees t, y[dim], dy[dim] and p[len(keys)] — e.g. for a damped oscillator
        Rhs.custom(2, "dy[0] = y[1]; dy[1] = -p[0]*y[0] - p[1]*y[1];", keys=("k", "c")).
        per_component=True: `body` returns dy_c for the component index `c` (nnhip_ode_rhs_compile_comp); systems of
        8 / 16 / 32 components then run on the lanes-per-system (LDS-staged) kernels.
        halo=(lo, hi) (per-component bodies): the body reads components c - lo .. c + hi of its system only (cyclically:
        y[(c + 1) % dim] is one to the right) — stencils, rings, banded couplings.  Systems whose size is a power of two then take their
        neighbours from the adjacent lanes (DPP) instead of the LDS stage vector (nnhip_ode_rhs_set_halo): the same bits, up to 2x faster.
        The declaration is checked once against the undeclared form on a random batch (a body that reads outside its window raises).

        NumContext in full (commonTypes.nim:4-27; nnhip_ode_rhs_compile_ctx) — any of the following makes it a right-hand side with
        a context layout, bound to the `ctx` given to solveODE & co. at every call:
          keys       more than 8 of them (ctx.fValues, any count): p[k] as before
          tvalues    mapping NAME -> length: the ctx.tValues entries the body reads, as NAME[j]; names in `per_ivp` are per-IVP
                     vectors (ctx.tValues[NAME] is a [length, N] array / CUDA tensor: column i belongs to IVP i, the batch form of N
                     reference calls with their own ctx), the others are shared by the batch (1-D, `length` entries)
          n_aux      per-IVP mutable doubles aux(j) the body may read and write (the mutable ctx of ode.nim:599); they live in
                     ctx.tValues[aux_key], a [n_aux, N] float64 CUDA tensor updated in place"""
        tvalues = dict(tvalues or {})
        per_ivp = tuple(per_ivp)
        has_ctx = bool(tvalues) or n_aux > 0 or len(keys) > 8
        for nm in per_ivp:
            if nm not in tvalues:
                raise ValueError(f"per_ivp names '{nm}', which tvalues does not declare")
        if halo is not None:
            halo = (int(halo[0]), int(halo[1]))
            if not per_component:
                raise ValueError("halo needs per_component=True")
        key = (int(dim), body, len(keys), bool(per_component), tuple(tvalues.items()), per_ivp, int(n_aux), halo)
        k = Rhs._compiled.get(key)  # the same source is registered (and compiled) once per process
        if k is not None and not _lib.lib().nnhip_ode_supported(0, k, int(dim), LAYOUT_SOA, 0):
            k = None  # released in the meantime (nnhip_ode_rhs_release)
        if k is None:
            kind = C.c_int(0)
            if has_ctx:
                names = list(tvalues.keys())
                nv = len(names)
                c_names = (C.c_char_p * max(nv, 1))(*[n.encode() for n in names])
                c_lens = (C.c_int64 * max(nv, 1))(*[int(tvalues[n]) for n in names])
                c_per = (C.c_int * max(nv, 1))(*[1 if n in per_ivp else 0 for n in names])
                _check(_lib.lib().nnhip_ode_rhs_compile_ctx(str(name).encode(), int(dim), len(keys), body.encode(), 1 if per_component else 0, nv,
                                                            c_names, c_lens, c_per, int(n_aux), C.byref(kind)))
            else:
                fn = _lib.lib().nnhip_ode_rhs_compile_comp if per_component else _lib.lib().nnhip_ode_rhs_compile
                _check(fn(str(name).encode(), int(dim), len(keys), body.encode(), C.byref(kind)))
            k = kind.value
            if halo is not None:
                _check(_lib.lib().nnhip_ode_rhs_set_halo(k, halo[0], halo[1]))
                if not has_ctx:
                    Rhs._check_halo(k, int(dim), body, keys, defaults, name)
            Rhs._compiled[key] = k
        r = Rhs(k, keys, defaults)
        r.dim = int(dim)
        if has_ctx:
            r.ctx_layout = dict(tvalues=tvalues, per_ivp=per_ivp, n_aux=int(n_aux), aux_key=aux_key)
        return r

    @staticmethod
    def _check_halo(kind, dim, body, keys, defaults, name):
        """A declared halo against the undeclared form of the same body: three RK4 steps of 64 random systems through both must agree bit for
        bit (they evaluate the same expression on the same values unless the body reads outside its window).  Needs a GPU; skipped without.
        Not run for a right-hand side with a context block (its vectors have no stand-in values here): there the halo is the caller's promise."""
        try:
            import torch
            if not torch.cuda.is_available():
                return
        except ImportError:
            return
        rng = np.random.default_rng(20260928)
        # every declared key gets a value for the check (the caller's default where there is one): the check must not depend on defaults
        syn = {k: (defaults or {}).get(k, float(rng.uniform(0.5, 1.5))) for k in keys}
        try:
            plain = Rhs.custom(dim, body, keys=keys, defaults=syn, name=str(name) + "_nohalo", per_component=True)
            declared = Rhs(kind, keys, syn)
            declared.dim = dim
            y0 = torch.from_numpy(rng.uniform(-1.0, 1.0, (64, dim))).cuda()
            opt = newODEoptions(dt=1e-3)
            a = solveODE(declared, y0, [0.0, 3e-3], opt, integrator="rk4", layout=LAYOUT_AOS)[1]
            b = solveODE(plain, y0, [0.0, 3e-3], opt, integrator="rk4", layout=LAYOUT_AOS)[1]
            same = torch.equal(a, b)
        except Exception:
            _lib.lib().nnhip_ode_rhs_release(kind)  # a body that does not compile against the window, a launch failure: the kind is not kept
            raise
        if not same:
            _lib.lib().nnhip_ode_rhs_release(kind)
            raise ValueError("halo=(lo, hi) does not cover what the body reads: the banded form and the plain form of this right-hand side disagree")

    def bind(self, ctx, device=None):
        """Binds `ctx` to this right-hand side's context layout (nnhip_ode_rhs_bind_ctx_f64_dev): the closure capturing its ctx.
        Called by solveODE & co. with the ctx they are given; arrays are uploaded — to `device`, the device of the batch being solved —
        and CUDA tensors are used in place (a tensor on another device than the batch is refused).
        The binding belongs to the calling THREAD (the library keeps one per thread and rhs_kind): concurrent solves of the SAME source with
        DIFFERENT contexts from different host threads each read their own."""
        lay = getattr(self, "ctx_layout", None)
        if lay is None:
            return
        import torch
        if ctx is None:
            raise ValueError("this right-hand side reads a context block: pass ctx")
        tv = ctx.tValues
        dev = device if (device is not None and device.type == "cuda") else None
        for v in tv.values():
            if _is_torch(v) and v.is_cuda:
                if dev is not None and v.device != dev and (dev.index is not None or v.device.index != torch.cuda.current_device()):
                    raise ValueError(f"ctx tensor on {v.device}, batch on {dev}: the context block must live on the device that integrates")
                dev = dev or v.device
        dev = dev or torch.device("cuda", torch.cuda.current_device())

        def to_dev(v):
            if _is_torch(v):
                return v.to(device=dev, dtype=torch.float64).contiguous()
            return torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float64))).to(dev)

        shared_parts, ivp_parts, stride = [], [], 0
        if len(self.keys) > 8:
            shared_parts.append(to_dev(np.asarray(Rhs.params_all(self, ctx), dtype=np.float64)))
        for nm, ln in lay["tvalues"].items():
            if nm not in tv:
                raise KeyError(f"ctx.tValues has no '{nm}' for this RHS")
            t = to_dev(tv[nm])
            if nm in lay["per_ivp"]:
                if t.dim() != 2 or t.shape[0] != ln:
                    raise ValueError(f"ctx.tValues['{nm}'] must be [{ln}, N]")
                if stride and t.shape[1] != stride:
                    raise ValueError("per-IVP vectors of one ctx must agree in N")
                stride = int(t.shape[1])
                ivp_parts.append(t)
            else:
                if t.numel() != ln:
                    raise ValueError(f"ctx.tValues['{nm}'] must have {ln} entries")
                shared_parts.append(t.reshape(-1))
        aux = None
        if lay["n_aux"] > 0:
            aux = tv.get(lay["aux_key"])
            if aux is None or not _is_torch(aux) or not aux.is_cuda or aux.dtype != torch.float64 or not aux.is_contiguous() or aux.dim() != 2 \
                    or aux.shape[0] != lay["n_aux"]:
                raise ValueError(f"ctx.tValues['{lay['aux_key']}'] must be a contiguous float64 CUDA tensor [{lay['n_aux']}, N] (it is updated in place)")
            if stride and aux.shape[1] != stride:
                raise ValueError("aux and the per-IVP vectors of one ctx must agree in N")
            stride = int(aux.shape[1])
        shared = torch.cat(shared_parts) if len(shared_parts) > 1 else (shared_parts[0] if shared_parts else None)
        ivp = torch.cat(ivp_parts, dim=0) if len(ivp_parts) > 1 else (ivp_parts[0] if ivp_parts else None)
        _check(_lib.lib().nnhip_ode_rhs_bind_ctx_f64_dev(self.kind, shared.data_ptr() if shared is not None else None, int(shared.numel()) if shared is not None else 0,
                                                         ivp.data_ptr() if ivp is not None else None, int(ivp.shape[0]) if ivp is not None else 0,
                                                         aux.data_ptr() if aux is not None else None, lay["n_aux"], stride))
        # Keeps the device memory alive while calls are in flight — per THREAD, like the library's binding: one Rhs object is shared by the threads
        # that solve its source, and a slot on the object itself let thread A's bind free the tensors thread B's binding still pointed at (found on
        # the fake node, whose allocation registry refused the kernel's read of the freed block: tests/test_ctx_block.py, two-thread test).
        tls = self.__dict__.get("_bound_tls")
        if tls is None:
            tls = self.__dict__.setdefault("_bound_tls", threading.local())
        tls.bound = (shared, ivp, aux)

    def params_all(self, ctx):
        out = []
        for k in self.keys:
            if ctx is not None and k in ctx.fValues:
                out.append(float(ctx.fValues[k]))
            elif k in self.defaults:
                out.append(float(self.defaults[k]))
            else:
                raise KeyError(f"ctx.fValues has no '{k}' for this RHS")
        return out

    @staticmethod
    def neg_y():  # dy = -y (ode.nim:16-17)
        return Rhs(Rhs.NEG_Y)

    @staticmethod
    def linear(a=None):  # dy = a*y (tests/test_ode.nim:5-7 uses a = -0.1)
        return Rhs(Rhs.LINEAR, ("a",), {} if a is None else {"a": a})

    @staticmethod
    def lorenz(sigma=10.0, rho=28.0, beta=8.0 / 3.0):
        return Rhs(Rhs.LORENZ, ("sigma", "rho", "beta"), {"sigma": sigma, "rho": rho, "beta": beta})

    @staticmethod
    def ring(c=0.1):
        return Rhs(Rhs.RING, ("c",), {"c": c})

    @staticmethod
    def affine_t(a, b):
        return Rhs(Rhs.AFFINE_T, ("a", "b"), {"a": a, "b": b})

    @staticmethod
    def vanderpol(mu=1.0):
        return Rhs(Rhs.VANDERPOL, ("mu",), {"mu": mu})


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _shape_info(y0, layout):
    shp = tuple(y0.shape)
    if len(shp) == 1:
        return shp[0], 1, True
    if len(shp) != 2:
        raise ValueError("y0 must be [N], [dim, N] (SoA) or [N, dim] (AoS)")
    if layout == LAYOUT_SOA:
        return shp[1], shp[0], False
    return shp[0], shp[1], False


def _require_state(name, x, like=None):
    """A device batch handed to a per-step seam: a float64 tensor on a HIP device (the kernels read 8 bytes per value from the pointer they are given — a float32
    or CPU tensor would be misread, not converted), and — for a companion of `like` (FSAL, scratch) — of the same shape on the same device."""
    import torch
    if not _is_torch(x) or not x.is_cuda or x.dtype != torch.float64:
        raise ValueError(f"{name} must be a float64 tensor on a CUDA/HIP device")
    if like is not None and (tuple(x.shape) != tuple(like.shape) or x.device != like.device):
        raise ValueError(f"{name} must have y's shape {tuple(like.shape)} and live on y's device")


def _params_array(f, ctx, like=None):
    """like: the state batch of the call (its device is where a context block has to live)"""
    if getattr(f, "ctx_layout", None) is not None:
        f.bind(ctx, like.device if (like is not None and _is_torch(like)) else None)  # NumContext.tValues / many fValues / mutable slots: the closure captures its ctx
    elif ctx is None:  # parameters from the defaults only: marshalled once per Rhs object
        cached = getattr(f, "_marshalled", None)
        if cached is None or cached[0] != (f.kind, f.keys, tuple(sorted(f.defaults.items()))):
            p = np.asarray(f.params(None), dtype=np.float64)
            p.flags.writeable = False  # handed out by reference on every call: a caller's write would corrupt later calls
            cached = f._marshalled = ((f.kind, f.keys, tuple(sorted(f.defaults.items()))), p, (p.ctypes.data_as(C.POINTER(C.c_double)) if p.size else None))
        return cached[1], cached[2]
    p = np.asarray(f.params(ctx), dtype=np.float64)
    return p, (p.ctypes.data_as(C.POINTER(C.c_double)) if p.size else None)


_INTEG = {}  # name as given -> (id, useFSAL, adaptive): the per-step seams are called in loops, a C call per lookup is host time


def _integ_info(integrator):
    info = _INTEG.get(integrator)
    if info is None:
        L = _lib.lib()
        rc = L.nnhip_ode_integrator_id(str(integrator).encode())
        if rc < 0:
            raise ValueError(f"{integrator} is not a valid integrator")  # ode.nim:651
        use_fsal, adaptive = C.c_int(), C.c_int()
        _check(L.nnhip_ode_integrator_traits(rc, C.byref(use_fsal), None, C.byref(adaptive)))
        info = _INTEG[integrator] = (rc, bool(use_fsal.value), bool(adaptive.value))
    return info


def integrator_id(integrator):
    return _integ_info(integrator)[0]


def _device_scope(dev):
    """The stream a launch on `dev` goes to, and a context manager only when `dev` is not the current device (entering torch.cuda.device costs
    ~10 us of host time per call; a per-step driver calls the seams thousands of times)."""
    import torch
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if idx == torch.cuda.current_device():
        return None, (raw(idx) if raw is not None else torch.cuda.current_stream().cuda_stream)
    return torch.cuda.device(idx), None


def solveODE(f, y0, tspan, options=None, ctx=None, integrator="dopri54", layout=LAYOUT_SOA, max_steps=0, stats=None,
             return_counts=False, sweep=None, sort_by=None, out=None, probe_steps=0):
    """Batched solveODE (ode.nim:589-651): returns (t, y) with t = the sorted output grid (ndarray) and
    y = [n_t, *y0.shape] holding the state of every IVP at every t (rows the reference would not
    return for an IVP are NaN; see include/nnhip_ode.h).

    return_counts=True appends a dict(ny, steps, rejected) of per-IVP int arrays/tensors.
    sweep: optional CUDA tensor [k, N] of PER-IVP values for the first k RHS parameters (a parameter sweep: IVP i is
    integrated with parameters sweep[:, i], exactly as if it were its own solveODE call with its own ctx).
    sort_by: optional CUDA float64 tensor [N], or the string "auto"; the batch is integrated in ascending (binned) order of sort_by and
    every result is written at the IVP's own index (nnhip_ode_solve_batch_sorted_f64_dev: the ordering lives below the C ABI).
    Results are bit-identical; neighbouring lanes of a wavefront then take similar step sequences, which removes most of the
    divergence of adaptive methods on heterogeneous batches (1.9x on a Van der Pol mu-sweep, scripts/bench_divergence.py).  Put the IVPs
    with the MOST work first (e.g. sort_by = -mu): another 8 % over the same key ascending.
    "auto" ranks the IVPs with a short probe solve (probe_steps accepted steps per IVP; 0 = the library's default) instead of a user key.
    out: optional result array for numpy batches, float64 C-contiguous of shape [len(tspan), *y0.shape], returned as y.  Reusing
    it across calls avoids the first-touch page faults of a fresh 100+ MB array inside the device-to-host copy (2x on C2), and if
    both y0 and out are page-locked the transfers are overlapped with the kernel (another 1.6x)."""
    L = _lib.lib()
    options = options if options is not None else _default_options()
    ctx = ctx if ctx is not None else NumContext()  # ode.nim:604-606
    integ = integrator_id(integrator)
    tspan = np.ascontiguousarray(np.asarray(tspan, dtype=np.float64))
    n_t = int(tspan.size)
    p, pp = _params_array(f, ctx, y0)
    t_out = np.empty(max(n_t, 1), dtype=np.float64)
    tp = t_out.ctypes.data_as(C.POINTER(C.c_double))
    tsp = tspan.ctypes.data_as(C.POINTER(C.c_double))
    N, dim, scalar = _shape_info(y0, layout)
    ntout = C.c_int(0)
    _check(L.nnhip_ode_time_grid(C.byref(options), tsp, n_t, tp, C.byref(ntout)))
    if _is_torch(y0):
        import torch
        if not y0.is_cuda:
            raise ValueError("torch y0 must live on a CUDA/HIP device (numericalnim_amd has no CPU path)")
        if y0.dtype != torch.float64:
            raise ValueError("y0 must be float64")
        if out is not None:
            raise ValueError("out is for numpy batches (torch results already live on the device)")
        y0c = y0.contiguous()
        with torch.cuda.device(y0c.device):
            y = torch.empty((n_t,) + tuple(y0c.shape), dtype=torch.float64, device=y0c.device)
            ny = torch.empty(N, dtype=torch.int32, device=y0c.device) if return_counts else None
            st = torch.empty(N, dtype=torch.int64, device=y0c.device) if return_counts else None
            rj = torch.empty(N, dtype=torch.int64, device=y0c.device) if return_counts else None
            wsb = int(L.nnhip_ode_solve_workspace_bytes(n_t))
            ws = torch.empty(wsb, dtype=torch.uint8, device=y0c.device)
            stream = torch.cuda.current_stream().cuda_stream
            sw = None
            if sweep is not None:
                sw = sweep.contiguous()
                if sw.dim() != 2 or sw.shape[1] != N or sw.dtype != torch.float64 or not sw.is_cuda:
                    raise ValueError("sweep must be a CUDA float64 tensor of shape [k, N]")
            if sort_by is not None:
                auto = isinstance(sort_by, str)
                if auto and sort_by != "auto":
                    raise ValueError('sort_by must be a CUDA float64 tensor of shape [N] or "auto"')
                key = None
                if not auto:
                    key = sort_by.contiguous()
                    if key.dim() != 1 or key.shape[0] != N or key.dtype != torch.float64 or not key.is_cuda:
                        raise ValueError("sort_by must be a CUDA float64 tensor of shape [N]")
                wsb = int(L.nnhip_ode_solve_sorted_workspace_bytes(N, n_t))
                ws = torch.empty(wsb, dtype=torch.uint8, device=y0c.device)
                _check(L.nnhip_ode_solve_batch_sorted_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), sw.data_ptr() if sw is not None else None,
                                                              int(sw.shape[0]) if sw is not None else 0, y0c.data_ptr(), N, dim, layout, tsp, n_t, tp,
                                                              y.data_ptr(), ny.data_ptr() if return_counts else None,
                                                              st.data_ptr() if return_counts else None, rj.data_ptr() if return_counts else None,
                                                              int(max_steps), key.data_ptr() if key is not None else None, int(probe_steps), ws.data_ptr(), wsb, stream))
            elif sweep is not None:
                _check(L.nnhip_ode_solve_batch_sweep_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), sw.data_ptr(), int(sw.shape[0]),
                                                             y0c.data_ptr(), N, dim, layout, tsp, n_t, tp, y.data_ptr(),
                                                             ny.data_ptr() if return_counts else None, st.data_ptr() if return_counts else None,
                                                             rj.data_ptr() if return_counts else None, int(max_steps), ws.data_ptr(), wsb, stream))
            else:
                _check(L.nnhip_ode_solve_batch_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), y0c.data_ptr(), N, dim, layout,
                                                       tsp, n_t, tp, y.data_ptr(), ny.data_ptr() if return_counts else None,
                                                       st.data_ptr() if return_counts else None,
                                                       rj.data_ptr() if return_counts else None, int(max_steps), ws.data_ptr(), wsb,
                                                       stream))
    else:
        y0c = np.ascontiguousarray(np.asarray(y0, dtype=np.float64))
        if out is not None:
            if not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.flags.c_contiguous and out.shape == (n_t,) + y0c.shape):
                raise ValueError(f"out must be a C-contiguous float64 ndarray of shape {(n_t,) + y0c.shape}")
            y = out
        else:
            y = np.empty((n_t,) + y0c.shape, dtype=np.float64)
        ny = np.empty(N, dtype=np.int32) if return_counts else None
        st = np.empty(N, dtype=np.int64) if return_counts else None
        rj = np.empty(N, dtype=np.int64) if return_counts else None
        s = stats if stats is not None else Stats()
        swh, kh = None, 0
        if sweep is not None:  # per-IVP parameters in host memory
            swa = np.ascontiguousarray(np.asarray(sweep, dtype=np.float64))
            if swa.ndim != 2 or swa.shape[1] != N:
                raise ValueError("sweep must have shape [k, N]")
            swh, kh = swa.ctypes.data, int(swa.shape[0])
        if sort_by is not None:  # host-pointer form of the sorted solve (the entry a Nim host holding seqs would call)
            keyh = None
            if not isinstance(sort_by, str):
                keya = np.ascontiguousarray(np.asarray(sort_by, dtype=np.float64))
                if keya.shape != (N,):
                    raise ValueError("sort_by must have shape [N]")
                keyh = keya.ctypes.data
            elif sort_by != "auto":
                raise ValueError('sort_by must be an array of shape [N] or "auto"')
            _check(L.nnhip_ode_solve_batch_sorted_f64(C.byref(options), integ, f.kind, pp, int(p.size), swh, kh, y0c.ctypes.data, N, dim, layout, tsp, n_t, tp,
                                                      y.ctypes.data, ny.ctypes.data if return_counts else None, st.ctypes.data if return_counts else None,
                                                      rj.ctypes.data if return_counts else None, int(max_steps), keyh, int(probe_steps), 0))
            t = t_out[:ntout.value].copy()
            return (t, y, dict(ny=ny, steps=st, rejected=rj)) if return_counts else (t, y)
        _check(L.nnhip_ode_solve_batch_sweep_f64(C.byref(options), integ, f.kind, pp, int(p.size), swh, kh, y0c.ctypes.data, N, dim, layout, tsp,
                                                 n_t, tp, y.ctypes.data, ny.ctypes.data if return_counts else None,
                                                 st.ctypes.data if return_counts else None, rj.ctypes.data if return_counts else None,
                                                 int(max_steps), C.byref(s), 0))
    t = t_out[:ntout.value].copy()
    if return_counts:
        return t, y, dict(ny=ny, steps=st, rejected=rj)
    return t, y


def integratorStep(f, t, y, FSAL, dt, options=None, ctx=None, integrator="dopri54", layout=LAYOUT_SOA, negate_time=False,
                   out=None):
    """One IntegratorProc call (ode.nim:38) over a device batch: returns (yNew, FSAL', dtUsed, error).

    y / FSAL: torch CUDA float64 batches; t / dt: Python floats (uniform) or [N] CUDA tensors.  out: a preallocated yNew tensor, or a
    tuple (yNew, FSAL', dtUsed, error) of preallocated tensors (None entries are allocated) — a loop that reuses its buffers then
    pays no allocations per call."""
    import torch
    L = _lib.lib()
    options = options if options is not None else _default_options()
    integ, use_fsal, adaptive = _integ_info(integrator)
    _require_state("y", y)
    if FSAL is not None:
        _require_state("FSAL", FSAL, y)
    p, pp = _params_array(f, ctx, y)
    N, dim, scalar = _shape_info(y, layout)
    yc = y if y.is_contiguous() else y.contiguous()
    scope, stream = _device_scope(yc.device)
    if scope is not None:
        scope.__enter__()
    try:
        o = tuple(out) + (None,) * (4 - len(out)) if isinstance(out, (tuple, list)) else (out, None, None, None)
        for b in o:
            if b is not None and (not b.is_contiguous() or b.dtype != torch.float64 or b.device != yc.device):
                raise ValueError("out buffers must be contiguous float64 tensors on y's device")
        y_new = o[0] if o[0] is not None else torch.empty_like(yc)
        fs_in = (FSAL if FSAL.is_contiguous() else FSAL.contiguous()) if FSAL is not None else None
        fs_new = (o[1] if o[1] is not None else torch.empty_like(yc)) if (use_fsal or FSAL is not None) else None
        t_dev = t.contiguous() if (_is_torch(t) and t.is_cuda) else None
        dt_dev = dt.contiguous() if (_is_torch(dt) and dt.is_cuda) else None
        for nm, v in (("t", t_dev), ("dt", dt_dev)):
            if v is not None and (v.dtype != torch.float64 or v.dim() != 1 or v.shape[0] != N or v.device != yc.device):
                raise ValueError(f"{nm} must be a Python float or a float64 tensor [N] on y's device")
        # host scalars that are not Python floats (np.float32, np.int64, 0-d arrays, 0-d CPU tensors): ctypes refuses them
        if t_dev is None and type(t) is not float:
            t = float(t)
        if dt_dev is None and type(dt) is not float:
            dt = float(dt)
        dt_used = (o[2] if o[2] is not None else torch.empty(N, dtype=torch.float64, device=yc.device)) if adaptive else None
        err = (o[3] if o[3] is not None else torch.empty(N, dtype=torch.float64, device=yc.device)) if adaptive else None
        if stream is None:
            stream = torch.cuda.current_stream().cuda_stream
        rc = L.nnhip_ode_step_batch_f64_dev(options, integ, f.kind, pp, p.size, N, dim, layout,
                                            t_dev.data_ptr() if t_dev is not None else None, 0.0 if t_dev is not None else t,
                                            dt_dev.data_ptr() if dt_dev is not None else None,
                                            0.0 if dt_dev is not None else dt, yc.data_ptr(),
                                            fs_in.data_ptr() if fs_in is not None else None, y_new.data_ptr(),
                                            fs_new.data_ptr() if fs_new is not None else None,
                                            dt_used.data_ptr() if dt_used is not None else None,
                                            err.data_ptr() if err is not None else None, 1 if negate_time else 0, stream)
        if rc:
            _check(rc)
    finally:
        if scope is not None:
            scope.__exit__(None, None, None)
    return y_new, fs_new, dt_used, err


def fixedStream(f, y, t0, tEnd, options=None, ctx=None, integrator="rk4", layout=LAYOUT_SOA, scratch=None):
    """ODESolver's fixed-step loop (ode.nim:511-532) over the step-streaming kernel; y (CUDA tensor) is
    advanced from t0 to tEnd.  Returns (y_final_tensor, n_steps); y_final is `y` or `scratch`."""
    import torch
    L = _lib.lib()
    options = options if options is not None else _default_options()
    integ = integrator_id(integrator)
    _require_state("y", y)
    p, pp = _params_array(f, ctx, y)
    N, dim, scalar = _shape_info(y, layout)
    if not y.is_contiguous():
        raise ValueError("y must be contiguous (it is updated in place)")
    if scratch is not None:
        _require_state("scratch", scratch, y)
        if not scratch.is_contiguous() or scratch.data_ptr() == y.data_ptr():
            raise ValueError("scratch must be a contiguous tensor of its own (the ping-pong partner of y)")
    nsteps = C.c_int64(0)
    yfin = C.c_void_p(0)
    with torch.cuda.device(y.device):
        stream = torch.cuda.current_stream().cuda_stream
        _check(L.nnhip_ode_fixed_stream_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), N, dim, layout, float(t0),
                                                float(tEnd), y.data_ptr(), scratch.data_ptr() if scratch is not None else None,
                                                C.byref(nsteps), C.byref(yfin), stream))
    final = scratch if (scratch is not None and yfin.value == scratch.data_ptr()) else y
    return final, nsteps.value


def solveODEPerIvpEnd(f, y0, t_end, options=None, ctx=None, integrator="dopri54", layout=LAYOUT_SOA, max_steps=0, sweep=None,
                      t_start=None, absTol=None, relTol=None, dtMax=None, dtMin=None, dt=None):
    """A batch of solveODE(f, y0_i, [tStart_i, t_end[i]], options_i, ctx, integrator) calls — every IVP with its own tspan and,
    optionally, its own ODEoptions fields (ode.nim:589-591, 26-34: each reference call owns both).  t_end and the optional
    t_start / absTol / relTol / dtMax / dtMin / dt: CUDA float64 tensors [N].  Returns (y [2, *y0.shape], counts): the two rows the
    reference returns for the sorted 2-point tspan of each IVP; ny = -1 marks a call the reference would have refused
    (nnhip_ode_solve_batch_calls_f64_dev)."""
    import torch
    L = _lib.lib()
    options = options if options is not None else _default_options()
    integ = integrator_id(integrator)
    _require_state("y0", y0)
    p, pp = _params_array(f, ctx, y0)
    N, dim, scalar = _shape_info(y0, layout)
    y0c = y0.contiguous()
    te = t_end.contiguous()
    if te.dim() != 1 or te.shape[0] != N or te.dtype != torch.float64 or not te.is_cuda:
        raise ValueError("t_end must be a CUDA float64 tensor of shape [N]")
    sw = None
    if sweep is not None:
        sw = sweep.contiguous()
        if sw.dim() != 2 or sw.shape[1] != N or sw.dtype != torch.float64 or not sw.is_cuda:
            raise ValueError("sweep must be a CUDA float64 tensor of shape [k, N]")
    with torch.cuda.device(y0c.device):
        y = torch.empty((2,) + tuple(y0c.shape), dtype=torch.float64, device=y0c.device)
        ny = torch.empty(N, dtype=torch.int32, device=y0c.device)
        st = torch.empty(N, dtype=torch.int64, device=y0c.device)
        rj = torch.empty(N, dtype=torch.int64, device=y0c.device)
        opts = []
        for name, v in (("t_start", t_start), ("absTol", absTol), ("relTol", relTol), ("dtMax", dtMax), ("dtMin", dtMin), ("dt", dt)):
            if v is not None:
                v = v.contiguous()
                if v.dim() != 1 or v.shape[0] != N or v.dtype != torch.float64 or not v.is_cuda:
                    raise ValueError(f"{name} must be a CUDA float64 tensor of shape [N]")
            opts.append(v)
        _check(L.nnhip_ode_solve_batch_calls_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), sw.data_ptr() if sw is not None else None,
                                                     int(sw.shape[0]) if sw is not None else 0, y0c.data_ptr(), N, dim, layout, te.data_ptr(),
                                                     *[v.data_ptr() if v is not None else None for v in opts], y.data_ptr(),
                                                     ny.data_ptr(), st.data_ptr(), rj.data_ptr(), int(max_steps), torch.cuda.current_stream().cuda_stream))
    return y, dict(ny=ny, steps=st, rejected=rj)


def solveODEPerIvpTspan(f, y0, tspans, options=None, ctx=None, integrator="dopri54", layout=LAYOUT_SOA, max_steps=0, sweep=None,
                        t_start=None, absTol=None, relTol=None, dtMax=None, dtMin=None, dt=None):
    """A batch of solveODE(f, y0_i, tspans[i], options_i, ctx, integrator) calls — every IVP its own n_t-point tspan (any order, both
    sides of its tStart, duplicates) and, optionally, its own ODEoptions fields (nnhip_ode_solve_batch_tspans_f64_dev).  tspans: CUDA
    float64 tensor [N, n_t].  Returns (t [N, n_t] — row i = the sorted times the reference returns for call i, NaN beyond —,
    y [n_t, *y0.shape], counts); ny = -1 marks a call the reference would refuse."""
    import torch
    L = _lib.lib()
    options = options if options is not None else _default_options()
    integ = integrator_id(integrator)
    _require_state("y0", y0)
    p, pp = _params_array(f, ctx, y0)
    N, dim, scalar = _shape_info(y0, layout)
    y0c = y0.contiguous()
    ts = tspans.contiguous()
    if ts.dim() != 2 or ts.shape[0] != N or ts.dtype != torch.float64 or not ts.is_cuda:
        raise ValueError("tspans must be a CUDA float64 tensor of shape [N, n_t]")
    n_t = int(ts.shape[1])
    sw = None
    if sweep is not None:
        sw = sweep.contiguous()
        if sw.dim() != 2 or sw.shape[1] != N or sw.dtype != torch.float64 or not sw.is_cuda:
            raise ValueError("sweep must be a CUDA float64 tensor of shape [k, N]")
    with torch.cuda.device(y0c.device):
        t_out = torch.empty((N, n_t), dtype=torch.float64, device=y0c.device)
        y = torch.empty((n_t,) + tuple(y0c.shape), dtype=torch.float64, device=y0c.device)
        ny = torch.empty(N, dtype=torch.int32, device=y0c.device)
        st = torch.empty(N, dtype=torch.int64, device=y0c.device)
        rj = torch.empty(N, dtype=torch.int64, device=y0c.device)
        wsb = int(L.nnhip_ode_solve_tspans_workspace_bytes(N, n_t))
        ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=y0c.device)
        opts = []
        for name, v in (("t_start", t_start), ("absTol", absTol), ("relTol", relTol), ("dtMax", dtMax), ("dtMin", dtMin), ("dt", dt)):
            if v is not None:
                v = v.contiguous()
                if v.dim() != 1 or v.shape[0] != N or v.dtype != torch.float64 or not v.is_cuda:
                    raise ValueError(f"{name} must be a CUDA float64 tensor of shape [N]")
            opts.append(v)
        _check(L.nnhip_ode_solve_batch_tspans_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), sw.data_ptr() if sw is not None else None,
                                                      int(sw.shape[0]) if sw is not None else 0, y0c.data_ptr(), N, dim, layout, ts.data_ptr(), n_t,
                                                      *[v.data_ptr() if v is not None else None for v in opts], t_out.data_ptr(), y.data_ptr(),
                                                      ny.data_ptr(), st.data_ptr(), rj.data_ptr(), int(max_steps), ws.data_ptr(), wsb,
                                                      torch.cuda.current_stream().cuda_stream))
    return t_out, y, dict(ny=ny, steps=st, rejected=rj)


def solveODECalls(f, y0, t_end, options=None, ctx=None, integrator="dopri54", layout=LAYOUT_SOA, max_steps=0, sweep=None, device=0):
    """N reference calls `solveODE(f, y0_i, [options_i.tStart, t_end[i]], options_i, ctx, integrator)` in one launch, host arrays in
    and out (nnhip_ode_solve_batch_calls_f64).  y0: numpy [dim, N] (SoA) / [N, dim] (AoS) / [N]; t_end: numpy [N]; options: one
    ODEoptions object, a sequence of N of them, or None.  Returns (y [2, *y0.shape], counts) like solveODEPerIvpEnd."""
    L = _lib.lib()
    integ = integrator_id(integrator)
    p, pp = _params_array(f, ctx, y0)
    y0c = np.ascontiguousarray(np.asarray(y0, dtype=np.float64))
    N, dim, scalar = _shape_info(y0c, layout)
    te = np.ascontiguousarray(np.asarray(t_end, dtype=np.float64))
    if te.shape != (N,):
        raise ValueError("t_end needs one value per IVP")
    each = None
    if options is None:
        base = _default_options()
    elif isinstance(options, _lib.Options):
        base = options
    else:
        options = list(options)
        if len(options) != N:
            raise ValueError("options: one object, or one per IVP")
        base = options[0] if N else _default_options()
        each = (_lib.Options * N)(*options)
    sw = None
    if sweep is not None:
        sw = np.ascontiguousarray(np.asarray(sweep, dtype=np.float64))
        if sw.ndim != 2 or sw.shape[1] != N:
            raise ValueError("sweep must have shape [k, N]")
    y = np.empty((2,) + y0c.shape, dtype=np.float64)
    ny = np.empty(N, dtype=np.int32)
    st = np.empty(N, dtype=np.int64)
    rj = np.empty(N, dtype=np.int64)
    _check(L.nnhip_ode_solve_batch_calls_f64(C.byref(base), C.cast(each, C.c_void_p) if each is not None else None, integ, f.kind, pp, int(p.size),
                                             sw.ctypes.data if sw is not None else None, int(sw.shape[0]) if sw is not None else 0,
                                             y0c.ctypes.data, N, dim, layout, te.ctypes.data, y.ctypes.data, ny.ctypes.data, st.ctypes.data,
                                             rj.ctypes.data, int(max_steps), int(device)))
    return y, dict(ny=ny, steps=st, rejected=rj)


def solveODECallsTspan(f, y0, tspans, options=None, ctx=None, integrator="dopri54", layout=LAYOUT_SOA, max_steps=0, sweep=None, device=0):
    """N reference calls `solveODE(f, y0_i, tspans[i], options_i, ctx, integrator)` in one launch, host arrays in and out
    (nnhip_ode_solve_batch_tspans_f64).  y0: numpy; tspans: numpy [N, n_t]; options: one ODEoptions object, a sequence of N, or None.
    Returns (t [N, n_t], y [n_t, *y0.shape], counts) like solveODEPerIvpTspan."""
    L = _lib.lib()
    integ = integrator_id(integrator)
    p, pp = _params_array(f, ctx, y0)
    y0c = np.ascontiguousarray(np.asarray(y0, dtype=np.float64))
    N, dim, scalar = _shape_info(y0c, layout)
    ts = np.ascontiguousarray(np.asarray(tspans, dtype=np.float64))
    if ts.ndim != 2 or ts.shape[0] != N:
        raise ValueError("tspans must have shape [N, n_t]")
    n_t = int(ts.shape[1])
    each = None
    if options is None:
        base = _default_options()
    elif isinstance(options, _lib.Options):
        base = options
    else:
        options = list(options)
        if len(options) != N:
            raise ValueError("options: one object, or one per IVP")
        base = options[0] if N else _default_options()
        each = (_lib.Options * N)(*options)
    sw = None
    if sweep is not None:
        sw = np.ascontiguousarray(np.asarray(sweep, dtype=np.float64))
        if sw.ndim != 2 or sw.shape[1] != N:
            raise ValueError("sweep must have shape [k, N]")
    t_out = np.empty((N, n_t), dtype=np.float64)
    y = np.empty((n_t,) + y0c.shape, dtype=np.float64)
    ny = np.empty(N, dtype=np.int32)
    st = np.empty(N, dtype=np.int64)
    rj = np.empty(N, dtype=np.int64)
    _check(L.nnhip_ode_solve_batch_tspans_f64(C.byref(base), C.cast(each, C.c_void_p) if each is not None else None, integ, f.kind, pp, int(p.size),
                                              sw.ctypes.data if sw is not None else None, int(sw.shape[0]) if sw is not None else 0,
                                              y0c.ctypes.data, N, dim, layout, ts.ctypes.data, n_t, t_out.ctypes.data, y.ctypes.data, ny.ctypes.data,
                                              st.ctypes.data, rj.ctypes.data, int(max_steps), int(device)))
    return t_out, y, dict(ny=ny, steps=st, rejected=rj)


def fixedStreamSolve(f, y0, tspan, options=None, ctx=None, integrator="rk4", layout=LAYOUT_SOA, max_steps=0):
    """solveODE (ode.nim:589-651) for a fixed-step integrator THROUGH THE IntegratorProc SEAM: the whole ODESolver driver — both
    directions, dense Hermite rows, output assembly — over the step-streaming kernels, state in HBM between steps
    (nnhip_ode_fixed_stream_dense_f64_dev).  Returns (t, y [n_t, *y0.shape], ny, n_steps); bitwise equal to solveODE.  When max_steps ends a
    direction short of its end time a RuntimeWarning is issued (the last row is then the state reached, as with solveODE's stats.truncated)."""
    import torch
    L = _lib.lib()
    options = options if options is not None else _default_options()
    integ = integrator_id(integrator)
    _require_state("y0", y0)
    p, pp = _params_array(f, ctx, y0)
    N, dim, scalar = _shape_info(y0, layout)
    tspan = np.ascontiguousarray(np.asarray(tspan, dtype=np.float64))
    n_t = int(tspan.size)
    t_out = np.empty(max(n_t, 1), dtype=np.float64)
    ntout = C.c_int(0)
    dpt = C.POINTER(C.c_double)
    _check(L.nnhip_ode_time_grid(C.byref(options), tspan.ctypes.data_as(dpt), n_t, t_out.ctypes.data_as(dpt), C.byref(ntout)))
    y0c = y0.contiguous()
    ny = C.c_int(0)
    ns = C.c_int64(0)
    with torch.cuda.device(y0c.device):
        y = torch.empty((n_t,) + tuple(y0c.shape), dtype=torch.float64, device=y0c.device)
        wsb = int(L.nnhip_ode_fixed_stream_dense_workspace_bytes(N, dim))
        ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=y0c.device)
        rc = L.nnhip_ode_fixed_stream_dense_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), y0c.data_ptr(), N, dim, layout,
                                                    tspan.ctypes.data_as(dpt), n_t, t_out.ctypes.data_as(dpt), y.data_ptr(), C.byref(ny), int(max_steps),
                                                    ws.data_ptr(), wsb, C.byref(ns), torch.cuda.current_stream().cuda_stream)
        if rc == _lib.NNHIP_TRUNCATED:
            import warnings
            warnings.warn(_lib.last_error(), RuntimeWarning)
        else:
            _check(rc)
    return t_out[:ntout.value].copy(), y, ny.value, ns.value


def adaptiveStreamSolve(f, y0, tspan, options=None, ctx=None, integrator="dopri54", layout=LAYOUT_SOA, check_every=0, max_launches=0):
    """solveODE (ode.nim:589-651) for an adaptive integrator THROUGH THE IntegratorProc SEAM: the whole ODESolver driver over the
    HBM-resident advance kernel, requested rows interpolated inside the launch that steps past them (nnhip_ode_adaptive_stream_dense_f64_dev).
    Returns (t, y, ny, launches); bitwise equal to solveODE.  max_launches > 0 bounds each direction's loop exactly as solveODE's
    max_steps does (a warning is issued when it cut an integration short).  check_every <= 0: polling groups of 8 launches, or the
    library's own schedule under tuning(adv_auto_poll=1)."""
    import torch
    L = _lib.lib()
    options = options if options is not None else _default_options()
    integ = integrator_id(integrator)
    _require_state("y0", y0)
    p, pp = _params_array(f, ctx, y0)
    N, dim, scalar = _shape_info(y0, layout)
    tspan = np.ascontiguousarray(np.asarray(tspan, dtype=np.float64))
    n_t = int(tspan.size)
    t_out = np.empty(max(n_t, 1), dtype=np.float64)
    ntout = C.c_int(0)
    dpt = C.POINTER(C.c_double)
    _check(L.nnhip_ode_time_grid(C.byref(options), tspan.ctypes.data_as(dpt), n_t, t_out.ctypes.data_as(dpt), C.byref(ntout)))
    y0c = y0.contiguous()
    nl = C.c_int64(0)
    with torch.cuda.device(y0c.device):
        y = torch.empty((n_t,) + tuple(y0c.shape), dtype=torch.float64, device=y0c.device)
        ny = torch.empty(max(N, 1), dtype=torch.int32, device=y0c.device)
        wsb = int(L.nnhip_ode_adaptive_stream_dense_workspace_bytes(N, dim, n_t))
        ws = torch.empty(max(wsb, 8), dtype=torch.uint8, device=y0c.device)
        rc = L.nnhip_ode_adaptive_stream_dense_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), y0c.data_ptr(), N, dim, layout,
                                                       tspan.ctypes.data_as(dpt), n_t, t_out.ctypes.data_as(dpt), y.data_ptr(), ny.data_ptr(),
                                                       ws.data_ptr(), wsb, int(check_every), int(max_launches), C.byref(nl),
                                                       torch.cuda.current_stream().cuda_stream)
        if rc == _lib.NNHIP_TRUNCATED:
            import warnings
            warnings.warn("adaptiveStreamSolve: max_launches ended an integration short of its end time", RuntimeWarning)
        else:
            _check(rc)
    return t_out[:ntout.value].copy(), y, ny[:N], nl.value


def adaptiveStream(f, y, t0, tEnd, options=None, ctx=None, integrator="dopri54", layout=LAYOUT_SOA, check_every=0, steps_per_launch=None):
    """ODESolver's adaptive loop (ode.nim:506-542) over the HBM-resident `advance` kernel; y (CUDA tensor) is advanced
    in place from t0 to tEnd.  Returns (y, number of launches).  Bitwise equal to solveODE(f, y0, [t0, tEnd])[1][-1].
    steps_per_launch (None = leave the process-wide knob "adv_steps_per_launch" as it is, default 1): loop iterations per IVP and
    launch; K > 1 keeps the state in registers for K iterations (same bits, 1/K of the launches, 1/K of the HBM traffic per step).
    check_every <= 0: polling groups of 8 launches, or the library's own schedule under tuning(adv_auto_poll=1); tuning(adv_lean=1)
    selects the lean kernels (same bits), tuning(fp_contract=1) their FMA-contracted build (within 1e-6, not bit-equal)."""
    import torch
    L = _lib.lib()
    options = options if options is not None else _default_options()
    integ = integrator_id(integrator)
    _require_state("y", y)
    p, pp = _params_array(f, ctx, y)
    N, dim, scalar = _shape_info(y, layout)
    if not y.is_contiguous():
        raise ValueError("y must be contiguous (it is updated in place)")
    nl = C.c_int64(0)
    if steps_per_launch is not None:
        _check(L.nnhip_tune_set(b"adv_steps_per_launch", int(steps_per_launch)))
    with torch.cuda.device(y.device):
        wsb = int(L.nnhip_ode_adaptive_stream_workspace_bytes(N, dim))
        ws = torch.empty(wsb, dtype=torch.uint8, device=y.device)
        _check(L.nnhip_ode_adaptive_stream_f64_dev(C.byref(options), integ, f.kind, pp, int(p.size), N, dim, layout, float(t0), float(tEnd),
                                                   y.data_ptr(), ws.data_ptr(), wsb, int(check_every), 0, C.byref(nl),
                                                   torch.cuda.current_stream().cuda_stream))
    return y, nl.value


class tuning:
    """with tuning(adv_lean=1, adv_auto_poll=1): ... — process-wide tuning knobs (nnhip_tune_set) for the duration of a block, then back to what the LIBRARY says they
    were before it (nnhip_tune_get: a knob set by a direct nnhip_tune_set call, or by another block, is restored to that value, not to a default).  Blocks nest.  Any key
    of nnhip_tune_set is accepted.  Opt-in settings of the adaptive streaming loop: adv_lean (its lean kernels, same bits), adv_auto_poll (its own polling schedule when
    check_every <= 0), fp_contract (FMA-contracted kernels: within 1e-10 / 1e-6, not the reference's bits)."""

    def __init__(self, **knobs):
        self.knobs = knobs
        self.before = []

    def __enter__(self):
        L = _lib.lib()
        try:
            for k, v in self.knobs.items():
                old = C.c_int(0)
                rc = L.nnhip_tune_get(k.encode(), C.byref(old))
                if rc:
                    raise ValueError("tuning(): " + _lib.last_error())
                _check(L.nnhip_tune_set(k.encode(), int(v)))
                self.before.append((k, old.value))
        except Exception:
            self.__exit__()
            raise
        return self

    def __exit__(self, *exc):
        L = _lib.lib()
        while self.before:  # in reverse: rk4_stream_vec / _mode switch the automatic choice off, rk4_stream_auto set earlier in the block must come back last
            k, v = self.before.pop()
            L.nnhip_tune_set(k.encode(), v)
        return False


def tuneGet(key):
    """The current value of a process-wide tuning knob (nnhip_tune_get)."""
    v = C.c_int(0)
    if _lib.lib().nnhip_tune_get(str(key).encode(), C.byref(v)):
        raise ValueError(_lib.last_error())
    return v.value


def hostLibmMatchesDevicePow(n=20000, seed=1234):
    """Does THIS host's C library evaluate the controller's pow(1/error, 1/order) (ode.nim:71,537) to the bits the device evaluates?
    The device restates glibc's table-driven pow (x86-64, glibc >= 2.28, the FMA variant its ifunc resolver picks on every CPU with
    FMA3 + AVX2) operation for operation; a host with another libm (musl, macOS, aarch64, an x86-64 CPU without FMA3) rounds a fraction
    of the calls differently, and an adaptive solve can then take a different step sequence on knife-edge steps — inside the 1e-6
    tolerance, but not bit for bit.  Checked on a sample of error norms in the pow's working range, against Python's math.pow (= the
    process's libm).  Needs a device.  Used by smoke() and by the GPU tests to decide whether bit-level comparisons are meaningful."""
    import math
    import torch
    L = _lib.lib()
    rng = np.random.default_rng(seed)
    ok = True
    for order in (2, 3, 5, 6):
        err = np.concatenate([10.0 ** rng.uniform(-6.0, 2.0, n), rng.uniform(0.5, 1.5, n // 4)])
        e = torch.from_numpy(err).cuda()
        out = torch.empty_like(e)
        _check(L.nnhip_ode_controller_factor_f64_dev(order, e.data_ptr(), out.data_ptr(), e.numel(), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        ref = np.array([min(4.0, max(0.125, 0.9 * math.pow(1.0 / x, 1.0 / order))) for x in err])
        ok = ok and bool(np.array_equal(got, ref))
    return ok

"""ctypes binding of include/nnhip_ode.h.  Loading fails loudly if the HIP library is not built."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NNHIP_LIB: developer override used by the A/B scripts (scripts/build_variant.sh builds the same sources with other flags)
SO_PATH = os.environ.get("NNHIP_LIB") or os.path.join(_HERE, "csrc", "libnnhip_ode.so")

NNHIP_OK, NNHIP_EVALUE, NNHIP_EINTEGRATOR, NNHIP_EHIP, NNHIP_EUNSUPPORTED, NNHIP_ENOMEM = 0, -1, -2, -3, -4, -5
NNHIP_TRUNCATED = 1  # not an error: max_steps ended an integration short of tEnd


class Options(C.Structure):  # nnhip_ode_options == ODEoptions (ode.nim:26-34)
    _fields_ = [(n, C.c_double) for n in ("dt", "dtMax", "dtMin", "tStart", "absTol", "relTol", "scaleMax", "scaleMin")]

    def __repr__(self):
        return "ODEoptions(" + ", ".join(f"{n}={getattr(self, n)!r}" for n, _ in self._fields_) + ")"


class Stats(C.Structure):  # nnhip_ode_stats
    _fields_ = [("steps_total", C.c_int64), ("rejected_total", C.c_int64), ("steps_max", C.c_int64), ("n_t_out", C.c_int32),
                ("ny_min", C.c_int32), ("nan_aborts", C.c_int32), ("truncated", C.c_int32), ("kernel_ms", C.c_double)]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); every function include/nnhip_ode.h declares
SIGNATURES = {
    "nnhip_abi_version": (C.c_int, []),
    "nnhip_device_count": (C.c_int, []),
    "nnhip_last_error": (C.c_char_p, []),
    "nnhip_build_info": (C.c_char_p, []),
    "nnhip_rtc_compiler": (C.c_char_p, []),
    "nnhip_release": (C.c_int, []),
    "nnhip_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64]),
    "nnhip_host_free": (C.c_int, [C.c_void_p]),
    "nnhip_tune_set": (C.c_int, [C.c_char_p, C.c_int]),
    "nnhip_tune_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "nnhip_ode_rk4_stream_variant": (C.c_int, [C.c_int64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nnhip_ode_new_options": (C.c_int, [C.POINTER(Options)] + [C.c_double] * 8),
    "nnhip_ode_default_options": (C.c_int, [C.POINTER(Options)]),
    "nnhip_ode_integrator_id": (C.c_int, [C.c_char_p]),
    "nnhip_ode_integrator_name": (C.c_char_p, [C.c_int]),
    "nnhip_ode_integrator_traits": (C.c_int, [C.c_int, C.POINTER(C.c_int), _dp, C.POINTER(C.c_int)]),
    "nnhip_ode_tableau_f64": (C.c_int, [C.c_int, C.c_int, _dp, C.c_int]),
    "nnhip_ode_time_grid": (C.c_int, [C.POINTER(Options), _dp, C.c_int, _dp, C.POINTER(C.c_int)]),
    "nnhip_ode_supported": (C.c_int, [C.c_int] * 5),
    "nnhip_ode_solve_batch_f64": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int,
                                            _dp, C.c_int, _dp, _vp, _vp, _vp, _vp, C.c_int64, C.POINTER(Stats), C.c_int]),
    "nnhip_ode_solve_batch_sweep_f64": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int,
                                                  _dp, C.c_int, _dp, _vp, _vp, _vp, _vp, C.c_int64, C.POINTER(Stats), C.c_int]),
    "nnhip_ode_solve_workspace_bytes": (C.c_int64, [C.c_int]),
    "nnhip_ode_solve_batch_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int64, C.c_int,
                                                C.c_int, _dp, C.c_int, _dp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "nnhip_ode_solve_batch_sweep_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int,
                                                      C.c_int, _dp, C.c_int, _dp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "nnhip_ode_solve_batch_calls_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int,
                                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "nnhip_ode_solve_batch_calls_f64": (C.c_int, [C.POINTER(Options), _vp, C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int,
                                                  _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int]),
    "nnhip_ode_solve_tspans_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "nnhip_ode_solve_batch_tspans_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int,
                                                       _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int64, _vp]),
    "nnhip_ode_solve_batch_tspans_f64": (C.c_int, [C.POINTER(Options), _vp, C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int,
                                                   _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int]),
    "nnhip_ode_solve_batch_tend_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int,
                                                     _vp, _vp, _vp, _vp, _vp, C.c_int64, _vp]),
    "nnhip_ode_bin_order_f64_dev": (C.c_int, [_vp, C.c_int64, _vp, _vp]),
    "nnhip_ode_solve_sorted_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "nnhip_ode_solve_batch_sorted_f64": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int,
                                                   C.c_int, _dp, C.c_int, _dp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int, C.c_int]),
    "nnhip_ode_solve_batch_sorted_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int,
                                                       C.c_int, _dp, C.c_int, _dp, _vp, _vp, _vp, _vp, C.c_int64, _vp, C.c_int, _vp, C.c_int64, _vp]),
    "nnhip_ode_step_batch_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, C.c_int64, C.c_int, C.c_int,
                                               _vp, C.c_double, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    "nnhip_ode_fixed_stream_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, C.c_int64, C.c_int, C.c_int,
                                                 C.c_double, C.c_double, _vp, _vp, C.POINTER(C.c_int64), C.POINTER(_vp), _vp]),
    "nnhip_ode_fixed_stream_dense_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "nnhip_ode_fixed_stream_dense_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int, _dp, C.c_int,
                                                       _dp, _vp, C.POINTER(C.c_int), C.c_int64, _vp, C.c_int64, C.POINTER(C.c_int64), _vp]),
    "nnhip_ode_adaptive_stream_dense_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int, C.c_int]),
    "nnhip_ode_adaptive_stream_dense_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int, _dp, C.c_int,
                                                          _dp, _vp, _vp, _vp, C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_int64), _vp]),
    "nnhip_ode_adaptive_stream_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "nnhip_ode_adaptive_stream_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, C.c_int64, C.c_int, C.c_int,
                                                    C.c_double, C.c_double, _vp, _vp, C.c_int64, C.c_int, C.c_int64, C.POINTER(C.c_int64), _vp]),
    "nnhip_ode_solve_batch_multi_gpu_sweep_f64": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int, _vp, C.c_int64, C.c_int,
                                                            C.c_int, _dp, C.c_int, _dp, _vp, _vp, _vp, _vp, C.c_int64, C.POINTER(Stats), C.c_int]),
    "nnhip_ode_solve_batch_multi_gpu_f64": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, _vp, C.c_int64, C.c_int,
                                                      C.c_int, _dp, C.c_int, _dp, _vp, _vp, C.c_int64, C.POINTER(Stats), C.c_int]),
    "nnhip_allgather_states_f64_dev": (C.c_int, [C.c_int, C.POINTER(_vp), C.POINTER(C.c_int64), C.c_int, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    "nnhip_ode_fixed_stream_multi_gpu_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int, C.c_double,
                                                           C.c_double, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                                           C.POINTER(C.c_int64), C.POINTER(_vp)]),
    "nnhip_ode_solve_batch_multi_gpu_f64_dev": (C.c_int, [C.POINTER(Options), C.c_int, C.c_int, _dp, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int, _dp, C.c_int,
                                                          _dp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.c_int64, C.POINTER(_vp), C.c_int64,
                                                          C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "nnhip_multigpu_last_error": (C.c_char_p, []),
    "nnhip_hermite_spline_f64_dev": (C.c_int, [C.c_double] * 3 + [_vp] * 5 + [C.c_int64, _vp]),
    "nnhip_ode_rhs_compile": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    "nnhip_ode_rhs_compile_comp": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_int)]),
    "nnhip_ode_rhs_compile_ctx": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                            C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "nnhip_ode_rhs_bind_ctx_f64_dev": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int64]),
    "nnhip_ode_rhs_bind_ctx_f64": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_double), C.c_int, C.c_int64, C.c_int]),
    "nnhip_ode_rhs_read_aux_f64": (C.c_int, [C.c_int, C.POINTER(C.c_double)]),
    "nnhip_ode_rhs_set_halo": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "nnhip_ode_rhs_release": (C.c_int, [C.c_int]),
    "nnhip_hermite_spline_eval_batch_f64_dev": (C.c_int, [_dp, C.c_int, _vp, _vp, C.c_int64, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, _vp]),
    "nnhip_hermite_spline_slopes_f64_dev": (C.c_int, [_dp, C.c_int, _vp, C.c_int64, _vp, _vp]),
    "nnhip_sort_and_trim_dataset_f64_dev": (C.c_int, [_dp, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int64, _dp, C.POINTER(C.c_void_p), C.POINTER(C.c_int), _vp]),
    "nnhip_dataset_rows_f64": (C.c_int, [_dp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "nnhip_cumtrapz_batch_f64_dev": (C.c_int, [_dp, C.c_int, _vp, C.c_int64, _vp, _vp]),
    "nnhip_cumsimpson_batch_f64_dev": (C.c_int, [_dp, C.c_int, _vp, C.c_int64, _vp, _vp]),
    "nnhip_cumtrapz_fn_batch_f64_dev": (C.c_int, [C.c_int, _dp, C.c_int, _vp, C.c_int, C.c_int64, C.c_int, C.c_int, _dp, C.c_int, C.c_double, _vp,
                                                  C.POINTER(C.c_int), _vp]),
    "nnhip_cumsimpson_fn_batch_f64_dev": (C.c_int, [C.c_int, _dp, C.c_int, _vp, C.c_int, C.c_int64, C.c_int, C.c_int, _dp, C.c_int, C.c_double, _vp,
                                                    C.POINTER(C.c_int), _vp]),
    "nnhip_cumtrapz_batch_f64": (C.c_int, [_dp, C.c_int, _dp, C.c_int64, _dp, C.c_int]),
    "nnhip_cumsimpson_batch_f64": (C.c_int, [_dp, C.c_int, _dp, C.c_int64, _dp, C.c_int]),
    "nnhip_hermite_spline_eval_batch_f64": (C.c_int, [_dp, C.c_int, _dp, _dp, C.c_int64, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp, C.c_int]),
    "nnhip_cumtrapz_fn_batch_f64": (C.c_int, [C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int64, C.c_int, C.c_int, _dp, C.c_int, C.c_double, _dp,
                                              C.POINTER(C.c_int), C.c_int]),
    "nnhip_cumsimpson_fn_batch_f64": (C.c_int, [C.c_int, _dp, C.c_int, _dp, C.c_int, C.c_int64, C.c_int, C.c_int, _dp, C.c_int, C.c_double, _dp,
                                                C.POINTER(C.c_int), C.c_int]),
    "nnhip_ode_controller_factor_f64_dev": (C.c_int, [C.c_int, _vp, _vp, C.c_int64, _vp]),
    "nnhip_ode_rhs_batch_f64_dev": (C.c_int, [C.c_int, _dp, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_double, _vp, _vp, _vp]),
}

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's).
    A process must use ONE HIP runtime: device pointers and streams handed over from torch are only valid in
    the runtime that created them.  If our library were loaded first it would bind to /opt/rocm's copy and
    torch would then load a second runtime next to it (symptom: hipErrorNoDevice from our launches).  So when
    torch is installed, load ITS runtime first; our NEEDED libamdhip64.so.7 then resolves to it by SONAME.
    Non-Python consumers (the Nim shim, C++) simply use the system runtime the library is linked against."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C numericalnim_amd/csrc`). numericalnim_amd has no CPU fallback.")
        _preload_torch_hip_runtime()
        h = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def last_error():
    return lib().nnhip_last_error().decode()

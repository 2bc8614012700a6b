/*
 * nnhip_ode.h — C ABI of the MI355X-native batched ODE backend for numericalnim.
 *
 * This is the drop-in boundary for the reference's ODE hot path.  The reference (Nim, CPU, one IVP per
 * call) has no FFI today; these are the entry points its Nim side would bind with {.importc, cdecl.}
 * (see INTEGRATION.md for the Nim stub).  Each entry cites the reference interface it replaces,
 * relative to /root/reference/src/numericalnim/.
 *
 * CORE ABI (frozen at NNHIP_ABI_VERSION 1 — what a Nim host needs; everything else in this file is an extension built on the same
 * conventions and may grow): entries marked [core] below —
 *   nnhip_abi_version, nnhip_last_error, nnhip_device_count, nnhip_release                       library
 *   nnhip_ode_new_options, nnhip_ode_integrator_id                                               newODEoptions / the `integrator` string
 *   nnhip_ode_solve_batch_f64                                                                    solveODE over a batch, host arrays
 *   nnhip_ode_solve_workspace_bytes, nnhip_ode_solve_batch_f64_dev                               the same on device-resident batches
 *   nnhip_ode_step_batch_f64_dev                                                                 one IntegratorProc call (the plugin seam)
 *   nnhip_ode_fixed_stream_f64_dev, nnhip_ode_adaptive_stream_workspace_bytes, nnhip_ode_adaptive_stream_f64_dev   ODESolver's loops over that seam
 *   nnhip_ode_rhs_compile, nnhip_ode_rhs_release                                                 a user right-hand side (from nim/rhs_macro.nim's output)
 *   nnhip_ode_solve_batch_multi_gpu_f64, nnhip_allgather_states_f64_dev                          shards over the devices of one node
 *
 * Conventions
 *   - A *batch* is N independent IVPs that share (f, tspan, options, integrator); IVP i has its own y0.
 *   - State arrays are float64.  dim = components per IVP; the scalar `float` state path of the
 *     reference is dim = 1 (bit-identical arithmetic: ode.nim:45-55 makes size=1, sum=id).
 *   - layout NNHIP_LAYOUT_SOA: y[c*N + i]              trajectories y_out[(j*dim + c)*N + i]
 *     layout NNHIP_LAYOUT_AOS: y[i*dim + c]            trajectories y_out[(j*N + i)*dim + c]
 *   - The user RHS closure f(t, y, ctx) (ODEProc[T], ode.nim:36) becomes (rhs_kind, rhs_params[]):
 *     a compiled-in __device__ RHS selected by enum, parameters = ctx.fValues flattened.
 *   - Every function returns NNHIP_OK (0) or a negative nnhip_status; nnhip_last_error() gives the
 *     thread-local message.  Nim exceptions map as: ValueError -> NNHIP_EVALUE / NNHIP_EINTEGRATOR.
 *   - "_dev" entry points take DEVICE pointers and a hipStream_t (as void*; NULL = default stream)
 *     and are asynchronous.  The others take HOST pointers and return when the result is in them.
 *   - The library owns no pointer after a call returns (plans excepted, until destroyed).
 *   - There is NO CPU fallback: without a usable HIP device every compute entry fails with NNHIP_EHIP.
 */
#ifndef NNHIP_ODE_H
#define NNHIP_ODE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNHIP_ABI_VERSION 1

typedef enum nnhip_status {
  NNHIP_OK = 0,
  NNHIP_EVALUE = -1,       /* ValueError: bad option / size / argument (ode.nim:95-100, utils.nim:26)        */
  NNHIP_EINTEGRATOR = -2,  /* ValueError: "... is not a valid integrator" (ode.nim:651)                      */
  NNHIP_EHIP = -3,         /* HIP runtime / device failure (incl. no device)                                 */
  NNHIP_EUNSUPPORTED = -4, /* (integrator, rhs_kind, dim, layout) combination has no compiled kernel         */
  NNHIP_ENOMEM = -5,
  NNHIP_TRUNCATED = 1,     /* NOT an error: the call completed, but max_steps ended an integration short of its tEnd (the last
                            * row holds the state reached, not the state at tEnd).  Only entries whose text says so return it. */
} nnhip_status;

/* ODEoptions, field for field (ode.nim:26-34). Build with nnhip_ode_new_options to get the reference's
 * abs() normalisation and validation (ode.nim:78-102). scaleMax/scaleMin are stored but, as in the
 * reference, never read by the controller (it hard-codes min(4, max(0.125, ...)), ode.nim:71,537). */
typedef struct nnhip_ode_options {
  double dt, dtMax, dtMin, tStart, absTol, relTol, scaleMax, scaleMin;
} nnhip_ode_options;

/* Integrator ids; names are the reference's strings (ode.nim:40-42, 607-649). */
typedef enum nnhip_integrator {
  NNHIP_RK4 = 0, NNHIP_DOPRI54 = 1, NNHIP_TSIT54 = 2, NNHIP_VERN65 = 3, NNHIP_BS32 = 4, NNHIP_RK21 = 5,
  NNHIP_HEUN2 = 6, NNHIP_RALSTON2 = 7, NNHIP_KUTTA3 = 8, NNHIP_HEUN3 = 9, NNHIP_RALSTON3 = 10,
  NNHIP_SSPRK3 = 11, NNHIP_RALSTON4 = 12, NNHIP_KUTTA4 = 13, NNHIP_N_INTEGRATORS = 14
} nnhip_integrator;

/* Compiled-in right-hand sides.  p[] = rhs_params.  Expressions are evaluated in exactly this order
 * (no FMA contraction in the default build), so they are part of the numerical contract. */
typedef enum nnhip_rhs_kind {
  NNHIP_RHS_NEG_Y = 0,     /* any dim : dy_c = -y_c                                    (ode.nim:16-17)      */
  NNHIP_RHS_LINEAR = 1,    /* any dim : dy_c = y_c * p0      (p0 = -0.1: tests/test_ode.nim:5-7)            */
  NNHIP_RHS_LORENZ = 2,    /* dim 3   : dx = p0*(y-x); dy = x*(p1-z) - y; dz = x*y - p2*z                   */
  NNHIP_RHS_RING = 3,      /* dim d   : dy_c = -((c+1)/d)*y_c + p0*y_{(c+1) mod d}                          */
  NNHIP_RHS_AFFINE_T = 4,  /* any dim : dy_c = p0*y_c + p1*t                                                */
  NNHIP_RHS_VANDERPOL = 5, /* dim 2   : dx = v; dv = p0*((1 - x*x)*v) - x                                   */
  NNHIP_N_RHS = 6,
  NNHIP_RHS_USER_BASE = 1000 /* rhs_kind values >= this are user right-hand sides from nnhip_ode_rhs_compile */
} nnhip_rhs_kind;

typedef enum nnhip_layout { NNHIP_LAYOUT_SOA = 0, NNHIP_LAYOUT_AOS = 1 } nnhip_layout;

/* Aggregate statistics of one batch solve (all optional outputs). */
typedef struct nnhip_ode_stats {
  int64_t steps_total;    /* accepted integrator calls summed over the batch (both directions)              */
  int64_t rejected_total; /* in-step retries (ode.nim:58-76) summed over the batch                          */
  int64_t steps_max;      /* max accepted steps of any single IVP                                           */
  int32_t n_t_out;        /* entries written to t_out (= n_t unless tStart is duplicated in tspan)          */
  int32_t ny_min;         /* min over IVPs of the number of y rows the reference would return               */
  int32_t nan_aborts;     /* IVPs aborted because the error norm became NaN (reference would hang)          */
  int32_t truncated;      /* IVPs stopped by max_steps                                                      */
  double kernel_ms;       /* device time of the compute kernels (hipEvent), host-pointer entry points only  */
} nnhip_ode_stats;

/* ---- library / device ------------------------------------------------------------------------ */
/* [core] */
int nnhip_abi_version(void);
/* [core] */
int nnhip_device_count(void);            /* >=0, or NNHIP_EHIP                                              */
/* [core] */
const char* nnhip_last_error(void);      /* thread-local, never NULL                                        */
const char* nnhip_build_info(void);
/* Which libhiprtc (path, version) compiles right-hand sides given as source.  A host process may bundle an older ROCm than the one this library
 * was built with (PyTorch wheels do); when the system's libhiprtc is newer than the process's own, it is loaded into a link namespace of its
 * own and used instead — the same compiler generation as the library's ahead-of-time kernels (1.7-2.2x on the 16-component systems).
 * Environment: NNHIP_HIPRTC=process keeps the process's own, NNHIP_HIPRTC=/path/to/libhiprtc.so names one. */
const char* nnhip_rtc_compiler(void);

/* Frees what the library caches between calls: the calling thread's pinned staging buffer, hipGraph caches (they are per
 * thread: other threads keep theirs until they call this or change a tuning knob) and side streams, the idle
 * stream / event contexts of the host-pointer solve, the RCCL communicators.  Everything is rebuilt on demand; compiled user
 * right-hand sides are released one by one with nnhip_ode_rhs_release.  Call it from each thread that used the library if a
 * clean shutdown matters; not calling it is harmless. */
/* [core] */
int nnhip_release(void);

/* Page-locked host memory for the buffers handed to nnhip_ode_solve_batch_f64: with page-locked y0 and y_out the transfers
 * are overlapped with the kernel (automatic; 1.6x on config C2).  Plain hipHostMalloc / hipHostFree underneath — memory from
 * any other page-locking allocator (or hipHostRegister) is recognised just the same. */
int nnhip_host_alloc(void** out, int64_t bytes);
int nnhip_host_free(void* p);

/* Performance tuning knobs (process-wide; results are bit-identical for every setting):
 *   "rk4_stream_auto" 0|1 (default 1: choose vec/mode from the working-set size), "rk4_stream_vec" 1|2|4|8,
 *   "rk4_stream_mode" 0..3 (0 plain, 1 non-temporal, 2 persistent, 3 both), "rk4_stream_blocks_per_cu" 1..64,
 *   "stream_graph" 0|1|2 (0 eager launches; 1 capture nnhip_ode_fixed_stream_f64_dev's launch sequence in a hipGraph
 *   and replay it; default 2 = do so for launch-bound batches — up to 2e6 states, 16..1e4 steps, a non-default stream — from the
 *   second identical call on; the polling groups of nnhip_ode_adaptive_stream[_dense]_f64_dev are replayed from a graph only with 1 —
 *   measured: eager launches are as fast or faster at every batch size),
 *   "adv_nontemporal" -1|0|1 (non-temporal instantiations of the streaming kernels; -1 = automatic: when the state of one launch
 *   exceeds 192 MiB), "adv_split" 0|1|2|4 (index ranges of the adaptive streaming loop on separate streams; measured slower, default 1),
 *   "adv_block" 0|64|128|256 (workgroup size of the thread-per-IVP advance kernel; 0 = automatic: 64),
 *   "adv_recompute_fsal" -1|0|1 (the streaming loops of DOPRI54 / Tsit54 re-evaluate FSAL = f(t, y) per launch instead of carrying it through
 *   HBM: 16*dim bytes per step less for one more evaluation of f, the same bits; -1 = automatic: on unless the right-hand side has mutable slots),
 *   "adv_steps_per_launch" 1..1024 (loop iterations of ode.nim:525-541 per IVP and launch of nnhip_ode_adaptive_stream_f64_dev; default 1 = one
 *   IntegratorProc call per launch, state through HBM between any two; K > 1 keeps an IVP's state in registers for up to K iterations:
 *   the same bits, 1/K of the launches and 1/K of the bytes per step — a different traffic model, never quoted against the one-per-launch figures),
 *   "sort_copy" 0|1 (1: the binned solve nnhip_ode_solve_batch_sorted_f64[_dev] gathers the batch into integration order, solves it with
 *   coalesced accesses and scatters the results back; default 0: the solve kernel follows the order array itself — measured faster),
 *   "sort_min_spread_permille" 0..1000 (the binned solve sorts only when its keys differ by more than this fraction of their magnitude; default 50 =
 *   5 %; 0 = always sort),
 *   "sort_auto_key" 0|1 (what the automatic binned solve ranks the IVPs by after its probe: 1, default, the steps still to take (tEnd - t) / dt on a
 *   forward tspan; 0 the distance the probe covered — also what backward / two-sided tspans use; most work first either way),
 *   "calls_bin" 0|1 (1, default: nnhip_ode_solve_batch_calls_f64[_dev] / _tend_ / _tspans_ of 32768 calls or more integrate the longest spans first, binned by
 *   the time each call integrates over — same bits, same order of results; 0: in the caller's order),
 *   "sort_rebin_steps" 0..1000000 (S > 0: the automatic binned solve — where it can resume, see "sort_resume" — takes S more accepted steps per IVP after its
 *   probe, bins the batch again by the steps still to take, and finishes: for step sizes that change late in the span; default 0 = the probe's order throughout),
 *   "sort_resume" 0|1 (1: the automatic binned solve continues from its probe's state — forward 2-point tspans, DOPRI54 / Tsit54 / BS32 / RK21 —
 *   instead of integrating the probed steps twice; default 0: within 1 % either way, the resumed pass needs the per-call instantiation of the kernel),
 *   "adv_lean" 0|1 (1: the adaptive streaming loop runs its lean kernels — the driver's own layout as the kernel's contract — where they apply; same bits.
 *   Default 0: opt-in until an MI355X has timed them, the hardware record of round 4 is of the general kernels),
 *   "adv_auto_poll" 0|1 (1: check_every <= 0 selects the library's own polling schedule, see nnhip_ode_adaptive_stream_f64_dev; default 0: check_every <= 0
 *   means uniform groups of 8 launches, the behaviour with a hardware record),
 *   "fixed_vec_ipl" 0|2 (0 = one IVP per lane instead of the vectorised fixed-step streaming kernel),
 *   "multi_gpu_oversubscribe" 0|1 (nnhip_ode_solve_batch_multi_gpu_f64 accepts more shards than devices: shard r on device r mod #devices),
 *   "dim16_variant" 0..4 (A/B mappings of the fused 16-component kernels), "fp_contract" 0|1 (opt-in FMA-contracted kernels — the fused solves
 *   of RK4 / DOPRI54 / Tsit54 / Vern65 and the lean kernels of the adaptive streaming loop, DOPRI54 / Tsit54, wherever a launch has their layout
 *   (whatever "adv_lean" says): NOT bit-exact, within 1e-10 / 1e-6), "host_chunks" 0..64 (0 = automatic: 8 when the caller's buffers are page-locked, else 1) and
 *   "host_register" 0|1 (pipelining of the host-pointer solve) */
/* Changing a knob drops the calling thread's hipGraph caches so that the new setting takes effect on its next call.  Knobs are process-wide atomics:
 * setting one while other host threads are inside the library is defined (no torn value; a call already running may finish under the old setting). */
int nnhip_tune_set(const char* key, int value);
/* The current value of a knob (the same keys; booleans as 0 / 1): what a scoped setter saves and restores. */
int nnhip_tune_get(const char* key, int* value);
/* Which instantiation of the headline kernel (rk4_stream_vec_kernel<RHS, NEG, VEC, MODE>) the scalar RK4 step entry launches for
 * `n_states` flat float64 states (in_place: y_out == y_in, else two buffers): the knobs above, or the automatic choice by working set.
 * Lets a profile be matched to the variant it was taken on (bench.py checks profiles/pmc_traffic.json's kernel name with it). */
int nnhip_ode_rk4_stream_variant(int64_t n_states, int in_place, int* vec, int* mode);

/* ---- options / dispatch (host only, no device needed) ---------------------------------------- */
/* newODEoptions (ode.nim:78-102): abs() of everything but tStart; NNHIP_EVALUE if |dtMax| < |dtMin|,
 * |scaleMax| < 1 or 1 < |scaleMin|.  Argument order = the Nim proc's. */
/* [core] */
int nnhip_ode_new_options(nnhip_ode_options* out, double dt, double absTol, double relTol, double dtMax, double dtMin,
                          double scaleMax, double scaleMin, double tStart);
/* DEFAULT_ODEoptions (ode.nim:104): dt=1e-4 absTol=relTol=1e-4 dtMax=1e-2 dtMin=1e-4 scale 4/0.1 tStart=0 */
int nnhip_ode_default_options(nnhip_ode_options* out);
/* `case integrator.toLower()` (ode.nim:607-651): id, or NNHIP_EINTEGRATOR. */
/* [core] */
int nnhip_ode_integrator_id(const char* name);
const char* nnhip_ode_integrator_name(int integrator);
/* (useFSAL, order, adaptive) triple the dispatch passes to ODESolver (ode.nim:608-649). */
int nnhip_ode_integrator_traits(int integrator, int* use_fsal, double* order, int* adaptive);
/* The Butcher tableau the kernels of a tableau method are compiled with — DOPRI54 (ode.nim:240-282), Tsit54 (:310-352), Vern65 (:380-443) —
 * for verification against the reference's text: out = [S, NB, c_1..c_S, a_21, a_31, a_32, .. a_S,S-1, b_1..b_NB, bHat_1..bHat_S]
 * (43 / 43 / 64 values).  device < 0: read on the host from the constexpr tables; device >= 0: written by a kernel on that device through
 * the accessors the steppers use.  Returns the number of values, NNHIP_EINTEGRATOR for the other 11 methods (their coefficients are
 * literals inside their step expressions, ode.nim:107-234), NNHIP_EVALUE if cap is too small. */
int nnhip_ode_tableau_f64(int integrator, int device, double* out, int cap);
/* Output time grid exactly as ODESolver assembles it (ode.nim:476-480, 585): tspan.sorted(), split
 * around tStart, tNegative.reversed ++ tZero ++ tPositive.  t_out has room for n_t doubles. */
int nnhip_ode_time_grid(const nnhip_ode_options* opt, const double* tspan, int n_t, double* t_out, int* n_t_out);
/* Is there a kernel for this combination?  1 / 0.  mode: 0 fused solve, 1 step-streaming.  Ahead-of-time kernels cover the
 * sizes of the reference's tests and of the BASELINE configs (dims 1-4, Lorenz, Van der Pol, 8/16/32-component systems); the
 * size-generic kinds NEG_Y, LINEAR, AFFINE_T and RING also run at every other dim in 1..256 (the reference's
 * Vector[float] has any length): those sizes are instantiated at run time from the same expressions on first use (hiprtc). */
int nnhip_ode_supported(int integrator, int rhs_kind, int dim, int layout, int mode);

/* ---- fused batch solve: replaces solveODE -> ODESolver (ode.nim:589-651, 471-586) -------------
 * One launch integrates every IVP from tStart across the whole (sorted) tspan, forward and backward
 * branches, dense Hermite output included; state, stage vectors and (t, dt) stay on chip.
 *   y0      [dim*N]        in `layout`
 *   tspan   [n_t]          any order (sorted internally, like tspan.sorted())
 *   t_out   [n_t]          (host) output grid; nullable
 *   y_out   [n_t*dim*N]    row j = state at t_out[j]; rows >= ny_out[i] are NaN for IVP i
 *   ny_out  [N] nullable   rows the reference would return for IVP i (it can drop requested times that
 *                          fall strictly inside the final step)
 *   steps_out / rejected_out [N] nullable per-IVP counters
 *   max_steps              safety cap on integrator calls per IVP and direction; <= 0 = unlimited
 * Host-pointer form; device = HIP device ordinal. */
/* [core] */
int nnhip_ode_solve_batch_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                              int n_params, const double* y0, int64_t N, int dim, int layout, const double* tspan,
                              int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                              int64_t* rejected_out, int64_t max_steps, nnhip_ode_stats* stats, int device);
/* The same with PER-IVP right-hand-side parameters in host memory (see nnhip_ode_solve_batch_sweep_f64_dev): parameter k of IVP i
 * = per_ivp_params[k*N + i] for k < n_per_ivp, overriding rhs_params[k]. */
int nnhip_ode_solve_batch_sweep_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                    int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N,
                                    int dim, int layout, const double* tspan, int n_t, double* t_out, double* y_out,
                                    int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps,
                                    nnhip_ode_stats* stats, int device);

/* Device-pointer form (asynchronous on `stream`).  y0/y_out/ny_out/steps_out/rejected_out are device
 * pointers on the current device; tspan/t_out stay host pointers (tiny, shared by the batch).
 * `ws` is a device workspace of at least nnhip_ode_solve_workspace_bytes(n_t) bytes. */
/* [core] */
int64_t nnhip_ode_solve_workspace_bytes(int n_t);
/* [core] */
int nnhip_ode_solve_batch_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                  int n_params, const double* y0, int64_t N, int dim, int layout, const double* tspan,
                                  int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                                  int64_t* rejected_out, int64_t max_steps, void* ws, int64_t ws_bytes, void* stream);

/* Parameter sweeps: the same fused solve with PER-IVP right-hand-side parameters.  In the reference every IVP is its own
 * solveODE call and may carry its own ctx (ode.nim:589-591, 599); here `per_ivp_params` is a device table [n_per_ivp][N]
 * whose column i overrides rhs_params[0 .. n_per_ivp) for IVP i (n_per_ivp <= n_params; remaining parameters stay
 * batch-wide).  Everything else as nnhip_ode_solve_batch_f64_dev. */
int nnhip_ode_solve_batch_sweep_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                        int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N,
                                        int dim, int layout, const double* tspan, int n_t, double* t_out, double* y_out,
                                        int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps, void* ws,
                                        int64_t ws_bytes, void* stream);

/* Every IVP its own tspan end: tspan_i = [options.tStart, t_end[i]], t_end a device array [N] (in the reference every IVP is a
 * solveODE call with its own tspan, ode.nim:589-591, 476-480).  y_out [2][dim][N] / [2][N][dim] holds per IVP the rows the reference
 * returns for tspan_i.sorted(): (y0, y(tEnd)) when tEnd > tStart; (y(tEnd), y0) when tEnd < tStart (backward branch, :544-584);
 * the single row y0 when they coincide (ny_out[i] = 1, second row NaN — the reference returns one state for two times there).
 * A non-finite t_end[i] gives ny_out[i] = -1 and NaN rows (the reference would never return from that call). */
int nnhip_ode_solve_batch_tend_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                       int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim,
                                       int layout, const double* t_end, double* y_out, int32_t* ny_out, int64_t* steps_out,
                                       int64_t* rejected_out, int64_t max_steps, void* stream);

/* Every IVP its own solveODE call: per-IVP tspan ends AND per-IVP ODEoptions fields (each reference call owns both, ode.nim:589-591,
 * 26-34).  t_end is required; t_start, abs_tol, rel_tol, dt_max, dt_min, dt_fixed are nullable device arrays [N] (NULL = the value in
 * `opt`).  Per-IVP option values go through abs() like newODEoptions does (ode.nim:101-102).  An IVP whose options newODEoptions would
 * reject (dtMax < dtMin), that could never finish (fixed-step dt == 0; dtMin == 0 without max_steps) or whose span is not finite gets
 * ny_out[i] = -1 and NaN rows (the reference raises ValueError / never returns for that call; the other calls are unaffected). */
int nnhip_ode_solve_batch_calls_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                        int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim,
                                        int layout, const double* t_end, const double* t_start, const double* abs_tol,
                                        const double* rel_tol, const double* dt_max, const double* dt_min, const double* dt_fixed,
                                        double* y_out, int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps,
                                        void* stream);

/* Every IVP its own n_t-point tspan (and, optionally, its own option fields): N reference calls solveODE(f, y0_i, tspans[i], options_i)
 * in one launch (ode.nim:589-591, 476-487, 609).  tspans: device array [N][n_t], every row in any order, on either side of its tStart,
 * with duplicates, with or without tStart itself; t_start, abs_tol, rel_tol, dt_max, dt_min, dt_fixed: nullable device arrays [N] as in
 * nnhip_ode_solve_batch_calls_f64_dev.  t_out (nullable): device array [N][n_t], row i = the times the reference returns for call i
 * (tspan_i sorted, ode.nim:585), NaN beyond; y_out [n_t][dim][N] / [n_t][N][dim]: row j of IVP i belongs to t_out[i][j]; ny_out[i] =
 * rows the reference returns (NaN beyond), -1 for a call it would refuse or never finish (non-finite tspan, dtMax < dtMin, ...).
 * `ws`: nnhip_ode_solve_tspans_workspace_bytes(N, n_t) bytes of device scratch.  Bitwise equal to one fused solve per IVP. */
int64_t nnhip_ode_solve_tspans_workspace_bytes(int64_t N, int n_t);
int nnhip_ode_solve_batch_tspans_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                         int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim,
                                         int layout, const double* tspans, int n_t, const double* t_start, const double* abs_tol,
                                         const double* rel_tol, const double* dt_max, const double* dt_min, const double* dt_fixed,
                                         double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out,
                                         int64_t max_steps, void* ws, int64_t ws_bytes, void* stream);

/* Host-pointer form of the per-IVP-tspan solve: tspans [N][n_t], t_out [N][n_t] (nullable), y0, y_out, counters in host memory;
 * opt_per_ivp: N option objects (NULL: every call uses `opt`), as in nnhip_ode_solve_batch_calls_f64. */
int nnhip_ode_solve_batch_tspans_f64(const nnhip_ode_options* opt, const nnhip_ode_options* opt_per_ivp, int integrator, int rhs_kind,
                                     const double* rhs_params, int n_params, const double* per_ivp_params, int n_per_ivp,
                                     const double* y0, int64_t N, int dim, int layout, const double* tspans, int n_t, double* t_out,
                                     double* y_out, int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps,
                                     int device);

/* Host-pointer form: N reference calls solveODE(f, y0_i, [options_i.tStart, t_end[i]], options_i) (ode.nim:589-591) in one launch.
 * opt_per_ivp: array of N option objects in host memory (NULL: every call uses `opt`; when given, `opt` still supplies nothing but
 * validation defaults).  t_end, y0, per_ivp_params, y_out [2][dim][N] / [2][N][dim], ny_out, steps_out, rejected_out: host arrays,
 * staged through `device` in one piece.  The objects' fields become the per-IVP tables of nnhip_ode_solve_batch_calls_f64_dev, whose
 * semantics apply (an adaptive call with dtMax < dtMin fails alone: ny_out[i] = -1, NaN rows). */
int nnhip_ode_solve_batch_calls_f64(const nnhip_ode_options* opt, const nnhip_ode_options* opt_per_ivp, int integrator, int rhs_kind,
                                    const double* rhs_params, int n_params, const double* per_ivp_params, int n_per_ivp,
                                    const double* y0, int64_t N, int dim, int layout, const double* t_end, double* y_out,
                                    int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps, int device);

/* Divergence binning: the same fused solve for batches whose members take very different step sequences (the reference runs
 * them one after the other, ode.nim:589-591; on a wavefront they share an instruction stream).  The IVPs are INTEGRATED in
 * ascending order of `sort_key` (device array [N]) and every result is WRITTEN at the IVP's own index, so outputs are in the
 * caller's order and bit-identical to nnhip_ode_solve_batch_sweep_f64_dev.  "Order" means bins: the keys are counted into 4096 bins over the range they
 * cover and integrated bin after bin (inside a bin in no particular order) — neighbouring lanes need similar step sequences, not a total order.
 * Choose the key so that the IVPs with the MOST work come first (e.g. minus the stiffness parameter): the workgroups dispatched last then hold the
 * short solves (1e6 Van der Pol IVPs: 1.33 ms against 1.44 ms for the same key ascending, profiles/r04_bench_divergence.json).
 * sort_key == NULL: automatic two-pass mode — a probe solve of `probe_steps` accepted steps per IVP (<= 0: 8) ranks the IVPs by the steps they still have
 * to take (knob "sort_auto_key"), most first, then the batch is integrated in that order.  Fixed-step integrators run unsorted (no divergence), and so does a batch (of 4096 IVPs or more) whose keys lie
 * within 5 % of each other (knob "sort_min_spread_permille"): nothing to gain, and the sort + indirection would cost ~20 % of such a solve.
 * Reading the keys' range synchronises `stream` once.  per_ivp_params may be NULL (n_per_ivp = 0).
 * `ws`: device workspace of nnhip_ode_solve_sorted_workspace_bytes(N, n_t) bytes.  N < 2^31. */
int64_t nnhip_ode_solve_sorted_workspace_bytes(int64_t N, int n_t);
/* The order of integration the binned solve derives from a key: order_out[k] (device, uint32 [N]) = index of the k-th IVP integrated.  The keys are
 * counted into 4096 slices of their range — by their order-preserving bit image (logarithmic across binades) when all finite keys have one sign and
 * none is zero, LINEARLY IN VALUE when the range touches or straddles zero (a finished IVP's "0 steps left", a zero-length span, a centred
 * parameter) — ascending from slice to slice, unordered inside one, non-finite keys last.  Asynchronous on `stream`. */
int nnhip_ode_bin_order_f64_dev(const double* sort_key, int64_t N, uint32_t* order_out, void* stream);
/* host-pointer form (all arrays in host memory, incl. sort_key; staged through `device` in one piece) */
int nnhip_ode_solve_batch_sorted_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                     const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim, int layout,
                                     const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                                     int64_t* rejected_out, int64_t max_steps, const double* sort_key, int probe_steps, int device);
int nnhip_ode_solve_batch_sorted_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                         int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N,
                                         int dim, int layout, const double* tspan, int n_t, double* t_out, double* y_out,
                                         int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps,
                                         const double* sort_key, int probe_steps, void* ws, int64_t ws_bytes, void* stream);

/* ---- step-streaming: replaces one IntegratorProc call (ode.nim:38, call sites :531,:573) ------
 * (yNew, FSAL', dtUsed, error) = integrator(f, t, y, FSAL, dt, options, ctx) for every IVP of the batch,
 * state resident in HBM between calls.  t/dt are per-IVP device arrays [N], or — when t_dev / dt_dev is
 * NULL — the uniform host scalars t_uniform / dt_uniform (fixed-step methods: every IVP shares them).
 * fsal_in/fsal_out may be NULL for non-FSAL methods; dt_used/error may be NULL for fixed-step methods
 * (they return their input dt and 0.0).  In-place (y_out == y_in, fsal_out == fsal_in) is allowed.
 * negate_time != 0 integrates g(t,y) = -f(-t,y) (backward branch, ode.nim:545). */
/* [core] */
int nnhip_ode_step_batch_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                 int n_params, int64_t N, int dim, int layout, const double* t_dev, double t_uniform,
                                 const double* dt_dev, double dt_uniform, const double* y_in, const double* fsal_in,
                                 double* y_out, double* fsal_out, double* dt_used, double* error, int negate_time,
                                 void* stream);

/* Fixed-step time loop of ODESolver (ode.nim:511-542 with adaptive=false, tspan.len == 2) driven from
 * the host over the step-streaming kernel: while t < tEnd: dt = min(dt, tEnd - t); step; t += dt.
 * y (device, in `layout`) is advanced in place from t0 to tEnd; `scratch` (device, dim*N doubles,
 * nullable) enables ping-pong instead of in-place.  Returns the number of steps via n_steps_out.
 * Works for every fixed-step integrator; asynchronous on `stream`. */
/* [core] */
int nnhip_ode_fixed_stream_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                   int n_params, int64_t N, int dim, int layout, double t0, double tEnd, double* y,
                                   double* scratch, int64_t* n_steps_out, double** y_final, void* stream);

/* The WHOLE of ODESolver (ode.nim:471-586) for fixed-step integrators through the IntegratorProc seam: both directions around
 * options.tStart, any tspan, requested-time rows by Hermite interpolation between consecutive steps (ode.nim:512-524), output
 * assembly yNegative.reversed ++ yZero ++ yPositive (:585).  The state stays in HBM between IntegratorProc calls (one
 * step-streaming launch per time step); when requested times fall into the step just taken, f(lastT, lastY), f(t, y) and one
 * Hermite kernel per requested time are launched — the two ping-pong buffers are (lastIter.y, y).  y0 / y_out [n_t][dim][N] (SoA)
 * or [n_t][N][dim] (AoS) are device pointers, tspan / t_out host.  (t, dt) are shared by the batch, so the number of rows the
 * reference would return is the same for every IVP: *ny_out (host, nullable); rows beyond it are NaN.  `ws`: device scratch of
 * nnhip_ode_fixed_stream_dense_workspace_bytes(N, dim).  Bitwise equal to nnhip_ode_solve_batch_f64_dev.  Returns NNHIP_TRUNCATED (> 0, all
 * outputs written) when max_steps cut a direction short of its end time — what the fused solve reports as stats.truncated. */
int64_t nnhip_ode_fixed_stream_dense_workspace_bytes(int64_t N, int dim);
int nnhip_ode_fixed_stream_dense_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                         int n_params, const double* y0, int64_t N, int dim, int layout, const double* tspan, int n_t,
                                         double* t_out, double* y_out, int* ny_out, int64_t max_steps, void* ws, int64_t ws_bytes,
                                         int64_t* n_steps_out, void* stream);

/* The WHOLE of ODESolver (ode.nim:471-586) for ADAPTIVE integrators through the IntegratorProc seam: per launch every unfinished
 * IVP takes one step (ode.nim:525-541) and emits the requested times that step passed (the emission block :511-524 of the next loop
 * iteration, Hermite interpolation from lastIter = (t, y, dy) — both ends of the step are in the kernel's registers, so the history never
 * goes to HBM); y, (t, dt), denseIndex and — as in nnhip_ode_adaptive_stream_f64_dev — FSAL where it is carried are resident in HBM
 * between launches: 8*(4*dim+4) + 4 or 8*(2*dim+4) + 4 bytes per step and IVP, 8*dim per emitted row.  Both directions around options.tStart, any tspan, the reference's row assembly and quirks.  y0 / y_out / ny_out
 * (int32 [N], required: rows the reference returns for IVP i; rows beyond are NaN) are device pointers, tspan / t_out host.
 * Every right-hand side kind (thread-per-IVP and lanes-per-system, compiled-in and run-time compiled).  `ws`:
 * nnhip_ode_adaptive_stream_dense_workspace_bytes(N, dim, n_t) bytes.  The host polls one group of `check_every` launches behind the
 * device (launches_out counts the group issued past the end as well; check_every <= 0: as
 * nnhip_ode_adaptive_stream_f64_dev's, per direction).  max_launches > 0 bounds the loop of EACH direction exactly as
 * max_steps bounds the fused solve's (same rows, same ny_out); the call then returns NNHIP_TRUNCATED (> 0, all outputs written) if an
 * integration was cut short.
 * Bitwise equal to nnhip_ode_solve_batch_f64_dev. */
int64_t nnhip_ode_adaptive_stream_dense_workspace_bytes(int64_t N, int dim, int n_t);
int nnhip_ode_adaptive_stream_dense_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                            int n_params, const double* y0, int64_t N, int dim, int layout, const double* tspan, int n_t,
                                            double* t_out, double* y_out, int32_t* ny_out, void* ws, int64_t ws_bytes, int check_every,
                                            int64_t max_launches, int64_t* launches_out, void* stream);

/* Adaptive time loop of ODESolver (ode.nim:506-542 with adaptive=true, tspan.len == 2) driven from the host over an
 * `advance` kernel: per launch, every unfinished IVP does dt = min(dt, tEnd-t); step; t += dt; controller — with y, (t, dt) and,
 * where the next step reads it, FSAL resident in HBM between launches.  Algorithmic bytes per step: 8*(4*dim+4) with FSAL carried
 * (Vern65; DOPRI54 / Tsit54 with knob "adv_recompute_fsal" = 0, the IntegratorProc signature as the reference passes it),
 * 8*(2*dim+4) without (BS32 / RK21 never read the slot; DOPRI54 / Tsit54 by default re-evaluate FSAL = f(t, y) at the start of the
 * launch — it is the previous step's last stage f(t + dt, yNew), the same bits — unless the right-hand side has mutable slots).
 * y (device, in `layout`) is
 * advanced in place from t0 to tEnd; `ws` is device scratch of nnhip_ode_adaptive_stream_workspace_bytes(N, dim) (16-byte aligned, as
 * every allocator's blocks are, it holds (t, dt) of an IVP side by side — one 16-byte access each way per launch; otherwise two columns).  The host
 * learns whether anyone is still integrating every `check_every` launches and always has the next group enqueued
 * before it waits (groups can be replayed from a hipGraph on a non-default stream: knob "stream_graph" = 1), so up to 2*check_every trailing launches
 * find nothing to do (they read t only).  check_every <= 0: uniform groups of 8; with knob "adv_auto_poll" = 1 the library's own schedule — no step is longer than dtMax, so nobody finishes within
 * the first ceil((tEnd - t0) / dtMax) launches, which go out unpolled; then groups of 2, 2, 4, 8, 8 ... (graph replay: uniform groups of 8).  Results are bitwise those of the fused solve.  Thread-per-IVP kernels for small
 * systems, lanes-per-system kernels for Vector[float] states of 8 / 16 / 32 ... components (ahead of time or run-time compiled). */
/* [core] */
int64_t nnhip_ode_adaptive_stream_workspace_bytes(int64_t N, int dim);
/* [core] */
int nnhip_ode_adaptive_stream_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                      int n_params, int64_t N, int dim, int layout, double t0, double tEnd, double* y, void* ws,
                                      int64_t ws_bytes, int check_every, int64_t max_launches, int64_t* launches_out, void* stream);

/* ---- multi-GPU (one process, n_gpus devices): contiguous shards of the IVP index range, no exchange
 * during integration, final trajectory tensor reassembled on every device's host view.  Host pointers. */
/* [core] */
int nnhip_ode_solve_batch_multi_gpu_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind,
                                        const double* rhs_params, int n_params, const double* y0, int64_t N, int dim,
                                        int layout, const double* tspan, int n_t, double* t_out, double* y_out,
                                        int32_t* ny_out, int64_t max_steps, nnhip_ode_stats* stats, int n_gpus);

/* The same with per-IVP right-hand-side parameters (table [n_per_ivp][N] in host memory, as nnhip_ode_solve_batch_sweep_f64) and the
 * per-IVP step counters (nullable): a parameter sweep — N solveODE calls with their own ctx each, ode.nim:589-591, 599 — sharded
 * over n_gpus devices; every device reads its columns of the caller's table in place. */
int nnhip_ode_solve_batch_multi_gpu_sweep_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                              int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N,
                                              int dim, int layout, const double* tspan, int n_t, double* t_out, double* y_out,
                                              int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps,
                                              nnhip_ode_stats* stats, int n_gpus);


/* ---- user-supplied right-hand side (run-time compiled) ------------------------------------------
 * The reference accepts any closure f(t, y, ctx) (ODEProc[T], ode.nim:36).  A host closure cannot run on the
 * device, its source can: `body` is the HIP C++ body of
 *     __device__ void rhs(double t, const double* y, double* dy, const double* p)   // y, dy: dim components; p: rhs_params
 * e.g. "dy[0] = y[1]; dy[1] = -p[0]*y[0];".  It is compiled with hiprtc into the SAME stepper / driver kernel
 * templates the built-in RHS use (same -ffp-contract=off numerics), once per (rhs, integrator), cached in-process.
 * *rhs_kind_out receives a handle (>= NNHIP_RHS_USER_BASE) accepted by every entry that takes rhs_kind
 * (thread-per-IVP kernels; dim 1..16).  The body is syntax-checked at registration; NNHIP_EVALUE + the compiler
 * log in nnhip_last_error() on failure. */
/* [core] */
int nnhip_ode_rhs_compile(const char* name, int dim, int n_params, const char* body, int* rhs_kind_out);
/* Per-component form: `comp_body` is the body of
 *     __device__ double rhs_comp(double t, int c, const double* y, const double* p)   // returns dy_c
 * e.g. "return -((c+1)/16.0)*y[c] + p[0]*y[(c+1)%16];" (`dim` is visible in the body).  dim 1..256: systems of 8 or 16
 * components and every system wider than 16 run on the lanes-per-system kernels (stage vector in LDS, up to a whole
 * wavefront of lanes x 4 components per system; sizes that are not a power of two use the next one with the tail slots
 * switched off); the other sizes up to 16 run thread-per-IVP. */
int nnhip_ode_rhs_compile_comp(const char* name, int dim, int n_params, const char* comp_body, int* rhs_kind_out);
/* NumContext in full (src/numericalnim/common/commonTypes.nim:4-27; the ctx the solver passes through to f, ode.nim:36,498,506,521,
 * 530,531, "IT IS MUTABLE", :599): a user right-hand side compiled with a CONTEXT LAYOUT.  Besides t, y, dy (or c) its body sees
 *   p[k], k < n_params       ctx.fValues in key order — ANY count: up to 8 travel as kernel arguments (rhs_params of each call, as
 *                            before); with more than 8 they are the first n_params doubles of the shared block and calls pass n_params = 0
 *   NAME[j]                  a SHARED vector (a ctx.tValues entry every IVP of the batch uses: a forcing vector, a coefficient table),
 *                            vec_lens[k] doubles, any length
 *   NAME(j) or NAME[j]       a PER-IVP vector (vec_per_ivp[k] != 0): entry j of THIS IVP's own vector — the batch form of N reference
 *                            calls that each had their own ctx (e.g. a 16 x 16 matrix per system: length 256)
 *   aux(j), j < n_aux        per-IVP MUTABLE doubles the body may read and write (event counters, running maxima): the mutable-ctx
 *                            analogue.  The device evaluates f exactly where ODESolver does for a 2-point tspan — FSAL = f(t0, y0)
 *                            (:506), then the stages of every attempt in order (k1 = FSAL is not re-evaluated) — so a body that
 *                            accumulates over its calls sees the reference's call sequence; with dense output (tspan.len > 2) the
 *                            non-FSAL methods' extra evaluations (:521,530) are made lazily and the call COUNT differs.  Whole-vector
 *                            bodies only (per_component bodies run once per lane of a system).
 * per_component: 0 = body of `void rhs(t, y[dim], dy[dim], ...)` as nnhip_ode_rhs_compile, 1 = body returning dy_c as
 * nnhip_ode_rhs_compile_comp.  Names must be C identifiers other than p, y, dy, t, c, aux, dim, size. */
int nnhip_ode_rhs_compile_ctx(const char* name, int dim, int n_params, const char* body, int per_component, int n_vectors,
                              const char* const* vec_names, const int64_t* vec_lens, const int* vec_per_ivp, int n_aux, int* rhs_kind_out);
/* Binds device memory to the layout of `rhs_kind` (the closure capturing its ctx); every later call that takes this rhs_kind reads it,
 * until the next bind.  The caller owns the memory and keeps it alive while such calls are in flight.
 *   shared   [shared_len]: the n_params scalars (only when n_params > 8) followed by the shared vectors in declaration order
 *   per_ivp  [per_ivp_rows][stride]: the per-IVP vectors in declaration order, row r of IVP i at per_ivp[r*stride + i]
 *   aux      [n_aux][stride], updated in place
 * shared_len / per_ivp_rows / n_aux must equal what the layout declares (NNHIP_EVALUE otherwise); stride = IVPs of the bound batch
 * (calls may solve any N <= stride: IVP i of a call reads column i).  A binding belongs to the THREAD that made it: that thread's later calls
 * read it whatever other threads bind in the meantime (two host threads can solve one compiled source with different contexts at the same
 * time); a thread that never bound this rhs_kind reads the most recent binding of any thread.  Device pointers belong to one device's
 * memory: the multi-GPU entries refuse a block bound this way — bind it from host arrays (below) and they cut it along the shards. */
int nnhip_ode_rhs_bind_ctx_f64_dev(int rhs_kind, const double* shared, int64_t shared_len, const double* per_ivp, int64_t per_ivp_rows,
                                   double* aux, int n_aux, int64_t stride);
/* The same for hosts without device-memory management (the Nim shim): the three parts are HOST arrays, copied into device memory of
 * `device` that the library owns until the binding is replaced / nnhip_ode_rhs_release; aux_init [n_aux][stride] seeds the mutable slots and
 * nnhip_ode_rhs_read_aux_f64 copies their current content back ([n_aux][stride], after synchronising the device).  The library keeps the host
 * values too: the multi-GPU entries (nnhip_ode_solve_batch_multi_gpu[_sweep]_f64 and the two ..._multi_gpu_f64_dev ones) upload to shard r's
 * device the columns [lo_r, hi_r) of the per-IVP rows and of the mutable slots (the shared block whole) — every member of the batch keeps its
 * own ctx wherever it is integrated (commonTypes.nim:4-27, ode.nim:599) — and the slots are gathered back into the caller's order before
 * anything reads them again (read_aux, a single-device call, the next sharded one). */
int nnhip_ode_rhs_bind_ctx_f64(int rhs_kind, const double* shared, int64_t shared_len, const double* per_ivp, int64_t per_ivp_rows,
                               const double* aux_init, int n_aux, int64_t stride, int device);
int nnhip_ode_rhs_read_aux_f64(int rhs_kind, double* aux_out);
/* A per-component body that reads only components c - lo .. c + hi of its system (cyclically: y[(c + 1) % dim] is one to the right) may say
 * so: on the lanes-per-system kernels (systems of 8, 16, 32 ... components whose size is a power of two) the neighbours then come from the
 * adjacent lanes through DPP rotations instead of the LDS stage vector — the form the compiled-in ring system uses; the same expression,
 * hence the same bits (2x on the streamed 16-component ring).  0 <= lo, hi <= 4.  A body that reads outside the declared window computes
 * with clamped neighbours, i.e. wrong numbers: declare what the body reads (the Python mirror checks a declaration against the undeclared
 * form on a random batch).  Call before the first solve; code objects compiled earlier are dropped. */
int nnhip_ode_rhs_set_halo(int rhs_kind, int lo, int hi);
/* [core] */
int nnhip_ode_rhs_release(int rhs_kind);

/* Device-resident reassembly (BASELINE.json config C5): one process, n_gpus devices, RCCL over xGMI.  shard[r] lives on
 * device r and holds that device's contiguous IVP range (counts[r] IVPs) of a state tensor in `layout`; full[r] on device r
 * receives the whole tensor ([dim][N] / [N][dim], N = sum counts).  Equal shards use one ncclAllGather per component
 * plane (SoA) or one in total (AoS); ragged shards one ncclBroadcast per shard.  Enqueued on streams[r] (nullable).
 * RCCL is loaded lazily (dlopen); failure text: nnhip_multigpu_last_error(). */
/* [core] */
int nnhip_allgather_states_f64_dev(int n_gpus, const double* const* shard, const int64_t* counts, int dim, int layout,
                                   double* const* full, void* const* streams);
/* BASELINE.json config C5 behind ONE call: "shards across the 8 GPUs with an RCCL all-gather over xGMI only to reassemble".  One
 * process, n_gpus devices; device r holds its contiguous shard y[r] (counts[r] IVPs, `layout`) resident in its own memory.  Every
 * shard is integrated by the step-streaming loop of nnhip_ode_fixed_stream_f64_dev (one reference call per IVP, ode.nim:589-591:
 * nothing couples trajectories) on streams[r] — one non-default stream per device, required — and the final shards are then
 * reassembled on EVERY device into full[r] ([dim][N] SoA / [N][dim] AoS, N = sum counts; nullable array: no gather) with one
 * ncclAllGather per component plane (equal counts) or grouped ncclBroadcasts (ragged).  The collective is enqueued behind the
 * solve on streams[r], or on gather_streams[r] (nullable array) behind an event: the caller's next solve on streams[r] then overlaps
 * the gather (rotate the state buffers so that a shard being gathered is not overwritten).  scratch (nullable array): ping-pong
 * buffers as in the single-device entry.  y_final[r] (nullable array) = the buffer that holds shard r's final state (y[r] or
 * scratch[r]); *n_steps_out = steps taken.  Asynchronous: returns after enqueueing, the caller synchronises its streams.  No host
 * copy of state anywhere; RCCL is loaded lazily (failure text in nnhip_last_error()). */
int nnhip_ode_fixed_stream_multi_gpu_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                             int n_gpus, const int64_t* counts, int dim, int layout, double t0, double tEnd, double* const* y,
                                             double* const* scratch, double* const* full, void* const* streams, void* const* gather_streams,
                                             int64_t* n_steps_out, double** y_final);
/* The same composition around the fused solve (any integrator, any tspan): y0[r] / y_out[r] ([n_t][dim][counts[r]] SoA,
 * [n_t][counts[r]][dim] AoS) / ny_out[r] (nullable array) / ws[r] (nnhip_ode_solve_workspace_bytes(n_t) bytes) live on device r; the
 * whole trajectory tensor is reassembled on every device into full[r] ([n_t][dim][N] / [n_t][N][dim]; nullable array: no gather) —
 * plane by plane in one RCCL group for SoA, one block per row for AoS.  t_out host, as nnhip_ode_solve_batch_f64_dev. */
int nnhip_ode_solve_batch_multi_gpu_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                            int n_gpus, const int64_t* counts, int dim, int layout, const double* tspan, int n_t, double* t_out,
                                            const double* const* y0, double* const* y_out, int32_t* const* ny_out, int64_t max_steps,
                                            void* const* ws, int64_t ws_bytes, double* const* full, void* const* streams, void* const* gather_streams);
const char* nnhip_multigpu_last_error(void);

/* ---- consumers either side of the path ------------------------------------------------------- */
/* hermiteSpline (utils.nim:273-279), batched on device: out[i] = H(x; x1, x2, y1[i], y2[i], dy1[i], dy2[i]) */
int nnhip_hermite_spline_f64_dev(double x, double x1, double x2, const double* y1, const double* y2, const double* dy1,
                                 const double* dy2, double* out, int64_t n, void* stream);
/* Output consumer: newHermiteSpline(X, Y, dY) + eval / derivEval (src/numericalnim/interpolate.nim:114-115,
 * 186-240, 299-390) over M independent series — e.g. M = dim*N for a trajectory tensor returned by the solver, with
 * dY = f(t_j, y_j) from nnhip_ode_rhs_batch_f64_dev (the README's (t, y, dy) recipe).
 *   X [n_knots] host, in any order: the constructor's sortAndTrimDataset(@X, @[@Y, @dY]) (interpolate.nim:231) runs here when X is not strictly
 *   ascending — per call: a spline that is evaluated often is sorted once with nnhip_sort_and_trim_dataset_f64_dev (then nothing moves here);
 *   NaN in X, impure duplicates (the same x with different y anywhere in the batch; ValueError utils.nim:372) and fewer than 2 distinct knots: NNHIP_EVALUE
 *   Y, dY [n_knots][M] device;  xq [n_q] host;  out [n_q][M] device
 *   deriv 0 = eval, 1 = derivEval;  extrap: 0 Constant(extrap_value) 1 Edge 2 Linear 3 Native 4 Error (ExtrapolateKind :89-90;
 *   Error out of range -> NNHIP_EVALUE, as the reference raises ValueError) */
int nnhip_hermite_spline_eval_batch_f64_dev(const double* X, int n_knots, const double* Y, const double* dY, int64_t M,
                                            const double* xq, int n_q, int deriv, int extrap, double extrap_value, double* out,
                                            void* stream);
/* newHermiteSpline(X, Y) WITHOUT derivatives (interpolate.nim:241-253): the three-point difference slopes the reference
 * estimates — (Y1-Y0)/(X1-X0) at the ends, 0.5*((Y[i+1]-Y[i])/(X[i+1]-X[i]) + (Y[i]-Y[i-1])/(X[i]-X[i-1])) inside — over M series.
 * X [n_knots] host, 2 <= n_knots <= 65535; Y, dY [n_knots][M] device.  X in any order: the slopes are those of the sorted, trimmed data
 * (interpolate.nim:244), dY row k belongs to sortAndTrimDataset's row k (rows beyond its count: NaN) — use nnhip_sort_and_trim_dataset_f64_dev for the
 * matching X / Y.  Feed dY to the eval entry above.  Synchronises `stream`. */
int nnhip_hermite_spline_slopes_f64_dev(const double* X, int n_knots, const double* Y, int64_t M, double* dY, void* stream);
/* sortAndTrimDataset(x, @[y_0, ...]) (src/numericalnim/utils.nim:384-413: sorted ascending by x — equal x by original position —, of every run of equal x
 * the first row kept, the others required to be pure duplicates of it), the front end of every discrete consumer below, as an entry of its own:
 *   X [n] host;  Y: host array of n_y (<= 8) device pointers, each [n][M];  X_out [n] host;  Y_out: n_y device pointers [n][M], not aliasing Y;
 *   *n_out = rows of the result; rows beyond it are NaN.  NaN in X (no defined order in the reference's comparison sort) and impure duplicates anywhere
 *   in the batch (ValueError :372; NaN values count as different, as `!=` does there): NNHIP_EVALUE.  Strictly ascending X: a copy. */
int nnhip_sort_and_trim_dataset_f64_dev(const double* X, int n, const double* const* Y, int n_y, int64_t M, double* X_out, double* const* Y_out, int* n_out,
                                        void* stream);
/* Host only: how many rows the discrete consumers return for this X — *n_sorted_trimmed (nullable) = distinct abscissae = rows of cumtrapz(Y, X) and of a
 * spline's knots; *n_cumsimpson_rows (nullable) = rows of cumsimpson(Y, X): len(X) for unsorted X, and for sorted X the rows below the maximum plus ONE for
 * the maximum however often it is repeated (hermiteInterpolate, utils.nim:290-301). */
int nnhip_dataset_rows_f64(const double* X, int n, int* n_sorted_trimmed, int* n_cumsimpson_rows);
/* Output consumer: cumtrapz(Y, X) for discrete points (src/numericalnim/integrate.nim:120-135) over M series (a trajectory
 * tensor's columns).  X [n] host, in any order (sortAndTrimDataset first, :131; strictly ascending X moves nothing); Y, out [n][M] device; the result has
 * nnhip_dataset_rows_f64's n_sorted_trimmed rows, in ascending x (rows beyond: NaN); out[0] = Y[0]-Y[0].  trapz(Y, X) (:104-117) is the
 * last row for finite data. */
int nnhip_cumtrapz_batch_f64_dev(const double* X, int n, const double* Y, int64_t M, double* out, void* stream);
/* cumsimpson(Y, X) for discrete points (integrate.nim:329-375: composite Simpson with non-uniform weights on interval pairs,
 * three-point closure for an odd number of intervals, then hermiteInterpolate (utils.nim:282-312) with dy = Y); same layout as
 * cumtrapz; at least 3 distinct abscissae (else NNHIP_EVALUE, as the reference raises ValueError).  X in any order: the rule runs on the sorted,
 * trimmed data (:340) and the result is returned AT THE CALLER's abscissae, in the caller's order (:375) — nnhip_dataset_rows_f64's n_cumsimpson_rows rows,
 * rows beyond NaN.  Synchronises `stream` before returning. */
int nnhip_cumsimpson_batch_f64_dev(const double* X, int n, const double* Y, int64_t M, double* out, void* stream);
/* The function-argument forms: cumtrapz(f, X, ctx, dx) (src/numericalnim/integrate.nim:138-175) and cumsimpson(f, X, ctx, dx)
 * (:377-400) — sample f on a grid of spacing dx over [min X, max X (+1)], accumulate the rule left to right, resample the running
 * integral at X with hermiteInterpolate (utils.nim:282-312).  The integrand is a function of x alone (NumContextProc,
 * integrate.nim:9): f(x) := rhs(x, y = 0, params) for any thread-per-IVP rhs_kind, compiled-in or registered from source
 * (nnhip_ode_rhs_compile) with dim components.  The batch axis is a parameter sweep: item i uses rhs_params overridden by
 * per_item_params[k*N + i] (device, nullable) for k < n_per_item — N calls of the reference, each with its own ctx.
 *   X [n_x] host, any order (the reference's sorted / unsorted branches are both reproduced, including rows it drops);
 *   out [n_x][dim][N] (SoA) / [n_x][N][dim] (AoS) device; *n_rows_out (nullable) = rows the reference returns (<= n_x);
 *   NNHIP_EVALUE where the reference raises ValueError or would never terminate (dx <= 0).  Synchronises `stream`. */
int nnhip_cumtrapz_fn_batch_f64_dev(int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params,
                                    int n_per_item, int64_t N, int dim, int layout, const double* X, int n_x, double dx,
                                    double* out, int* n_rows_out, void* stream);
int nnhip_cumsimpson_fn_batch_f64_dev(int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params,
                                      int n_per_item, int64_t N, int dim, int layout, const double* X, int n_x, double dx,
                                      double* out, int* n_rows_out, void* stream);
/* Host-pointer forms of the consumers above, for hosts without device-memory management (the Nim shim): every array lives in
 * host memory, the call stages it, runs the device-pointer entry on a private stream and copies the result back; same bits.
 * dY == NULL in the spline entry means newHermiteSpline(X, Y): the slopes are estimated (interpolate.nim:241-253). */
int nnhip_cumtrapz_batch_f64(const double* X, int n, const double* Y, int64_t M, double* out, int device);
int nnhip_cumsimpson_batch_f64(const double* X, int n, const double* Y, int64_t M, double* out, int device);
int nnhip_hermite_spline_eval_batch_f64(const double* X, int n_knots, const double* Y, const double* dY, int64_t M, const double* xq,
                                        int n_q, int deriv, int extrap, double extrap_value, double* out, int device);
int nnhip_cumtrapz_fn_batch_f64(int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params, int n_per_item,
                                int64_t N, int dim, int layout, const double* X, int n_x, double dx, double* out, int* n_rows_out,
                                int device);
int nnhip_cumsimpson_fn_batch_f64(int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params,
                                  int n_per_item, int64_t N, int dim, int layout, const double* X, int n_x, double dx, double* out,
                                  int* n_rows_out, int device);
/* The adaptive controller's step-size factor min(4, max(0.125, 0.9 * pow(1/error, 1/order))) (ode.nim:71, 537) over an
 * array of error norms; order in {2, 3, 5, 6} (rk21, bs32, dopri54/tsit54, vern65). */
int nnhip_ode_controller_factor_f64_dev(int order, const double* error, double* out, int64_t n, void* stream);
/* RHS evaluation alone over a batch: dy = f(t, y) (pins the compiled-in RHS library). */
int nnhip_ode_rhs_batch_f64_dev(int rhs_kind, const double* rhs_params, int n_params, int64_t N, int dim, int layout,
                                double t, const double* y, double* dy, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNHIP_ODE_H */

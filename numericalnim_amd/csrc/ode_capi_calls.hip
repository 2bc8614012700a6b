// ode_capi_calls.hip — the extern "C" boundary, part 3: batches whose members are different calls.  N reference calls with their own tEnd / tspan and
// option fields in one launch (nnhip_ode_solve_batch_calls_f64[_dev], _tend_, _tspans_: ode.nim:589-591 — every call owns its tspan and its ODEoptions), and
// divergence binning below the boundary (nnhip_ode_solve_batch_sorted_f64[_dev]): the order of integration is free because the calls are independent, so
// the batch is integrated in bins of similar work, most work first, and every result is written at the call's own index.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ode_capi_internal.hpp"

using namespace nnhip_capi;

// (not while the caller's stream is being captured into a graph: the scratch comes from the stream-ordered allocator, which a capture may refuse)
static bool stream_is_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}
constexpr int64_t kCallsBinMinN = 32768;  // five small launches (~30 us) in front of the solve: worth it where the solve is long

extern "C" {

// ---- every IVP its own solveODE call ------------------------------------------------------------------------------------
// In the reference every IVP is a solveODE call with its own tspan AND its own ODEoptions (ode.nim:589-591, 476-480, 26-34).
// Here: tspan_i = [t_start[i] (or options.tStart), t_end[i]] and, optionally, per-IVP absTol / relTol / dtMax / dtMin / dt (device
// arrays [N]; NULL = the batch-wide value of `opt`).  Output y_out [2][dim][N] (SoA) / [2][N][dim] (AoS) holds, per IVP, the rows the
// reference returns for tspan.sorted(): (y0, y(tEnd)) when tEnd > tStart, (y(tEnd), y0) when tEnd < tStart (backward branch), the
// single row y0 when they coincide (ny_out[i] = 1, second row NaN).  Per-IVP option values go through abs() like newODEoptions
// does; an IVP whose options the reference's newODEoptions would reject (dtMax < dtMin), that could never finish (fixed-step dt == 0;
// dtMin == 0 without max_steps) or whose span is not finite gets ny_out[i] = -1 and NaN rows.
int nnhip_ode_solve_batch_calls_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                        const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim, int layout,
                                        const double* t_end, const double* t_start, const double* abs_tol, const double* rel_tol,
                                        const double* dt_max, const double* dt_min, const double* dt_fixed, double* y_out, int32_t* ny_out,
                                        int64_t* steps_out, int64_t* rejected_out, int64_t max_steps, void* stream) {
  if (N > 0 && !t_end) return fail(NNHIP_EVALUE, "t_end is NULL");
  if (!opt) return fail(NNHIP_EVALUE, "options is NULL");
  // a 2-point placeholder tspan on the forward side: validation, dispatch and the batch-wide option fields; the rest is per IVP
  const double tspan[2] = {opt->tStart, opt->tStart + 1.0};
  nnhip_ode_options o = *opt;
  if (integrator >= 0 && integrator < NNHIP_N_INTEGRATORS) {  // per-IVP values replace the fields prepare_solve would refuse as batch-wide ones
    if (!kMethods[integrator].adaptive && dt_fixed && !(o.dt > 0.0)) o.dt = 1.0;
    if (kMethods[integrator].adaptive && dt_min && !(o.dtMin > 0.0)) o.dtMin = o.dtMax > 0.0 ? o.dtMax : 1.0;
  }
  PreparedSolve ps;
  int rc = prepare_solve(&o, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y0, N, dim, layout, tspan, 2, nullptr, y_out, ny_out,
                         steps_out, rejected_out, max_steps, nullptr, 0, nullptr, nullptr, (hipStream_t)stream, ps);
  if (rc) return rc;
  ps.a.perCall.tEnd = t_end; ps.a.perCall.tStart = t_start; ps.a.perCall.absTol = abs_tol; ps.a.perCall.relTol = rel_tol;
  ps.a.perCall.dtMax = dt_max; ps.a.perCall.dtMin = dt_min; ps.a.perCall.dt = dt_fixed;
  ps.a.nZero = 1;                                   // tStart_i is in every tspan_i
  ps.a.uniformFull[0] = ps.a.uniformFull[1] = -1;   // no batch-wide step schedule: the spans differ
  ps.a.nTail[0] = ps.a.nTail[1] = 0;
  // N separate calls rarely take the same number of steps: lanes of a wavefront finish together when the calls are integrated longest span first, binned
  // by |tEnd_i - tStart_i| (results are written at the call's own index: the same bits, in the caller's order).  Nothing here waits for the device: whether
  // the spans differ enough to bother (5 %, knob "sort_min_spread_permille") is decided by the binning kernel itself; the scratch comes from the stream-ordered
  // allocator, and without it the calls run in the caller's order.
  hipStream_t s = (hipStream_t)stream;
  if (g_calls_bin && N >= kCallsBinMinN && N < ((int64_t)1 << 31) && !stream_is_capturing(s)) {
    const size_t colKey = ((size_t)N * 8 + 255) & ~(size_t)255, colPerm = ((size_t)N * 4 + 255) & ~(size_t)255;
    const int64_t sortBytes = nnhip::argsort_workspace_bytes(N);
    char* d = nullptr;
    if (hipMallocAsync((void**)&d, colKey + colPerm + (size_t)sortBytes, s) == hipSuccess && d) {
      double* key = (double*)d;
      uint32_t* perm = (uint32_t*)(d + colKey);
      void* sortWs = d + colKey + colPerm;
      bool ok = nnhip::span_key_f64(t_end, t_start, opt->tStart, key, N, s) == hipSuccess;
      ok = ok && nnhip::key_range_f64(key, N, sortWs, nullptr, s) == hipSuccess;
      ok = ok && nnhip::argsort_f64(key, N, perm, sortWs, sortBytes, s, std::max(g_sort_min_spread.load(), 1e-300)) == hipSuccess;
      if (ok) ps.a.perm = perm;
      rc = launch_solve_range(ps, 0, N, s);
      (void)hipFreeAsync(d, s);
      return rc;
    }
    (void)hipGetLastError();
  }
  return launch_solve_range(ps, 0, N, s);
}

int nnhip_ode_solve_batch_tend_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                       const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim, int layout,
                                       const double* t_end, double* y_out, int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out,
                                       int64_t max_steps, void* stream) {
  return nnhip_ode_solve_batch_calls_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y0, N, dim, layout, t_end, nullptr,
                                             nullptr, nullptr, nullptr, nullptr, nullptr, y_out, ny_out, steps_out, rejected_out, max_steps, stream);
}

// ---- every IVP its own n_t-point tspan ------------------------------------------------------------------------------
// N reference calls solveODE(f, y0_i, tspan_i, options_i) with tspan_i = tspans[i][0 .. n_t) — any order, both sides of tStart_i,
// duplicates, tStart_i inside or not (ode.nim:589-591, 476-487, 609).  A device pre-pass sorts and splits every row
// (nnhip::prepare_tspans); the fused kernels then read their own requested times.  Workspace: the prepared rows + the counts.
int64_t nnhip_ode_solve_tspans_workspace_bytes(int64_t N, int n_t) {
  if (!batch_size_sane(N, (int64_t)n_t + 2)) return 0;
  return (((int64_t)N * n_t * 8 + 255) & ~(int64_t)255) + (((int64_t)N * 3 * 4 + 255) & ~(int64_t)255) + 256;
}

int nnhip_ode_solve_batch_tspans_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                         const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim, int layout,
                                         const double* tspans, int n_t, const double* t_start, const double* abs_tol, const double* rel_tol,
                                         const double* dt_max, const double* dt_min, const double* dt_fixed, double* t_out, double* y_out,
                                         int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps, void* ws, int64_t ws_bytes,
                                         void* stream) {
  if (!opt) return fail(NNHIP_EVALUE, "options is NULL");
  if (n_t < 0 || N < 0) return fail(NNHIP_EVALUE, "bad sizes");
  if (N > 0 && n_t > 0 && (!tspans || !y_out)) return fail(NNHIP_EVALUE, "tspans / y_out is NULL");
  if (N > 0 && n_t > 0 && (!ws || ws_bytes < nnhip_ode_solve_tspans_workspace_bytes(N, n_t)))
    return fail(NNHIP_EVALUE, "workspace missing or too small: need %lld bytes", (long long)nnhip_ode_solve_tspans_workspace_bytes(N, n_t));
  hipStream_t s = (hipStream_t)stream;
  const double tspan2[2] = {opt->tStart, opt->tStart + 1.0};  // placeholder for validation, dispatch and the batch-wide option fields
  nnhip_ode_options o = *opt;
  if (integrator >= 0 && integrator < NNHIP_N_INTEGRATORS) {  // per-IVP values replace the fields prepare_solve would refuse as batch-wide ones
    if (!kMethods[integrator].adaptive && dt_fixed && !(o.dt > 0.0)) o.dt = 1.0;
    if (kMethods[integrator].adaptive && dt_min && !(o.dtMin > 0.0)) o.dtMin = o.dtMax > 0.0 ? o.dtMax : 1.0;
  }
  PreparedSolve ps;
  int rc = prepare_solve(&o, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y0, N, dim, layout, tspan2, 2, nullptr,
                         y_out ? y_out : (double*)y0 /* n_t == 0: nothing is written */, ny_out, steps_out, rejected_out, max_steps, nullptr, 0, nullptr,
                         nullptr, s, ps);
  if (rc) return rc;
  if (N == 0) return NNHIP_OK;
  if (n_t == 0) {  // solveODE with an empty tspan returns no rows
    if (ny_out) HIP_TRY(hipMemsetAsync(ny_out, 0, (size_t)N * sizeof(int32_t), s));
    if (steps_out) HIP_TRY(hipMemsetAsync(steps_out, 0, (size_t)N * sizeof(int64_t), s));
    if (rejected_out) HIP_TRY(hipMemsetAsync(rejected_out, 0, (size_t)N * sizeof(int64_t), s));
    return NNHIP_OK;
  }
  double* grid = (double*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  int32_t* counts = (int32_t*)((char*)grid + (((size_t)N * (size_t)n_t * 8 + 255) & ~(size_t)255));
  // as for the per-IVP tEnd solves: from kCallsBinMinN calls on, the calls that integrate over the longest time go first (the key comes out of the pre-pass)
  char* d = nullptr;
  double* spanKey = nullptr;
  const size_t colKey = ((size_t)N * 8 + 255) & ~(size_t)255, colPerm = ((size_t)N * 4 + 255) & ~(size_t)255;
  const int64_t sortBytes = nnhip::argsort_workspace_bytes(N);
  if (g_calls_bin && N >= kCallsBinMinN && N < ((int64_t)1 << 31) && !stream_is_capturing(s)) {
    if (hipMallocAsync((void**)&d, colKey + colPerm + (size_t)sortBytes, s) == hipSuccess && d) spanKey = (double*)d;
    else { d = nullptr; (void)hipGetLastError(); }
  }
  auto done = [&](int code) { if (d) (void)hipFreeAsync(d, s); return code; };
  if (nnhip::prepare_tspans(tspans, n_t, N, t_start, opt->tStart, grid, counts, t_out, s, spanKey) != hipSuccess) return done(fail(NNHIP_EHIP, "preparing the per-IVP tspans failed"));
  ps.a.n_t = n_t;
  ps.a.useDense = n_t != 2 ? 1 : 0;  // :499-502
  ps.a.perCall.tGrid = grid; ps.a.perCall.tCounts = counts;
  ps.a.perCall.tStart = t_start; ps.a.perCall.absTol = abs_tol; ps.a.perCall.relTol = rel_tol;
  ps.a.perCall.dtMax = dt_max; ps.a.perCall.dtMin = dt_min; ps.a.perCall.dt = dt_fixed;
  ps.a.uniformFull[0] = ps.a.uniformFull[1] = -1;   // no batch-wide step schedule: the spans differ
  ps.a.nTail[0] = ps.a.nTail[1] = 0;
  if (spanKey) {
    uint32_t* perm = (uint32_t*)(d + colKey);
    void* sortWs = d + colKey + colPerm;
    bool ok = nnhip::key_range_f64(spanKey, N, sortWs, nullptr, s) == hipSuccess;
    ok = ok && nnhip::argsort_f64(spanKey, N, perm, sortWs, sortBytes, s, std::max(g_sort_min_spread.load(), 1e-300)) == hipSuccess;
    if (ok) ps.a.perm = perm;
  }
  return done(launch_solve_range(ps, 0, N, s));
}

// ---- divergence binning below the boundary -----------------------------------------------------------------------
// Workspace of nnhip_ode_solve_batch_sorted_f64_dev: requested times + order of integration (4N) + probe progress / key (8N)
// + the device sort's scratch.
int64_t nnhip_ode_solve_sorted_workspace_bytes(int64_t N, int n_t) {
  if (!batch_size_sane(N, 8)) return 0;
  // [requested times / schedule][perm: 4N][key: 8N][resume state t, dt, tEnd: 3 x 8N][argsort workspace]
  return ((nnhip_ode_solve_workspace_bytes(n_t) + 255) & ~(int64_t)255) + (((int64_t)N * 4 + 255) & ~(int64_t)255) + 4 * (((int64_t)N * 8 + 255) & ~(int64_t)255) +
         nnhip::argsort_workspace_bytes(N) + 256;
}

// The order of integration the binned solve derives from `sort_key`: order_out[k] = index of the k-th IVP to be integrated (a binning of the keys
// into 4096 slices of their range, see ode_sort.hip — ascending from slice to slice, unordered inside one; non-finite keys last).  What the
// sorted entry does internally, exposed so that a caller can look at it (tests: bin occupancy for keys containing 0 / of both signs) or reuse it.
int nnhip_ode_bin_order_f64_dev(const double* sort_key, int64_t N, uint32_t* order_out, void* stream) {
  if (N < 0 || N >= ((int64_t)1 << 31)) return fail(NNHIP_EVALUE, "N must be in [0, 2^31)");
  if (N == 0) return NNHIP_OK;
  if (!sort_key || !order_out) return fail(NNHIP_EVALUE, "sort_key / order_out is NULL");
  hipStream_t s = (hipStream_t)stream;
  const int64_t bytes = nnhip::argsort_workspace_bytes(N);
  void* ws = nullptr;
  HIP_TRY(hipMallocAsync(&ws, (size_t)bytes, s));
  bool ok = nnhip::key_range_f64(sort_key, N, ws, nullptr, s) == hipSuccess;
  ok = ok && nnhip::argsort_f64(sort_key, N, order_out, ws, bytes, s) == hipSuccess;
  (void)hipFreeAsync(ws, s);
  return ok ? NNHIP_OK : fail(NNHIP_EHIP, "binning the keys failed");
}

// solveODE over a batch whose members take very different step sequences (heterogeneous parameters / initial states): the
// IVPs are integrated in ascending order of `sort_key` — neighbouring lanes of a wavefront then agree on accept / reject and
// finish together — and every result is written at the IVP's own index (SolveArgs::perm), so the output is in the caller's
// order and bit-identical to the unsorted solve.  sort_key == NULL selects the automatic two-pass mode: a probe solve of
// `probe_steps` accepted steps per IVP (default 8) measures how far each IVP gets, which ranks the step sizes the controller
// settles on; the batch is then integrated in that order.  Fixed-step methods have no divergence: they run unsorted.
int nnhip_ode_solve_batch_sorted_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                         const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim, int layout,
                                         const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                                         int64_t* rejected_out, int64_t max_steps, const double* sort_key, int probe_steps, void* ws,
                                         int64_t ws_bytes, void* stream) {
  if (N < 0 || N >= ((int64_t)1 << 31)) return fail(NNHIP_EVALUE, "N must be in [0, 2^31)");
  if (integrator < 0 || integrator >= NNHIP_N_INTEGRATORS) return fail(NNHIP_EINTEGRATOR, "%d is not a valid integrator", integrator);
  if (N > 0 && (!ws || ws_bytes < nnhip_ode_solve_sorted_workspace_bytes(N, n_t))) return fail(NNHIP_EVALUE, "workspace missing or too small: need %lld bytes", (long long)nnhip_ode_solve_sorted_workspace_bytes(N, n_t));
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)ws;
  const int64_t wsTimes = (nnhip_ode_solve_workspace_bytes(n_t) + 255) & ~(int64_t)255;
  auto at = [base](int64_t off) -> char* { return base ? base + off : nullptr; };  // (N == 0 comes without a workspace: nothing below is touched then)
  uint32_t* perm = (uint32_t*)at(wsTimes);
  const int64_t colBytes = ((int64_t)N * 8 + 255) & ~(int64_t)255;
  const int64_t keyOff = wsTimes + (((int64_t)N * 4 + 255) & ~(int64_t)255);
  double* key = (double*)at(keyOff);
  double* resT = (double*)at(keyOff + colBytes);    // resume state of the automatic mode: t, dt where the probe stopped, and tEnd as a column
  double* resDt = (double*)at(keyOff + 2 * colBytes);
  double* resEnd = (double*)at(keyOff + 3 * colBytes);
  void* sortWs = at(keyOff + 4 * colBytes);
  const int64_t sortWsBytes = ws_bytes - (keyOff + 4 * colBytes);
  const bool adaptive = kMethods[integrator].adaptive != 0;
  PreparedSolve ps;
  bool sorted = adaptive && N > 1;  // integrate in the order of `perm`
  double g0End = 0.0;               // tEnd of the forward span (resumed automatic mode)
  bool resumed = false;             // the probe's steps are kept: the sorted pass continues from where the probe stopped
  int rc = NNHIP_OK;
  if (adaptive && N > 1) {
    if (!sort_key) {  // pass 1: the probe.  The same solve, cut off after probe_steps accepted steps; its progress is the key.
      if (probe_steps <= 0) probe_steps = 8;  // scripts/ab_probe_steps.py (1e6 Van der Pol IVPs, round 4): <= 4 steps do not rank (every controller is still ramping up from dtInit: the keys tie and the batch runs unsorted, 2.48 ms), 6 -> 1.47 ms, 8 -> 1.45 ms, 12 -> 1.55 ms, 16 -> 1.81 ms
      if (max_steps > 0 && probe_steps > max_steps) probe_steps = (int)max_steps;
      // Resume instead of restart where the loop's whole state is (t, dt, y): a forward 2-point tspan [tStart, tEnd], a method whose FSAL is f(t, y) of
      // the state it returns (DOPRI54, Tsit54, BS32) or that has none (RK21), a right-hand side without mutable slots.  The probe then writes its
      // state into the caller's own output rows and the sorted pass takes it up as N per-IVP calls [t_i, tEnd] with dtInit_i = the probe's dt (the
      // per-call instantiation of the solve kernel, nnhip_ode_solve_batch_calls_f64_dev): the same loop iterations on the same values, 8 of ~130 steps
      // not integrated twice.  Everything else restarts after the probe, as before.
      TimeGrid g0;
      if (opt && tspan && n_t >= 1) make_grid(opt, tspan, n_t, g0);
      const bool fsalIsF = integrator == NNHIP_DOPRI54 || integrator == NNHIP_TSIT54 || integrator == NNHIP_BS32 || integrator == NNHIP_RK21;
      const bool canResume = (g_sort_resume || g_sort_rebin_steps > 0) && n_t == 2 && opt && opt->dtMin > 0.0 && opt->dtMax >= opt->dtMin && g0.tNeg.empty() && g0.tPos.size() == 1 && g0.nZero == 1 && fsalIsF && !g_sort_copy &&
                             !nnhip::rtc_has_aux(rhs_kind) && (max_steps <= 0 || max_steps > probe_steps) && y_out != nullptr;
      rc = prepare_solve(opt, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y0, N, dim, layout, tspan, n_t, canResume ? t_out : nullptr, y_out,
                         nullptr, canResume ? steps_out : nullptr, canResume ? rejected_out : nullptr, probe_steps, ws, wsTimes, nullptr, nullptr, s, ps);
      if (rc) return rc;
      ps.a.progress_out = key;
      const bool byRemaining = g_sort_auto_key == 1 && g0.tNeg.empty() && !g0.tPos.empty();
      if (canResume || byRemaining) { ps.a.tfinal_out = resT; ps.a.dtfinal_out = resDt; }
      rc = launch_solve_range(ps, 0, N, s);
      if (rc) return rc;
      // The IVPs with the most work left go FIRST (the workgroups dispatched last then hold the short solves and the kernel's tail is short:
      // 1.35 -> 1.27 ms on the pre-sorted sweep of scripts/bench_divergence.py): ascending progress, or descending estimate of the steps still to
      // take.  Non-finite keys sort last; the order among equal keys is the caller's (stable sort).
      if (byRemaining) HIP_TRY(nnhip::remaining_key_f64(resT, resDt, g0.tEndPos, key, N, s));
      sort_key = key;
      resumed = canResume;
      if (resumed) { g0End = g0.tEndPos; HIP_TRY(nnhip::launch_fill_f64(resEnd, N, g0.tEndPos, s)); }
    }
    // Nothing to gain?  Keys within 5 % of each other (relative to their magnitude: a parameter sweep over [100, 101], a probe in which every
    // IVP got equally far) promise no better lane utilisation than the caller's order, and the sort + the indirection would cost ~20 % of such a
    // solve (profiles/r03_bench_divergence.json: 0.78 vs 0.65 ms).  The key's range is reduced on the device into page-locked host memory;
    // reading it synchronises `stream` once (the one place this entry waits for the device).
    static_assert(sizeof(unsigned long long) == sizeof(double), "the key range shares the pinned staging buffer");
    bool worthSorting = true;
    if (g_sort_min_spread > 0.0 && N >= 4096) {  // (a small batch is binned in any case: the check would cost as much)
      rc = stage_reserve(2);
      if (rc) return rc;
      unsigned long long* img = (unsigned long long*)g_stage.host;
      HIP_TRY(nnhip::key_range_f64(sort_key, N, sortWs, img, s));
      HIP_TRY(hipStreamSynchronize(s));
      double mn = 0.0, mx = 0.0;
      nnhip::key_range_decode(img, &mn, &mx);
      const double scale = std::fabs(mn) > std::fabs(mx) ? std::fabs(mn) : std::fabs(mx);
      worthSorting = mn <= mx && scale > 0.0 && (mx - mn) > g_sort_min_spread * scale;
    } else {
      HIP_TRY(nnhip::key_range_f64(sort_key, N, sortWs, nullptr, s));  // the binning reads the range on the device; nothing waits
    }
    if (worthSorting) HIP_TRY(nnhip::argsort_f64(sort_key, N, perm, sortWs, sortWsBytes, s));
    else sorted = false;
  }
  if (sorted && g_sort_copy && dim >= 1) {
    // The batch in integration order, physically: gather y0 (and the per-IVP parameter table), solve with coalesced accesses, scatter the
    // rows and counters back to the caller's order.  Temporaries come from the stream-ordered allocator; if that fails the solve
    // kernel follows `perm` itself (SolveArgs::perm, below).
    const size_t nState = (size_t)N * (size_t)dim, nOut = nState * (size_t)(n_t > 0 ? n_t : 0);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t oY0 = 0, oOut = oY0 + up(nState * 8), oPer = oOut + up(nOut * 8), oNy = oPer + up((size_t)(n_per_ivp > 0 ? n_per_ivp : 0) * (size_t)N * 8),
                 oSt = oNy + up((size_t)N * 4), oRj = oSt + up((size_t)N * 8), oInv = oRj + up((size_t)N * 8), total = oInv + up((size_t)N * 4);
    char* d = nullptr;
    if (hipMallocAsync((void**)&d, total, s) == hipSuccess && d) {
      const int R0 = layout == NNHIP_LAYOUT_SOA ? dim : 1, W0 = layout == NNHIP_LAYOUT_SOA ? 1 : dim;
      auto bail = [&](int code) { (void)hipFreeAsync(d, s); return code; };
      if (nnhip::gather_f64(y0, (double*)(d + oY0), perm, N, R0, W0, s) != hipSuccess) return bail(fail(NNHIP_EHIP, "gathering y0 into integration order failed"));
      if (n_per_ivp > 0 && nnhip::gather_f64(per_ivp_params, (double*)(d + oPer), perm, N, n_per_ivp, 1, s) != hipSuccess)
        return bail(fail(NNHIP_EHIP, "gathering the per-IVP parameters into integration order failed"));
      int rc = prepare_solve(opt, integrator, rhs_kind, rhs_params, n_params, n_per_ivp > 0 ? (const double*)(d + oPer) : nullptr, n_per_ivp, (const double*)(d + oY0), N, dim,
                             layout, tspan, n_t, t_out, (double*)(d + oOut), ny_out ? (int32_t*)(d + oNy) : nullptr, steps_out ? (int64_t*)(d + oSt) : nullptr,
                             rejected_out ? (int64_t*)(d + oRj) : nullptr, max_steps, ws, wsTimes, nullptr, nullptr, s, ps);
      if (rc) return bail(rc);
      if (ps.a.P.ivp || ps.a.P.aux) return bail(fail(NNHIP_EUNSUPPORTED, "a right-hand side with a per-IVP context block is not available in the binned solve"));
      rc = launch_solve_range(ps, 0, N, s);
      if (rc) return bail(rc);
      // back to the caller's order through the inverse order, as gathers (full-line writes; see invert_perm)
      const uint32_t* inv = (const uint32_t*)(d + oInv);
      bool ok = nnhip::invert_perm(perm, (uint32_t*)(d + oInv), N, s) == hipSuccess;
      ok = ok && nnhip::gather_f64((const double*)(d + oOut), y_out, inv, N, (layout == NNHIP_LAYOUT_SOA ? dim : 1) * n_t, W0, s) == hipSuccess;
      if (ny_out) ok = ok && nnhip::gather_i32((const int32_t*)(d + oNy), ny_out, inv, N, s) == hipSuccess;
      if (steps_out) ok = ok && nnhip::gather_i64((const int64_t*)(d + oSt), steps_out, inv, N, s) == hipSuccess;
      if (rejected_out) ok = ok && nnhip::gather_i64((const int64_t*)(d + oRj), rejected_out, inv, N, s) == hipSuccess;
      (void)hipFreeAsync(d, s);
      return ok ? NNHIP_OK : fail(NNHIP_EHIP, "scattering the results back to the caller's order failed");
    }
    (void)hipGetLastError();
  }
  if (resumed) {
    // pass 2 of the automatic mode, resuming: N per-IVP calls [t_i, tEnd] from the probe's state (row 1 of the caller's output, where the probe left it),
    // first step size = the probe's dt, in sorted order; per-IVP counters are added to the probe's; row 0 is restored from y0 afterwards.
    // With knob "sort_rebin_steps" = S > 0 there is a pass in between: every IVP takes up to S more accepted steps in the probe's order, the batch is
    // binned again by the steps still to take where each IVP stands NOW, and the last pass finishes it — for batches whose step sizes change late in
    // the span, where the order found after 8 steps has gone stale.  Cut-offs by step count leave the step sequence alone (as the probe's does), so
    // the bits stay those of the plain solve.
    const double span2[2] = {opt->tStart, opt->tStart + 1.0};  // placeholder for validation and dispatch: the spans are per IVP
    int64_t used = probe_steps;
    const bool rebin = g_sort_rebin_steps > 0 && (max_steps <= 0 || max_steps - used > g_sort_rebin_steps);
    for (int pass = rebin ? 0 : 1; pass < 2; ++pass) {
      const bool last = pass == 1;
      rc = prepare_solve(opt, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y_out + (int64_t)dim * N, N, dim, layout, span2, 2, nullptr, y_out,
                         ny_out, steps_out, rejected_out, last ? (max_steps > 0 ? max_steps - used : 0) : (int64_t)g_sort_rebin_steps, nullptr, 0, nullptr, nullptr, s, ps);
      if (rc) return rc;
      ps.a.perCall.tEnd = resEnd; ps.a.perCall.tStart = resT; ps.a.perCall.dtInit = resDt; ps.a.perCall.resume = 1;
      ps.a.nZero = 1;
      ps.a.uniformFull[0] = ps.a.uniformFull[1] = -1;
      ps.a.nTail[0] = ps.a.nTail[1] = 0;
      ps.a.accumulate = 1;
      if (sorted) ps.a.perm = perm;
      if (!last) { ps.a.tfinal_out = resT; ps.a.dtfinal_out = resDt; }  // (each work item reads its own start before it writes where it stopped)
      rc = launch_solve_range(ps, 0, N, s);
      if (rc) return rc;
      if (!last) {
        used += g_sort_rebin_steps;
        HIP_TRY(nnhip::remaining_key_f64(resT, resDt, g0End, key, N, s));
        HIP_TRY(nnhip::key_range_f64(key, N, sortWs, nullptr, s));
        HIP_TRY(nnhip::argsort_f64(key, N, perm, sortWs, sortWsBytes, s));
        sorted = true;
      }
    }
    HIP_TRY(hipMemcpyAsync(y_out, y0, (size_t)dim * (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, s));  // row 0 = y0 (`t0 in tspan`, ode.nim:485-487)
    return NNHIP_OK;
  }
  rc = prepare_solve(opt, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y0, N, dim, layout, tspan, n_t, t_out, y_out, ny_out,
                     steps_out, rejected_out, max_steps, ws, wsTimes, nullptr, nullptr, s, ps);
  if (rc) return rc;
  if (sorted) ps.a.perm = perm;
  return launch_solve_range(ps, 0, N, s);
}

// Host-pointer form of the sorted solve (what a Nim host holding `seq`s calls): everything staged through the device in one piece —
// the order of integration needs the whole batch resident, so the chunked transfer overlap of the plain host entry does not apply.
int nnhip_ode_solve_batch_sorted_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                     const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim, int layout,
                                     const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                                     int64_t* rejected_out, int64_t max_steps, const double* sort_key, int probe_steps, int device) {
  if (N < 0 || dim < 1 || n_t < 0) return fail(NNHIP_EVALUE, "bad sizes");
  if (N >= ((int64_t)1 << 31) || !batch_size_sane(N, (int64_t)dim * ((int64_t)n_t + 1) + (n_per_ivp > 0 ? n_per_ivp : 0))) return fail(NNHIP_EVALUE, "N must be in [0, 2^31) and the batch addressable");
  if (N > 0 && (!y0 || (n_t > 0 && !y_out))) return fail(NNHIP_EVALUE, "y0 / y_out is NULL");  // before anything is allocated for them
  int ndev = nnhip_device_count();
  if (ndev < 0) return ndev;
  if (ndev == 0) return fail(NNHIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(NNHIP_EVALUE, "device %d out of range [0,%d)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  HostSolveCtx* hc = nullptr;
  int rc = host_ctx_acquire(device, 0, &hc);
  if (rc) { host_ctx_release(hc); return rc; }
  hipStream_t st = hc->s[0];
  const size_t nState = (size_t)N * dim, nOut = nState * (size_t)n_t;
  const int64_t wsBytes = nnhip_ode_solve_sorted_workspace_bytes(N, n_t);
  // one allocation: y0 | out | ny | steps | rejected | key | per-IVP table | workspace
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t oY0 = 0, oOut = oY0 + up(nState * 8), oNy = oOut + up(nOut * 8), oSt = oNy + up((size_t)N * 4), oRj = oSt + up((size_t)N * 8),
               oKey = oRj + up((size_t)N * 8), oPer = oKey + up((size_t)N * 8), oWs = oPer + up((size_t)(n_per_ivp > 0 ? n_per_ivp : 0) * (size_t)N * 8),
               total = oWs + up((size_t)wsBytes) + 256;
  char* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, total);
  if (e != hipSuccess) { host_ctx_release(hc); return fail(e == hipErrorOutOfMemory ? NNHIP_ENOMEM : NNHIP_EHIP, "hipMalloc(%zu) failed: %s", total, hipGetErrorString(e)); }
  auto done = [&](int code) { (void)hipStreamSynchronize(st); (void)hipFree(d); host_ctx_release(hc); return code; };
#define HIP_TRY_S(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return done(fail(NNHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e))); } while (0)
  if (nState) HIP_TRY_S(hipMemcpyAsync(d + oY0, y0, nState * 8, hipMemcpyHostToDevice, st));
  if (sort_key && N) HIP_TRY_S(hipMemcpyAsync(d + oKey, sort_key, (size_t)N * 8, hipMemcpyHostToDevice, st));
  if (n_per_ivp > 0 && N) {
    if (!per_ivp_params) return done(fail(NNHIP_EVALUE, "bad per-IVP parameter table"));
    HIP_TRY_S(hipMemcpyAsync(d + oPer, per_ivp_params, (size_t)n_per_ivp * (size_t)N * 8, hipMemcpyHostToDevice, st));
  }
  rc = nnhip_ode_solve_batch_sorted_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, n_per_ivp > 0 ? (const double*)(d + oPer) : nullptr, n_per_ivp,
                                            (const double*)(d + oY0), N, dim, layout, tspan, n_t, t_out, (double*)(d + oOut), ny_out ? (int32_t*)(d + oNy) : nullptr,
                                            steps_out ? (int64_t*)(d + oSt) : nullptr, rejected_out ? (int64_t*)(d + oRj) : nullptr, max_steps,
                                            sort_key ? (const double*)(d + oKey) : nullptr, probe_steps, d + oWs, wsBytes, st);
  if (rc) return done(rc);
  if (nOut) HIP_TRY_S(hipMemcpyAsync(y_out, d + oOut, nOut * 8, hipMemcpyDeviceToHost, st));
  if (ny_out && N) HIP_TRY_S(hipMemcpyAsync(ny_out, d + oNy, (size_t)N * 4, hipMemcpyDeviceToHost, st));
  if (steps_out && N) HIP_TRY_S(hipMemcpyAsync(steps_out, d + oSt, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  if (rejected_out && N) HIP_TRY_S(hipMemcpyAsync(rejected_out, d + oRj, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY_S(hipStreamSynchronize(st));
#undef HIP_TRY_S
  return done(NNHIP_OK);
}

// Host-pointer form of the per-call solve: N reference calls `solveODE(f, y0_i, [options_i.tStart, t_end[i]], options_i)` in one
// launch (ode.nim:589-591).  opt_per_ivp is an array of N option objects (NULL: every call uses `opt`); their fields are transposed
// into the per-IVP tables of nnhip_ode_solve_batch_calls_f64_dev here, whose per-field semantics apply (scaleMax / scaleMin are unused
// after construction, ode.nim:97-102).
int nnhip_ode_solve_batch_calls_f64(const nnhip_ode_options* opt, const nnhip_ode_options* opt_per_ivp, int integrator, int rhs_kind,
                                    const double* rhs_params, int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0,
                                    int64_t N, int dim, int layout, const double* t_end, double* y_out, int32_t* ny_out, int64_t* steps_out,
                                    int64_t* rejected_out, int64_t max_steps, int device) {
  if (N < 0 || dim < 1) return fail(NNHIP_EVALUE, "bad sizes");
  if (N > 0 && (!t_end || !y0 || !y_out)) return fail(NNHIP_EVALUE, "t_end / y0 / y_out is NULL");
  if (n_per_ivp < 0 || n_per_ivp > nnhip::kMaxParams || (n_per_ivp > 0 && !per_ivp_params && N > 0)) return fail(NNHIP_EVALUE, "bad per-IVP parameter table");
  int ndev = nnhip_device_count();
  if (ndev < 0) return ndev;
  if (ndev == 0) return fail(NNHIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(NNHIP_EVALUE, "device %d out of range [0,%d)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  HostSolveCtx* hc = nullptr;
  int rc = host_ctx_acquire(device, 0, &hc);
  if (rc) { host_ctx_release(hc); return rc; }
  hipStream_t st = hc->s[0];
  const size_t nState = (size_t)N * dim, nOut = nState * 2;
  const int nOpt = opt_per_ivp ? 6 : 0;  // tStart, absTol, relTol, dtMax, dtMin, dt
  std::vector<double> cols;
  if (nOpt && N) {
    cols.resize((size_t)nOpt * (size_t)N);
    for (int64_t i = 0; i < N; ++i) {
      const nnhip_ode_options& o = opt_per_ivp[i];
      cols[0 * (size_t)N + i] = o.tStart;
      cols[1 * (size_t)N + i] = o.absTol;
      cols[2 * (size_t)N + i] = o.relTol;
      cols[3 * (size_t)N + i] = o.dtMax;
      cols[4 * (size_t)N + i] = o.dtMin;
      cols[5 * (size_t)N + i] = o.dt;
    }
  }
  // one allocation: y0 | out | ny | steps | rejected | t_end | option columns | per-IVP table
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t oY0 = 0, oOut = oY0 + up(nState * 8), oNy = oOut + up(nOut * 8), oSt = oNy + up((size_t)N * 4), oRj = oSt + up((size_t)N * 8),
               oEnd = oRj + up((size_t)N * 8), oCols = oEnd + up((size_t)N * 8), oPer = oCols + up((size_t)nOpt * (size_t)N * 8),
               total = oPer + up((size_t)n_per_ivp * (size_t)N * 8) + 256;
  char* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, total);
  if (e != hipSuccess) { host_ctx_release(hc); return fail(e == hipErrorOutOfMemory ? NNHIP_ENOMEM : NNHIP_EHIP, "hipMalloc(%zu) failed: %s", total, hipGetErrorString(e)); }
  auto done = [&](int code) { (void)hipStreamSynchronize(st); (void)hipFree(d); host_ctx_release(hc); return code; };
#define HIP_TRY_S(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return done(fail(NNHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e))); } while (0)
  if (nState) HIP_TRY_S(hipMemcpyAsync(d + oY0, y0, nState * 8, hipMemcpyHostToDevice, st));
  if (N) HIP_TRY_S(hipMemcpyAsync(d + oEnd, t_end, (size_t)N * 8, hipMemcpyHostToDevice, st));
  if (!cols.empty()) HIP_TRY_S(hipMemcpyAsync(d + oCols, cols.data(), cols.size() * 8, hipMemcpyHostToDevice, st));
  if (n_per_ivp > 0 && N) HIP_TRY_S(hipMemcpyAsync(d + oPer, per_ivp_params, (size_t)n_per_ivp * (size_t)N * 8, hipMemcpyHostToDevice, st));
  auto col = [&](int k) -> const double* { return nOpt && N ? (const double*)(d + oCols) + (size_t)k * (size_t)N : nullptr; };
  rc = nnhip_ode_solve_batch_calls_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, n_per_ivp > 0 ? (const double*)(d + oPer) : nullptr, n_per_ivp,
                                           (const double*)(d + oY0), N, dim, layout, (const double*)(d + oEnd), col(0), col(1), col(2), col(3), col(4), col(5),
                                           (double*)(d + oOut), ny_out ? (int32_t*)(d + oNy) : nullptr, steps_out ? (int64_t*)(d + oSt) : nullptr,
                                           rejected_out ? (int64_t*)(d + oRj) : nullptr, max_steps, st);
  if (rc) return done(rc);
  if (nOut) HIP_TRY_S(hipMemcpyAsync(y_out, d + oOut, nOut * 8, hipMemcpyDeviceToHost, st));
  if (ny_out && N) HIP_TRY_S(hipMemcpyAsync(ny_out, d + oNy, (size_t)N * 4, hipMemcpyDeviceToHost, st));
  if (steps_out && N) HIP_TRY_S(hipMemcpyAsync(steps_out, d + oSt, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  if (rejected_out && N) HIP_TRY_S(hipMemcpyAsync(rejected_out, d + oRj, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY_S(hipStreamSynchronize(st));
#undef HIP_TRY_S
  return done(NNHIP_OK);
}

// Host-pointer form of the per-IVP-tspan solve: N reference calls `solveODE(f, y0_i, tspans[i], options_i)` in one launch
// (ode.nim:589-591); tspans [N][n_t] and t_out [N][n_t] in host memory.  opt_per_ivp is an array of N option objects (NULL: every call uses `opt`); their fields are transposed
// into the per-IVP tables of nnhip_ode_solve_batch_calls_f64_dev here, whose per-field semantics apply (scaleMax / scaleMin are unused
// after construction, ode.nim:97-102).
int nnhip_ode_solve_batch_tspans_f64(const nnhip_ode_options* opt, const nnhip_ode_options* opt_per_ivp, int integrator, int rhs_kind,
                                     const double* rhs_params, int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0,
                                     int64_t N, int dim, int layout, const double* tspans, int n_t, double* t_out, double* y_out, int32_t* ny_out,
                                     int64_t* steps_out, int64_t* rejected_out, int64_t max_steps, int device) {
  if (N < 0 || dim < 1 || n_t < 0) return fail(NNHIP_EVALUE, "bad sizes");
  if (N > 0 && (!y0 || (n_t > 0 && (!tspans || !y_out)))) return fail(NNHIP_EVALUE, "tspans / y0 / y_out is NULL");
  if (n_per_ivp < 0 || n_per_ivp > nnhip::kMaxParams || (n_per_ivp > 0 && !per_ivp_params && N > 0)) return fail(NNHIP_EVALUE, "bad per-IVP parameter table");
  int ndev = nnhip_device_count();
  if (ndev < 0) return ndev;
  if (ndev == 0) return fail(NNHIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(NNHIP_EVALUE, "device %d out of range [0,%d)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  HostSolveCtx* hc = nullptr;
  int rc = host_ctx_acquire(device, 0, &hc);
  if (rc) { host_ctx_release(hc); return rc; }
  hipStream_t st = hc->s[0];
  const size_t nState = (size_t)N * dim, nOut = nState * (size_t)n_t, nGrid = (size_t)N * (size_t)n_t;
  const int64_t wsBytes = nnhip_ode_solve_tspans_workspace_bytes(N, n_t);
  const int nOpt = opt_per_ivp ? 6 : 0;  // tStart, absTol, relTol, dtMax, dtMin, dt
  std::vector<double> cols;
  if (nOpt && N) {
    cols.resize((size_t)nOpt * (size_t)N);
    for (int64_t i = 0; i < N; ++i) {
      const nnhip_ode_options& o = opt_per_ivp[i];
      cols[0 * (size_t)N + i] = o.tStart;
      cols[1 * (size_t)N + i] = o.absTol;
      cols[2 * (size_t)N + i] = o.relTol;
      cols[3 * (size_t)N + i] = o.dtMax;
      cols[4 * (size_t)N + i] = o.dtMin;
      cols[5 * (size_t)N + i] = o.dt;
    }
  }
  // one allocation: y0 | out | ny | steps | rejected | tspans | t_out | option columns | per-IVP table | workspace
  auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t oY0 = 0, oOut = oY0 + up(nState * 8), oNy = oOut + up(nOut * 8), oSt = oNy + up((size_t)N * 4), oRj = oSt + up((size_t)N * 8),
               oEnd = oRj + up((size_t)N * 8), oTout = oEnd + up(nGrid * 8), oCols = oTout + up(nGrid * 8), oPer = oCols + up((size_t)nOpt * (size_t)N * 8),
               oWs = oPer + up((size_t)n_per_ivp * (size_t)N * 8), total = oWs + up((size_t)wsBytes) + 256;
  char* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, total);
  if (e != hipSuccess) { host_ctx_release(hc); return fail(e == hipErrorOutOfMemory ? NNHIP_ENOMEM : NNHIP_EHIP, "hipMalloc(%zu) failed: %s", total, hipGetErrorString(e)); }
  auto done = [&](int code) { (void)hipStreamSynchronize(st); (void)hipFree(d); host_ctx_release(hc); return code; };
#define HIP_TRY_S(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return done(fail(NNHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e))); } while (0)
  if (nState) HIP_TRY_S(hipMemcpyAsync(d + oY0, y0, nState * 8, hipMemcpyHostToDevice, st));
  if (nGrid) HIP_TRY_S(hipMemcpyAsync(d + oEnd, tspans, nGrid * 8, hipMemcpyHostToDevice, st));
  if (!cols.empty()) HIP_TRY_S(hipMemcpyAsync(d + oCols, cols.data(), cols.size() * 8, hipMemcpyHostToDevice, st));
  if (n_per_ivp > 0 && N) HIP_TRY_S(hipMemcpyAsync(d + oPer, per_ivp_params, (size_t)n_per_ivp * (size_t)N * 8, hipMemcpyHostToDevice, st));
  auto col = [&](int k) -> const double* { return nOpt && N ? (const double*)(d + oCols) + (size_t)k * (size_t)N : nullptr; };
  rc = nnhip_ode_solve_batch_tspans_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, n_per_ivp > 0 ? (const double*)(d + oPer) : nullptr, n_per_ivp,
                                            (const double*)(d + oY0), N, dim, layout, (const double*)(d + oEnd), n_t, col(0), col(1), col(2), col(3), col(4), col(5),
                                            t_out ? (double*)(d + oTout) : nullptr, (double*)(d + oOut), ny_out ? (int32_t*)(d + oNy) : nullptr,
                                            steps_out ? (int64_t*)(d + oSt) : nullptr, rejected_out ? (int64_t*)(d + oRj) : nullptr, max_steps, d + oWs, wsBytes, st);
  if (rc) return done(rc);
  if (nOut) HIP_TRY_S(hipMemcpyAsync(y_out, d + oOut, nOut * 8, hipMemcpyDeviceToHost, st));
  if (t_out && nGrid) HIP_TRY_S(hipMemcpyAsync(t_out, d + oTout, nGrid * 8, hipMemcpyDeviceToHost, st));
  if (ny_out && N) HIP_TRY_S(hipMemcpyAsync(ny_out, d + oNy, (size_t)N * 4, hipMemcpyDeviceToHost, st));
  if (steps_out && N) HIP_TRY_S(hipMemcpyAsync(steps_out, d + oSt, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  if (rejected_out && N) HIP_TRY_S(hipMemcpyAsync(rejected_out, d + oRj, (size_t)N * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY_S(hipStreamSynchronize(st));
#undef HIP_TRY_S
  return done(NNHIP_OK);
}
}  // extern "C"

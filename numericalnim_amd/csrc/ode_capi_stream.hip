// ode_capi_stream.hip — the extern "C" boundary, part 2: the IntegratorProc seam (ode.nim:38, call sites :531, :573).  One stepper call over a
// device-resident batch (nnhip_ode_step_batch_f64_dev), ODESolver's fixed-step and adaptive loops over it with the state in HBM between launches
// (nnhip_ode_fixed_stream_f64_dev, nnhip_ode_adaptive_stream_f64_dev; hipGraph caches, host-resident "anyone left?" flags), and the whole driver
// with dense output through the same seam (nnhip_ode_fixed_stream_dense_f64_dev, nnhip_ode_adaptive_stream_dense_f64_dev).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "adv_poll_schedule.hpp"
#include "ode_capi_internal.hpp"

using namespace nnhip_capi;

// The variant of the headline kernel a streamed working set of `workingSet` bytes gets.
// Measured on MI355X (profiles/r01_stream_tuning.txt): while the streamed working set fits the 256 MiB
// Infinity Cache plain accesses with one 16-B load per lane win (6.9 TB/s); beyond it, non-temporal
// accesses with 4 loads in flight per lane do (6.4 TB/s vs 5.9).
static nnhip::StreamTune rk4_stream_tune_for(int64_t workingSet) {
  nnhip::StreamTune tune = tune_snapshot();
  if (g_tune_auto) {
    if (workingSet <= (192LL << 20)) { tune.vec = 1; tune.mode = 0; } else { tune.vec = 4; tune.mode = 1; }
  }
  return tune;
}

extern "C" {

int nnhip_ode_rk4_stream_variant(int64_t n_states, int in_place, int* vec, int* mode) {
  if (n_states < 0 || !vec || !mode) return fail(NNHIP_EVALUE, "n_states < 0 or vec / mode is NULL");
  const nnhip::StreamTune tune = rk4_stream_tune_for(8 * n_states * (in_place ? 1 : 2));
  *vec = tune.vec; *mode = tune.mode;
  return NNHIP_OK;
}

// ---- step-streaming ---------------------------------------------------------------------------------
int nnhip_ode_step_batch_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                 int n_params, int64_t N, int dim, int layout, const double* t_dev, double t_uniform,
                                 const double* dt_dev, double dt_uniform, const double* y_in, const double* fsal_in,
                                 double* y_out, double* fsal_out, double* dt_used, double* error, int negate_time,
                                 void* stream) {
  nnhip::Params P;
  int rc = check_common(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, P);
  if (rc) return rc;
  if (!kMethods[integrator].implemented) return fail(NNHIP_EUNSUPPORTED, "integrator %s has no HIP kernel yet", kMethods[integrator].name);
  if (N == 0) return NNHIP_OK;
  if (!y_in || !y_out) return fail(NNHIP_EVALUE, "y_in / y_out is NULL");
  if (kMethods[integrator].useFSAL && (!fsal_in || !fsal_out)) return fail(NNHIP_EVALUE, "FSAL methods need fsal_in and fsal_out");
  if (kMethods[integrator].adaptive && !fsal_out) return fail(NNHIP_EVALUE, "adaptive methods need fsal_out");
  // scalar elementwise RK4 with uniform (t, dt): the vectorised streaming kernel over N*dim flat states
  if (integrator == NNHIP_RK4 && elementwise_rhs(rhs_kind) && !t_dev && !dt_dev && !fsal_out && !dt_used && !error &&
      (((uintptr_t)y_in | (uintptr_t)y_out) & 15) == 0) {
    const nnhip::StreamTune tune = rk4_stream_tune_for(8 * N * dim * (y_in == y_out ? 1 : 2));
    HIP_TRY(nnhip::launch_rk4_stream(rhs_kind, y_in, y_out, N * dim, t_uniform, dt_uniform, P, negate_time, tune, (hipStream_t)stream));
    return NNHIP_OK;
  }
  // any fixed-step method over a thread-per-IVP system: 16-byte lane accesses, several IVPs per lane (uniform or per-IVP t, dt)
  if (!kMethods[integrator].adaptive && g_fixed_vec_ipl && rhs_kind < NNHIP_RHS_USER_BASE && !dt_used && !error &&
      (((uintptr_t)y_in | (uintptr_t)y_out | (uintptr_t)fsal_out | (uintptr_t)t_dev | (uintptr_t)dt_dev) & 15) == 0 &&
      (layout == NNHIP_LAYOUT_AOS || dim == 1 || (N & 1) == 0)) {
    if (nnhip::FixedVecLaunchFn vf = find_fixed_vec(integrator, rhs_kind, dim)) {
      nnhip::FixedVecArgs va{};
      va.yin = y_in; va.yout = y_out; va.fsalOut = fsal_out; va.tDev = t_dev; va.dtDev = dt_dev; va.N = N;
      va.aos = layout == NNHIP_LAYOUT_AOS && dim > 1 ? 1 : 0;
      va.t = t_uniform; va.dt = dt_uniform; va.P = P;
      // arrays beyond the Infinity Cache: non-temporal hint (knob "adv_nontemporal"; 5.06 -> see profiles/r02_bench_extra.json)
      const int64_t bytes = (int64_t)sizeof(double) * N * ((y_in == y_out ? 1 : 2) * dim + (fsal_out ? dim : 0) + (t_dev ? 1 : 0) + (dt_dev ? 1 : 0));
      const int ntv = knob_or(g_adv_nt, 0, (bytes > (192LL << 20) ? 1 : 0));
      HIP_TRY(vf(va, negate_time, ntv, (hipStream_t)stream));
      return NNHIP_OK;
    }
  }
  bool user = rhs_kind >= NNHIP_RHS_USER_BASE;
  nnhip::StepLaunchFn fn = user ? nullptr : find_step(integrator, rhs_kind, dim);
  if (!fn && !user) {
    const int k = nnhip::rtc_builtin_kind(rhs_kind, dim);
    if (k >= 0) { user = true; rhs_kind = k; }
  }
  if (!fn && !user) return fail(NNHIP_EUNSUPPORTED, "no step kernel for integrator=%s rhs_kind=%d dim=%d", kMethods[integrator].name, rhs_kind, dim);
  nnhip::StepArgs a{};
  a.N = N;
  if (layout == NNHIP_LAYOUT_SOA) { a.ivpStride = 1; a.compStride = N; } else { a.ivpStride = dim; a.compStride = 1; }
  a.t_dev = t_dev; a.t_uniform = t_uniform; a.dt_dev = dt_dev; a.dt_uniform = dt_uniform;
  a.y_in = y_in; a.fsal_in = fsal_in; a.y_out = y_out; a.fsal_out = fsal_out; a.dt_used = dt_used; a.error = error;
  a.ctl = ctl_of(opt); a.P = P;
  // state beyond the Infinity Cache: non-temporal hint (adaptive thread-per-IVP kernels; knob "adv_nontemporal")
  a.nontemporal = knob_or(g_adv_nt, 0, ((dim <= 4 && (int64_t)sizeof(double) * (4 * dim + 5) * N > (192LL << 20)) ? 1 : 0));
  if (user) {
    if (nnhip::rtc_launch_step(rhs_kind, integrator, a, negate_time, (hipStream_t)stream) != hipSuccess)
      return fail(NNHIP_EHIP, "user RHS launch failed: %s", nnhip::rtc_last_error());
    return NNHIP_OK;
  }
  HIP_TRY(fn(a, negate_time, (hipStream_t)stream));
  return NNHIP_OK;
}

}  // extern "C"

namespace nnhip_capi {
// hipGraph cache for the fixed-step streaming loop: for small / mid-size batches the loop is launch-bound
// (a 1e5-IVP RK4 step runs ~2 us, a launch costs ~4 us of host time), so the whole sequence of step launches is
// captured once per (buffers, sizes, times, options, RHS) and replayed with one hipGraphLaunch.
struct StreamGraphKey {
  int integrator, rhs_kind, dim, layout, n_params;
  int64_t N;
  double t0, tEnd, dt, p[nnhip::kMaxParams];
  const void *y, *scratch;
  hipStream_t stream;
  int device;
  bool operator==(const StreamGraphKey& o) const { return std::memcmp(this, &o, sizeof(*this)) == 0; }
};
void sync_device_of(int device);
// An instantiated graph, shared between the cache and whoever is launching it right now: eviction, nnhip_release() or a knob change on another
// thread only drop the cache's reference; the executable is destroyed (after its device has drained) when the last launcher lets go of it.
struct GraphExec {
  hipGraphExec_t exec = nullptr;
  int device = -1;
  GraphExec(hipGraphExec_t e, int d) : exec(e), device(d) {}
  GraphExec(const GraphExec&) = delete;
  GraphExec& operator=(const GraphExec&) = delete;
  ~GraphExec() { if (exec) { sync_device_of(device); (void)hipGraphExecDestroy(exec); } }
};
struct StreamGraphEntry {
  StreamGraphKey key;
  std::shared_ptr<GraphExec> exec;
  int64_t nSteps = 0;
  double* yFinal = nullptr;
};
// Process-wide since round 3 (they were per thread): nnhip_release() and a knob change free every thread's captures, and the worker
// threads of the multi-GPU entries find what an earlier call's workers captured.  The mutex covers lookup / insert / erase only; a launcher
// holds a reference to the executable it found (GraphExec) while it launches it outside the lock.
std::mutex g_graph_mu;
std::vector<StreamGraphEntry> g_graphs;
std::vector<StreamGraphKey> g_graph_seen;  // automatic mode: keys that ran eagerly once (a repeat is worth capturing)
thread_local bool g_capturing = false;
void sync_device_of(int device) {  // a cached graph may still be executing; its stream handle may be gone: synchronise its device
  int prev = 0;
  const bool have = hipGetDevice(&prev) == hipSuccess;
  if (device >= 0 && hipSetDevice(device) == hipSuccess) (void)hipDeviceSynchronize();
  if (have) (void)hipSetDevice(prev);
}
void release_stream_graphs() {
  std::vector<StreamGraphEntry> dropped;  // destroyed outside the lock (the destructor synchronises a device)
  {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    dropped.swap(g_graphs);
    g_graph_seen.clear();
  }
}
}  // namespace nnhip_capi

extern "C" {

int nnhip_ode_fixed_stream_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                   int n_params, int64_t N, int dim, int layout, double t0, double tEnd, double* y,
                                   double* scratch, int64_t* n_steps_out, double** y_final, void* stream) {
  nnhip::Params P;
  int rc = check_common(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, P);
  if (rc) return rc;
  // Graph replay pays when the loop is launch-bound: a dependent launch costs >= 3.2 us, the step kernel less than that below
  // ~2e6 states.  Automatic mode replays such batches from the second identical call on (the first runs eagerly and
  // is remembered), so one-off calls never pay for a capture.
  const bool graphAuto = g_stream_graph == 2 && !kMethods[integrator].adaptive && opt->dt > 0.0 && N * (int64_t)dim <= 2000000 &&
                         (tEnd - t0) / opt->dt >= 16.0 && (tEnd - t0) / opt->dt <= 10000.0;  // <= 1e4 kernel nodes per graph
  if ((g_stream_graph == 1 || graphAuto) && !g_capturing && N > 0 && stream != nullptr) {  // the legacy default stream cannot be captured
    StreamGraphKey key;
    std::memset(&key, 0, sizeof(key));
    key.integrator = integrator; key.rhs_kind = rhs_kind; key.dim = dim; key.layout = layout; key.n_params = n_params; key.N = N;
    key.t0 = t0; key.tEnd = tEnd; key.dt = opt->dt;
    for (int k = 0; k < nnhip::kMaxParams; ++k) key.p[k] = P.p[k];
    key.y = y; key.scratch = scratch; key.stream = (hipStream_t)stream;
    HIP_TRY(hipGetDevice(&key.device));
    bool eager = false;
    {
      StreamGraphEntry hit;
      bool found = false;
      {
        std::lock_guard<std::mutex> lk(g_graph_mu);
        for (auto& e : g_graphs)
          if (e.key == key) { hit = e; found = true; break; }
        if (!found && g_stream_graph == 2) {
          bool seen = false;
          for (auto& k2 : g_graph_seen) seen = seen || k2 == key;
          if (!seen) {
            if (g_graph_seen.size() >= 64) g_graph_seen.erase(g_graph_seen.begin());
            g_graph_seen.push_back(key);
            eager = true;  // first sight of this call: run it eagerly below
          }
        }
      }
      if (found) {
        HIP_TRY(hipGraphLaunch(hit.exec->exec, (hipStream_t)stream));
        if (n_steps_out) *n_steps_out = hit.nSteps;
        if (y_final) *y_final = hit.yFinal;
        return NNHIP_OK;
      }
    }
    hipGraph_t graph = nullptr;
    if (!eager && hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      if (g_stream_graph == 1) return fail(NNHIP_EHIP, "hipStreamBeginCapture failed (stream already capturing?)");
      eager = true;  // automatic mode: e.g. the caller is capturing this stream itself — just enqueue the launches
    }
    if (!eager) {
    StreamGraphEntry e;
    e.key = key;
    g_capturing = true;
    rc = nnhip_ode_fixed_stream_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, t0, tEnd, y, scratch, &e.nSteps,
                                        &e.yFinal, stream);
    g_capturing = false;
    hipError_t ce = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (ce != hipSuccess) return fail(NNHIP_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
    hipGraphExec_t raw = nullptr;
    const hipError_t ie = hipGraphInstantiate(&raw, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ie != hipSuccess) return fail(NNHIP_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie));
    e.exec = std::make_shared<GraphExec>(raw, key.device);
    {
      std::shared_ptr<GraphExec> evicted;  // the oldest; it may still be executing or being launched: released outside the lock
      std::lock_guard<std::mutex> lk(g_graph_mu);
      if (g_graphs.size() >= 32) {
        evicted = std::move(g_graphs.front().exec);
        g_graphs.erase(g_graphs.begin());
      }
      g_graphs.push_back(e);
    }
    HIP_TRY(hipGraphLaunch(e.exec->exec, (hipStream_t)stream));
    if (n_steps_out) *n_steps_out = e.nSteps;
    if (y_final) *y_final = e.yFinal;
    return NNHIP_OK;
    }  // !eager
  }
  if (kMethods[integrator].adaptive) return fail(NNHIP_EVALUE, "nnhip_ode_fixed_stream_f64_dev needs a fixed-step integrator");
  if (!kMethods[integrator].implemented) return fail(NNHIP_EUNSUPPORTED, "integrator %s has no HIP kernel yet", kMethods[integrator].name);
  if (!(opt->dt > 0.0)) return fail(NNHIP_EVALUE, "fixed-step integrators need options.dt > 0 (the reference would loop forever)");
  if (N > 0 && !y) return fail(NNHIP_EVALUE, "y is NULL");
  if (!std::isfinite(t0) || !std::isfinite(tEnd)) return fail(NNHIP_EVALUE, "t0 / tEnd must be finite");
  // ODESolver forward loop, adaptive = false, no dense output (ode.nim:509-532)
  double t = t0;
  double dt = opt->dt;
  int64_t n = 0;
  double* cur = y;
  double* nxt = scratch ? scratch : y;
  while (t < tEnd) {             // :511
    dt = nmin_h(dt, tEnd - t);   // :525
    rc = nnhip_ode_step_batch_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, nullptr, t, nullptr, dt,
                                      cur, nullptr, nxt, nullptr, nullptr, nullptr, 0, stream);  // :531
    if (rc) return rc;
    if (scratch) std::swap(cur, nxt);
    t += dt;                     // :532
    ++n;
  }
  if (n_steps_out) *n_steps_out = n;
  if (y_final) *y_final = cur;
  return NNHIP_OK;
}

// ---- ODESolver through the IntegratorProc seam, fixed-step methods, WITH dense output -----------------------------------------
// The reference's whole driver (ode.nim:471-586: both directions, requested-time rows by Hermite interpolation :512-524, output
// assembly :585) over the step-streaming kernels: the state lives in HBM between IntegratorProc calls, (t, dt) are shared by the
// batch — so the host replays the time loop and, whenever requested times fall into the step just taken, launches f(lastT, lastY),
// f(t, y) and one Hermite kernel per requested time.  The two state buffers of the ping-pong ARE (lastIter.y, y).  Bitwise equal
// to the fused solve, rows, NaN fill and the reference's dropped-rows quirk included (they are uniform over the batch here).
int64_t nnhip_ode_fixed_stream_dense_workspace_bytes(int64_t N, int dim) {
  if (dim < 1 || !batch_size_sane(N, 4 * (int64_t)dim)) return 0;
  return 4 * N * dim * (int64_t)sizeof(double);  // ping, pong, f(lastT, lastY), f(t, y)
}

int nnhip_ode_fixed_stream_dense_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                         const double* y0, int64_t N, int dim, int layout, const double* tspan, int n_t, double* t_out,
                                         double* y_out, int* ny_out, int64_t max_steps, void* ws, int64_t ws_bytes, int64_t* n_steps_out,
                                         void* stream) {
  nnhip::Params P;
  int rc = check_common(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, P);
  if (rc) return rc;
  if (kMethods[integrator].adaptive) return fail(NNHIP_EVALUE, "nnhip_ode_fixed_stream_dense_f64_dev needs a fixed-step integrator");
  if (!(opt->dt > 0.0)) return fail(NNHIP_EVALUE, "fixed-step integrators need options.dt > 0 (the reference would loop forever)");
  if (n_t < 0 || (n_t > 0 && !tspan)) return fail(NNHIP_EVALUE, "bad tspan");
  for (int j = 0; j < n_t; ++j) if (!std::isfinite(tspan[j])) return fail(NNHIP_EVALUE, "tspan[%d] is not finite", j);
  if (!std::isfinite(opt->tStart)) return fail(NNHIP_EVALUE, "options.tStart is not finite");
  if (N > 0 && (!y0 || (!y_out && n_t > 0) || !ws || ws_bytes < nnhip_ode_fixed_stream_dense_workspace_bytes(N, dim))) return fail(NNHIP_EVALUE, "y0 / y_out / workspace missing or too small");
  TimeGrid g;
  make_grid(opt, tspan, n_t, g);
  if (t_out) std::copy(g.tOut.begin(), g.tOut.end(), t_out);
  hipStream_t s = (hipStream_t)stream;
  const int64_t nState = N * dim;
  double* bufA = (double*)ws;
  double* bufB = bufA + nState;
  double* d1 = bufB + nState;
  double* d2 = d1 + nState;
  const bool useDense = n_t != 2;  // ode.nim:499-502
  const bool exactCalls = nnhip::rtc_has_aux(rhs_kind);  // a right-hand side with mutable slots: every evaluation the reference makes, in its order
  int64_t stepsTotal = 0;
  bool truncated = false;
  auto row = [&](int j) { return y_out + (int64_t)j * nState; };
  // one direction of ODESolver's loop; returns the number of rows it produced (<= nReq), rows go to row(rowOf(k))
  auto run_dir = [&](bool neg, double tStartEff, double tEnd, const std::vector<double>& req, auto rowOf, int& produced) -> int {
    const int nReq = (int)req.size(), high = nReq - 1;
    double* cur = bufA;
    double* nxt = bufB;
    const double* lastBuf = nullptr;
    if (nState) HIP_TRY(hipMemcpyAsync(cur, y0, (size_t)nState * 8, hipMemcpyDeviceToDevice, s));  // y = y0.clone() (:482)
    double t = tStartEff, dt = opt->dt, lastT = tStartEff;
    int denseIndex = 0;
    int64_t steps = 0;
    auto evalF = [&](double tEff, const double* yy, double* out) { return nnhip_ode_rhs_batch_f64_dev(rhs_kind, rhs_params, n_params, N, dim, layout, neg ? -tEff : tEff, yy, out, stream); };
    if (exactCalls && nState) {  // a mutating f: lastIter.dy (:498) and FSAL (:506) before the forward loop, g(-t0, y0) (:546) before the backward one
      int r0 = evalF(t, cur, d1);
      if (!r0 && !neg) r0 = evalF(t, cur, d1);
      if (r0) return r0;
    }
    while (t < tEnd) {  // :511
      if (useDense) {
        if (high < denseIndex) break;  // :513-514
        double treq = neg ? -req[denseIndex] : req[denseIndex];
        if (treq <= t) {
          const double* lb = lastBuf ? lastBuf : cur;
          bool fused = false;
          if (!exactCalls && nState) {
            // compiled-in right-hand sides: every row due now in one pass over (lastY, y) — f at both ends and hermiteSpline inside the kernel
            nnhip::DenseRows rows;
            int di = denseIndex;
            double tq = treq;
            bool stop = false;
            while (!stop && tq <= t) {
              rows.n = 0;
              while (rows.n < 8 && tq <= t) {  // :515
                rows.treq[rows.n] = tq;
                rows.out[rows.n] = row(rowOf(di));
                rows.n += 1;
                di += 1;
                if (high < di) { stop = true; break; }  // :523-524
                tq = neg ? -req[di] : req[di];
              }
              hipError_t le = hipSuccess;
              if (!nnhip::launch_dense_rows_kind(rhs_kind, dim, N, layout == NNHIP_LAYOUT_SOA ? 1 : dim, layout == NNHIP_LAYOUT_SOA ? N : 1, lastT, t, neg ? 1 : 0,
                                                 lb, cur, rows, P, s, &le)) break;  // no instantiation (first batch): the separate launches below
              HIP_TRY(le);
              fused = true;
            }
            if (fused) { denseIndex = di; treq = tq; }
          }
          if (!fused) {
          // lastIter.dy = f(lastT, lastY) (:530) and f(t, y) (:521); for the backward branch the kernels negate them (g = -f(-t, y)).
          // Lazily — only when a requested time has been passed — unless f mutates its ctx (then d1 was evaluated with the step, :530)
          int r2 = exactCalls ? NNHIP_OK : evalF(lastT, lb, d1);
          if (!r2 && !exactCalls) r2 = evalF(t, cur, d2);
          if (r2) return r2;
          while (treq <= t) {  // :515
            if (exactCalls) { const int r4 = evalF(t, cur, d2); if (r4) return r4; }  // once per emitted point (:521)
            HIP_TRY(nnhip::launch_hermite(treq, lastT, t, lb, cur, d1, d2, row(rowOf(denseIndex)), nState, neg ? 1 : 0, s));
            denseIndex += 1;
            if (high < denseIndex) break;  // :523-524
            treq = neg ? -req[denseIndex] : req[denseIndex];
          }
          }
        }
      }
      dt = nmin_h(dt, tEnd - t);  // :525
      if (exactCalls && useDense && nState) { const int r5 = evalF(t, cur, d1); if (r5) return r5; }  // lastIter.dy = f(t, y, ctx) (:530)
      const int r3 = nnhip_ode_step_batch_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, nullptr, t, nullptr, dt, cur, nullptr,
                                                  nxt, nullptr, nullptr, nullptr, neg ? 1 : 0, stream);  // :531
      if (r3) return r3;
      lastT = t;       // lastIter = (t, y, ...) (:526-530)
      lastBuf = cur;
      std::swap(cur, nxt);
      t += dt;  // :532
      ++steps;
      if (max_steps > 0 && steps >= max_steps) { truncated = truncated || t < tEnd; break; }
    }
    // yPositive.add(y) / yNegative.add(y) (:542, :584): appended after whatever was emitted
    if (denseIndex < nReq && nState) HIP_TRY(hipMemcpyAsync(row(rowOf(denseIndex)), cur, (size_t)nState * 8, hipMemcpyDeviceToDevice, s));
    produced = denseIndex + 1 < nReq ? denseIndex + 1 : nReq;
    stepsTotal += steps;
    return NNHIP_OK;
  };
  int rowBase = 0;
  int mPosFirst = -1;  // both directions: the forward one runs first, as in the reference (:508-542) — observable through a mutable ctx
  const int fwdBase = (int)g.tNeg.size() + (g.nZero ? 1 : 0);
  if (!g.tNeg.empty() && !g.tPos.empty()) {
    int m = 0;
    rc = run_dir(false, opt->tStart, g.tEndPos, g.tPos, [&](int k) { return fwdBase + k; }, m);
    if (rc) return rc;
    mPosFirst = m;
  }
  if (!g.tNeg.empty()) {  // backward branch (:544-584): element k of yNegative lands in row nNeg-1-k (yNegative.reversed, :585)
    const int nNeg = (int)g.tNeg.size();
    int m = 0;
    rc = run_dir(true, -opt->tStart, g.tEndNeg, g.tNeg, [&](int k) { return nNeg - 1 - k; }, m);
    if (rc) return rc;
    if (m < nNeg && nState) {  // reference quirk: fewer rows than requested -> they close up
      const int shift = nNeg - m;
      for (int j = 0; j < m; ++j) HIP_TRY(hipMemcpyAsync(row(j), row(j + shift), (size_t)nState * 8, hipMemcpyDeviceToDevice, s));
    }
    rowBase = m;
  }
  if (g.nZero) {  // `if t0 in tspan` (:485-487)
    if (nState) HIP_TRY(hipMemcpyAsync(row(rowBase), y0, (size_t)nState * 8, hipMemcpyDeviceToDevice, s));
    rowBase += 1;
  }
  if (mPosFirst >= 0) {  // the forward rows were written for a backward branch that returns all of its rows: close up if it did not
    if (rowBase != fwdBase && nState)
      for (int j = 0; j < mPosFirst; ++j) HIP_TRY(hipMemcpyAsync(row(rowBase + j), row(fwdBase + j), (size_t)nState * 8, hipMemcpyDeviceToDevice, s));
    rowBase += mPosFirst;
  } else if (!g.tPos.empty()) {
    int m = 0;
    const int rb = rowBase;
    rc = run_dir(false, opt->tStart, g.tEndPos, g.tPos, [&](int k) { return rb + k; }, m);
    if (rc) return rc;
    rowBase += m;
  }
  if (rowBase < n_t) HIP_TRY(nnhip::launch_fill_f64(row(rowBase), (int64_t)(n_t - rowBase) * nState, std::nan(""), s));
  if (ny_out) *ny_out = rowBase;
  if (n_steps_out) *n_steps_out = stepsTotal;
  if (truncated) { (void)fail(NNHIP_TRUNCATED, "max_steps = %lld ended the integration before tEnd: the last row is the state reached, not y(tEnd)", (long long)max_steps); return NNHIP_TRUNCATED; }
  return NNHIP_OK;
}

int64_t nnhip_ode_adaptive_stream_workspace_bytes(int64_t N, int dim) {
  if (dim < 1 || !batch_size_sane(N, (int64_t)dim + 4)) return 0;
  return (int64_t)sizeof(double) * (N * dim /*FSAL*/ + 3 * N /*(t, dt) + one spare column*/) + (int64_t)sizeof(unsigned int) * nnhip::kAggSlots;
}

}  // extern "C"

namespace nnhip_capi {
// hipGraph cache of the adaptive streaming loop: one graph = one polling group (flag reset + `check_every` advance launches).
// Every group of a solve is the same graph (the state is advanced in place), so it is replayed until the batch is done, and kept
// for the next identical call.  Why: at C3's own size (1e6 Lorenz IVPs) an eagerly launched advance kernel runs 20 us but costs
// 27 us per loop iteration — the rest is the dispatch gap between dependent launches, which a graph replay removes.
struct AdvGraphKey {
  nnhip::StepArgs a;
  const void* fn;
  const void* active;  // which half of the pinned flag block the group's last launch writes
  int userKind, integrator, checkEvery, device, split;
  hipStream_t stream;
};
struct AdvGraphEntry {
  AdvGraphKey key;
  std::shared_ptr<GraphExec> exec;
};
// The "anyone still integrating?" flags live in page-locked, device-visible HOST memory and the last launch of a polling group stores
// into them directly (a few thousand 4-byte writes over PCIe, once per group).  Round 2 kept them in device memory: every group then
// carried a memset node in front and a device-to-host copy behind (~13 us per group, 1.5 us per loop iteration at C3's size).  Two
// halves for the two groups in flight; the host zeroes a half itself before it issues the group that writes it (the previous
// group on that half has been waited for by then).
struct AdvPoll {
  unsigned int* h = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  // side streams + fork/join events for interleaving index ranges of the batch (see adv_issue_group)
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
  int device = -1;
};
std::vector<AdvGraphEntry> g_adv_graphs;  // process-wide, under g_graph_mu (see g_graphs)
thread_local AdvPoll g_adv_poll;          // the calling thread's flag block, events and side streams (its graphs bake the flag addresses in)

void release_adv_graphs();
void free_adv_poll(AdvPoll& p);
int adv_poll_reserve() {
  int device = 0;
  HIP_TRY(hipGetDevice(&device));
  if (g_adv_poll.device != device) free_adv_poll(g_adv_poll);  // streams and events belong to one device
  AdvPoll& p = g_adv_poll;
  p.device = device;
  if (!p.h) HIP_TRY(hipHostMalloc((void**)&p.h, 2 * nnhip::kAggSlots * sizeof(unsigned int), hipHostMallocMapped | hipHostMallocCoherent));
  for (hipEvent_t& e : p.ev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (hipStream_t& st : p.side) if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  if (!p.fork) HIP_TRY(hipEventCreateWithFlags(&p.fork, hipEventDisableTiming));
  for (hipEvent_t& e : p.join) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return NNHIP_OK;
}
void release_adv_graphs() {
  {
    std::vector<AdvGraphEntry> dropped;
    {
      std::lock_guard<std::mutex> lk(g_graph_mu);
      dropped.swap(g_adv_graphs);
    }
  }
  free_adv_poll(g_adv_poll);
}
void free_adv_poll(AdvPoll& p) {
  if (p.device >= 0) {  // graphs that bake this block's flag addresses in go with it (they are this thread's own: nobody else launches them)
    std::vector<std::shared_ptr<GraphExec>> dropped;
    {
      std::lock_guard<std::mutex> lk(g_graph_mu);
      for (size_t k = 0; k < g_adv_graphs.size();) {
        const unsigned int* f = (const unsigned int*)g_adv_graphs[k].key.active;
        if (p.h && f >= p.h && f < p.h + 2 * nnhip::kAggSlots) {
          dropped.push_back(std::move(g_adv_graphs[k].exec));
          g_adv_graphs.erase(g_adv_graphs.begin() + (long)k);
        } else ++k;
      }
    }
  }
  if (p.h) (void)hipHostFree(p.h);
  for (hipEvent_t e : p.ev) if (e) (void)hipEventDestroy(e);
  for (hipStream_t st : p.side) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  if (p.fork) (void)hipEventDestroy(p.fork);
  for (hipEvent_t e : p.join) if (e) (void)hipEventDestroy(e);
  p = AdvPoll();
}

// The arguments of the sub-batch [lo, lo + n) of a prepared advance launch (strides keep addressing the full arrays).
nnhip::StepArgs adv_range(const nnhip::StepArgs& full, int64_t lo, int64_t n) {
  nnhip::StepArgs a = full;
  a.N = n;
  a.y_in += lo * a.ivpStride; a.y_out += lo * a.ivpStride; a.fsal_in += lo * a.ivpStride; a.fsal_out += lo * a.ivpStride;
  if (a.dt_io) { a.t_io += lo; a.dt_io += lo; }
  else a.t_io += 2 * lo;  // packed layout: (t, dt) of IVP i side by side, [N][2]
  if (a.error) a.error += lo;
  if (a.steps_io) a.steps_io += lo;
  if (a.perIvpParams) a.perIvpParams += lo;
  if (a.P.ivp) a.P.ivp += lo;
  if (a.P.aux) a.P.aux += lo;
  return a;
}

// One polling group: reset the flags, then `checkEvery` loop iterations, the last one reporting whether work is left.
// With split > 1 the batch is cut into `split` index ranges whose launch chains run on separate streams (fork / join by
// events; under stream capture they become parallel branches of the graph): while one range's kernel drains its last
// waves, the other range's next kernel is already filling the freed CUs, which hides the ramp-down / ramp-up gap between
// DEPENDENT launches (6 of 27 us per iteration at C3's own size).  The ranges are independent IVPs: same bits.
int adv_issue_group(nnhip::StepLaunchFn fn, int userKind, int integrator, const nnhip::StepArgs& full, unsigned int* active, int checkEvery,
                    int split, hipStream_t s) {
  AdvPoll& p = g_adv_poll;
  if (split > 1) {
    HIP_TRY(hipEventRecord(p.fork, s));
    for (int r = 1; r < split; ++r) HIP_TRY(hipStreamWaitEvent(p.side[r - 1], p.fork, 0));
  }
  for (int k = 0; k < checkEvery; ++k) {
    for (int r = 0; r < split; ++r) {
      const int64_t lo = full.N * r / split, hi = full.N * (r + 1) / split;
      nnhip::StepArgs a = split > 1 ? adv_range(full, lo, hi - lo) : full;
      a.active = k == checkEvery - 1 ? active : nullptr;
      hipStream_t st = r == 0 ? s : p.side[r - 1];
      if (fn) HIP_TRY(fn(a, knob_or(g_adv_block, 1, 64), st));  // one-wave workgroups retire and refill sooner: 1e7 Lorenz IVPs 208 -> 203 us, 1e6 24.0 -> 23.0 us (mb_adv c3a)
      else if (nnhip::rtc_launch_advance(userKind, integrator, a, st) != hipSuccess) return fail(NNHIP_EHIP, "user RHS launch failed: %s", nnhip::rtc_last_error());
    }
  }
  for (int r = 1; r < split; ++r) {
    HIP_TRY(hipEventRecord(p.join[r - 1], p.side[r - 1]));
    HIP_TRY(hipStreamWaitEvent(s, p.join[r - 1], 0));
  }
  return NNHIP_OK;
}

// The polling group `issue()` enqueues on `s`, as a cached hipGraph (nullptr: not available — the caller issues eagerly).  rc receives
// the status of `issue` when it had to be run for the capture.
template <class IssueFn>
std::shared_ptr<GraphExec> adv_cached_graph(const AdvGraphKey& key, hipStream_t s, IssueFn&& issue, int& rc) {
  rc = NNHIP_OK;
  {
    std::lock_guard<std::mutex> lk(g_graph_mu);
    for (auto& e : g_adv_graphs)
      if (std::memcmp(&e.key, &key, sizeof(key)) == 0) return e.exec;
  }
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return nullptr; }  // e.g. the caller is capturing this stream itself
  rc = issue();
  const hipError_t ce = hipStreamEndCapture(s, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return nullptr; }
  if (ce != hipSuccess || !graph) { (void)hipGetLastError(); return nullptr; }
  const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ie != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  AdvGraphEntry e;
  e.key = key; e.exec = std::make_shared<GraphExec>(exec, key.device);
  std::shared_ptr<GraphExec> evicted;  // released after the lock
  std::lock_guard<std::mutex> lk(g_graph_mu);
  if (g_adv_graphs.size() >= 32) {  // evict the oldest; it may still be executing
    evicted = std::move(g_adv_graphs.front().exec);
    g_adv_graphs.erase(g_adv_graphs.begin());
  }
  g_adv_graphs.push_back(e);
  return e.exec;
}
}  // namespace nnhip_capi

extern "C" {

// ODESolver's adaptive forward loop (ode.nim:506-542, tspan.len == 2) over the `advance` kernels: per-IVP (t, dt, FSAL)
// live in `ws`; every launch performs one loop iteration of every unfinished IVP (thread-per-IVP for small systems,
// lanes-per-system for Vector[float] states of 8 / 16 / 32 ... components).  The host learns whether anyone is still
// integrating once per group of `check_every` launches, and always has the NEXT group enqueued before it waits for the
// answer of the current one (a launch over finished IVPs only reads their t: 8 B per IVP), so the device never idles on
// the host.  Groups are replayed from a hipGraph unless tuning knob "stream_graph" is 0 or `stream` is the legacy default stream.
int nnhip_ode_adaptive_stream_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                      int n_params, int64_t N, int dim, int layout, double t0, double tEnd, double* y, void* ws,
                                      int64_t ws_bytes, int check_every, int64_t max_launches, int64_t* launches_out, void* stream) {
  nnhip::Params P;
  int rc = check_common(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, P);
  if (rc) return rc;
  if (!kMethods[integrator].adaptive) return fail(NNHIP_EVALUE, "nnhip_ode_adaptive_stream_f64_dev needs an adaptive integrator");
  if (!std::isfinite(t0) || !std::isfinite(tEnd)) return fail(NNHIP_EVALUE, "t0 / tEnd must be finite");
  if (!(opt->dtMin > 0.0) && max_launches <= 0) return fail(NNHIP_EVALUE, "adaptive integrators need options.dtMin > 0 or max_launches > 0");
  if (launches_out) *launches_out = 0;
  if (N == 0 || !(t0 < tEnd)) return NNHIP_OK;
  if (!y || !ws || ws_bytes < nnhip_ode_adaptive_stream_workspace_bytes(N, dim)) return fail(NNHIP_EVALUE, "y / workspace missing or too small");
  int userKind = rhs_kind >= NNHIP_RHS_USER_BASE ? rhs_kind : -1;
  nnhip::StepLaunchFn fn = userKind >= 0 ? nullptr : find_advance(integrator, rhs_kind, dim);
  if (!fn && userKind < 0) userKind = nnhip::rtc_builtin_kind(rhs_kind, dim);  // built-in kind at a size without an ahead-of-time kernel
  if (!fn && userKind < 0)
    return fail(NNHIP_EUNSUPPORTED, "no advance kernel for integrator=%s rhs_kind=%d dim=%d", kMethods[integrator].name, rhs_kind, dim);
  hipStream_t s = (hipStream_t)stream;
  // workspace: (t, dt) of IVP i side by side when `ws` is 16-byte aligned (one 16-byte access each way per launch instead of two of 8;
  // every allocator's blocks are), two columns otherwise; then FSAL.  `error` (ode.nim:531) is a local of the loop: it is not stored.
  const bool packed = ((uintptr_t)ws & 15u) == 0;
  double* tArr = (double*)ws;
  double* dtArr = packed ? nullptr : tArr + N;
  double* fsal = tArr + 2 * N;
  // DOPRI54 / Tsit54: FSAL re-evaluated by each launch instead of carried through HBM (knob "adv_recompute_fsal"; see adv_fsal_in_hbm in ode_kernels.hpp)
  const bool fsalRecomputable = integrator == NNHIP_DOPRI54 || integrator == NNHIP_TSIT54;
  const int recomputeFsal = !fsalRecomputable ? 0 : knob_or(g_adv_refsal, 0, nnhip::rtc_has_aux(rhs_kind) ? 0 : 1);
  const bool fsalInHbm = !(recomputeFsal || integrator == NNHIP_BS32 || integrator == NNHIP_RK21);  // BS32 / RK21 never read the slot
  // FSAL = f(t0, y) (:506); t = t0; dt = sqrt(dtMax*dtMin) (:491-493)
  if (nnhip::rtc_has_aux(rhs_kind)) {  // lastIter.dy = f(t0, y, ctx) (:498): the first of the reference's two evaluations at t0, observable through aux
    rc = nnhip_ode_rhs_batch_f64_dev(rhs_kind, rhs_params, n_params, N, dim, layout, t0, y, fsal, stream);
    if (rc) return fail(rc, "initial RHS evaluation failed");
  }
  if (fsalInHbm || nnhip::rtc_has_aux(rhs_kind)) {
    rc = nnhip_ode_rhs_batch_f64_dev(rhs_kind, rhs_params, n_params, N, dim, layout, t0, y, fsal, stream);
    if (rc) return fail(rc, "initial RHS evaluation failed");
  }
  if (packed)
    HIP_TRY(nnhip::launch_kernel(nnhip::fill_td_kernel<0>, dim3((unsigned)((N + nnhip::kBlock - 1) / nnhip::kBlock)), dim3(nnhip::kBlock), s, (double2*)tArr, N, t0,
                                 std::sqrt(opt->dtMax * opt->dtMin)));
  else
    HIP_TRY(nnhip::launch_kernel(nnhip::fill_t_dt_kernel<0>, dim3((unsigned)((N + nnhip::kBlock - 1) / nnhip::kBlock)), dim3(nnhip::kBlock), s, tArr, dtArr, N, t0,
                                 std::sqrt(opt->dtMax * opt->dtMin)));
  nnhip::StepArgs a{};
  a.N = N;
  if (layout == NNHIP_LAYOUT_SOA) { a.ivpStride = 1; a.compStride = N; } else { a.ivpStride = dim; a.compStride = 1; }
  a.y_in = y; a.y_out = y; a.fsal_in = fsal; a.fsal_out = fsal; a.error = nullptr;
  a.ctl = ctl_of(opt); a.P = P;
  a.tEnd = tEnd; a.t_io = tArr; a.dt_io = dtArr; a.active = nullptr; a.steps_io = nullptr;
  a.stepsPerLaunch = g_adv_steps;
  a.recomputeFsal = recomputeFsal;
  a.noLean = g_adv_lean ? 0 : 1;
  // thread-per-IVP kernels (the lanes-per-system ones are not memory-bound enough to gain: measured -3 %); the state of one launch = y, (t, dt) and FSAL if carried
  a.nontemporal = knob_or(g_adv_nt, 0, ((dim <= 4 && (int64_t)sizeof(double) * ((fsalInHbm ? 2 : 1) * dim + 2) * N > (192LL << 20)) ? 1 : 0));
  const bool autoPoll = check_every <= 0 && g_adv_auto_poll;  // the polling schedule is the library's (below, knob "adv_auto_poll"); a caller's check_every is taken as given
  if (check_every <= 0) check_every = 8;
  rc = adv_poll_reserve();
  if (rc) return rc;
  AdvPoll& poll = g_adv_poll;
  // Interleaved index ranges pay while a launch is short enough for its ramp-down to matter and long enough to fill the chip twice
  int split = g_adv_split;
  if (split == 0) split = 1;  // measured and rejected as a default: 1e6 Lorenz IVPs 29 us per iteration unsplit, 35 us in 2 ranges, 44 us in 4
                              // (profiles/r02_pow_tables_ab.txt) — the launches were never gap-bound: the kernel itself takes 27 us
  if ((int64_t)split > N) split = 1;
  // knob "fp_contract": where the launch has the lean kernels' layout, the FMA-contracted lean kernel advances it (within north_star's 1e-6, not the reference's bits)
  if (fn && split == 1 && nnhip::adv_lean_layout_ok(a, dim, layout == NNHIP_LAYOUT_AOS)) {
    if (nnhip::StepLaunchFn contracted = find_advance_lean_contracted(integrator, rhs_kind, dim)) fn = contracted;
  }

  // ---- the polling group as a graph (cached per thread; key = everything the launches depend on), one per half of the flag block ----
  std::shared_ptr<GraphExec> execs[2];  // held for the whole call: another thread's eviction / nnhip_release() cannot destroy them under it
  // Replayed only when asked for (knob 1) since round 3: with (t, dt) interleaved and FSAL re-evaluated a launch takes 5-300 us, the command
  // processor pipelines eager launches behind each other, and a graph's node-to-node hand-over costs 1-2 us more than that (Lorenz 1e4 ... 3e6
  // IVPs and 16-component rings: eager 1-5 % faster at every size, scripts/ab_stream_graph_vs_eager.py)
  if (g_stream_graph == 1 && s != nullptr) {
    int device = 0;
    HIP_TRY(hipGetDevice(&device));
    for (int half = 0; half < 2; ++half) {
      unsigned int* flags = poll.h + half * nnhip::kAggSlots;
      AdvGraphKey key;
      std::memset(&key, 0, sizeof(key));
      std::memcpy(&key.a, &a, sizeof(a));
      key.fn = (const void*)fn; key.active = flags; key.userKind = userKind; key.integrator = integrator; key.checkEvery = check_every; key.device = device; key.split = split; key.stream = s;
      if (fn == nullptr) {  // run-time compiled kernels: make sure the module is loaded before the stream goes into capture mode
        nnhip::StepArgs warm = a;
        warm.N = 0;
        (void)nnhip::rtc_launch_advance(userKind, integrator, warm, s);
      }
      execs[half] = adv_cached_graph(key, s, [&]() { return adv_issue_group(fn, userKind, integrator, a, flags, check_every, split, s); }, rc);
      if (rc) return rc;
    }
    if (!execs[0] || !execs[1]) execs[0] = execs[1] = nullptr;
  }
  // How many launches before the host asks "is anyone still integrating?": adv_poll_schedule.hpp (a caller's check_every and graph replay: uniform groups;
  // check_every <= 0 with eager launches: the first ceil((tEnd - t0) / dtMax) launches unpolled, then 2, 2, 4, 8 ...)
  nnhip::AdvPollSchedule sched = nnhip::AdvPollSchedule::make(!autoPoll || (execs[0] != nullptr), check_every, t0, tEnd, opt->dtMax, a.stepsPerLaunch, max_launches);
  auto issue = [&](int n, int half) -> int {
    unsigned int* flags = poll.h + half * nnhip::kAggSlots;
    std::memset(flags, 0, nnhip::kAggSlots * sizeof(unsigned int));  // host memory; the group that last wrote this half has been waited for
    if (execs[half]) HIP_TRY(hipGraphLaunch(execs[half]->exec, s));
    else { const int r = adv_issue_group(fn, userKind, integrator, a, flags, n, split, s); if (r) return r; }
    HIP_TRY(hipEventRecord(poll.ev[half], s));
    return NNHIP_OK;
  };
  auto wait = [&](int half) -> int {
    HIP_TRY(hipEventSynchronize(poll.ev[half]));
    unsigned int any = 0;
    for (int k = 0; k < nnhip::kAggSlots; ++k) any |= poll.h[half * nnhip::kAggSlots + k];
    return any ? 1 : 0;
  };
  int64_t launches = 0;
  rc = nnhip::adv_poll_loop(sched, issue, wait, &launches);
  if (rc) return rc;
  if (launches_out) *launches_out = launches;
  return NNHIP_OK;
}

// ---- ODESolver through the IntegratorProc seam, adaptive methods, WITH dense output --------------------------------------------
int64_t nnhip_ode_adaptive_stream_dense_workspace_bytes(int64_t N, int dim, int n_t) {
  if (dim < 1 || !batch_size_sane(N, 2 * (int64_t)dim + 4)) return 0;
  const int64_t nt = n_t < 0 ? 0 : n_t;
  // y, FSAL [dim*N]; (t, dt) [N][2]; denseIndex [N] and the forward direction's row count [N] (int32); requested times (lastIter = (t, y, dy)
  // lives in the kernel's registers)
  return (int64_t)sizeof(double) * (2 * N * dim + 2 * N + nt + 8) + (int64_t)sizeof(int32_t) * (2 * N + 4) + 64;
}

// The whole ODESolver driver (ode.nim:471-586) for adaptive integrators over the HBM-resident `advance` kernel: both directions,
// per-IVP (t, dt, FSAL), per-IVP Hermite history and denseIndex, requested rows emitted by the kernel as each IVP's steps pass them
// (:512-524).  y0 / y_out / ny_out are device pointers, tspan / t_out host.  ny_out[i] (required, int32 [N]) = rows the
// reference returns for IVP i; rows beyond are NaN.  Every right-hand side kind: thread-per-IVP and lanes-per-system, compiled-in and
// run-time compiled.  Bitwise equal to nnhip_ode_solve_batch_f64_dev.
int nnhip_ode_adaptive_stream_dense_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                            const double* y0, int64_t N, int dim, int layout, const double* tspan, int n_t, double* t_out,
                                            double* y_out, int32_t* ny_out, void* ws, int64_t ws_bytes, int check_every, int64_t max_launches,
                                            int64_t* launches_out, void* stream) {
  nnhip::Params P;
  int rc = check_common(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, P);
  if (rc) return rc;
  if (!kMethods[integrator].adaptive) return fail(NNHIP_EVALUE, "nnhip_ode_adaptive_stream_dense_f64_dev needs an adaptive integrator");
  if (n_t < 0 || (n_t > 0 && !tspan)) return fail(NNHIP_EVALUE, "bad tspan");
  for (int j = 0; j < n_t; ++j) if (!std::isfinite(tspan[j])) return fail(NNHIP_EVALUE, "tspan[%d] is not finite", j);
  if (!std::isfinite(opt->tStart)) return fail(NNHIP_EVALUE, "options.tStart is not finite");
  if (!(opt->dtMin > 0.0) && max_launches <= 0) return fail(NNHIP_EVALUE, "adaptive integrators need options.dtMin > 0 or max_launches > 0");
  nnhip::DenseAdvLaunch fn = rhs_kind < NNHIP_RHS_USER_BASE ? find_advance_dense(integrator, rhs_kind, dim) : nnhip::DenseAdvLaunch{nullptr, nullptr};
  int userKind = rhs_kind >= NNHIP_RHS_USER_BASE ? rhs_kind : -1;
  if (!fn.advance && userKind < 0) userKind = nnhip::rtc_builtin_kind(rhs_kind, dim);  // a built-in right-hand side at a size without an ahead-of-time kernel
  if (!fn.advance && userKind < 0) return fail(NNHIP_EUNSUPPORTED, "no dense advance kernel for integrator=%s rhs_kind=%d dim=%d", kMethods[integrator].name, rhs_kind, dim);
  TimeGrid g;
  make_grid(opt, tspan, n_t, g);
  if (t_out) std::copy(g.tOut.begin(), g.tOut.end(), t_out);
  if (launches_out) *launches_out = 0;
  if (N == 0) return NNHIP_OK;
  if (!y0 || (!y_out && n_t > 0) || !ny_out || !ws || ws_bytes < nnhip_ode_adaptive_stream_dense_workspace_bytes(N, dim, n_t))
    return fail(NNHIP_EVALUE, "y0 / y_out / ny_out / workspace missing or too small");
  if (((uintptr_t)ws & 15u) != 0) return fail(NNHIP_EVALUE, "the workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const int64_t nState = N * dim;
  double* yW = (double*)ws;
  double* fsal = yW + nState;
  double* tdArr = fsal + nState;                     // (t, dt) of IVP i side by side: [N][2] (16-byte aligned: ws is, nState * 16 is)
  double* tReqDev = tdArr + 2 * N;                   // n_t doubles (+ padding)
  int32_t* denseIdx = (int32_t*)(tReqDev + n_t + 8);
  int32_t* fwdRows = denseIdx + N + 2;  // rows the forward direction produced (both directions asked for: it runs first)
  // requested times of both directions, as the reference holds them
  const int nPos = (int)g.tPos.size(), nNeg = (int)g.tNeg.size();
  if (nPos + nNeg > 0) {
    rc = stage_reserve((size_t)(nPos + nNeg));
    if (rc) return rc;
    std::copy(g.tPos.begin(), g.tPos.end(), g_stage.host);
    std::copy(g.tNeg.begin(), g.tNeg.end(), g_stage.host + nPos);
    HIP_TRY(hipMemcpyAsync(tReqDev, g_stage.host, (size_t)(nPos + nNeg) * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(hipEventRecord(g_stage.ev, s));
    g_stage.pending = true;
  }
  rc = adv_poll_reserve();
  if (rc) return rc;
  AdvPoll& poll = g_adv_poll;
  nnhip::StepArgs a{};
  a.N = N;
  if (layout == NNHIP_LAYOUT_SOA) { a.ivpStride = 1; a.compStride = N; } else { a.ivpStride = dim; a.compStride = 1; }
  a.y_in = yW; a.y_out = yW; a.fsal_in = fsal; a.fsal_out = fsal; a.error = nullptr;
  a.ctl = ctl_of(opt); a.P = P;
  a.t_io = tdArr; a.dt_io = nullptr;
  a.denseIdx_io = denseIdx; a.emitAfter = 1;
  {
    const bool fsalRecomputable = integrator == NNHIP_DOPRI54 || integrator == NNHIP_TSIT54;
    a.recomputeFsal = !fsalRecomputable ? 0 : knob_or(g_adv_refsal, 0, nnhip::rtc_has_aux(rhs_kind) ? 0 : 1);
  }
  const bool fsalInHbm = !(a.recomputeFsal || integrator == NNHIP_RK21);
  a.rows = y_out; a.rowStride = nState;
  // state of one launch beyond the Infinity Cache: non-temporal instantiation (thread-per-IVP kernels; knob "adv_nontemporal")
  a.nontemporal = knob_or(g_adv_nt, 0, ((dim <= 4 && (int64_t)sizeof(double) * ((fsalInHbm ? 2 : 1) * dim + 2) * N > (192LL << 20)) ? 1 : 0));
  const bool autoPoll = check_every <= 0 && g_adv_auto_poll;  // the library's own polling schedule (adv_poll_schedule.hpp, knob "adv_auto_poll"), per direction
  if (check_every <= 0) check_every = 8;
  const double dtInit = std::sqrt(opt->dtMax * opt->dtMin);  // :491-493
  const dim3 grid((unsigned)((N + nnhip::kBlock - 1) / nnhip::kBlock)), block(nnhip::kBlock);
  int64_t launches = 0;
  bool truncated = false;
  auto finalize = [&](int mode) -> int {
    HIP_TRY(nnhip::launch_kernel(nnhip::advance_dense_finalize_kernel<0>, grid, block, s, a, mode, dim, y0, ny_out, n_t, fwdRows));
    return NNHIP_OK;
  };
  auto advance = [&](const nnhip::StepArgs& run) -> int {
    if (fn.advance) { HIP_TRY(fn.advance(run, s)); return NNHIP_OK; }
    if (nnhip::rtc_launch_advance_dense(userKind, integrator, run, s) != hipSuccess) return fail(NNHIP_EHIP, "run-time compiled dense advance kernel: %s", nnhip::rtc_last_error());
    return NNHIP_OK;
  };
  // y = y0, FSAL = f(t0, y0) / g(-t0, y0) = -f(t0, y0) (:506,:546), t, dt, denseIndex = 0
  auto init = [&](const nnhip::StepArgs& run, bool neg, double tStartEff) -> int {
    if (fn.init) { HIP_TRY(fn.init(run, y0, tStartEff, dtInit, s)); return NNHIP_OK; }
    const size_t bytes = (size_t)nState * sizeof(double);
    HIP_TRY(hipMemcpyAsync(yW, y0, bytes, hipMemcpyDeviceToDevice, s));
    if (!neg && nnhip::rtc_has_aux(rhs_kind)) {  // lastIter.dy = f(t0, y, ctx) (:498): the first of the reference's two evaluations at t0, observable through aux
      const int r0 = nnhip_ode_rhs_batch_f64_dev(rhs_kind, rhs_params, n_params, N, dim, layout, tStartEff, y0, fsal, stream);
      if (r0) return r0;
    }
    const int r = nnhip_ode_rhs_batch_f64_dev(rhs_kind, rhs_params, n_params, N, dim, layout, neg ? -tStartEff : tStartEff, y0, fsal, stream);
    if (r) return r;
    if (neg) HIP_TRY(nnhip::negate_f64(fsal, fsal, nState, s));
    HIP_TRY(nnhip::launch_kernel(nnhip::fill_td_kernel<0>, grid, block, s, (double2*)tdArr, N, tStartEff, dtInit));
    HIP_TRY(hipMemsetAsync(denseIdx, 0, (size_t)N * sizeof(int32_t), s));
    return NNHIP_OK;
  };
  auto run_dir = [&](bool neg, double tStartEff, double tEnd, const double* req, int nReq, const int32_t* rowBase, int rowBase0) -> int {
    a.negate = neg ? 1 : 0; a.tEnd = tEnd; a.tReq = req; a.nReq = nReq; a.rowBase = rowBase; a.rowBase0 = rowBase0;
    a.useDense = n_t != 2 ? 1 : 0;  // :499-502: with a 2-point tspan the only row of a direction is the final yPositive.add(y)
    nnhip::StepArgs run = a;
    int r = init(run, neg, tStartEff);
    if (r) return r;
    // groups of check_every launches; the host reads group g's "anyone still integrating?" flags while group g + 1 runs (a retired
    // IVP returns at once, so the one group issued past the end costs only its launches)
    // the last launch of a group stores the flags straight into page-locked host memory (no memset / copy nodes); the groups replay
    // as hipGraphs, one per half of the flag block, as the loop without dense output does
    auto issue_group = [&](unsigned int* flags, int n, bool lastPermitted) -> int {
      for (int k = 0; k < n; ++k) {
        run.active = k == n - 1 ? flags : nullptr;
        run.emitAfter = (lastPermitted && k == n - 1) ? 0 : 1;  // the cut of max_launches falls where the fused solve's max_steps does
        const int ra = advance(run);
        if (ra) return ra;
      }
      run.emitAfter = 1;
      return NNHIP_OK;
    };
    int64_t dirLaunches = 0;  // max_launches bounds each direction's loop, as max_steps does in the fused solve
    std::shared_ptr<GraphExec> execs[2];
    if (g_stream_graph == 1 && s != nullptr) {  // as the loop without dense output: eager launches unless asked for
      int device = 0;
      HIP_TRY(hipGetDevice(&device));
      if (!fn.advance) {  // run-time compiled kernels: load the module before the stream goes into capture mode
        nnhip::StepArgs warm = run;
        warm.N = 0; warm.active = nullptr;
        (void)nnhip::rtc_launch_advance_dense(userKind, integrator, warm, s);
      }
      for (int half = 0; half < 2; ++half) {
        unsigned int* flags = poll.h + half * nnhip::kAggSlots;
        AdvGraphKey key;
        std::memset(&key, 0, sizeof(key));
        run.active = nullptr;
        std::memcpy(&key.a, &run, sizeof(run));
        key.fn = (const void*)fn.advance; key.active = flags; key.userKind = userKind; key.integrator = integrator; key.checkEvery = check_every;
        key.device = device; key.split = -1 /* dense */; key.stream = s;
        int rcg = NNHIP_OK;
        execs[half] = adv_cached_graph(key, s, [&]() { return issue_group(flags, check_every, false); }, rcg);
        if (rcg) return rcg;
      }
      if (!execs[0] || !execs[1]) execs[0] = execs[1] = nullptr;
    }
    // group sizes: a caller's check_every (and graph replay) uniform; check_every <= 0: nobody finishes within the first ceil((tEnd - tStartEff) / dtMax) launches
    // of this direction, which go out unpolled, then 2, 2, 4, 8 ... (the cut of max_launches stays this loop's own: it must fall exactly where the fused solve's does)
    nnhip::AdvPollSchedule sched = nnhip::AdvPollSchedule::make(!autoPoll || (execs[0] != nullptr), check_every, tStartEff, tEnd, opt->dtMax, 1, 0);
    auto issue = [&](int64_t grp) -> int {
      const int half = (int)(grp & 1);
      unsigned int* flags = poll.h + half * nnhip::kAggSlots;
      std::memset(flags, 0, nnhip::kAggSlots * sizeof(unsigned int));  // host memory; the group that last wrote this half has been waited for
      const int want = sched.next();
      const bool lastPermitted = max_launches > 0 && dirLaunches + want >= max_launches;
      const int n = lastPermitted ? (int)(max_launches - dirLaunches) : want;
      if (execs[half] && !lastPermitted) HIP_TRY(hipGraphLaunch(execs[half]->exec, s));
      else { const int ra = issue_group(flags, n, lastPermitted); if (ra) return ra; }
      sched.issued_group(n);
      launches += n;
      dirLaunches += n;
      HIP_TRY(hipEventRecord(poll.ev[half], s));
      return NNHIP_OK;
    };
    int64_t grp = 0;
    r = issue(0);
    if (r) return r;
    for (;;) {
      const bool more = !(max_launches > 0 && dirLaunches >= max_launches);
      if (more) {
        r = issue(grp + 1);
        if (r) return r;
      }
      HIP_TRY(hipEventSynchronize(poll.ev[grp & 1]));
      unsigned int any = 0;
      for (int k = 0; k < nnhip::kAggSlots; ++k) any |= poll.h[(grp & 1) * nnhip::kAggSlots + k];
      if (!any || !more) {
        if (more) HIP_TRY(hipEventSynchronize(poll.ev[(grp + 1) & 1]));
        else if (any) truncated = true;
        break;
      }
      ++grp;
    }
    return NNHIP_OK;
  };
  HIP_TRY(hipMemsetAsync(ny_out, 0, (size_t)N * sizeof(int32_t), s));
  const bool bothDirections = nNeg > 0 && nPos > 0;
  const int fwdBase = nNeg + (g.nZero ? 1 : 0);
  if (bothDirections) {  // forward branch FIRST, as the reference runs them (:508-542): its rows go where they belong if the backward branch returns all of its rows
    rc = run_dir(false, opt->tStart, g.tEndPos, tReqDev, nPos, nullptr, fwdBase);
    if (rc) return rc;
    rc = finalize(5);
    if (rc) return rc;
  }
  if (nNeg > 0) {  // backward branch (:544-584)
    rc = run_dir(true, -opt->tStart, g.tEndNeg, tReqDev + nPos, nNeg, nullptr, 0);
    if (rc) return rc;
    rc = finalize(0);
    if (rc) return rc;
  }
  if (g.nZero) {  // `if t0 in tspan` (:485-487)
    rc = finalize(1);
    if (rc) return rc;
  }
  if (bothDirections) {  // the forward rows follow whatever the backward branch and `t0 in tspan` produced (it may return fewer rows than asked: close up)
    a.rowBase0 = fwdBase;
    rc = finalize(4);
    if (rc) return rc;
  } else if (nPos > 0) {
    rc = run_dir(false, opt->tStart, g.tEndPos, tReqDev, nPos, nullptr, g.nZero ? 1 : 0);
    if (rc) return rc;
    rc = finalize(2);
    if (rc) return rc;
  }
  rc = finalize(3);
  if (rc) return rc;
  if (launches_out) *launches_out = launches;
  return truncated ? NNHIP_TRUNCATED : NNHIP_OK;
}

}  // extern "C"


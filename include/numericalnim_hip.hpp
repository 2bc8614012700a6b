// numericalnim_hip.hpp — C++17 host-side mirror of the reference's ODE interface over the C ABI (nnhip_ode.h).
//
// The reference's host language is Nim; no Nim toolchain exists in the build image, so the host side above the
// C ABI is provided (a) as the Nim shim a maintainer would add (nim/numericalnim_hip.nim) and
// (b) as this header for compiled hosts.  Names, argument order, defaults and error behaviour follow
// /root/reference/src/numericalnim/ode.nim:
//   ODEoptions / newODEoptions / DEFAULT_ODEoptions   ode.nim:26-34, 78-104     (ValueError -> std::invalid_argument)
//   NumContext[T, float]                              common/commonTypes.nim:4-39
//   solveODE(f, y0, tspan, options, ctx, integrator)  ode.nim:589-651
// What changes is only what must: y0 is a *batch* of initial states and f is an RhsSpec naming a compiled-in
// device RHS whose parameters are pulled from ctx.fValues (the reference's parameter channel).
#pragma once
#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "nnhip_ode.h"

namespace numericalnim {

using ODEoptions = nnhip_ode_options;

inline void throwOn(int rc) {
  if (rc == NNHIP_OK) return;
  const std::string msg = nnhip_last_error();
  if (rc == NNHIP_EVALUE || rc == NNHIP_EINTEGRATOR) throw std::invalid_argument(msg);  // Nim: ValueError
  if (rc == NNHIP_ENOMEM) throw std::bad_alloc();
  throw std::runtime_error("nnhip error " + std::to_string(rc) + ": " + msg);
}

// ode.nim:78-102 — same parameter order and defaults as the Nim proc
inline ODEoptions newODEoptions(double dt = 1e-4, double absTol = 1e-4, double relTol = 1e-4, double dtMax = 1e-2,
                                double dtMin = 1e-4, double scaleMax = 4.0, double scaleMin = 0.1, double tStart = 0.0) {
  ODEoptions o;
  throwOn(nnhip_ode_new_options(&o, dt, absTol, relTol, dtMax, dtMin, scaleMax, scaleMin, tStart));
  return o;
}
inline const ODEoptions& DEFAULT_ODEoptions() {  // ode.nim:104
  static const ODEoptions o = newODEoptions();
  return o;
}

// commonTypes.nim:4-39 — the context handed to the RHS; fValues is the float-parameter table
template <class T>
struct NumContext {
  std::map<std::string, double> fValues;
  std::map<std::string, T> tValues;
  double getF(const std::string& key) const { return fValues.at(key); }
  void setF(const std::string& key, double v) { fValues[key] = v; }
  T& operator[](const std::string& key) { return tValues[key]; }
};

// Stand-in for the closure ODEProc[T] (ode.nim:36): a compiled-in device RHS + the ctx.fValues keys it reads
struct RhsSpec {
  nnhip_rhs_kind kind;
  std::vector<std::string> keys;
  std::map<std::string, double> defaults;
  bool scalarsInBlock = false;  // a right-hand side with a context layout and more than 8 keys: its scalars lead the shared block (bindCtx)
  template <class T>
  std::vector<double> params(const NumContext<T>* ctx) const {
    if (scalarsInBlock) return {};
    return allParams(ctx);
  }
  template <class T>
  std::vector<double> allParams(const NumContext<T>* ctx) const {
    std::vector<double> p;
    for (const auto& k : keys) {
      if (ctx) {
        auto it = ctx->fValues.find(k);
        if (it != ctx->fValues.end()) { p.push_back(it->second); continue; }
      }
      auto d = defaults.find(k);
      if (d == defaults.end()) throw std::out_of_range("ctx.fValues has no '" + k + "'");  // Nim: KeyError
      p.push_back(d->second);
    }
    return p;
  }
};
inline RhsSpec rhsNegY() { return {NNHIP_RHS_NEG_Y, {}, {}}; }                                  // ode.nim:16-17
inline RhsSpec rhsLinear() { return {NNHIP_RHS_LINEAR, {"a"}, {}}; }                            // tests/test_ode.nim:5
inline RhsSpec rhsLinear(double a) { return {NNHIP_RHS_LINEAR, {"a"}, {{"a", a}}}; }
inline RhsSpec rhsLorenz(double sigma = 10.0, double rho = 28.0, double beta = 8.0 / 3.0) {
  return {NNHIP_RHS_LORENZ, {"sigma", "rho", "beta"}, {{"sigma", sigma}, {"rho", rho}, {"beta", beta}}};
}
inline RhsSpec rhsRing(double c = 0.1) { return {NNHIP_RHS_RING, {"c"}, {{"c", c}}}; }
inline RhsSpec rhsAffineT(double a, double b) { return {NNHIP_RHS_AFFINE_T, {"a", "b"}, {{"a", a}, {"b", b}}}; }
inline RhsSpec rhsVanDerPol(double mu = 1.0) { return {NNHIP_RHS_VANDERPOL, {"mu"}, {{"mu", mu}}}; }
// An arbitrary right-hand side from HIP C++ source (compiled at run time, nnhip_ode_rhs_compile): `body` is the body of
// rhs(double t, const double* y, double* dy, const double* p) with p = the values of `keys` in ctx.fValues order.
inline RhsSpec rhsFromSource(int dim, const std::string& body, std::vector<std::string> keys = {},
                             std::map<std::string, double> defaults = {}, const std::string& name = "user") {
  int kind = 0;
  throwOn(nnhip_ode_rhs_compile(name.c_str(), dim, (int)keys.size(), body.c_str(), &kind));
  return {static_cast<nnhip_rhs_kind>(kind), std::move(keys), std::move(defaults)};
}

// A per-component right-hand side from source (`body` returns dy_c for the component index c) that reads components c - lo .. c + hi of
// its system only (cyclically): stencils, rings, banded couplings.  On the lanes-per-system kernels the neighbours then come from the
// adjacent lanes instead of the LDS stage vector (nnhip_ode_rhs_compile_comp + nnhip_ode_rhs_set_halo) — the same bits, up to 2x faster.
inline RhsSpec rhsFromSourcePerComponent(int dim, const std::string& body, std::vector<std::string> keys = {}, std::map<std::string, double> defaults = {},
                                         const std::string& name = "user", int haloLo = -1, int haloHi = -1) {
  int kind = 0;
  throwOn(nnhip_ode_rhs_compile_comp(name.c_str(), dim, (int)keys.size(), body.c_str(), &kind));
  if (haloLo >= 0 && haloHi >= 0) throwOn(nnhip_ode_rhs_set_halo(kind, haloLo, haloHi));
  return {static_cast<nnhip_rhs_kind>(kind), std::move(keys), std::move(defaults)};
}

// NumContext in full (commonTypes.nim:4-27; nnhip_ode_rhs_compile_ctx): a right-hand side from source whose body also reads named
// ctx.tValues entries — NAME[j] of a vector the batch shares, NAME(j) of the IVP's own vector — any number of fValues (p[k]) and
// per-IVP mutable slots aux(j) (the mutable ctx of ode.nim:599).
struct CtxVector { std::string name; int64_t len; bool perIvp; };
inline RhsSpec rhsFromSourceCtx(int dim, const std::string& body, std::vector<std::string> keys, const std::vector<CtxVector>& vectors, int nAux = 0,
                                std::map<std::string, double> defaults = {}, const std::string& name = "user", bool perComponent = false) {
  int kind = 0;
  std::vector<const char*> names;
  std::vector<int64_t> lens;
  std::vector<int> per;
  for (const auto& v : vectors) { names.push_back(v.name.c_str()); lens.push_back(v.len); per.push_back(v.perIvp ? 1 : 0); }
  throwOn(nnhip_ode_rhs_compile_ctx(name.c_str(), dim, (int)keys.size(), body.c_str(), perComponent ? 1 : 0, (int)vectors.size(), names.data(), lens.data(),
                                    per.data(), nAux, &kind));
  RhsSpec r{static_cast<nnhip_rhs_kind>(kind), std::move(keys), std::move(defaults)};
  r.scalarsInBlock = r.keys.size() > 8;
  return r;
}
// The closure capturing its ctx: `shared` = the shared vectors concatenated in declaration order (after the scalars, when there are
// more than 8 keys: pass f.allParams(&ctx) in front), perIvp [rows][N], auxInit [nAux][N] (host arrays, uploaded to `device`).
inline void bindCtx(const RhsSpec& f, const std::vector<double>& shared, const std::vector<double>& perIvp, const std::vector<double>& auxInit, int nAux,
                    int64_t N, int device = 0) {
  throwOn(nnhip_ode_rhs_bind_ctx_f64(f.kind, shared.empty() ? nullptr : shared.data(), (int64_t)shared.size(), perIvp.empty() ? nullptr : perIvp.data(),
                                     N > 0 ? (int64_t)perIvp.size() / N : 0, auxInit.empty() ? nullptr : auxInit.data(), nAux, N, device));
}
inline std::vector<double> readAux(const RhsSpec& f, int nAux, int64_t N) {
  std::vector<double> a((size_t)nAux * (size_t)N);
  throwOn(nnhip_ode_rhs_read_aux_f64(f.kind, a.data()));
  return a;
}

// A batch of N initial states of `dim` float64 components (host memory).  dim == 1 is the reference's
// scalar `float` state; dim > 1 its Vector[float] / seq[float] state.
struct OdeBatch {
  int64_t N = 0;
  int dim = 1;
  nnhip_layout layout = NNHIP_LAYOUT_SOA;
  std::vector<double> data;  // [dim][N] (SoA) or [N][dim] (AoS)
  double& at(int64_t i, int c) { return layout == NNHIP_LAYOUT_SOA ? data[(size_t)c * N + i] : data[(size_t)i * dim + c]; }
  double at(int64_t i, int c) const { return layout == NNHIP_LAYOUT_SOA ? data[(size_t)c * N + i] : data[(size_t)i * dim + c]; }
  static OdeBatch zeros(int64_t N, int dim = 1, nnhip_layout layout = NNHIP_LAYOUT_SOA) {
    OdeBatch b; b.N = N; b.dim = dim; b.layout = layout; b.data.assign((size_t)N * dim, 0.0); return b;
  }
};

// Result of a batched solveODE: y[j] is the batch state at t[j] (same layout as y0).  ny[i] = number of rows the
// reference returns for IVP i (rows beyond are NaN; see nnhip_ode.h).
struct OdeSolution {
  std::vector<double> t;
  std::vector<OdeBatch> y;
  std::vector<int32_t> ny;
  nnhip_ode_stats stats{};
};

// solveODE — ode.nim:589-651.  integrator is the reference's case-insensitive string; unknown -> invalid_argument.
template <class T = double>
inline OdeSolution solveODE(const RhsSpec& f, const OdeBatch& y0, const std::vector<double>& tspan,
                            const ODEoptions& options = DEFAULT_ODEoptions(), const NumContext<T>* ctx = nullptr,
                            const std::string& integrator = "dopri54", int device = 0, int n_gpus = 1,
                            const std::vector<std::vector<double>>& sweep = {}, const std::vector<double>& sortBy = {}, bool autoSort = false) {
  // sweep[k][i]: value of RHS parameter k for IVP i (a parameter sweep: every IVP its own ctx); empty = one ctx for the batch
  // sortBy (one key per IVP) / autoSort: divergence binning below the C ABI (nnhip_ode_solve_batch_sorted_f64) for heterogeneous
  // batches of adaptive solves; results stay in the caller's order and are bit-identical
  const int integ = nnhip_ode_integrator_id(integrator.c_str());
  if (integ < 0) throw std::invalid_argument(integrator + " is not a valid integrator");  // ode.nim:651
  const std::vector<double> p = f.params(ctx);
  const int n_t = (int)tspan.size();
  OdeSolution sol;
  sol.t.assign((size_t)std::max(n_t, 1), 0.0);
  std::vector<double> yout((size_t)n_t * y0.N * y0.dim);
  sol.ny.assign((size_t)y0.N, 0);
  int rc;
  if (n_gpus > 1) {  // contiguous shards of the batch (and of the sweep table) per device
    std::vector<double> flat;
    for (const auto& row : sweep) {
      if ((int64_t)row.size() != y0.N) throw std::invalid_argument("sweep rows must have one value per IVP");
      flat.insert(flat.end(), row.begin(), row.end());
    }
    rc = nnhip_ode_solve_batch_multi_gpu_sweep_f64(&options, integ, f.kind, p.data(), (int)p.size(), flat.empty() ? nullptr : flat.data(), (int)sweep.size(),
                                                   y0.data.data(), y0.N, y0.dim, y0.layout, tspan.data(), n_t, sol.t.data(), yout.data(), sol.ny.data(),
                                                   nullptr, nullptr, 0, &sol.stats, n_gpus);
  } else if (!sortBy.empty() || autoSort) {
    if (!sortBy.empty() && (int64_t)sortBy.size() != y0.N) throw std::invalid_argument("sortBy needs one key per IVP");
    std::vector<double> flat;
    for (const auto& row : sweep) {
      if ((int64_t)row.size() != y0.N) throw std::invalid_argument("sweep rows must have one value per IVP");
      flat.insert(flat.end(), row.begin(), row.end());
    }
    rc = nnhip_ode_solve_batch_sorted_f64(&options, integ, f.kind, p.data(), (int)p.size(), flat.empty() ? nullptr : flat.data(), (int)sweep.size(),
                                          y0.data.data(), y0.N, y0.dim, y0.layout, tspan.data(), n_t, sol.t.data(), yout.data(), sol.ny.data(), nullptr,
                                          nullptr, 0, sortBy.empty() ? nullptr : sortBy.data(), 0, device);
    int nt = 0;
    if (rc == 0) rc = nnhip_ode_time_grid(&options, tspan.data(), n_t, nullptr, &nt);
    sol.stats.n_t_out = nt;
  } else {
    std::vector<double> flat;
    for (const auto& row : sweep) {
      if ((int64_t)row.size() != y0.N) throw std::invalid_argument("sweep rows must have one value per IVP");
      flat.insert(flat.end(), row.begin(), row.end());
    }
    rc = nnhip_ode_solve_batch_sweep_f64(&options, integ, f.kind, p.data(), (int)p.size(), flat.empty() ? nullptr : flat.data(), (int)sweep.size(),
                                         y0.data.data(), y0.N, y0.dim, y0.layout, tspan.data(), n_t, sol.t.data(), yout.data(), sol.ny.data(), nullptr,
                                         nullptr, 0, &sol.stats, device);
  }
  throwOn(rc);
  sol.t.resize((size_t)sol.stats.n_t_out);
  sol.y.resize((size_t)n_t);
  const size_t row = (size_t)y0.N * y0.dim;
  for (int j = 0; j < n_t; ++j) {
    sol.y[j].N = y0.N; sol.y[j].dim = y0.dim; sol.y[j].layout = y0.layout;
    sol.y[j].data.assign(yout.begin() + (size_t)j * row, yout.begin() + (size_t)(j + 1) * row);
  }
  return sol;
}

// N separate reference calls in one launch: result row set i == solveODE(f, y0_i, {options_i.tStart, tEnd[i]}, options_i)
// (ode.nim:589-591: every call owns its tspan and its options).  `options` holds one object per IVP, or one for all, or none
// (DEFAULT_ODEoptions).  y[0], y[1] are the two rows the reference returns per call (y0 first when tEnd > tStart, last when
// tEnd < tStart); ny[i] = 1 when the span is empty, -1 for a call the reference would refuse (see nnhip_ode.h); sol.t stays empty
// (the times differ per IVP: they are {min, max} of (tStart_i, tEnd[i])).
template <class T = double>
inline OdeSolution solveODECalls(const RhsSpec& f, const OdeBatch& y0, const std::vector<double>& tEnd,
                                 const std::vector<ODEoptions>& options = {}, const NumContext<T>* ctx = nullptr,
                                 const std::string& integrator = "dopri54", int device = 0,
                                 const std::vector<std::vector<double>>& sweep = {}) {
  const int integ = nnhip_ode_integrator_id(integrator.c_str());
  if (integ < 0) throw std::invalid_argument(integrator + " is not a valid integrator");  // ode.nim:651
  if ((int64_t)tEnd.size() != y0.N) throw std::invalid_argument("tEnd needs one value per IVP");
  if (options.size() > 1 && (int64_t)options.size() != y0.N) throw std::invalid_argument("options: one object, or one per IVP");
  const std::vector<double> p = f.params(ctx);
  std::vector<double> flat;
  for (const auto& row : sweep) {
    if ((int64_t)row.size() != y0.N) throw std::invalid_argument("sweep rows must have one value per IVP");
    flat.insert(flat.end(), row.begin(), row.end());
  }
  OdeSolution sol;
  const size_t row = (size_t)y0.N * y0.dim;
  std::vector<double> yout(2 * row);
  sol.ny.assign((size_t)y0.N, 0);
  const ODEoptions& base = options.size() == 1 ? options[0] : DEFAULT_ODEoptions();
  throwOn(nnhip_ode_solve_batch_calls_f64(&base, options.size() > 1 ? options.data() : nullptr, integ, f.kind, p.data(), (int)p.size(),
                                          flat.empty() ? nullptr : flat.data(), (int)sweep.size(), y0.data.data(), y0.N, y0.dim, y0.layout,
                                          tEnd.data(), yout.data(), sol.ny.data(), nullptr, nullptr, 0, device));
  sol.y.resize(2);
  for (int j = 0; j < 2; ++j) {
    sol.y[j].N = y0.N; sol.y[j].dim = y0.dim; sol.y[j].layout = y0.layout;
    sol.y[j].data.assign(yout.begin() + (size_t)j * row, yout.begin() + (size_t)(j + 1) * row);
  }
  return sol;
}

// The same for n_t-point tspans: result row set i == solveODE(f, y0_i, tspans[i], options_i) — tspans[i] in any order, on both sides
// of options_i.tStart, with duplicates.  sol.y[j] is the batch at output slot j; the times of IVP i are tOut[i] (its sorted tspan,
// what the reference returns as `t`; shorter than n_t when tspan_i holds tStart_i more than once).
template <class T = double>
inline OdeSolution solveODECalls(const RhsSpec& f, const OdeBatch& y0, const std::vector<std::vector<double>>& tspans,
                                 std::vector<std::vector<double>>& tOut, const std::vector<ODEoptions>& options = {},
                                 const NumContext<T>* ctx = nullptr, const std::string& integrator = "dopri54", int device = 0) {
  const int integ = nnhip_ode_integrator_id(integrator.c_str());
  if (integ < 0) throw std::invalid_argument(integrator + " is not a valid integrator");  // ode.nim:651
  if ((int64_t)tspans.size() != y0.N) throw std::invalid_argument("tspans needs one row per IVP");
  if (options.size() > 1 && (int64_t)options.size() != y0.N) throw std::invalid_argument("options: one object, or one per IVP");
  const int n_t = tspans.empty() ? 0 : (int)tspans[0].size();
  std::vector<double> flat;
  for (const auto& row : tspans) {
    if ((int)row.size() != n_t) throw std::invalid_argument("every tspan needs the same number of points");
    flat.insert(flat.end(), row.begin(), row.end());
  }
  const std::vector<double> p = f.params(ctx);
  OdeSolution sol;
  const size_t rowSz = (size_t)y0.N * y0.dim;
  std::vector<double> yout((size_t)n_t * rowSz), tflat((size_t)y0.N * n_t);
  sol.ny.assign((size_t)y0.N, 0);
  const ODEoptions& base = options.size() == 1 ? options[0] : DEFAULT_ODEoptions();
  throwOn(nnhip_ode_solve_batch_tspans_f64(&base, options.size() > 1 ? options.data() : nullptr, integ, f.kind, p.data(), (int)p.size(), nullptr, 0,
                                           y0.data.data(), y0.N, y0.dim, y0.layout, flat.data(), n_t, tflat.data(), yout.data(), sol.ny.data(), nullptr,
                                           nullptr, 0, device));
  tOut.assign((size_t)y0.N, {});
  for (int64_t i = 0; i < y0.N; ++i)
    for (int j = 0; j < n_t; ++j)
      if (tflat[(size_t)i * n_t + j] == tflat[(size_t)i * n_t + j]) tOut[(size_t)i].push_back(tflat[(size_t)i * n_t + j]);  // NaN = beyond the returned times
  sol.y.resize((size_t)n_t);
  for (int j = 0; j < n_t; ++j) {
    sol.y[j].N = y0.N; sol.y[j].dim = y0.dim; sol.y[j].layout = y0.layout;
    sol.y[j].data.assign(yout.begin() + (size_t)j * rowSz, yout.begin() + (size_t)(j + 1) * rowSz);
  }
  return sol;
}

// ---- the consumers on either side of the solver: same names as the reference's procs, batched, over the
// host-pointer entries (arrays are staged through the device per call) ---------------------------------------------------------
namespace detail {
inline std::vector<double> flatten(const std::vector<OdeBatch>& Y) {
  std::vector<double> f;
  for (const auto& b : Y) f.insert(f.end(), b.data.begin(), b.data.end());
  return f;
}
inline std::vector<OdeBatch> rows(const std::vector<double>& flat, size_t nRows, const OdeBatch& proto) {
  std::vector<OdeBatch> out(nRows);
  const size_t m = proto.data.size();
  for (size_t j = 0; j < nRows; ++j) {
    out[j].N = proto.N; out[j].dim = proto.dim; out[j].layout = proto.layout;
    out[j].data.assign(flat.begin() + j * m, flat.begin() + (j + 1) * m);
  }
  return out;
}
}  // namespace detail

// cumtrapz(Y, X) / cumsimpson(Y, X) for discrete points (integrate.nim:120-135, 329-375): Y[j] = the batch at X[j]
inline std::vector<OdeBatch> cumtrapz(const std::vector<OdeBatch>& Y, const std::vector<double>& X, int device = 0) {
  if (Y.size() != X.size()) throw std::invalid_argument("X and Y must have the same length");
  const std::vector<double> in = detail::flatten(Y);
  std::vector<double> out(in.size());
  throwOn(nnhip_cumtrapz_batch_f64(X.data(), (int)X.size(), in.data(), (int64_t)Y.at(0).data.size(), out.data(), device));
  int nRows = 0;  // X in any order: sorted and trimmed below the boundary (integrate.nim:131) — one row per distinct abscissa
  throwOn(nnhip_dataset_rows_f64(X.data(), (int)X.size(), &nRows, nullptr));
  return detail::rows(out, (size_t)nRows, Y[0]);
}
inline std::vector<OdeBatch> cumsimpson(const std::vector<OdeBatch>& Y, const std::vector<double>& X, int device = 0) {
  if (Y.size() != X.size()) throw std::invalid_argument("X and Y must have the same length");
  const std::vector<double> in = detail::flatten(Y);
  std::vector<double> out(in.size());
  throwOn(nnhip_cumsimpson_batch_f64(X.data(), (int)X.size(), in.data(), (int64_t)Y.at(0).data.size(), out.data(), device));
  int nRows = 0;  // the rows hermiteInterpolate yields at the caller's abscissae (integrate.nim:375)
  throwOn(nnhip_dataset_rows_f64(X.data(), (int)X.size(), nullptr, &nRows));
  return detail::rows(out, (size_t)nRows, Y[0]);
}

// cumtrapz(f, X, ctx, dx) / cumsimpson(f, X, ctx, dx) (integrate.nim:138-175, 377-400) for N parameter sets at once:
// sweep[k][i] overrides parameter k for item i (empty: one item with ctx's parameters).  result[j] = the batch at X[j];
// the reference can return fewer rows than X.size() and so does this.
template <class T = double>
inline std::vector<OdeBatch> cumQuadFn(bool simpson, const RhsSpec& f, const std::vector<double>& X, const NumContext<T>* ctx, double dx,
                                       const std::vector<std::vector<double>>& sweep, int dim, int device) {
  const std::vector<double> p = f.params(ctx);
  const int64_t N = sweep.empty() ? 1 : (int64_t)sweep[0].size();
  std::vector<double> flat;
  for (const auto& row : sweep) {
    if ((int64_t)row.size() != N) throw std::invalid_argument("sweep rows must have the same length");
    flat.insert(flat.end(), row.begin(), row.end());
  }
  OdeBatch proto = OdeBatch::zeros(N, dim);
  std::vector<double> out(X.size() * proto.data.size());
  int rows = 0;
  const auto fn = simpson ? nnhip_cumsimpson_fn_batch_f64 : nnhip_cumtrapz_fn_batch_f64;
  throwOn(fn(f.kind, p.data(), (int)p.size(), flat.empty() ? nullptr : flat.data(), (int)sweep.size(), N, dim, NNHIP_LAYOUT_SOA, X.data(), (int)X.size(),
             dx, out.data(), &rows, device));
  return detail::rows(out, (size_t)rows, proto);
}
template <class T = double>
inline std::vector<OdeBatch> cumtrapz(const RhsSpec& f, const std::vector<double>& X, const NumContext<T>* ctx = nullptr, double dx = 1e-5,
                                      const std::vector<std::vector<double>>& sweep = {}, int dim = 1, int device = 0) {
  return cumQuadFn<T>(false, f, X, ctx, dx, sweep, dim, device);
}
template <class T = double>
inline std::vector<OdeBatch> cumsimpson(const RhsSpec& f, const std::vector<double>& X, const NumContext<T>* ctx = nullptr, double dx = 1e-5,
                                        const std::vector<std::vector<double>>& sweep = {}, int dim = 1, int device = 0) {
  return cumQuadFn<T>(true, f, X, ctx, dx, sweep, dim, device);
}

// newHermiteSpline(X, Y[, dY]) + eval / derivEval (interpolate.nim:216-257, 299-390) for a whole batch
enum class ExtrapolateKind { Constant = 0, Edge = 1, Linear = 2, Native = 3, Error = 4 };  // interpolate.nim:89-90
class HermiteSpline {
 public:
  HermiteSpline(const std::vector<double>& X, const std::vector<OdeBatch>& Y, const std::vector<OdeBatch>& dY = {})
      : X_(X), Y_(detail::flatten(Y)), dY_(detail::flatten(dY)), proto_(Y.at(0)) {
    if (X.size() != Y.size() || (!dY.empty() && dY.size() != X.size())) throw std::invalid_argument("X and Y and dY must have the same length.");
  }
  std::vector<OdeBatch> eval(const std::vector<double>& x, ExtrapolateKind extrap = ExtrapolateKind::Native, double extrapValue = 0.0, int device = 0) const {
    return run(x, 0, extrap, extrapValue, device);
  }
  std::vector<OdeBatch> derivEval(const std::vector<double>& x, ExtrapolateKind extrap = ExtrapolateKind::Native, double extrapValue = 0.0, int device = 0) const {
    return run(x, 1, extrap, extrapValue, device);
  }

 private:
  std::vector<OdeBatch> run(const std::vector<double>& x, int deriv, ExtrapolateKind extrap, double extrapValue, int device) const {
    std::vector<double> out(x.size() * proto_.data.size());
    throwOn(nnhip_hermite_spline_eval_batch_f64(X_.data(), (int)X_.size(), Y_.data(), dY_.empty() ? nullptr : dY_.data(), (int64_t)proto_.data.size(),
                                                x.data(), (int)x.size(), deriv, (int)extrap, extrapValue, out.data(), device));
    return detail::rows(out, x.size(), proto_);
  }
  std::vector<double> X_, Y_, dY_;
  OdeBatch proto_;
};
inline HermiteSpline newHermiteSpline(const std::vector<double>& X, const std::vector<OdeBatch>& Y, const std::vector<OdeBatch>& dY = {}) {
  return HermiteSpline(X, Y, dY);
}

}  // namespace numericalnim

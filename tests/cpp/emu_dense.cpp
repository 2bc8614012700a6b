// emu_dense.cpp — TEST INFRASTRUCTURE: the adaptive streaming driver WITH dense output (nnhip_ode_adaptive_stream_dense_f64_dev: the whole ODESolver of
// ode.nim:471-586 over one-iteration-per-launch kernels, per-IVP (t, dt), denseIndex and rows in memory) with its kernel BODIES executed on the host
// (tests/cpp/hip_cpu_emu.hpp) — advance_dense_init_kernel, advance_dense_tpi_kernel / advance_dense_lps_kernel, advance_dense_finalize_kernel (all six modes),
// rhs_batch_kernel + fill_td_kernel (the generic initialisation of the lanes-per-system form).  The sequence of launches is the library's (ode_capi_stream.hip),
// restated here without streams, graphs and polling: forward direction first when both are asked for, one launch per loop iteration until no workgroup reports
// work left.  Input and output as tests/cpp/emu_solve.cpp (the step counters print as 0: the streaming driver does not keep them); adaptive integrators only.
#include "solve_plan.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace nnhip;

struct Ws {
  std::vector<double> yW, fsal, td, tReq;
  std::vector<int32_t> denseIdx, fwdRows;
  std::vector<unsigned int> active = std::vector<unsigned int>(kAggSlots, 0u);
};
static bool any(std::vector<unsigned int>& f) {
  bool a = false;
  for (auto& x : f) { a = a || x != 0; x = 0; }
  return a;
}

template <int METHOD, class RHS>
struct Tpi {
  static void init(const StepArgs& a, const double* y0, double tS, double dtInit, int) {
    hipemu::launch(advance_dense_init_kernel<RHS>, dim3((unsigned)((a.N + kBlock - 1) / kBlock)), dim3(kBlock), a, y0, tS, dtInit);
  }
  static void advance(const StepArgs& a) {
    const int bs = adv_dense_block<false>();
    hipemu::launch(advance_dense_tpi_kernel<METHOD, RHS, false>, dim3((unsigned)((a.N + bs - 1) / bs)), dim3(bs), a);
  }
};
template <int METHOD, class RHS, int CPL>
struct Lps {
  // the driver's generic initialisation (no init kernel for the lanes-per-system form): y = y0, FSAL = f(t0, y0) or -f(-(-t0) ...) = -f(t0, y0), (t, dt), denseIndex = 0
  static void init(const StepArgs& a, const double* y0, double tS, double dtInit, int dim) {
    const int64_t n = a.N * dim;
    std::memcpy(a.y_out, y0, (size_t)n * sizeof(double));
    const bool neg = a.negate != 0;
    hipemu::launch(rhs_batch_kernel<RHS>, dim3((unsigned)((a.N + kBlock - 1) / kBlock)), dim3(kBlock), a.N, a.ivpStride, a.compStride, neg ? -tS : tS, y0, a.fsal_out, a.P);
    if (neg) for (int64_t k = 0; k < n; ++k) a.fsal_out[k] = -a.fsal_out[k];
    hipemu::launch(fill_td_kernel<0>, dim3((unsigned)((a.N + kBlock - 1) / kBlock)), dim3(kBlock), reinterpret_cast<double2*>(a.t_io), a.N, tS, dtInit);
    std::memset(a.denseIdx_io, 0, (size_t)a.N * sizeof(int32_t));
  }
  static void advance(const StepArgs& a) {
    constexpr int perBlock = kBlock / (RHS::dim / CPL);
    hipemu::launch(advance_dense_lps_kernel<METHOD, RHS, CPL>, dim3((unsigned)((a.N + perBlock - 1) / perBlock)), dim3(kBlock), a);
  }
};

template <class K>
static int drive(int integrator, const nnhip_ode_options& opt, const Params& P, const double* y0, int64_t N, int dim, int layout, const double* tspan, int n_t,
                 std::vector<double>& yOut, std::vector<int32_t>& ny, nnhip_capi::TimeGrid& g, int64_t& launches) {
  nnhip_capi::make_grid(&opt, tspan, n_t, g);
  const int64_t nState = N * dim;
  Ws w;
  w.yW.assign((size_t)nState, 0.0); w.fsal.assign((size_t)nState, 0.0); w.td.assign((size_t)2 * N, 0.0);
  w.denseIdx.assign((size_t)N + 2, 0); w.fwdRows.assign((size_t)N, 0);
  const int nPos = (int)g.tPos.size(), nNeg = (int)g.tNeg.size();
  w.tReq.assign(g.tPos.begin(), g.tPos.end());
  w.tReq.insert(w.tReq.end(), g.tNeg.begin(), g.tNeg.end());
  w.tReq.resize(w.tReq.size() + 8, 0.0);
  StepArgs a{};
  a.N = N;
  if (layout == NNHIP_LAYOUT_SOA) { a.ivpStride = 1; a.compStride = N; } else { a.ivpStride = dim; a.compStride = 1; }
  a.y_in = w.yW.data(); a.y_out = w.yW.data(); a.fsal_in = w.fsal.data(); a.fsal_out = w.fsal.data(); a.error = nullptr;
  a.ctl = nnhip_capi::ctl_of(&opt); a.P = P;
  a.t_io = w.td.data(); a.dt_io = nullptr;
  a.denseIdx_io = w.denseIdx.data(); a.emitAfter = 1;
  a.recomputeFsal = (integrator == NNHIP_DOPRI54 || integrator == NNHIP_TSIT54) ? 1 : 0;
  a.rows = yOut.data(); a.rowStride = nState;
  a.nontemporal = 0;
  const double dtInit = std::sqrt(opt.dtMax * opt.dtMin);  // :491-493
  const dim3 grid((unsigned)((N + kBlock - 1) / kBlock)), block(kBlock);
  auto finalize = [&](int mode) { hipemu::launch(advance_dense_finalize_kernel<0>, grid, block, a, mode, dim, y0, ny.data(), n_t, w.fwdRows.data()); };
  auto run_dir = [&](bool neg, double tStartEff, double tEnd, const double* req, int nReq, int rowBase0) -> int {
    a.negate = neg ? 1 : 0; a.tEnd = tEnd; a.tReq = req; a.nReq = nReq; a.rowBase = nullptr; a.rowBase0 = rowBase0;
    a.useDense = n_t != 2 ? 1 : 0;
    StepArgs run = a;
    K::init(run, y0, tStartEff, dtInit, dim);
    run.active = w.active.data();
    for (int64_t k = 0;; ++k) {
      if (k > 1000000) return 70;
      K::advance(run);
      ++launches;
      if (!any(w.active)) break;
    }
    return 0;
  };
  std::fill(ny.begin(), ny.end(), 0);
  const bool both = nNeg > 0 && nPos > 0;
  const int fwdBase = nNeg + (g.nZero ? 1 : 0);
  int rc = 0;
  if (both) {
    if ((rc = run_dir(false, opt.tStart, g.tEndPos, w.tReq.data(), nPos, fwdBase))) return rc;
    finalize(5);
  }
  if (nNeg > 0) {
    if ((rc = run_dir(true, -opt.tStart, g.tEndNeg, w.tReq.data() + nPos, nNeg, 0))) return rc;
    finalize(0);
  }
  if (g.nZero) finalize(1);
  if (both) {
    a.rowBase0 = fwdBase;
    finalize(4);
  } else if (nPos > 0) {
    if ((rc = run_dir(false, opt.tStart, g.tEndPos, w.tReq.data(), nPos, g.nZero ? 1 : 0))) return rc;
    finalize(2);
  }
  finalize(3);
  return 0;
}

template <int METHOD>
static int dispatch(int kind, int dim, const nnhip_ode_options& opt, const Params& P, const double* y0, int64_t N, int layout, const double* tspan, int n_t,
                    std::vector<double>& yOut, std::vector<int32_t>& ny, nnhip_capi::TimeGrid& g, int64_t& launches) {
#define GO(...) return drive<__VA_ARGS__>(METHOD, opt, P, y0, N, dim, layout, tspan, n_t, yOut, ny, g, launches)
  if (kind == NNHIP_RHS_NEG_Y && dim == 1) GO(Tpi<METHOD, RhsNegY<1>>);
  if (kind == NNHIP_RHS_LINEAR && dim == 1) GO(Tpi<METHOD, RhsLinear<1>>);
  if (kind == NNHIP_RHS_LINEAR && dim == 3) GO(Tpi<METHOD, RhsLinear<3>>);
  if (kind == NNHIP_RHS_AFFINE_T && dim == 1) GO(Tpi<METHOD, RhsAffineT<1>>);
  if (kind == NNHIP_RHS_LORENZ && dim == 3) GO(Tpi<METHOD, RhsLorenz>);
  if (kind == NNHIP_RHS_VANDERPOL && dim == 2) GO(Tpi<METHOD, RhsVanDerPol>);
  if (kind == NNHIP_RHS_RING && dim == 4) GO(Tpi<METHOD, RhsRing<4>>);
  if (kind == NNHIP_RHS_RING && dim == 16) GO(Lps<METHOD, RhsRing<16>, 4>);  // components per lane as find_advance_dense_tpi has them
#undef GO
  return 66;
}

int main() {
  int method, kind, dim, layout, n_t, n_params;
  long long N, max_steps;
  if (std::scanf("%d %d %d %d %lld %d %lld %d", &method, &kind, &dim, &layout, &N, &n_t, &max_steps, &n_params) != 8) return 64;
  nnhip_ode_options opt{};
  if (std::scanf("%la %la %la %la %la %la %la %la", &opt.dt, &opt.dtMax, &opt.dtMin, &opt.tStart, &opt.absTol, &opt.relTol, &opt.scaleMax, &opt.scaleMin) != 8) return 64;
  const int dimv = dim > 0 ? dim : 1;
  std::vector<double> params(n_params), tspan(n_t), y0((size_t)N * dimv);
  for (double& v : params) if (std::scanf("%la", &v) != 1) return 64;
  for (double& v : tspan) if (std::scanf("%la", &v) != 1) return 64;
  for (double& v : y0) if (std::scanf("%la", &v) != 1) return 64;
  Params P{};
  for (int k = 0; k < n_params && k < kMaxParams; ++k) P.p[k] = params[k];
  std::vector<double> yOut((size_t)n_t * dimv * N, -7.0);
  std::vector<int32_t> ny(N, -1);
  nnhip_capi::TimeGrid g;
  int64_t launches = 0;
  int rc;
  switch (method) {
    case NNHIP_RK21: rc = dispatch<NNHIP_RK21>(kind, dimv, opt, P, y0.data(), N, layout, tspan.data(), n_t, yOut, ny, g, launches); break;
    case NNHIP_BS32: rc = dispatch<NNHIP_BS32>(kind, dimv, opt, P, y0.data(), N, layout, tspan.data(), n_t, yOut, ny, g, launches); break;
    case NNHIP_DOPRI54: rc = dispatch<NNHIP_DOPRI54>(kind, dimv, opt, P, y0.data(), N, layout, tspan.data(), n_t, yOut, ny, g, launches); break;
    case NNHIP_TSIT54: rc = dispatch<NNHIP_TSIT54>(kind, dimv, opt, P, y0.data(), N, layout, tspan.data(), n_t, yOut, ny, g, launches); break;
    case NNHIP_VERN65: rc = dispatch<NNHIP_VERN65>(kind, dimv, opt, P, y0.data(), N, layout, tspan.data(), n_t, yOut, ny, g, launches); break;
    default: return 65;
  }
  if (rc) { std::fprintf(stderr, "emu_dense: rc %d (method %d kind %d dim %d)\n", rc, method, kind, dimv); return rc; }
  const int64_t rowStride = (int64_t)dimv * N, is = layout == NNHIP_LAYOUT_SOA ? 1 : dimv, cs = layout == NNHIP_LAYOUT_SOA ? N : 1;
  std::printf("t %zu", g.tOut.size());
  for (double v : g.tOut) std::printf(" %a", v);
  std::printf("\n");
  for (long long i = 0; i < N; ++i) {
    std::printf("ivp %d 0 0", ny[i]);
    for (int j = 0; j < ny[i] && j < n_t; ++j)
      for (int c = 0; c < dimv; ++c) std::printf(" %a", yOut[(size_t)j * rowStride + (size_t)i * is + (size_t)c * cs]);
    std::printf("\n");
    // rows beyond ny are NaN (mode 3)
    for (int j = ny[i]; j < n_t; ++j)
      for (int c = 0; c < dimv; ++c)
        if (!std::isnan(yOut[(size_t)j * rowStride + (size_t)i * is + (size_t)c * cs])) { std::fprintf(stderr, "row %d of IVP %lld is not NaN\n", j, i); return 71; }
  }
  std::printf("launches %lld\n", (long long)launches);
  return 0;
}

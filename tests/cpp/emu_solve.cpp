// emu_solve.cpp — TEST INFRASTRUCTURE: the fused solve (solveODE -> ODESolver per IVP, ode.nim:589-651, 471-586) through the BODIES of solve_tpi_kernel /
// solve_lps_kernel, executed on the host (tests/cpp/hip_cpu_emu.hpp: lanes as threads), with the launch record the library itself would build
// (numericalnim_amd/csrc/solve_plan.hpp: time grid, first step, the fixed-step methods' host-replayed schedule).  One case per run, read from stdin:
//   method kind dim layout N n_t max_steps n_params
//   dt dtMax dtMin tStart absTol relTol scaleMax scaleMin          (the fields of nnhip_ode_options, already abs()'d as newODEoptions leaves them)
//   params ... / tspan ... / y0 ... (N * max(dim, 1) values in `layout`)                                              — all doubles as hex floats
// Prints: "t <n>" + times, then per IVP "ivp <ny> <steps> <rejected>" + its ny rows (dim values each).
// Instantiated: the (integrator, right-hand side, size) combinations of tests/golden/ode_golden.json — all 14 integrators x {-y, a y, a y + b t (scalar and
// 3 components), Lorenz, Van der Pol, the 4- and the 16-component ring (the latter on the lanes-per-system kernel)}, lean (2-point tspan) and dense.
#include "solve_plan.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace nnhip;

template <int METHOD, class RHS, int MODE>
static void run_tpi(const SolveArgs& a) { hipemu::launch(solve_tpi_kernel<METHOD, RHS, MODE>, dim3((unsigned)((a.N + kBlock - 1) / kBlock)), dim3(kBlock), a); }
template <int METHOD, class RHS, int CPL, int MODE>
static void run_lps(const SolveArgs& a) {
  constexpr int perBlock = kBlock / (RHS::dim / CPL);
  hipemu::launch(solve_lps_kernel<METHOD, RHS, CPL, false, MODE>, dim3((unsigned)((a.N + perBlock - 1) / perBlock)), dim3(kBlock), a);
}
template <int METHOD, int MODE>
static bool dispatch_rhs(int kind, int dim, const SolveArgs& a) {
  constexpr bool ad = MethodTraits<METHOD>::adaptive;
  if (kind == NNHIP_RHS_NEG_Y && dim == 1) { run_tpi<METHOD, RhsNegY<1>, MODE>(a); return true; }
  if (kind == NNHIP_RHS_LINEAR && dim == 1) { run_tpi<METHOD, RhsLinear<1>, MODE>(a); return true; }
  if (kind == NNHIP_RHS_LINEAR && dim == 3) { run_tpi<METHOD, RhsLinear<3>, MODE>(a); return true; }
  if (kind == NNHIP_RHS_AFFINE_T && dim == 1) { run_tpi<METHOD, RhsAffineT<1>, MODE>(a); return true; }
  if (kind == NNHIP_RHS_LORENZ && dim == 3) { run_tpi<METHOD, RhsLorenz, MODE>(a); return true; }
  if (kind == NNHIP_RHS_VANDERPOL && dim == 2) { run_tpi<METHOD, RhsVanDerPol, MODE>(a); return true; }
  if (kind == NNHIP_RHS_RING && dim == 4) { run_tpi<METHOD, RhsRing<4>, MODE>(a); return true; }
  if (kind == NNHIP_RHS_RING && dim == 16) { run_lps<METHOD, RhsRing<16>, (ad ? 4 : 2), MODE>(a); return true; }  // components per lane as NNHIP_FOR_EACH_LPS_RHS has them
  return false;
}
template <int METHOD>
static bool dispatch_mode(int kind, int dim, const SolveArgs& a) { return a.useDense ? dispatch_rhs<METHOD, 1>(kind, dim, a) : dispatch_rhs<METHOD, 0>(kind, dim, a); }

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

  SolveArgs a{};
  std::vector<double> yOut((size_t)n_t * dimv * N, NAN);
  std::vector<int32_t> ny(N, -1);
  std::vector<int64_t> steps(N, -1), rej(N, -1);
  a.y0 = y0.data(); a.y_out = yOut.data(); a.ny_out = ny.data(); a.steps_out = steps.data(); a.rejected_out = rej.data();
  a.N = N;
  if (layout == NNHIP_LAYOUT_SOA) { a.ivpStride = 1; a.compStride = N; } else { a.ivpStride = dimv; a.compStride = 1; }
  a.rowStride = (int64_t)dimv * N;
  for (int k = 0; k < n_params && k < kMaxParams; ++k) a.P.p[k] = params[k];
  a.perIvpStride = N;
  bool adaptive = false;
  switch (method) {
#define M(id) case id: adaptive = MethodTraits<id>::adaptive; break;
    M(NNHIP_HEUN2) M(NNHIP_RALSTON2) M(NNHIP_KUTTA3) M(NNHIP_HEUN3) M(NNHIP_RALSTON3) M(NNHIP_SSPRK3) M(NNHIP_RALSTON4) M(NNHIP_KUTTA4) M(NNHIP_RK4)
    M(NNHIP_RK21) M(NNHIP_BS32) M(NNHIP_DOPRI54) M(NNHIP_TSIT54) M(NNHIP_VERN65)
#undef M
    default: return 65;
  }
  nnhip_capi::TimeGrid g;
  std::vector<double> emitW[2];
  std::vector<int64_t> emitStep[2];
  nnhip_capi::plan_solve(&opt, adaptive, tspan.data(), n_t, max_steps, a, g, emitW, emitStep);
  // the requested times and the emission tables where the kernel reads them (the library: one device workspace, same order)
  std::vector<double> tPos(g.tPos), tNeg(g.tNeg);
  if (a.useDense && (a.nPos + a.nNeg) > 0) {
    a.tPos = tPos.data(); a.tNeg = tNeg.data();
    a.emitW[0] = emitW[0].data(); a.emitW[1] = emitW[1].data(); a.emitStep[0] = emitStep[0].data(); a.emitStep[1] = emitStep[1].data();
  }
  bool ok = false;
  switch (method) {
#define M(id) case id: ok = dispatch_mode<id>(kind, dimv, a); break;
    M(NNHIP_HEUN2) M(NNHIP_RALSTON2) M(NNHIP_KUTTA3) M(NNHIP_HEUN3) M(NNHIP_RALSTON3) M(NNHIP_SSPRK3) M(NNHIP_RALSTON4) M(NNHIP_KUTTA4) M(NNHIP_RK4)
    M(NNHIP_RK21) M(NNHIP_BS32) M(NNHIP_DOPRI54) M(NNHIP_TSIT54) M(NNHIP_VERN65)
#undef M
  }
  if (!ok) { std::fprintf(stderr, "no instantiation for method %d kind %d dim %d\n", method, kind, dimv); return 66; }
  std::printf("t %zu", g.tOut.size());
  for (double v : g.tOut) std::printf(" %a", v);
  std::printf("\n");
  for (long long i = 0; i < N; ++i) {
    std::printf("ivp %d %lld %lld", ny[i], (long long)steps[i], (long long)rej[i]);
    for (int j = 0; j < ny[i] && j < n_t; ++j)
      for (int c = 0; c < dimv; ++c) std::printf(" %a", yOut[(size_t)j * a.rowStride + (size_t)i * a.ivpStride + (size_t)c * a.compStride]);
    std::printf("\n");
  }
  return 0;
}

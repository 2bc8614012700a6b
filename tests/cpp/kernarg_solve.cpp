// kernarg_solve.cpp — TEST INFRASTRUCTURE: the launch record of a fused solve (SolveArgs) exactly as the library's planning code fills it (solve_plan.hpp:
// time grid, first step, host-replayed fixed-step schedule), printed as hex bytes for tools/gfx950_isa_interp.py's callers, which execute the COMPILED fused
// kernels on the host.  Device addresses are the caller's (fake) ones.  2-point tspans only (no emission tables).
//   kernarg_solve <adaptive 0|1> <N> <dim> <layout> <max_steps> <t0> <t1> <8 option fields> <n_params> <params...> <y0> <y_out> <ny> <steps> <rejected>   (addresses decimal)
#include "solve_plan.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace nnhip;
int main(int argc, char** argv) {
  int k = 1;
  auto I = [&]() { return std::strtoll(argv[k++], nullptr, 0); };
  auto D = [&]() { return std::strtod(argv[k++], nullptr); };
  if (argc < 19) return 2;
  const int adaptive = (int)I();
  const int64_t N = I();
  const int dim = (int)I(), layout = (int)I();
  const int64_t max_steps = I();
  double tspan[2] = {D(), D()};
  nnhip_ode_options opt{};
  opt.dt = D(); opt.dtMax = D(); opt.dtMin = D(); opt.tStart = D(); opt.absTol = D(); opt.relTol = D(); opt.scaleMax = D(); opt.scaleMin = D();
  const int np = (int)I();
  SolveArgs a{};
  for (int j = 0; j < np; ++j) a.P.p[j] = D();
  a.y0 = (const double*)(uintptr_t)I(); a.y_out = (double*)(uintptr_t)I(); a.ny_out = (int32_t*)(uintptr_t)I();
  a.steps_out = (int64_t*)(uintptr_t)I(); a.rejected_out = (int64_t*)(uintptr_t)I();
  a.N = N;
  if (layout == NNHIP_LAYOUT_SOA) { a.ivpStride = 1; a.compStride = N; } else { a.ivpStride = dim; a.compStride = 1; }
  a.rowStride = (int64_t)dim * N;
  a.perIvpStride = N;
  nnhip_capi::TimeGrid g;
  std::vector<double> emitW[2];
  std::vector<int64_t> emitStep[2];
  nnhip_capi::plan_solve(&opt, adaptive != 0, tspan, 2, max_steps, a, g, emitW, emitStep);
  std::printf("%zu ", sizeof(a));
  const unsigned char* b = reinterpret_cast<const unsigned char*>(&a);
  for (size_t j = 0; j < sizeof(a); ++j) std::printf("%02x", b[j]);
  std::printf("\n");
  return 0;
}

// budget_tu.hip — TEST INFRASTRUCTURE: the hot kernels of the BASELINE configurations instantiated in one small translation unit, cross-compiled for gfx950 by
// tests/test_device_code_budget.py (hipcc needs no GPU) so that register / scratch / LDS budgets — what decides how many wavefronts a SIMD holds — are checked
// at every CPU test run.  Same flags as the library's Makefile.
#include "ode_kernels.hpp"
namespace nnhip {
// C2 / C5 headline (ode.nim:180-189): the variants the tuner picks inside and beyond the Infinity Cache
template __global__ void rk4_stream_vec_kernel<RhsNegY<1>, false, 1, 0>(const double*, double*, int64_t, double, double, double, double, const Params);
template __global__ void rk4_stream_vec_kernel<RhsNegY<1>, false, 4, 1>(const double*, double*, int64_t, double, double, double, double, const Params);
// streamed C3 / C4: one iteration of ODESolver's adaptive loop per launch (ode.nim:525-541)
template __global__ void advance_tpi_lean_kernel<NNHIP_DOPRI54, RhsLorenz>(const AdvLeanArgs);
template __global__ void advance_lps_lean_kernel<NNHIP_TSIT54, RhsRing<16>, 4>(const AdvLeanArgs);
template __global__ void advance_tpi_kernel<NNHIP_DOPRI54, RhsLorenz, false, false>(const StepArgs);
template __global__ void advance_lps_kernel<NNHIP_TSIT54, RhsRing<16>, 4, false>(const StepArgs);
// fused C1 / C2 / C3 / C4 (ode.nim:471-586 per IVP)
template __global__ void solve_tpi_kernel<NNHIP_RK4, RhsNegY<1>, 0>(const SolveArgs);
template __global__ void solve_tpi_kernel<NNHIP_DOPRI54, RhsLorenz, 0>(const SolveArgs);
template __global__ void solve_lps_kernel<NNHIP_TSIT54, RhsRing<16>, 4, false, 0>(const SolveArgs);
}

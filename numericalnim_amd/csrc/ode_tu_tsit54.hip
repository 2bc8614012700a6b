// ode_tu_tsit54.hip — Tsit54 instantiations (ode.nim:307-374).
#include "ode_kernels.hpp"
namespace nnhip {
SolveLaunchFn find_solve_tsit54(int rhs_kind, int dim) { return find_solve_tpi<NNHIP_TSIT54>(rhs_kind, dim); }
StepLaunchFn find_step_tsit54(int rhs_kind, int dim) { return find_step_tpi<NNHIP_TSIT54>(rhs_kind, dim); }
}  // namespace nnhip

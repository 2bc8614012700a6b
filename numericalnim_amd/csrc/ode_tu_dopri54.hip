// ode_tu_dopri54.hip — DOPRI54 instantiations (ode.nim:237-305).
#include "ode_kernels.hpp"
namespace nnhip {
SolveLaunchFn find_solve_dopri54(int rhs_kind, int dim) { return find_solve_tpi<NNHIP_DOPRI54>(rhs_kind, dim); }
StepLaunchFn find_step_dopri54(int rhs_kind, int dim) { return find_step_tpi<NNHIP_DOPRI54>(rhs_kind, dim); }
}  // namespace nnhip

// ode_tu_lean_fast.hip — the opt-in FMA-contracted instantiations of the streaming path (tuning knob "fp_contract"): advance_tpi_lean_kernel and
// advance_lps_lean_kernel for DOPRI54 and Tsit54 over the compiled-in right-hand sides, nothing else.  Compiled with -ffp-contract=fast
// -DNNHIP_NS=nnhip_fast (Makefile).  With a*b+c fused the streamed C4 kernel issues ~1/3 fewer FP64 instructions, which takes its issue floor below
// its HBM floor; the results stay inside north_star's tolerance for adaptive methods (1e-6), they are not the reference's bits.
#include "ode_kernels.hpp"

namespace NNHIP_NS {
StepLaunchFn find_advance_lean_dopri54(int rhs_kind, int dim) { return find_advance_lean<NNHIP_DOPRI54>(rhs_kind, dim); }
StepLaunchFn find_advance_lean_tsit54(int rhs_kind, int dim) { return find_advance_lean<NNHIP_TSIT54>(rhs_kind, dim); }
}  // namespace NNHIP_NS

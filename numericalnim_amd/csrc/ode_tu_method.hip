// ode_tu_method.hip — all (RHS, dim) instantiations of ONE integrator; compiled once per integrator with
// -DNNHIP_TU_METHOD=<nnhip_integrator id> -DNNHIP_TU_NAME=<its reference name> (see Makefile).
#include "ode_kernels.hpp"

#define NNHIP_CAT2(a, b) a##b
#define NNHIP_CAT(a, b) NNHIP_CAT2(a, b)

namespace NNHIP_NS {
SolveLaunchFn NNHIP_CAT(find_solve_, NNHIP_TU_NAME)(int rhs_kind, int dim, int dim16_variant) {
  return find_solve_tpi<NNHIP_TU_METHOD>(rhs_kind, dim, dim16_variant);
}
#ifndef NNHIP_FAST_ROOT  // the opt-in FMA-contracted build (namespace nnhip_fast, knob "fp_contract") exists for the compute-bound FUSED solves only: the
                         // step-streaming / advance / dense kernels of that namespace were never dispatched to (8 MB of device code nothing selected)
StepLaunchFn NNHIP_CAT(find_step_, NNHIP_TU_NAME)(int rhs_kind, int dim) { return find_step_tpi<NNHIP_TU_METHOD>(rhs_kind, dim); }
StepLaunchFn NNHIP_CAT(find_advance_, NNHIP_TU_NAME)(int rhs_kind, int dim) {
  if constexpr (MethodTraits<NNHIP_TU_METHOD>::adaptive) return find_advance_tpi<NNHIP_TU_METHOD>(rhs_kind, dim);
  else return nullptr;
}
FixedVecLaunchFn NNHIP_CAT(find_fixed_vec_, NNHIP_TU_NAME)(int rhs_kind, int dim) {
  if constexpr (!MethodTraits<NNHIP_TU_METHOD>::adaptive) return find_fixed_vec_tpi<NNHIP_TU_METHOD>(rhs_kind, dim);
  else return nullptr;
}
DenseAdvLaunch NNHIP_CAT(find_advance_dense_, NNHIP_TU_NAME)(int rhs_kind, int dim) {
  if constexpr (MethodTraits<NNHIP_TU_METHOD>::adaptive) return find_advance_dense_tpi<NNHIP_TU_METHOD>(rhs_kind, dim);
  else return DenseAdvLaunch{nullptr, nullptr};
}
#endif
}  // namespace NNHIP_NS

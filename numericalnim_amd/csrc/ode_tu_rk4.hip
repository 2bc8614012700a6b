// ode_tu_rk4.hip — RK4 instantiations (ode.nim:180-189): fused solve, step-streaming, vectorised scalar stream.
#include "ode_kernels.hpp"

namespace nnhip {

SolveLaunchFn find_solve_rk4(int rhs_kind, int dim) { return find_solve_tpi<NNHIP_RK4>(rhs_kind, dim); }
StepLaunchFn find_step_rk4(int rhs_kind, int dim) { return find_step_tpi<NNHIP_RK4>(rhs_kind, dim); }

bool rk4_stream_supported(int rhs_kind) {
  return rhs_kind == NNHIP_RHS_NEG_Y || rhs_kind == NNHIP_RHS_LINEAR || rhs_kind == NNHIP_RHS_AFFINE_T;
}

template <class RHS1>
static hipError_t launch_variant(const double* yin, double* yout, int64_t n, double t, double dt, const Params& P, int negate,
                                 int variant, hipStream_t s) {
  switch (variant) {
    case 1: return launch_rk4_stream_vec<RHS1, 1>(yin, yout, n, t, dt, P, negate, s);
    case 2: return launch_rk4_stream_vec<RHS1, 2>(yin, yout, n, t, dt, P, negate, s);
    case 8: return launch_rk4_stream_vec<RHS1, 8>(yin, yout, n, t, dt, P, negate, s);
    default: return launch_rk4_stream_vec<RHS1, 4>(yin, yout, n, t, dt, P, negate, s);
  }
}

hipError_t launch_rk4_stream(int rhs_kind, const double* yin, double* yout, int64_t n, double t, double dt, const Params& P,
                             int negate, int variant, hipStream_t s) {
  switch (rhs_kind) {
    case NNHIP_RHS_NEG_Y: return launch_variant<RhsNegY<1>>(yin, yout, n, t, dt, P, negate, variant, s);
    case NNHIP_RHS_LINEAR: return launch_variant<RhsLinear<1>>(yin, yout, n, t, dt, P, negate, variant, s);
    case NNHIP_RHS_AFFINE_T: return launch_variant<RhsAffineT<1>>(yin, yout, n, t, dt, P, negate, variant, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace nnhip

// ode_tu_rk4_stream.hip — the headline kernel's instantiations: vectorised scalar RK4 step-streaming (ode.nim:180-189).
#include "ode_kernels.hpp"

namespace nnhip {

bool rk4_stream_supported(int rhs_kind) {
  return rhs_kind == NNHIP_RHS_NEG_Y || rhs_kind == NNHIP_RHS_LINEAR || rhs_kind == NNHIP_RHS_AFFINE_T;
}

template <class RHS1, int MODE>
static hipError_t launch_vec(const double* yin, double* yout, int64_t n, double t, double dt, const Params& P, int negate,
                             const StreamTune& tune, hipStream_t s) {
  switch (tune.vec) {
    case 1: return launch_rk4_stream_vec<RHS1, 1, MODE>(yin, yout, n, t, dt, P, negate, tune, s);
    case 2: return launch_rk4_stream_vec<RHS1, 2, MODE>(yin, yout, n, t, dt, P, negate, tune, s);
    case 8: return launch_rk4_stream_vec<RHS1, 8, MODE>(yin, yout, n, t, dt, P, negate, tune, s);
    default: return launch_rk4_stream_vec<RHS1, 4, MODE>(yin, yout, n, t, dt, P, negate, tune, s);
  }
}
template <class RHS1>
static hipError_t launch_mode(const double* yin, double* yout, int64_t n, double t, double dt, const Params& P, int negate,
                              const StreamTune& tune, hipStream_t s) {
  switch (tune.mode) {
    case 1: return launch_vec<RHS1, 1>(yin, yout, n, t, dt, P, negate, tune, s);
    case 2: return launch_vec<RHS1, 2>(yin, yout, n, t, dt, P, negate, tune, s);
    case 3: return launch_vec<RHS1, 3>(yin, yout, n, t, dt, P, negate, tune, s);
    default: return launch_vec<RHS1, 0>(yin, yout, n, t, dt, P, negate, tune, s);
  }
}

hipError_t launch_rk4_stream(int rhs_kind, const double* yin, double* yout, int64_t n, double t, double dt, const Params& P,
                             int negate, const StreamTune& tune, hipStream_t s) {
  switch (rhs_kind) {
    case NNHIP_RHS_NEG_Y: return launch_mode<RhsNegY<1>>(yin, yout, n, t, dt, P, negate, tune, s);
    case NNHIP_RHS_LINEAR: return launch_mode<RhsLinear<1>>(yin, yout, n, t, dt, P, negate, tune, s);
    case NNHIP_RHS_AFFINE_T: return launch_mode<RhsAffineT<1>>(yin, yout, n, t, dt, P, negate, tune, s);
  }
  return hipErrorInvalidValue;
}

}  // namespace nnhip

// ode_rtc.hpp — internal interface of the hiprtc user-RHS module (ode_rtc.hip).
#pragma once
#include "ode_kernels.hpp"
#include "quad_kernels.hpp"

namespace nnhip {
const char* rtc_last_error();
int rtc_register(const char* name, int dim, int n_params, const char* body, bool per_component, bool check_compiles);  // -> rhs_kind or -1
int rtc_release(int rhs_kind);
bool rtc_info(int rhs_kind, int* dim, int* n_params);
hipError_t rtc_launch_solve(int rhs_kind, int integrator, const SolveArgs& a, hipStream_t s);
hipError_t rtc_launch_step(int rhs_kind, int integrator, const StepArgs& a, int negate, hipStream_t s);
hipError_t rtc_launch_advance(int rhs_kind, int integrator, const StepArgs& a, hipStream_t s);  // adaptive streaming (thread-per-IVP kinds)
hipError_t rtc_launch_advance_dense(int rhs_kind, int integrator, const StepArgs& a, hipStream_t s);  // ... with dense output (any user kind)
bool rtc_is_thread_per_ivp(int rhs_kind);
hipError_t rtc_launch_rhs(int rhs_kind, int64_t N, int64_t is, int64_t cs, double t, const double* y, double* dy, const Params& P,
                          hipStream_t s);
// size-generic built-in right-hand sides (NEG_Y, LINEAR, AFFINE_T, RING) at a size without an ahead-of-time kernel: the user
// rhs_kind of their run-time instantiation (registered on first request), or -1
bool rtc_builtin_available(int rhs_kind, int dim);
int rtc_builtin_kind(int rhs_kind, int dim);
// cumtrapz (rule 0) / cumsimpson (rule 1) of a user integrand f(x) := rhs(x, 0, params)
hipError_t rtc_launch_quad(int rhs_kind, int rule, const QuadArgs& a, hipStream_t s);
}  // namespace nnhip

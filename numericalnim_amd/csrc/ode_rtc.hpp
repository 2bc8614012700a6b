// ode_rtc.hpp — internal interface of the hiprtc user-RHS module (ode_rtc.hip).
#pragma once
#include <memory>

#include "ode_kernels.hpp"
#include "quad_kernels.hpp"

namespace nnhip {
const char* rtc_last_error();
// context layout of a user right-hand side (NumContext beyond eight scalars): named vectors, shared by the batch or one per IVP, and
// per-IVP mutable slots
struct RtcCtxLayout {
  int n_vectors;
  const char* const* names;
  const int64_t* lens;
  const int* per_ivp;
  int n_aux;
};
int rtc_register(const char* name, int dim, int n_params, const char* body, bool per_component, bool check_compiles, const RtcCtxLayout* ctx = nullptr);  // -> rhs_kind or -1
int rtc_bind_ctx(int rhs_kind, const double* shared, int64_t shared_len, const double* per_ivp, int64_t per_ivp_rows, double* aux, int n_aux, int64_t stride);
int rtc_ctx_fill(int rhs_kind, int64_t N, Params& P, int* n_scalars_in_block);
int rtc_bind_ctx_host(int rhs_kind, const double* shared, int64_t shared_len, const double* per_ivp, int64_t per_ivp_rows, const double* aux_init, int n_aux,
                      int64_t stride, int device);  // uploads into device memory the library owns
int rtc_read_aux(int rhs_kind, double* aux_out);
void rtc_drop_owned_ctx(int rhs_kind);
bool rtc_has_per_ivp_ctx(int rhs_kind);
// a host-bound context block cut into column ranges, one per device (the multi-GPU entries); see ode_rtc.hip
int rtc_ctx_shards_prepare(int rhs_kind, int n_shards, const int* devices, const int64_t* lo, const int64_t* n, std::shared_ptr<void>* handle);
int rtc_ctx_shard_enter(int rhs_kind, const std::shared_ptr<void>& handle, int r);
void rtc_ctx_shard_leave(int rhs_kind);
int rtc_ctx_shards_collect(const std::shared_ptr<void>& handle);
int rtc_set_halo(int rhs_kind, int lo, int hi);  // per-component body reads components c - lo .. c + hi only (banded form: DPP instead of the LDS stage vector); 0 / -1
const char* rtc_compiler_origin();  // the libhiprtc in use (path, version), see rtc_api() in ode_rtc.hip
bool rtc_has_aux(int rhs_kind);  // the right-hand side mutates per-IVP slots: the number and order of its evaluations are observable
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

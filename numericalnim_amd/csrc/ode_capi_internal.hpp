// ode_capi_internal.hpp — what the translation units of the C ABI share (ode_capi.hip: library, options, dispatch, fused solve;
// ode_capi_calls.hip: per-IVP calls and the binned solves; ode_capi_stream.hip: the step-streaming entries).  Not part of the product's interface: include/nnhip_ode.h is.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <vector>

#include "ode_kernels.hpp"
#include "ode_rtc.hpp"
#include "solve_plan.hpp"

namespace nnhip_capi {

int fail(int code, const char* fmt, ...);  // sets the calling thread's nnhip_last_error() text, returns `code`

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return fail(NNHIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                      __FILE__, __LINE__);                                              \
  } while (0)

struct MethodInfo {
  const char* name;
  int useFSAL;
  double order;
  int adaptive;
  int implemented;
};
// solveODE's dispatch (ode.nim:607-649), indexed by nnhip_integrator
extern const MethodInfo kMethods[NNHIP_N_INTEGRATORS];

// tuning knobs (nnhip_tune_set; defined and documented in ode_capi.hip)
// Tuning knobs (nnhip_tune_set) are process-wide and may be set by one host thread while others are inside the library: atomics, each read once per call.
extern std::atomic<int> g_stream_graph, g_fixed_vec_ipl, g_adv_nt, g_adv_refsal, g_adv_block, g_adv_steps, g_adv_split, g_adv_lean, g_adv_auto_poll;
extern std::atomic<int> g_tune_vec, g_tune_mode, g_tune_blocks_per_cu;
extern std::atomic<bool> g_tune_auto;
// Workspace-size queries: N IVPs x `per` doubles each beyond 2^44 doubles (128 TiB) is no batch — the size functions answer 0 ("invalid", as for N < 0) instead of
// overflowing their arithmetic; the compute entries then refuse the call (workspace missing / too small).
inline bool batch_size_sane(int64_t N, int64_t per) { return N >= 0 && per >= 0 && (per == 0 || N <= ((int64_t)1 << 44) / per); }
// ONE read of a knob whose values below `lowest` mean "automatic"
inline int knob_or(const std::atomic<int>& k, int lowest, int automatic) { const int v = k.load(std::memory_order_relaxed); return v >= lowest ? v : automatic; }
inline nnhip::StreamTune tune_snapshot() { nnhip::StreamTune t; t.vec = g_tune_vec; t.mode = g_tune_mode; t.blocksPerCU = g_tune_blocks_per_cu; return t; }

nnhip::StepLaunchFn find_step(int integrator, int rhs_kind, int dim);
nnhip::StepLaunchFn find_advance(int integrator, int rhs_kind, int dim);
nnhip::StepLaunchFn find_advance_lean_contracted(int integrator, int rhs_kind, int dim);
nnhip::FixedVecLaunchFn find_fixed_vec(int integrator, int rhs_kind, int dim);
nnhip::DenseAdvLaunch find_advance_dense(int integrator, int rhs_kind, int dim);
bool elementwise_rhs(int k);
int check_common(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params, int64_t N, int dim, int layout,
                 nnhip::Params& P);

// Nim system.min/max (`if x <= y: x else: y`), host copy for the time loop

// TimeGrid, make_grid, plan_solve, ctl_of, nmin_h / nmax_h: solve_plan.hpp (plain C++, shared with the CPU test harness)

// pinned staging for the (tiny) requested-time arrays of the device-pointer entries
struct Staging {
  double* host = nullptr;
  size_t cap = 0;
  hipEvent_t ev = nullptr;
  bool pending = false;
};
extern thread_local Staging g_stage;
int stage_reserve(size_t n);

// host-pointer solves borrow their streams and events from a process-wide pool keyed by device (ode_capi.hip)
struct HostSolveCtx {
  int device = -1;
  bool busy = false;
  hipStream_t s[2] = {nullptr, nullptr};
  hipEvent_t evPrep = nullptr;
  std::vector<hipEvent_t> evs;  // timing events, grown on demand
};
int host_ctx_acquire(int device, int nEvents, HostSolveCtx** out);
void host_ctx_release(HostSolveCtx* c);

// the fused solve, prepared once and launched over index ranges of the batch (ode_capi.hip)
struct PreparedSolve {
  nnhip::SolveArgs a{};       // arguments for the FULL batch
  nnhip::SolveLaunchFn fn = nullptr;
  bool user = false;
  int integrator = 0, rhs_kind = 0;
};
int prepare_solve(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params, const double* per_ivp_params, int n_per_ivp,
                  const double* y0, int64_t N, int dim, int layout, const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                  int64_t* rejected_out, int64_t max_steps, void* ws, int64_t ws_bytes, unsigned long long* agg, int* n_t_out, hipStream_t stream, PreparedSolve& ps);
int launch_solve_range(const PreparedSolve& ps, int64_t lo, int64_t n, hipStream_t stream);
// knobs of the binned solves (defined and documented in ode_capi.hip)
extern std::atomic<int> g_sort_copy, g_sort_auto_key, g_calls_bin, g_sort_rebin_steps, g_sort_resume;
extern std::atomic<double> g_sort_min_spread;
// hipGraph caches and polling blocks of the streaming loops (ode_capi_stream.hip), released by nnhip_release() / knob changes
void release_stream_graphs();
void release_adv_graphs();

}  // namespace nnhip_capi

namespace nnhip {
// ode_capi_aux.hip
hipError_t launch_hermite(double x, double x1, double x2, const double* y1, const double* y2, const double* dy1, const double* dy2, double* out, int64_t n,
                          int negate_dy, hipStream_t s);
hipError_t launch_fill_f64(double* p, int64_t n, double v, hipStream_t s);
bool launch_dense_rows_kind(int rhs_kind, int dim, int64_t N, int64_t is, int64_t cs, double tA, double tB, int neg, const double* yA, const double* yB,
                            const DenseRows& r, const Params& P, hipStream_t s, hipError_t* err);
// ode_sort.hip
int64_t argsort_workspace_bytes(int64_t N);
hipError_t argsort_f64(const double* keys, int64_t N, uint32_t* perm_out, void* ws, int64_t ws_bytes, hipStream_t s, double min_spread_on_device = 0.0);
hipError_t span_key_f64(const double* tEnd, const double* tStart, double t0, double* out, int64_t N, hipStream_t s);
hipError_t remaining_key_f64(const double* t, const double* dt, double tEnd, double* out, int64_t N, hipStream_t s);
hipError_t negate_f64(const double* in, double* out, int64_t N, hipStream_t s);
hipError_t key_range_f64(const double* keys, int64_t N, void* ws, unsigned long long* pinned2, hipStream_t s);
void key_range_decode(const unsigned long long* img, double* mn, double* mx);
hipError_t gather_f64(const double* src, double* dst, const uint32_t* perm, int64_t N, int R, int W, hipStream_t s);
hipError_t scatter_f64(const double* src, double* dst, const uint32_t* perm, int64_t N, int R, int W, hipStream_t s);
hipError_t scatter_i32(const int32_t* src, int32_t* dst, const uint32_t* perm, int64_t N, hipStream_t s);
hipError_t scatter_i64(const int64_t* src, int64_t* dst, const uint32_t* perm, int64_t N, hipStream_t s);
hipError_t gather_i32(const int32_t* src, int32_t* dst, const uint32_t* perm, int64_t N, hipStream_t s);
hipError_t gather_i64(const int64_t* src, int64_t* dst, const uint32_t* perm, int64_t N, hipStream_t s);
hipError_t invert_perm(const uint32_t* perm, uint32_t* inv, int64_t N, hipStream_t s);
hipError_t prepare_tspans(const double* tspans, int n_t, int64_t N, const double* tStart, double t0, double* grid, int32_t* counts, double* t_out,
                          hipStream_t s, double* span_key = nullptr);
void multigpu_release();  // ode_multigpu.hip
int solve_host_range(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                     const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t NFull, int64_t lo0, int64_t N, int dim,
                     int layout, const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                     int64_t* rejected_out, int64_t max_steps, nnhip_ode_stats* stats, int device);
}

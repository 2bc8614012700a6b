// ode_capi.hip — the extern "C" boundary (include/nnhip_ode.h), part 1: library and option handling, dispatch, right-hand-side
// registration and the FUSED solve entries (device- and host-pointer forms).  Part 2 — the step-streaming entries of the IntegratorProc
// seam — is ode_capi_stream.hip; what the two share is declared in ode_capi_internal.hpp.  No CPU compute fallback exists here: every
// compute entry needs a HIP device and fails with NNHIP_EHIP otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <deque>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "ode_capi_internal.hpp"

namespace nnhip_fast {  // ode_tu_method.hip compiled with -ffp-contract=fast -DNNHIP_NS=nnhip_fast (Makefile)
nnhip_abi::SolveLaunchFn find_solve_rk4(int rhs_kind, int dim, int dim16_variant);
nnhip_abi::SolveLaunchFn find_solve_dopri54(int rhs_kind, int dim, int dim16_variant);
nnhip_abi::SolveLaunchFn find_solve_tsit54(int rhs_kind, int dim, int dim16_variant);
nnhip_abi::SolveLaunchFn find_solve_vern65(int rhs_kind, int dim, int dim16_variant);
nnhip_abi::StepLaunchFn find_advance_lean_dopri54(int rhs_kind, int dim);  // ode_tu_lean_fast.hip
nnhip_abi::StepLaunchFn find_advance_lean_tsit54(int rhs_kind, int dim);
}  // namespace nnhip_fast

namespace nnhip_capi {

thread_local char g_err[8192] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}



// solveODE's dispatch (ode.nim:607-649), indexed by nnhip_integrator
extern const MethodInfo kMethods[NNHIP_N_INTEGRATORS] = {
    {"rk4", 0, 4.0, 0, 1},      {"dopri54", 1, 5.0, 1, 1},  {"tsit54", 1, 5.0, 1, 1},   {"vern65", 1, 6.0, 1, 1},
    {"bs32", 1, 3.0, 1, 1},     {"rk21", 0, 2.0, 1, 1},     {"heun2", 0, 2.0, 0, 1},    {"ralston2", 0, 2.0, 0, 1},
    {"kutta3", 0, 3.0, 0, 1},   {"heun3", 0, 3.0, 0, 1},    {"ralston3", 0, 3.0, 0, 1}, {"ssprk3", 0, 3.0, 0, 1},
    {"ralston4", 0, 4.0, 0, 1}, {"kutta4", 0, 4.0, 0, 1},
};

std::atomic<int> g_host_chunks{0};    // tuning knob "host_chunks": 0 = auto (8 when the caller's buffers are page-locked, else 1; see nnhip_ode_solve_batch_f64)
std::atomic<int> g_host_register{0};  // tuning knob "host_register": page-lock the caller's buffers for the duration of a host-pointer solve
std::atomic<int> g_fast_math{0};     // tuning knob "fp_contract": 1 = FMA-contracted instantiations of the fused kernels (not bit-exact)
std::atomic<int> g_stream_graph{2};  // tuning knob "stream_graph": 0 eager launches; 1 hipGraph capture + replay of the streaming loop;
                         // 2 (default) = replay only launch-bound batches, from the second identical call on
std::atomic<int> g_fixed_vec_ipl{2};  // tuning knob "fixed_vec_ipl": IVPs per lane of the vectorised fixed-step streaming kernel (0 = off, 2)
std::atomic<int> g_mg_oversubscribe{0};  // tuning knob "multi_gpu_oversubscribe": the multi-GPU host entry accepts more shards than devices (shard r on device
                             // r mod #devices) — lets a one-GPU box exercise the sharded code path (index ranges, strided copies, empty shards)
std::atomic<int> g_adv_nt{-1};        // tuning knob "adv_nontemporal": -1 = automatic (thread-per-IVP state beyond 192 MiB), 0 / 1 = force
std::atomic<int> g_adv_refsal{-1};    // tuning knob "adv_recompute_fsal": DOPRI54 / Tsit54 streaming loops re-evaluate FSAL = f(t, y) per launch instead of carrying it through
                          // HBM (16*d bytes per step less, the same bits).  -1 = automatic (on, unless the right-hand side has mutable slots: the extra
                          // evaluation would be observable), 0 = carry (the IntegratorProc signature as the reference passes it), 1 = force
std::atomic<int> g_adv_block{0};      // tuning knob "adv_block": workgroup size of the thread-per-IVP advance kernel (0 = auto: 64)
std::atomic<int> g_sort_copy{0};      // tuning knob "sort_copy": the binned solve reorders the batch physically (gather, solve, scatter) instead of following perm[] inside the
                          // kernel.  Measured and NOT the default (profiles/r03_bench_divergence.json, 1e6 Van der Pol IVPs): 1.77 ms against 1.65 ms with the
                          // order array followed in the kernel (pre-sorted by the caller: 1.52) — the five extra kernels and the stream-ordered allocation cost
                          // more than the scattered first load and last stores they remove
std::atomic<int> g_adv_steps{1};      // tuning knob "adv_steps_per_launch": loop iterations per IVP and launch of nnhip_ode_adaptive_stream_f64_dev (1 = the IntegratorProc
                          // seam proper; K > 1 keeps the state in registers for K iterations: 8*(4d+5)/K bytes per attempted step, a different traffic model)
std::atomic<int> g_sort_auto_key{1};   // tuning knob "sort_auto_key": what the automatic binned solve ranks by — 0 the probe's progress, 1 the steps still to take (tEnd - t) / dt (forward spans)
std::atomic<int> g_calls_bin{1};       // tuning knob "calls_bin": the per-IVP-call solves (every IVP its own tEnd / tspan) of kCallsBinMinN calls or more integrate the longest spans first, binned by span
std::atomic<int> g_sort_rebin_steps{0};  // tuning knob "sort_rebin_steps": > 0 = the resumed automatic binned solve stops after that many further accepted steps per IVP, re-bins by the steps still to take, and finishes
std::atomic<int> g_sort_resume{0};     // tuning knob "sort_resume": the automatic binned solve continues from its probe's state instead of restarting (where the loop's state is (t, dt, y)).
                          // Built, bit-identical, measured and NOT the default (profiles/r04_bench_divergence.json): the resumed pass has to run on the per-call instantiation of
                          // the solve kernel (per-lane tStart / dt), which takes 1.61 ms for 122 steps where the lean one takes 1.46 ms for all 130: 1.70 vs 1.68 ms
std::atomic<double> g_sort_min_spread{0.05};  // tuning knob "sort_min_spread_permille": the binned solve sorts only when the keys differ by more than this fraction of their magnitude
std::atomic<int> g_adv_lean{0};       // tuning knob "adv_lean": 1 = the adaptive streaming loop runs its lean kernels (the driver's own layout as the kernel's contract) where they apply.
                          // Same bits as the general kernels.  Opt-in until an MI355X has timed them: round 4's hardware record is of the general kernels.
std::atomic<int> g_adv_auto_poll{0};  // tuning knob "adv_auto_poll": 1 = check_every <= 0 means the library's own polling schedule (adv_poll_schedule.hpp); 0 = uniform groups
                          // of 8 launches, the behaviour with a hardware record (round 4).  Opt-in for the same reason.
std::atomic<int> g_adv_split{0};      // tuning knob "adv_split": index ranges the adaptive streaming loop interleaves on separate streams (0 = auto, 1, 2, 4)
std::atomic<int> g_dim16_variant{0};  // tuning knob "dim16_variant": A/B mappings of the fused 16-component kernels (see ode_kernels.hpp)

nnhip::SolveLaunchFn find_solve(int integrator, int rhs_kind, int dim) {
  if (g_fast_math) {  // opt-in FMA-contracted build of the compute-bound fused kernels
    switch (integrator) {
      case NNHIP_RK4: return nnhip_fast::find_solve_rk4(rhs_kind, dim, g_dim16_variant);
      case NNHIP_DOPRI54: return nnhip_fast::find_solve_dopri54(rhs_kind, dim, g_dim16_variant);
      case NNHIP_TSIT54: return nnhip_fast::find_solve_tsit54(rhs_kind, dim, g_dim16_variant);
      case NNHIP_VERN65: return nnhip_fast::find_solve_vern65(rhs_kind, dim, g_dim16_variant);
    }
  }
  switch (integrator) {
#define X(id, name) \
  case id: return nnhip::find_solve_##name(rhs_kind, dim, g_dim16_variant);
    NNHIP_FOR_EACH_METHOD(X)
#undef X
  }
  return nullptr;
}
nnhip::StepLaunchFn find_step(int integrator, int rhs_kind, int dim) {
  switch (integrator) {
#define X(id, name) \
  case id: return nnhip::find_step_##name(rhs_kind, dim);
    NNHIP_FOR_EACH_METHOD(X)
#undef X
  }
  return nullptr;
}

// the FMA-contracted lean kernel of (integrator, right-hand side), or nullptr (knob "fp_contract" off, another integrator, no such instantiation)
nnhip::StepLaunchFn find_advance_lean_contracted(int integrator, int rhs_kind, int dim) {
  if (!g_fast_math) return nullptr;
  if (integrator == NNHIP_DOPRI54) return nnhip_fast::find_advance_lean_dopri54(rhs_kind, dim);
  if (integrator == NNHIP_TSIT54) return nnhip_fast::find_advance_lean_tsit54(rhs_kind, dim);
  return nullptr;
}

nnhip::StepLaunchFn find_advance(int integrator, int rhs_kind, int dim) {
  switch (integrator) {
#define X(id, name) \
  case id: return nnhip::find_advance_##name(rhs_kind, dim);
    NNHIP_FOR_EACH_METHOD(X)
#undef X
  }
  return nullptr;
}

nnhip::FixedVecLaunchFn find_fixed_vec(int integrator, int rhs_kind, int dim) {
  switch (integrator) {
#define X(id, name) \
  case id: return nnhip::find_fixed_vec_##name(rhs_kind, dim);
    NNHIP_FOR_EACH_METHOD(X)
#undef X
  }
  return nullptr;
}

nnhip::DenseAdvLaunch find_advance_dense(int integrator, int rhs_kind, int dim) {
  switch (integrator) {
#define X(id, name) \
  case id: return nnhip::find_advance_dense_##name(rhs_kind, dim);
    NNHIP_FOR_EACH_METHOD(X)
#undef X
  }
  return nnhip::DenseAdvLaunch{nullptr, nullptr};
}

bool elementwise_rhs(int k) { return k == NNHIP_RHS_NEG_Y || k == NNHIP_RHS_LINEAR || k == NNHIP_RHS_AFFINE_T; }

int check_common(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                 int64_t N, int dim, int layout, nnhip::Params& P) {
  if (!opt) return fail(NNHIP_EVALUE, "options is NULL");
  if (integrator < 0 || integrator >= NNHIP_N_INTEGRATORS) return fail(NNHIP_EINTEGRATOR, "%d is not a valid integrator", integrator);
  int userDim = 0, userParams = 0;
  const bool user = rhs_kind >= NNHIP_RHS_USER_BASE;
  if (user) {
    if (!nnhip::rtc_info(rhs_kind, &userDim, &userParams)) return fail(NNHIP_EVALUE, "unknown user rhs_kind %d", rhs_kind);
    if (dim != userDim) return fail(NNHIP_EVALUE, "user rhs_kind %d was compiled for dim %d, got %d", rhs_kind, userDim, dim);
  } else if (rhs_kind < 0 || rhs_kind >= NNHIP_N_RHS) {
    return fail(NNHIP_EVALUE, "unknown rhs_kind %d", rhs_kind);
  }
  if (N < 0) return fail(NNHIP_EVALUE, "N must be >= 0");
  if (dim < 1) return fail(NNHIP_EVALUE, "dim must be >= 1 (scalar state = dim 1)");
  if (layout != NNHIP_LAYOUT_SOA && layout != NNHIP_LAYOUT_AOS) return fail(NNHIP_EVALUE, "unknown layout %d", layout);
  if (n_params < 0 || n_params > nnhip::kMaxParams) return fail(NNHIP_EVALUE, "n_params must be in [0, %d]", nnhip::kMaxParams);
  if (n_params > 0 && !rhs_params) return fail(NNHIP_EVALUE, "rhs_params is NULL");
  static const int need[NNHIP_N_RHS] = {0, 1, 3, 1, 2, 1};
  int needed = user ? userParams : need[rhs_kind];
  // the context block of a run-time compiled right-hand side (NumContext beyond eight scalars): bound device pointers travel in P
  int inBlock = 0;
  if (nnhip::rtc_ctx_fill(rhs_kind, N, P, &inBlock) < 0) return fail(NNHIP_EVALUE, "rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  if (inBlock > 0) {
    if (n_params != 0) return fail(NNHIP_EVALUE, "rhs_kind %d keeps its %d scalars in its context block: pass n_params = 0", rhs_kind, inBlock);
    needed = 0;
  }
  if (n_params < needed) return fail(NNHIP_EVALUE, "rhs_kind %d needs %d parameters, got %d", rhs_kind, needed, n_params);
  for (int k = 0; k < nnhip::kMaxParams; ++k) P.p[k] = k < n_params ? rhs_params[k] : 0.0;
  return NNHIP_OK;
}



// ODESolver's bookkeeping before the loops (ode.nim:476-487, 510, 549, 585)

// tuning knobs of the headline streaming kernel (nnhip_tune_set); defaults = the measured best
// the headline kernel's (vec, mode, blocksPerCU): three independent knobs, read as a snapshot (tune_snapshot)
std::atomic<int> g_tune_vec{nnhip::StreamTune{}.vec}, g_tune_mode{nnhip::StreamTune{}.mode}, g_tune_blocks_per_cu{nnhip::StreamTune{}.blocksPerCU};
std::atomic<bool> g_tune_auto{true};  // pick (vec, mode) from the working-set size; any explicit nnhip_tune_set pins them

// pinned staging for the (tiny) requested-time arrays of the device-pointer solve
thread_local Staging g_stage;

bool is_page_locked(const void* p) {
  hipPointerAttribute_t a;
  const hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) { (void)hipGetLastError(); return false; }  // ordinary pageable memory is unknown to the runtime
  return a.type == hipMemoryTypeHost;
}

// `rows` segments of `width` bytes, `pitch` bytes apart in both buffers.  A few long rows go as plain 1-D copies (DMA engines, which
// overlap with kernels); hipMemcpy2DAsync is kept for many short rows.
hipError_t copy_rows(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows, hipMemcpyKind kind, hipStream_t st) {
  if (rows == 0 || width == 0) return hipSuccess;
  if (dpitch == width && spitch == width) return hipMemcpyAsync(dst, src, width * rows, kind, st);
  if (rows <= 64) {
    for (size_t r = 0; r < rows; ++r) {
      const hipError_t e = hipMemcpyAsync((char*)dst + r * dpitch, (const char*)src + r * spitch, width, kind, st);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, kind, st);
}

// Streams and events of the host-pointer solve come from a process-wide pool keyed by device: creating and destroying two streams
// and a handful of events costs ~7 ms on this platform — as much as moving C2's 240 MB over PCIe.  A call borrows a context and
// hands it back; the pool grows to the number of concurrent calls per device and is never torn down (process exit may come after
// the runtime's own teardown).
std::mutex g_host_ctx_mu;
std::deque<HostSolveCtx> g_host_ctx;  // deque: growing never moves a borrowed context
int host_ctx_acquire(int device, int nEvents, HostSolveCtx** out) {
  HostSolveCtx* c = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_host_ctx_mu);
    for (auto& x : g_host_ctx) if (x.device == device && !x.busy) { c = &x; break; }
    if (!c) { g_host_ctx.emplace_back(); c = &g_host_ctx.back(); c->device = device; }
    c->busy = true;
  }
  *out = c;  // from here on the context belongs to this call (released by the caller even when the creations below fail)
  for (hipStream_t& st : c->s) if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  if (!c->evPrep) HIP_TRY(hipEventCreateWithFlags(&c->evPrep, hipEventDisableTiming));
  while ((int)c->evs.size() < nEvents) {
    hipEvent_t e = nullptr;
    HIP_TRY(hipEventCreate(&e));
    c->evs.push_back(e);
  }
  return NNHIP_OK;
}
void host_ctx_release(HostSolveCtx* c) {
  if (!c) return;
  std::lock_guard<std::mutex> lk(g_host_ctx_mu);
  c->busy = false;
}

int stage_reserve(size_t n) {
  Staging& s = g_stage;
  if (s.pending) { HIP_TRY(hipEventSynchronize(s.ev)); s.pending = false; }
  if (!s.ev) HIP_TRY(hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
  if (n > s.cap) {
    if (s.host) HIP_TRY(hipHostFree(s.host));
    s.host = nullptr; s.cap = 0;
    HIP_TRY(hipHostMalloc((void**)&s.host, n * sizeof(double), hipHostMallocDefault));
    s.cap = n;
  }
  return NNHIP_OK;
}

}  // namespace nnhip_capi
using namespace nnhip_capi;

namespace nnhip {
// error reporting for the other translation units of the C ABI (ode_capi_quad.hip): same thread-local message buffer
int fail_msg(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace nnhip

extern "C" {

int nnhip_abi_version(void) { return NNHIP_ABI_VERSION; }

int nnhip_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(NNHIP_EHIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  return n;
}

const char* nnhip_last_error(void) { return g_err; }

const char* nnhip_build_info(void) {
  return "numericalnim-hip ODE backend; target gfx950 (CDNA4); device math -ffp-contract=off (bit-parity build); "
         "compiler " __VERSION__;
}

const char* nnhip_rtc_compiler(void) { return nnhip::rtc_compiler_origin(); }

int nnhip_release(void) {
  // calling thread: pinned staging buffer of the requested-time arrays and the hipGraph cache of the streaming loop
  Staging& st = g_stage;
  if (st.pending && st.ev) (void)hipEventSynchronize(st.ev);
  if (st.host) (void)hipHostFree(st.host);
  if (st.ev) (void)hipEventDestroy(st.ev);
  st = Staging();
  release_stream_graphs();
  release_adv_graphs();
  // process: idle stream / event contexts of the host-pointer solve, RCCL communicators
  {
    std::lock_guard<std::mutex> lk(g_host_ctx_mu);
    int prev = 0;
    const bool havePrev = hipGetDevice(&prev) == hipSuccess;
    for (auto& c : g_host_ctx) {
      if (c.busy) continue;
      if (c.device >= 0) (void)hipSetDevice(c.device);
      for (hipStream_t& x : c.s) if (x) { (void)hipStreamDestroy(x); x = nullptr; }
      if (c.evPrep) { (void)hipEventDestroy(c.evPrep); c.evPrep = nullptr; }
      for (hipEvent_t e : c.evs) (void)hipEventDestroy(e);
      c.evs.clear();
    }
    if (havePrev) (void)hipSetDevice(prev);
  }
  nnhip::multigpu_release();
  return NNHIP_OK;
}

int nnhip_host_alloc(void** out, int64_t bytes) {
  if (!out || bytes < 0) return fail(NNHIP_EVALUE, "out is NULL or bytes < 0");
  *out = nullptr;
  if (bytes == 0) return NNHIP_OK;
  const hipError_t e = hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault);
  if (e != hipSuccess) { *out = nullptr; return fail(e == hipErrorOutOfMemory ? NNHIP_ENOMEM : NNHIP_EHIP, "hipHostMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e)); }
  return NNHIP_OK;
}

int nnhip_host_free(void* p) {
  if (!p) return NNHIP_OK;
  HIP_TRY(hipHostFree(p));
  return NNHIP_OK;
}

int nnhip_tune_set(const char* key, int value) {
  if (!key) return fail(NNHIP_EVALUE, "key is NULL");
  const std::string k(key);
  // Captured graphs bake in the kernel variants the knobs select: drop this thread's caches so the new setting takes effect
  // on the next call (the caches are per thread; other threads keep theirs until they change a knob or call nnhip_release).
  release_stream_graphs();
  release_adv_graphs();
  if (k == "host_chunks") { if (value < 0 || value > 64) return fail(NNHIP_EVALUE, "host_chunks must be 0..64"); g_host_chunks = value; return NNHIP_OK; }
  if (k == "host_register") { g_host_register = value != 0; return NNHIP_OK; }
  if (k == "fp_contract") { g_fast_math = value != 0; release_adv_graphs(); return NNHIP_OK; }
  if (k == "stream_graph") { if (value < 0 || value > 2) return fail(NNHIP_EVALUE, "stream_graph must be 0, 1 or 2"); g_stream_graph = value; return NNHIP_OK; }
  if (k == "fixed_vec_ipl") { if (value != 0 && value != 2) return fail(NNHIP_EVALUE, "fixed_vec_ipl must be 0 (off) or 2"); g_fixed_vec_ipl = value; return NNHIP_OK; }
  if (k == "multi_gpu_oversubscribe") { g_mg_oversubscribe = value != 0; return NNHIP_OK; }
  if (k == "adv_recompute_fsal") { if (value < -1 || value > 1) return fail(NNHIP_EVALUE, "adv_recompute_fsal must be -1, 0 or 1"); g_adv_refsal = value; release_adv_graphs(); return NNHIP_OK; }
  if (k == "adv_nontemporal") { if (value < -1 || value > 1) return fail(NNHIP_EVALUE, "adv_nontemporal must be -1, 0 or 1"); g_adv_nt = value; return NNHIP_OK; }
  if (k == "adv_block") { if (value != 0 && value != 64 && value != 128 && value != 256) return fail(NNHIP_EVALUE, "adv_block must be 0, 64, 128 or 256"); g_adv_block = value; release_adv_graphs(); return NNHIP_OK; }
  if (k == "sort_copy") { g_sort_copy = value != 0; return NNHIP_OK; }
  if (k == "adv_steps_per_launch") { if (value < 1 || value > 1024) return fail(NNHIP_EVALUE, "adv_steps_per_launch must be 1..1024"); g_adv_steps = value; return NNHIP_OK; }
  if (k == "sort_auto_key") { if (value != 0 && value != 1) return fail(NNHIP_EVALUE, "sort_auto_key must be 0 or 1"); g_sort_auto_key = value; return NNHIP_OK; }
  if (k == "calls_bin") { if (value != 0 && value != 1) return fail(NNHIP_EVALUE, "calls_bin must be 0 or 1"); g_calls_bin = value; return NNHIP_OK; }
  if (k == "sort_rebin_steps") { if (value < 0 || value > 1000000) return fail(NNHIP_EVALUE, "sort_rebin_steps must be in 0..1000000"); g_sort_rebin_steps = value; return NNHIP_OK; }
  if (k == "sort_resume") { if (value != 0 && value != 1) return fail(NNHIP_EVALUE, "sort_resume must be 0 or 1"); g_sort_resume = value; return NNHIP_OK; }
  if (k == "sort_min_spread_permille") { if (value < 0 || value > 1000) return fail(NNHIP_EVALUE, "sort_min_spread_permille must be in 0..1000"); g_sort_min_spread = value / 1000.0; return NNHIP_OK; }
  if (k == "adv_auto_poll") { if (value != 0 && value != 1) return fail(NNHIP_EVALUE, "adv_auto_poll must be 0 or 1"); g_adv_auto_poll = value; return NNHIP_OK; }
  if (k == "adv_lean") { if (value != 0 && value != 1) return fail(NNHIP_EVALUE, "adv_lean must be 0 or 1"); g_adv_lean = value; release_adv_graphs(); return NNHIP_OK; }
  if (k == "adv_split") { if (value != 0 && value != 1 && value != 2 && value != 4) return fail(NNHIP_EVALUE, "adv_split must be 0, 1, 2 or 4"); g_adv_split = value; return NNHIP_OK; }
  if (k == "dim16_variant") { if (value < 0 || value > 4) return fail(NNHIP_EVALUE, "dim16_variant must be 0..4"); g_dim16_variant = value; return NNHIP_OK; }
  if (k == "rk4_stream_auto") { g_tune_auto = value != 0; return NNHIP_OK; }
  if (k == "rk4_stream_vec" || k == "rk4_stream_mode") g_tune_auto = false;
  if (k == "rk4_stream_vec") { if (value != 1 && value != 2 && value != 4 && value != 8) return fail(NNHIP_EVALUE, "rk4_stream_vec must be 1, 2, 4 or 8"); g_tune_vec = value; return NNHIP_OK; }
  if (k == "rk4_stream_mode") { if (value < 0 || value > 3) return fail(NNHIP_EVALUE, "rk4_stream_mode must be 0..3"); g_tune_mode = value; return NNHIP_OK; }
  if (k == "rk4_stream_blocks_per_cu") { if (value < 1 || value > 64) return fail(NNHIP_EVALUE, "rk4_stream_blocks_per_cu must be 1..64"); g_tune_blocks_per_cu = value; return NNHIP_OK; }
  return fail(NNHIP_EVALUE, "unknown tuning key %s", key);
}

int nnhip_tune_get(const char* key, int* value) {
  if (!key || !value) return fail(NNHIP_EVALUE, "key / value is NULL");
  const std::string k(key);
  static const struct { const char* name; const std::atomic<int>* knob; } kInt[] = {
      {"host_chunks", &g_host_chunks}, {"host_register", &g_host_register}, {"fp_contract", &g_fast_math}, {"stream_graph", &g_stream_graph},
      {"fixed_vec_ipl", &g_fixed_vec_ipl}, {"multi_gpu_oversubscribe", &g_mg_oversubscribe}, {"adv_recompute_fsal", &g_adv_refsal},
      {"adv_nontemporal", &g_adv_nt}, {"adv_block", &g_adv_block}, {"sort_copy", &g_sort_copy}, {"adv_steps_per_launch", &g_adv_steps},
      {"sort_auto_key", &g_sort_auto_key}, {"calls_bin", &g_calls_bin}, {"sort_rebin_steps", &g_sort_rebin_steps}, {"sort_resume", &g_sort_resume},
      {"adv_auto_poll", &g_adv_auto_poll}, {"adv_lean", &g_adv_lean}, {"adv_split", &g_adv_split}, {"dim16_variant", &g_dim16_variant},
      {"rk4_stream_vec", &g_tune_vec}, {"rk4_stream_mode", &g_tune_mode}, {"rk4_stream_blocks_per_cu", &g_tune_blocks_per_cu}};
  for (const auto& e : kInt) if (k == e.name) { *value = e.knob->load(); return NNHIP_OK; }
  if (k == "sort_min_spread_permille") { *value = (int)std::lround(g_sort_min_spread.load() * 1000.0); return NNHIP_OK; }
  if (k == "rk4_stream_auto") { *value = g_tune_auto.load() ? 1 : 0; return NNHIP_OK; }
  return fail(NNHIP_EVALUE, "unknown tuning key %s", key);
}

int nnhip_ode_new_options(nnhip_ode_options* out, double dt, double absTol, double relTol, double dtMax, double dtMin,
                          double scaleMax, double scaleMin, double tStart) {
  if (!out) return fail(NNHIP_EVALUE, "out is NULL");
  if (std::fabs(dtMax) < std::fabs(dtMin)) return fail(NNHIP_EVALUE, "dtMin must be less than dtMax");  // ode.nim:95-96
  if (std::fabs(scaleMax) < 1) return fail(NNHIP_EVALUE, "scaleMax must be bigger than 1");             // :97-98
  if (1 < std::fabs(scaleMin)) return fail(NNHIP_EVALUE, "scaleMin must be smaller than 1");            // :99-100
  out->dt = std::fabs(dt); out->absTol = std::fabs(absTol); out->relTol = std::fabs(relTol);             // :101-102
  out->dtMax = std::fabs(dtMax); out->dtMin = std::fabs(dtMin); out->scaleMax = std::fabs(scaleMax);
  out->scaleMin = std::fabs(scaleMin); out->tStart = tStart;
  return NNHIP_OK;
}

int nnhip_ode_default_options(nnhip_ode_options* out) {  // ode.nim:78-79,104
  return nnhip_ode_new_options(out, 1e-4, 1e-4, 1e-4, 1e-2, 1e-4, 4.0, 0.1, 0.0);
}

int nnhip_ode_integrator_id(const char* name) {
  if (!name) return fail(NNHIP_EINTEGRATOR, "(null) is not a valid integrator");
  std::string s(name);
  for (auto& ch : s) ch = (char)std::tolower((unsigned char)ch);  // integrator.toLower() (ode.nim:607)
  for (int i = 0; i < NNHIP_N_INTEGRATORS; ++i) if (s == kMethods[i].name) return i;
  return fail(NNHIP_EINTEGRATOR, "%s is not a valid integrator", name);  // ode.nim:651
}

const char* nnhip_ode_integrator_name(int integrator) {
  if (integrator < 0 || integrator >= NNHIP_N_INTEGRATORS) return "";
  return kMethods[integrator].name;
}

int nnhip_ode_integrator_traits(int integrator, int* use_fsal, double* order, int* adaptive) {
  if (integrator < 0 || integrator >= NNHIP_N_INTEGRATORS) return fail(NNHIP_EINTEGRATOR, "%d is not a valid integrator", integrator);
  if (use_fsal) *use_fsal = kMethods[integrator].useFSAL;
  if (order) *order = kMethods[integrator].order;
  if (adaptive) *adaptive = kMethods[integrator].adaptive;
  return NNHIP_OK;
}

int nnhip_ode_time_grid(const nnhip_ode_options* opt, const double* tspan, int n_t, double* t_out, int* n_t_out) {
  if (!opt || (!tspan && n_t > 0) || n_t < 0) return fail(NNHIP_EVALUE, "bad arguments");
  TimeGrid g;
  make_grid(opt, tspan, n_t, g);
  if (t_out) std::copy(g.tOut.begin(), g.tOut.end(), t_out);
  if (n_t_out) *n_t_out = (int)g.tOut.size();
  return NNHIP_OK;
}

int nnhip_ode_rhs_compile(const char* name, int dim, int n_params, const char* body, int* rhs_kind_out) {
  if (!body || !rhs_kind_out) return fail(NNHIP_EVALUE, "body / rhs_kind_out is NULL");
  if (dim < 1 || dim > 16) return fail(NNHIP_EVALUE, "user RHS dim must be in [1, 16]");
  if (n_params < 0 || n_params > nnhip::kMaxParams) return fail(NNHIP_EVALUE, "n_params must be in [0, %d]", nnhip::kMaxParams);
  const int k = nnhip::rtc_register(name, dim, n_params, body, false, true);
  if (k < 0) return fail(NNHIP_EVALUE, "%s", nnhip::rtc_last_error());
  *rhs_kind_out = k;
  return NNHIP_OK;
}

int nnhip_ode_rhs_compile_comp(const char* name, int dim, int n_params, const char* comp_body, int* rhs_kind_out) {
  if (!comp_body || !rhs_kind_out) return fail(NNHIP_EVALUE, "comp_body / rhs_kind_out is NULL");
  if (dim < 1 || dim > 256) return fail(NNHIP_EVALUE, "per-component user RHS: dim must be in [1, 256]");
  if (n_params < 0 || n_params > nnhip::kMaxParams) return fail(NNHIP_EVALUE, "n_params must be in [0, %d]", nnhip::kMaxParams);
  const int k = nnhip::rtc_register(name, dim, n_params, comp_body, true, true);
  if (k < 0) return fail(NNHIP_EVALUE, "%s", nnhip::rtc_last_error());
  *rhs_kind_out = k;
  return NNHIP_OK;
}

// NumContext beyond eight scalars (commonTypes.nim:4-27, ode.nim:36,599).  See include/nnhip_ode.h.
int nnhip_ode_rhs_compile_ctx(const char* name, int dim, int n_params, const char* body, int per_component, int n_vectors,
                              const char* const* vec_names, const int64_t* vec_lens, const int* vec_per_ivp, int n_aux, int* rhs_kind_out) {
  if (!body || !rhs_kind_out) return fail(NNHIP_EVALUE, "body / rhs_kind_out is NULL");
  if (per_component ? (dim < 1 || dim > 256) : (dim < 1 || dim > 16)) return fail(NNHIP_EVALUE, "user RHS dim must be in [1, %d]", per_component ? 256 : 16);
  if (n_params < 0 || n_params > (1 << 20)) return fail(NNHIP_EVALUE, "n_params must be in [0, 2^20]");
  if (n_vectors < 0 || n_vectors > 64 || (n_vectors > 0 && (!vec_names || !vec_lens || !vec_per_ivp))) return fail(NNHIP_EVALUE, "bad vector declarations");
  if (n_aux < 0 || n_aux > 1024) return fail(NNHIP_EVALUE, "n_aux must be in [0, 1024]");
  if (n_aux > 0 && per_component) return fail(NNHIP_EVALUE, "per-component bodies run once per lane of a system: mutable slots (n_aux) need a whole-vector body");
  static const char* reserved[] = {"p", "y", "dy", "t", "c", "aux", "dim", "size", "P_", "ys"};
  for (int k = 0; k < n_vectors; ++k) {
    const char* nm = vec_names[k];
    if (!nm || !*nm || !(std::isalpha((unsigned char)nm[0]) || nm[0] == '_')) return fail(NNHIP_EVALUE, "vector %d: the name must be a C identifier", k);
    for (const char* q = nm; *q; ++q) if (!(std::isalnum((unsigned char)*q) || *q == '_')) return fail(NNHIP_EVALUE, "vector %d: the name must be a C identifier", k);
    for (const char* r : reserved) if (std::strcmp(nm, r) == 0) return fail(NNHIP_EVALUE, "vector name '%s' is reserved", nm);
    for (int j = 0; j < k; ++j) if (std::strcmp(nm, vec_names[j]) == 0) return fail(NNHIP_EVALUE, "vector name '%s' is declared twice", nm);
    if (vec_lens[k] < 1) return fail(NNHIP_EVALUE, "vector '%s': length must be >= 1", nm);
  }
  nnhip::RtcCtxLayout lay{n_vectors, vec_names, vec_lens, vec_per_ivp, n_aux};
  const int k = nnhip::rtc_register(name, dim, n_params, body, per_component != 0, true, &lay);
  if (k < 0) return fail(NNHIP_EVALUE, "%s", nnhip::rtc_last_error());
  *rhs_kind_out = k;
  return NNHIP_OK;
}

int nnhip_ode_rhs_bind_ctx_f64_dev(int rhs_kind, const double* shared, int64_t shared_len, const double* per_ivp, int64_t per_ivp_rows,
                                   double* aux, int n_aux, int64_t stride) {
  if (nnhip::rtc_bind_ctx(rhs_kind, shared, shared_len, per_ivp, per_ivp_rows, aux, n_aux, stride) != 0)
    return fail(NNHIP_EVALUE, "rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  nnhip::rtc_drop_owned_ctx(rhs_kind);
  return NNHIP_OK;
}

int nnhip_ode_rhs_bind_ctx_f64(int rhs_kind, const double* shared, int64_t shared_len, const double* per_ivp, int64_t per_ivp_rows,
                               const double* aux_init, int n_aux, int64_t stride, int device) {
  if (nnhip::rtc_bind_ctx_host(rhs_kind, shared, shared_len, per_ivp, per_ivp_rows, aux_init, n_aux, stride, device) != 0)
    return fail(NNHIP_EVALUE, "rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  return NNHIP_OK;
}

int nnhip_ode_rhs_read_aux_f64(int rhs_kind, double* aux_out) {
  if (nnhip::rtc_read_aux(rhs_kind, aux_out) != 0) return fail(NNHIP_EVALUE, "rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  return NNHIP_OK;
}

int nnhip_ode_rhs_set_halo(int rhs_kind, int lo, int hi) {
  if (nnhip::rtc_set_halo(rhs_kind, lo, hi) != 0) return fail(NNHIP_EVALUE, "rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  release_adv_graphs();  // captured launches of the previous code objects
  return NNHIP_OK;
}

int nnhip_ode_rhs_release(int rhs_kind) {
  return nnhip::rtc_release(rhs_kind) == 0 ? NNHIP_OK : fail(NNHIP_EVALUE, "unknown user rhs_kind %d", rhs_kind);
}

int nnhip_ode_supported(int integrator, int rhs_kind, int dim, int layout, int mode) {
  if (integrator >= 0 && integrator < NNHIP_N_INTEGRATORS && rhs_kind >= NNHIP_RHS_USER_BASE) {
    int d = 0;
    return nnhip::rtc_info(rhs_kind, &d, nullptr) && d == dim;
  }
  if (integrator < 0 || integrator >= NNHIP_N_INTEGRATORS || rhs_kind < 0 || rhs_kind >= NNHIP_N_RHS) return 0;
  if (layout != NNHIP_LAYOUT_SOA && layout != NNHIP_LAYOUT_AOS) return 0;
  if (mode == 0) return find_solve(integrator, rhs_kind, dim) != nullptr || nnhip::rtc_builtin_available(rhs_kind, dim);
  if (mode == 1) {
    if (find_step(integrator, rhs_kind, dim) || nnhip::rtc_builtin_available(rhs_kind, dim)) return 1;
    return (!kMethods[integrator].adaptive && integrator == NNHIP_RK4 && elementwise_rhs(rhs_kind)) ? 1 : 0;
  }
  return 0;
}

int64_t nnhip_ode_solve_workspace_bytes(int n_t) {
  const int64_t n = n_t < 0 ? 0 : n_t;  // requested times + the fixed-step emission schedule (4 weights and a step index per row)
  return (6 * n + 8) * (int64_t)sizeof(double) + 8 * (int64_t)sizeof(unsigned long long);
}

// ---- fused solve ---------------------------------------------------------------------------------
}  // extern "C"
namespace nnhip_capi {

// Everything of solveODE / ODESolver that precedes the per-IVP loops: validation, time grid, dispatch (ode.nim:589-651, 476-510).
int prepare_solve(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                         const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim, int layout, const double* tspan, int n_t, double* t_out,
                         double* y_out, int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps, void* ws,
                         int64_t ws_bytes, unsigned long long* agg, int* n_t_out, hipStream_t stream, PreparedSolve& ps) {
  nnhip::Params P;
  int rc = check_common(opt, integrator, rhs_kind, rhs_params, n_params, N, dim, layout, P);
  if (rc) return rc;
  if (n_t < 0 || (n_t > 0 && !tspan)) return fail(NNHIP_EVALUE, "bad tspan");
  // Non-finite times make the reference's `while t < tEnd` loop spin forever (t += dt never reaches +inf; NaN breaks
  // tspan.sorted()).  On a GPU that is a hung device, so they are refused (deviation).
  for (int j = 0; j < n_t; ++j) if (!std::isfinite(tspan[j])) return fail(NNHIP_EVALUE, "tspan[%d] is not finite", j);
  if (!std::isfinite(opt->tStart)) return fail(NNHIP_EVALUE, "options.tStart is not finite");
  if (N > 0 && (!y0 || (!y_out && n_t > 0))) return fail(NNHIP_EVALUE, "y0 / y_out is NULL");
  if (!kMethods[integrator].implemented) return fail(NNHIP_EUNSUPPORTED, "integrator %s has no HIP kernel yet", kMethods[integrator].name);
  ps.user = rhs_kind >= NNHIP_RHS_USER_BASE;
  ps.integrator = integrator;
  ps.rhs_kind = rhs_kind;
  ps.fn = ps.user ? nullptr : find_solve(integrator, rhs_kind, dim);
  if (!ps.fn && !ps.user) {  // a built-in right-hand side at a size without an ahead-of-time kernel
    const int k = nnhip::rtc_builtin_kind(rhs_kind, dim);
    if (k >= 0) { ps.user = true; ps.rhs_kind = k; }
  }
  if (!ps.fn && !ps.user) return fail(NNHIP_EUNSUPPORTED, "no fused-solve kernel for integrator=%s rhs_kind=%d dim=%d", kMethods[integrator].name, rhs_kind, dim);
  const bool adaptive = kMethods[integrator].adaptive;
  // Deviations from the reference that keep the device from spinning forever (documented in DESIGN.md):
  if (!adaptive && !(opt->dt > 0.0)) return fail(NNHIP_EVALUE, "fixed-step integrators need options.dt > 0 (the reference would loop forever)");
  if (adaptive && !(opt->dtMin > 0.0) && max_steps <= 0) return fail(NNHIP_EVALUE, "adaptive integrators need options.dtMin > 0 or max_steps > 0");

  nnhip::SolveArgs& a = ps.a;
  a.y0 = y0; a.y_out = y_out; a.ny_out = ny_out; a.steps_out = steps_out; a.rejected_out = rejected_out; a.agg = agg;
  a.N = N;
  if (layout == NNHIP_LAYOUT_SOA) { a.ivpStride = 1; a.compStride = N; } else { a.ivpStride = dim; a.compStride = 1; }
  a.rowStride = (int64_t)dim * N;
  a.P = P;
  if (n_per_ivp < 0 || n_per_ivp > nnhip::kMaxParams || (n_per_ivp > 0 && !per_ivp_params && N > 0)) return fail(NNHIP_EVALUE, "bad per-IVP parameter table");
  a.perIvpParams = n_per_ivp > 0 ? per_ivp_params : nullptr;
  a.nPerIvp = n_per_ivp;
  a.perIvpStride = N;
  // everything of the launch record that follows from (options, tspan, integrator) alone — time grid, first step size and, for fixed-step methods, the
  // host-replayed step and emission schedule: solve_plan.hpp (plain C++; also what the CPU test suite feeds the kernel bodies with)
  TimeGrid g;
  std::vector<double> emitW[2];
  std::vector<int64_t> emitStep[2];
  plan_solve(opt, adaptive, tspan, n_t, max_steps, a, g, emitW, emitStep);
  if (t_out) std::copy(g.tOut.begin(), g.tOut.end(), t_out);
  if (n_t_out) *n_t_out = (int)g.tOut.size();
  a.tPos = nullptr; a.tNeg = nullptr;
  if (N > 0 && a.useDense && (a.nPos + a.nNeg) > 0) {
    const size_t n = (size_t)a.nPos + (size_t)a.nNeg;
    const size_t nE = (size_t)a.nEmit[0] + (size_t)a.nEmit[1];
    const size_t words = n + 5 * nE;  // [tPos][tNeg][weights fwd][weights bwd][step indices fwd][step indices bwd]
    if (!ws || ws_bytes < (int64_t)(words * sizeof(double))) return fail(NNHIP_EVALUE, "workspace too small: need %zu bytes (nnhip_ode_solve_workspace_bytes)", words * sizeof(double));
    rc = stage_reserve(words);
    if (rc) return rc;
    std::copy(g.tPos.begin(), g.tPos.end(), g_stage.host);
    std::copy(g.tNeg.begin(), g.tNeg.end(), g_stage.host + a.nPos);
    double* hw = g_stage.host + n;
    std::copy(emitW[0].begin(), emitW[0].end(), hw);
    std::copy(emitW[1].begin(), emitW[1].end(), hw + 4 * (size_t)a.nEmit[0]);
    static_assert(sizeof(int64_t) == sizeof(double), "step indices share the staging buffer");
    if (nE) {
      std::memcpy(hw + 4 * nE, emitStep[0].data(), sizeof(int64_t) * emitStep[0].size());
      std::memcpy(hw + 4 * nE + a.nEmit[0], emitStep[1].data(), sizeof(int64_t) * emitStep[1].size());
    }
    HIP_TRY(hipMemcpyAsync(ws, g_stage.host, words * sizeof(double), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipEventRecord(g_stage.ev, stream));
    g_stage.pending = true;
    a.tPos = (const double*)ws;
    a.tNeg = (const double*)ws + a.nPos;
    const double* dw = (const double*)ws + n;
    a.emitW[0] = dw; a.emitW[1] = dw + 4 * (size_t)a.nEmit[0];
    a.emitStep[0] = (const int64_t*)(dw + 4 * nE); a.emitStep[1] = a.emitStep[0] + a.nEmit[0];
  }
  return NNHIP_OK;
}

// Launch the fused kernel over the IVP index range [lo, lo + n) of a prepared batch (strides keep addressing the full arrays).
int launch_solve_range(const PreparedSolve& ps, int64_t lo, int64_t n, hipStream_t stream) {
  if (n <= 0) return NNHIP_OK;
  nnhip::SolveArgs a = ps.a;
  if (a.perm) {  // work items [lo, lo + n) of the integration order; every array keeps its full-batch addressing
    a.perm += lo;
  } else {
    a.y0 += lo * a.ivpStride;
    a.y_out += lo * a.ivpStride;
    if (a.ny_out) a.ny_out += lo;
    if (a.steps_out) a.steps_out += lo;
    if (a.rejected_out) a.rejected_out += lo;
    if (a.progress_out) a.progress_out += lo;
    if (a.tfinal_out) a.tfinal_out += lo;
    if (a.dtfinal_out) a.dtfinal_out += lo;
    if (a.perCall.dtInit) a.perCall.dtInit += lo;
    if (a.perIvpParams) a.perIvpParams += lo;
    if (a.P.ivp) a.P.ivp += lo;
    if (a.P.aux) a.P.aux += lo;
  }
  a.N = n;
  if (ps.user) {
    if (nnhip::rtc_launch_solve(ps.rhs_kind, ps.integrator, a, stream) != hipSuccess)
      return fail(NNHIP_EHIP, "user RHS launch failed: %s", nnhip::rtc_last_error());
    return NNHIP_OK;
  }
  if (!ps.fn) return fail(NNHIP_EUNSUPPORTED, "launch_solve_range: the batch was not prepared (no fused-solve kernel selected)");
  HIP_TRY(ps.fn(a, stream));
  return NNHIP_OK;
}
}  // namespace nnhip_capi
extern "C" {

int nnhip_ode_solve_batch_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                  int n_params, const double* y0, int64_t N, int dim, int layout, const double* tspan,
                                  int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                                  int64_t* rejected_out, int64_t max_steps, void* ws, int64_t ws_bytes, void* stream) {
  PreparedSolve ps;
  int rc = prepare_solve(opt, integrator, rhs_kind, rhs_params, n_params, nullptr, 0, y0, N, dim, layout, tspan, n_t, t_out, y_out, ny_out,
                         steps_out, rejected_out, max_steps, ws, ws_bytes, nullptr, nullptr, (hipStream_t)stream, ps);
  if (rc) return rc;
  return launch_solve_range(ps, 0, N, (hipStream_t)stream);
}

int nnhip_ode_solve_batch_sweep_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                        int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N,
                                        int dim, int layout, const double* tspan, int n_t, double* t_out, double* y_out,
                                        int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out, int64_t max_steps, void* ws,
                                        int64_t ws_bytes, void* stream) {
  PreparedSolve ps;
  int rc = prepare_solve(opt, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y0, N, dim, layout, tspan, n_t, t_out,
                         y_out, ny_out, steps_out, rejected_out, max_steps, ws, ws_bytes, nullptr, nullptr, (hipStream_t)stream, ps);
  if (rc) return rc;
  return launch_solve_range(ps, 0, N, (hipStream_t)stream);
}


int nnhip_ode_solve_batch_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                              int n_params, const double* y0, int64_t N, int dim, int layout, const double* tspan,
                              int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                              int64_t* rejected_out, int64_t max_steps, nnhip_ode_stats* stats, int device) {
  return nnhip_ode_solve_batch_sweep_f64(opt, integrator, rhs_kind, rhs_params, n_params, nullptr, 0, y0, N, dim, layout, tspan, n_t, t_out, y_out,
                                         ny_out, steps_out, rejected_out, max_steps, stats, device);
}

int nnhip_ode_solve_batch_sweep_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params,
                                    int n_params, const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim,
                                    int layout, const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out,
                                    int64_t* steps_out, int64_t* rejected_out, int64_t max_steps, nnhip_ode_stats* stats, int device) {
  return nnhip::solve_host_range(opt, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y0, N, 0, N, dim, layout, tspan, n_t,
                                 t_out, y_out, ny_out, steps_out, rejected_out, max_steps, stats, device);
}

}  // extern "C"

namespace nnhip {
void release_thread_staging() {  // worker threads of the multi-GPU entry call this before they exit
  Staging& st = g_stage;
  if (st.pending && st.ev) (void)hipEventSynchronize(st.ev);
  if (st.host) (void)hipHostFree(st.host);
  if (st.ev) (void)hipEventDestroy(st.ev);
  st = Staging();
}
const char* thread_error() { return g_err; }
bool multi_gpu_oversubscribe() { return g_mg_oversubscribe != 0; }

// The host-pointer solve over the IVP index range [lo, lo + N) of the caller's arrays, which hold NFull IVPs (SoA planes and
// per-IVP tables have pitch NFull): host buffers in, host buffers out, on `device`.  The single-GPU entry passes (NFull, 0, NFull);
// the multi-GPU entry hands every device its shard of the SAME arrays — no staging copies in between.
int solve_host_range(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                     const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t NFull, int64_t lo0, int64_t N, int dim,
                     int layout, const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                     int64_t* rejected_out, int64_t max_steps, nnhip_ode_stats* stats, int device) {
  if (N < 0 || NFull < N || lo0 < 0 || lo0 + N > NFull || dim < 1 || n_t < 0) return fail(NNHIP_EVALUE, "bad sizes");
  if (n_per_ivp < 0 || n_per_ivp > nnhip::kMaxParams || (n_per_ivp > 0 && !per_ivp_params && N > 0)) return fail(NNHIP_EVALUE, "bad per-IVP parameter table");
  if (!opt || (n_t > 0 && !tspan)) return fail(NNHIP_EVALUE, "options / tspan is NULL");
  if (N > 0 && (!y0 || (n_t > 0 && !y_out))) return fail(NNHIP_EVALUE, "y0 / y_out is NULL");
  for (int j = 0; j < n_t; ++j) if (!std::isfinite(tspan[j])) return fail(NNHIP_EVALUE, "tspan[%d] is not finite", j);
  if (!std::isfinite(opt->tStart)) return fail(NNHIP_EVALUE, "options.tStart is not finite");
  int ndev = nnhip_device_count();
  if (ndev < 0) return ndev;
  if (ndev == 0) return fail(NNHIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(NNHIP_EVALUE, "device %d out of range [0,%d)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  // Host buffers in, host buffers out.  The batch can be cut into chunks of the IVP index range that flow through two
  // streams (H2D of chunk c+1 overlaps the kernel and D2H of chunk c) and the user buffers can be page-locked for the
  // duration of the call (tuning knobs "host_chunks", "host_register"); see the measurement below for the defaults.
  const size_t nState = (size_t)N * dim, nOut = nState * (size_t)n_t;
  double *d_y0 = nullptr, *d_out = nullptr;
  int32_t* d_ny = nullptr;
  int64_t *d_steps = nullptr, *d_rej = nullptr;
  void* d_ws = nullptr;
  double* d_per = nullptr;  // per-IVP parameter table [n_per_ivp][N]
  unsigned long long* d_agg = nullptr;
  hipStream_t s[2] = {nullptr, nullptr};  // borrowed from the pool (host_ctx_acquire), not owned
  hipEvent_t* evs = nullptr;
  hipEvent_t evPrep = nullptr;
  HostSolveCtx* hc = nullptr;
  bool regIn = false, regOut = false;
  int rc = NNHIP_OK;
  int nTOut = 0;
  const int64_t wsBytes = nnhip_ode_solve_workspace_bytes(n_t);
  auto cleanup = [&]() {
    for (hipStream_t st : s) if (st) (void)hipStreamSynchronize(st);  // nothing may still be reading the buffers freed below
    if (regIn) (void)hipHostUnregister((void*)y0);
    if (regOut) (void)hipHostUnregister((void*)y_out);
    void* bufs[] = {d_y0, d_out, d_ny, d_steps, d_rej, d_ws, d_agg, d_per};
    for (void* b : bufs) if (b) (void)hipFree(b);
    host_ctx_release(hc);
    hc = nullptr;
  };
#define HIP_TRY_C(expr)                                                                                                   \
  do {                                                                                                                    \
    hipError_t _e = (expr);                                                                                               \
    if (_e != hipSuccess) { rc = fail(_e == hipErrorOutOfMemory ? NNHIP_ENOMEM : NNHIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(_e)); cleanup(); return rc; } \
  } while (0)
  // Measured on MI355X / EPYC 9575F (scripts/ab_host_pinned.py, C2 fused RK4, 80 MB in + 160 MB out, caller's buffers reused):
  // pageable buffers 8.9 ms whatever the chunk count (their copies block the host thread, nothing overlaps); page-locked buffers
  // 8.9 ms in one chunk, 6.6 ms in 4 and 5.5 ms in 8 chunks (copy engines overlap the kernel).  Page-locking pageable buffers for the
  // duration of the call costs more than it gains ("host_register", off).  Hence: automatic = 8 chunks when both buffers are
  // already page-locked (hipHostMalloc / hipHostRegister by the caller), else 1.  A wrapper that allocates a fresh result array
  // per call pays ~8-13 ms of first-touch page faults inside the copy on top of this.
  const int hostChunks = g_host_chunks;
  int nChunks = hostChunks > 0 ? hostChunks : 1;
  if (hostChunks == 0 && (nState + nOut) * sizeof(double) >= ((size_t)32 << 20) && is_page_locked(y0) && is_page_locked(y_out)) nChunks = 8;
  if ((int64_t)nChunks > N) nChunks = (int)std::max<int64_t>(1, N);
  rc = host_ctx_acquire(device, nChunks * 2, &hc);
  if (rc) { host_ctx_release(hc); return rc; }
  s[0] = hc->s[0]; s[1] = hc->s[1]; evPrep = hc->evPrep; evs = hc->evs.data();
  if (nState) HIP_TRY_C(hipMalloc((void**)&d_y0, nState * sizeof(double)));
  if (nOut) HIP_TRY_C(hipMalloc((void**)&d_out, nOut * sizeof(double)));
  if (ny_out && N) HIP_TRY_C(hipMalloc((void**)&d_ny, (size_t)N * sizeof(int32_t)));
  if (steps_out && N) HIP_TRY_C(hipMalloc((void**)&d_steps, (size_t)N * sizeof(int64_t)));
  if (rejected_out && N) HIP_TRY_C(hipMalloc((void**)&d_rej, (size_t)N * sizeof(int64_t)));
  HIP_TRY_C(hipMalloc(&d_ws, (size_t)wsBytes));
  if (n_per_ivp > 0 && N > 0) {  // this range's columns of the [k][NFull] table go up front as a dense [k][N] table
    HIP_TRY_C(hipMalloc((void**)&d_per, (size_t)n_per_ivp * (size_t)N * sizeof(double)));
    HIP_TRY_C(hipMemcpy2D(d_per, (size_t)N * 8, per_ivp_params + lo0, (size_t)NFull * 8, (size_t)N * 8, (size_t)n_per_ivp, hipMemcpyHostToDevice));
  }
  HIP_TRY_C(hipMalloc((void**)&d_agg, nnhip::kAggSlots * 8 * sizeof(unsigned long long)));
  {
    std::vector<unsigned long long> init((size_t)nnhip::kAggSlots * 8, 0ull);
    for (int k = 0; k < nnhip::kAggSlots; ++k) init[(size_t)k * 8 + 3] = ~0ull;
    HIP_TRY_C(hipMemcpy(d_agg, init.data(), init.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  if (g_host_register && N == NFull) {  // best effort: a failed registration just leaves the copies staged
    regIn = nState && hipHostRegister((void*)y0, nState * sizeof(double), hipHostRegisterDefault) == hipSuccess;
    regOut = nOut && hipHostRegister((void*)y_out, nOut * sizeof(double), hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
  }
  PreparedSolve ps;
  rc = prepare_solve(opt, integrator, rhs_kind, rhs_params, n_params, d_per, d_per ? n_per_ivp : 0, d_y0, N, dim, layout, tspan, n_t, t_out, d_out, d_ny, d_steps,
                     d_rej, max_steps, d_ws, wsBytes, d_agg, &nTOut, s[0], ps);
  if (rc) { cleanup(); return rc; }
  HIP_TRY_C(hipEventRecord(evPrep, s[0]));
  HIP_TRY_C(hipStreamWaitEvent(s[1], evPrep, 0));
  const bool soa = layout == NNHIP_LAYOUT_SOA;
  for (int cI = 0; cI < nChunks && N > 0; ++cI) {
    const int64_t lo = N * cI / nChunks, hi = N * (cI + 1) / nChunks, n = hi - lo;
    if (n <= 0) continue;
    hipStream_t st = s[cI & 1];
    const int64_t hlo = lo0 + lo;  // the chunk's first IVP in the caller's arrays
    if (soa) {  // component planes: `dim` rows of n doubles; device pitch N, host pitch NFull
      HIP_TRY_C(copy_rows(d_y0 + lo, (size_t)N * 8, y0 + hlo, (size_t)NFull * 8, (size_t)n * 8, (size_t)dim, hipMemcpyHostToDevice, st));
    } else {
      HIP_TRY_C(hipMemcpyAsync(d_y0 + lo * dim, y0 + hlo * dim, (size_t)n * dim * 8, hipMemcpyHostToDevice, st));
    }
    HIP_TRY_C(hipEventRecord(evs[(size_t)cI * 2], st));
    rc = launch_solve_range(ps, lo, n, st);
    if (rc) { cleanup(); return rc; }
    HIP_TRY_C(hipEventRecord(evs[(size_t)cI * 2 + 1], st));
    if (n_t > 0) {
      if (soa) {
        HIP_TRY_C(copy_rows(y_out + hlo, (size_t)NFull * 8, d_out + lo, (size_t)N * 8, (size_t)n * 8, (size_t)n_t * dim, hipMemcpyDeviceToHost, st));
      } else {
        HIP_TRY_C(copy_rows(y_out + hlo * dim, (size_t)NFull * dim * 8, d_out + lo * dim, (size_t)N * dim * 8, (size_t)n * dim * 8, (size_t)n_t, hipMemcpyDeviceToHost, st));
      }
    }
    if (d_ny) HIP_TRY_C(hipMemcpyAsync(ny_out + hlo, d_ny + lo, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (d_steps) HIP_TRY_C(hipMemcpyAsync(steps_out + hlo, d_steps + lo, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
    if (d_rej) HIP_TRY_C(hipMemcpyAsync(rejected_out + hlo, d_rej + lo, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
  }
  HIP_TRY_C(hipStreamSynchronize(s[0]));
  HIP_TRY_C(hipStreamSynchronize(s[1]));
  if (stats) {
    std::vector<unsigned long long> agg((size_t)nnhip::kAggSlots * 8, 0ull);
    HIP_TRY_C(hipMemcpy(agg.data(), d_agg, agg.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long tot[8] = {0, 0, 0, ~0ull, 0, 0, 0, 0};
    for (int k = 0; k < nnhip::kAggSlots; ++k) {
      const unsigned long long* a8 = &agg[(size_t)k * 8];
      tot[0] += a8[0]; tot[1] += a8[1]; tot[2] = std::max(tot[2], a8[2]); tot[3] = std::min(tot[3], a8[3]); tot[4] += a8[4]; tot[5] += a8[5];
    }
    double ms = 0.0;
    for (int cI = 0; cI < nChunks && N > 0; ++cI) {
      float m1 = 0.f;
      if (hipEventElapsedTime(&m1, evs[(size_t)cI * 2], evs[(size_t)cI * 2 + 1]) == hipSuccess) ms += m1;
    }
    stats->steps_total = (int64_t)tot[0]; stats->rejected_total = (int64_t)tot[1]; stats->steps_max = (int64_t)tot[2];
    stats->n_t_out = nTOut; stats->ny_min = N ? (int32_t)tot[3] : 0; stats->nan_aborts = (int32_t)tot[4];
    stats->truncated = (int32_t)tot[5]; stats->kernel_ms = ms;
  }
  cleanup();
  return NNHIP_OK;
#undef HIP_TRY_C
}
}  // namespace nnhip


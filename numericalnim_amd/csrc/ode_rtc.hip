// ode_rtc.hip — user-supplied right-hand sides, compiled at run time with hiprtc.
//
// The reference takes an arbitrary closure f(t, y, ctx) (ODEProc[T], ode.nim:36).  A host closure cannot run on the
// device, but its *source* can: the caller hands over the body of
//     __device__ void rhs(double t, const double* y, double* dy, const double* p)      // y, dy: dim components
// and this module instantiates the SAME kernel templates the built-in RHS use (ode_kernels.hpp, embedded in the library
// as text) for it — same steppers, same driver, same -ffp-contract=off numerics — one hiprtc program per
// (rhs, integrator), cached for the life of the process.
#include <hip/hip_runtime.h>
#include <hip/hip_version.h>
#include <hip/hiprtc.h>

#include <atomic>

#include <cstdio>
#include <climits>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ode_kernels.hpp"
#include "ode_rtc.hpp"

namespace nnhip {
namespace {

// Which compiler builds the user's right-hand sides.  A host process may carry its own, older copy of hiprtc + comgr (PyTorch wheels bundle the
// ROCm they were built with: 7.0.2 / clang 20 in this image, against the system's 7.2 / clang 22 that built this library) — and the run-time
// compiled kernels are then those of that compiler: the same templates come out at 179 instead of 168 VGPRs (2 instead of 3 waves per
// SIMD) and run 1.7-2.2x slower (profiles/r03_bench_rtc_stream.json; "faster under rocprofv3" was this: the profiler's environment puts
// /opt/rocm/lib first).  So: if the system's libhiprtc is another file than the one the process resolves, it is loaded into a link namespace of its
// own (dlmopen: its comgr comes with it, nothing the process already uses is touched) and used for the compilations; only source text goes
// in and a code object comes out.  (hiprtcVersion reports the API level, 9.0 for both, so the rule is not "newer" but "the ROCm this library was
// built with, if it is installed and is not what the process already uses".)  NNHIP_HIPRTC=process keeps the process's own;
// NNHIP_HIPRTC=/path/libhiprtc.so names another.
struct RtcApi {
  hiprtcResult (*createProgram)(hiprtcProgram*, const char*, const char*, int, const char* const*, const char* const*);
  hiprtcResult (*compileProgram)(hiprtcProgram, int, const char* const*);
  hiprtcResult (*getProgramLogSize)(hiprtcProgram, size_t*);
  hiprtcResult (*getProgramLog)(hiprtcProgram, char*);
  hiprtcResult (*getCodeSize)(hiprtcProgram, size_t*);
  hiprtcResult (*getCode)(hiprtcProgram, char*);
  hiprtcResult (*addNameExpression)(hiprtcProgram, const char*);
  hiprtcResult (*getLoweredName)(hiprtcProgram, const char*, const char**);
  hiprtcResult (*destroyProgram)(hiprtcProgram*);
  std::string origin;
};
#ifndef NNHIP_BUILD_ROCM_PATH
#define NNHIP_BUILD_ROCM_PATH "/opt/rocm"  // (the Makefile bakes in the ROCm the library is built with: `hipconfig --rocmpath`)
#endif
struct RtcState {
  RtcApi process;            // the libhiprtc the process resolves (always usable)
  RtcApi priv;               // the build's ROCm in a link namespace of its own, if it was loaded
  bool havePriv = false;
  std::atomic<bool> privDisabled{false};
  std::mutex mu;
  std::string why;           // why the private one is not (or no longer) used
};
RtcState& rtc_state() {
  static RtcState st;
  static std::once_flag once;
  std::call_once(once, [] {
    RtcState& s = st;
    s.process = RtcApi{&hiprtcCreateProgram, &hiprtcCompileProgram, &hiprtcGetProgramLogSize, &hiprtcGetProgramLog, &hiprtcGetCodeSize, &hiprtcGetCode,
                       &hiprtcAddNameExpression, &hiprtcGetLoweredName, &hiprtcDestroyProgram, "the process's libhiprtc"};
    const char* pref = std::getenv("NNHIP_HIPRTC");
    if (pref && std::strcmp(pref, "process") == 0) { s.why = "NNHIP_HIPRTC=process"; return; }
    const std::string path = pref && *pref ? pref : NNHIP_BUILD_ROCM_PATH "/lib/libhiprtc.so";  // the ROCm this library was built with
    Dl_info info;
    char mine[PATH_MAX] = "", theirs[PATH_MAX] = "";
    if (!realpath(path.c_str(), theirs)) { s.why = path + " does not exist on this host"; return; }
    if (dladdr((void*)&hiprtcCreateProgram, &info) && info.dli_fname && realpath(info.dli_fname, mine)) {
      s.process.origin = std::string(mine) + " (the process's own)";
      if (std::strcmp(mine, theirs) == 0) { s.why = "the process already uses the build's libhiprtc"; return; }
    }
    // A code object from the build's compiler is loaded by the PROCESS's runtime: only within one major version of HIP (the code-object
    // ABI and the kernel-descriptor layout are stable there); across majors the process's own compiler is the safe one.
    int rt = 0;
    if (!(pref && *pref) && hipRuntimeGetVersion(&rt) == hipSuccess && rt / 10000000 != HIP_VERSION_MAJOR) {
      s.why = "the process's HIP runtime is version " + std::to_string(rt / 10000000) + ".x, this library was built with " + std::to_string(HIP_VERSION_MAJOR) + "." +
              std::to_string(HIP_VERSION_MINOR) + ": code objects stay with the process's own compiler";
      return;
    }
    void* h = dlmopen(LM_ID_NEWLM, theirs, RTLD_NOW | RTLD_LOCAL);
    if (!h) { const char* de = dlerror(); s.why = std::string("dlmopen(") + theirs + ") failed: " + (de ? de : "?"); return; }
    RtcApi b;
    b.createProgram = (decltype(b.createProgram))dlsym(h, "hiprtcCreateProgram");
    b.compileProgram = (decltype(b.compileProgram))dlsym(h, "hiprtcCompileProgram");
    b.getProgramLogSize = (decltype(b.getProgramLogSize))dlsym(h, "hiprtcGetProgramLogSize");
    b.getProgramLog = (decltype(b.getProgramLog))dlsym(h, "hiprtcGetProgramLog");
    b.getCodeSize = (decltype(b.getCodeSize))dlsym(h, "hiprtcGetCodeSize");
    b.getCode = (decltype(b.getCode))dlsym(h, "hiprtcGetCode");
    b.addNameExpression = (decltype(b.addNameExpression))dlsym(h, "hiprtcAddNameExpression");
    b.getLoweredName = (decltype(b.getLoweredName))dlsym(h, "hiprtcGetLoweredName");
    b.destroyProgram = (decltype(b.destroyProgram))dlsym(h, "hiprtcDestroyProgram");
    if (!b.createProgram || !b.compileProgram || !b.getProgramLogSize || !b.getProgramLog || !b.getCodeSize || !b.getCode || !b.addNameExpression || !b.getLoweredName ||
        !b.destroyProgram) { s.why = std::string(theirs) + " lacks a hiprtc entry point"; return; }
    b.origin = std::string(theirs) + " (in a link namespace of its own; the process's own is " + (mine[0] ? mine : "unknown") + ")";
    s.priv = b;
    s.havePriv = true;
  });
  return st;
}
bool rtc_is_private(const RtcApi& api) { return &api == &rtc_state().priv; }
// Fault injection for the fall-back paths (tests/test_gpu_hiprtc_modes.py): NNHIP_HIPRTC_INJECT=compile makes the first compilation in the
// private namespace report failure, =load makes the first module load of a privately compiled code object report failure.
bool rtc_inject(const char* what) {
  static std::atomic<bool> spent{false};  // one injected failure per process
  const char* e = std::getenv("NNHIP_HIPRTC_INJECT");
  if (!e || std::strcmp(e, what) != 0) return false;
  return !spent.exchange(true);
}
const RtcApi& rtc_api() {
  RtcState& s = rtc_state();
  return s.havePriv && !s.privDisabled.load() ? s.priv : s.process;
}
// The private compiler produced something unusable here (its comgr / device libraries are not found, or the process's runtime refuses its
// code object): from now on the process's own libhiprtc compiles, and nnhip_rtc_compiler() says why.
void rtc_disable_private(const std::string& reason) {
  RtcState& s = rtc_state();
  std::lock_guard<std::mutex> lk(s.mu);
  if (!s.privDisabled.exchange(true)) s.why = reason;
}

struct Header { const char* name; const char* text; };
const Header kHeaders[] = {
#include "embedded_headers.inc"
};

struct Program {  // the kernels of one (rhs, integrator) code object as loaded on ONE device; shared_ptr-owned: a launch in flight keeps
                   // its module alive across a concurrent nnhip_ode_rhs_release(), the module is unloaded when the last holder lets go
  int device = -1;
  hipModule_t module = nullptr;
  Program() = default;
  Program(const Program&) = delete;
  Program& operator=(const Program&) = delete;
  ~Program() {
    if (!module) return;
    int prev = 0;
    const bool have = hipGetDevice(&prev) == hipSuccess;
    if (device >= 0) (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();  // kernels of this module may still be queued
    (void)hipModuleUnload(module);
    if (have) (void)hipSetDevice(prev);
  }
  hipFunction_t solve = nullptr, stepPos = nullptr, stepNeg = nullptr, advance = nullptr, advanceDense = nullptr, rhs = nullptr, quad[2] = {nullptr, nullptr};
  int ivpsPerBlockSolve = kBlock, ivpsPerBlockStep = kBlock, ivpsPerBlockAdvance = kBlock;  // thread-per-IVP: 256; lanes-per-system: 256 / lanes per system
};
struct CodeObject {  // compiled once per (rhs, integrator); hipModuleLoadData binds it to the device that is current at the time
  std::vector<char> code;
  bool fromPrivate = false;  // compiled by the build's libhiprtc in its private link namespace (see rtc_state)
  std::vector<std::string> lowered;
  int ivpsPerBlockSolve = kBlock, ivpsPerBlockStep = kBlock, ivpsPerBlockAdvance = kBlock;
  std::map<int, std::shared_ptr<Program>> loaded;  // by device ordinal
};
// One bind call's result (nnhip_ode_rhs_bind_ctx_f64[_dev]).  Shared by the registry entry (the latest bind of any thread) and by the thread
// that made it (thread-local: two host threads solving the same source with different contexts each read their own — bind and the calls that
// follow it on the same thread cannot be torn apart by another thread's bind).  Library-owned device memory lives and dies with it.
struct CtxShard { int device = -1; int64_t lo = 0, n = 0; void* d[3] = {nullptr, nullptr, nullptr}; };
struct CtxBinding {
  const double* shared = nullptr;  // device pointers (the caller's, or `owned`)
  const double* ivp = nullptr;
  double* aux = nullptr;
  int64_t stride = 0;
  int64_t sharedLen = 0, ivpRows = 0;
  int nAux = 0;
  // bound from HOST arrays (nnhip_ode_rhs_bind_ctx_f64): the device copies the library made, and the host copies they were made from — what the
  // multi-GPU entries cut into column ranges, one per device
  int ownedDevice = -1;
  void* owned[3] = {nullptr, nullptr, nullptr};
  bool haveHost = false;
  std::vector<double> hShared, hIvp, hAux;
  std::mutex mu;                 // shards / auxInShards
  std::vector<CtxShard> shards;  // per-device column ranges [lo, lo + n) of the per-IVP rows and the mutable slots (stride n), the shared block whole
  bool auxInShards = false;      // the mutable slots were last written by a sharded solve: gathered back before anything else reads them
  void free_shards() {
    int prev = 0;
    const bool have = hipGetDevice(&prev) == hipSuccess;
    for (CtxShard& sh : shards) {
      if (sh.device < 0) continue;
      (void)hipSetDevice(sh.device);
      (void)hipDeviceSynchronize();  // launches reading the block may still be queued
      for (void*& q : sh.d) { if (q) (void)hipFree(q); q = nullptr; }
    }
    shards.clear();
    if (have) (void)hipSetDevice(prev);
  }
  ~CtxBinding() {
    free_shards();
    if (ownedDevice < 0) return;
    int prev = 0;
    const bool have = hipGetDevice(&prev) == hipSuccess;
    (void)hipSetDevice(ownedDevice);
    (void)hipDeviceSynchronize();
    for (void*& q : owned) { if (q) (void)hipFree(q); q = nullptr; }
    if (have) (void)hipSetDevice(prev);
  }
};
struct CtxVec { std::string name; int64_t len = 0; bool perIvp = false; int64_t offset = 0; };  // offset: doubles into the shared block / row of the per-IVP block
struct UserRhsEntry {
  std::string name, body;
  int dim = 0, n_params = 0;
  bool perComponent = false;  // body computes ONE component (usable by the lanes-per-system kernels) instead of the whole vector
  int haloLo = -1, haloHi = -1;  // per-component body that reads components c - haloLo .. c + haloHi (cyclically) only: neighbours come through DPP lane rotations (RhsBanded) instead of the LDS stage vector; -1: not declared
  bool alive = false;
  // context layout (nnhip_ode_rhs_compile_ctx): NumContext beyond eight scalars
  bool hasCtx = false;
  std::vector<CtxVec> vecs;
  int n_aux = 0;
  int64_t sharedLen = 0, ivpRows = 0;  // doubles of the shared block (scalars beyond kMaxParams included) / rows of the per-IVP block
  // what is bound to the layout right now: the most recent bind of any thread (a thread that bound one itself reads its own, see t_bound)
  std::shared_ptr<CtxBinding> bound;
  std::map<int, CodeObject> programs;  // by integrator; key -1 = the rhs_batch kernel only, key -2 = the cumulative-quadrature kernels,
                                       // 1000 + integrator = the dense-output adaptive streaming kernel
};

std::mutex g_mu;
constexpr int kMultiKey = 500;   // programs[kMultiKey + integrator]: the advance kernel with several loop iterations per launch (StepArgs::stepsPerLaunch > 1)
constexpr int kDenseKey = 1000;  // programs[kDenseKey + integrator]: advance_dense_*_kernel of that integrator
constexpr int kCallsKey = 2000;  // programs[kCallsKey + integrator]: the solve kernel with per-IVP call data (MODE 2)
constexpr int kGridKey = 3000;   // programs[kGridKey + integrator]: the solve kernel with per-IVP n_t-point tspans (MODE 3)
constexpr int kLeanKey = 4000;   // programs[kLeanKey + integrator]: the solve kernel without dense output (MODE 0: no Hermite history in registers; 2-point tspans)
std::deque<UserRhsEntry> g_user;  // deque: registering a new RHS never moves existing entries (programs are handed out by pointer)
thread_local std::string g_rtc_err;

// Systems of 8 or 16 components, and every system wider than 16, given per component run on the lanes-per-system kernels:
// component slots are laid out for the next power of two (32 / 64 / 128 / 256), the slots beyond the real size stay zero and
// touch no memory (RhsSize, LpsOps::owns).  Everything else is thread-per-IVP.
bool uses_lps(const UserRhsEntry& e) { return e.perComponent && (e.dim == 8 || e.dim == 16 || e.dim > 16); }
int padded_dim(const UserRhsEntry& e) {
  if (!uses_lps(e)) return e.dim;
  int p = 8;
  while (p < e.dim) p *= 2;
  return p;
}
// components per lane: fused kernels from the A/B of the built-in systems (ode_kernels.hpp NNHIP_FOR_EACH_LPS_RHS); the
// step-streaming kernels keep one component per lane until a system would no longer fit one wavefront
int lps_cpl(const UserRhsEntry& e, bool adaptive) { return padded_dim(e) == 8 ? 2 : (padded_dim(e) >= 64 ? 4 : (adaptive ? 4 : 2)); }
int lps_step_cpl(const UserRhsEntry& e) { return padded_dim(e) <= 64 ? 1 : padded_dim(e) / 64; }
// the adaptive streaming (advance) kernels: as the built-in systems' (NNHIP_ADV_CPL: the per-lane controller arithmetic is replicated over a
// system's lanes, so fewer, fatter lanes win — 4 components per lane, 2 for 8-component systems)
int lps_adv_cpl(const UserRhsEntry& e) { const int c = padded_dim(e) == 8 ? 2 : 4; return c > lps_step_cpl(e) ? c : lps_step_cpl(e); }

// what a body sees of its context: `p` (scalars, wherever they live), one name per declared vector, `aux`
std::string ctx_preamble(const UserRhsEntry& e) {
  std::string s;
  if (e.hasCtx && e.n_params > kMaxParams) s += "    const double* p = P_.shared; (void)p;\n";
  else s += "    const double* p = P_.p; (void)p;\n";
  if (!e.hasCtx) return s;
  for (const CtxVec& v : e.vecs) {
    if (v.perIvp) s += "    const nnhip_ctx::IvpVec " + v.name + "{P_.ivp + " + std::to_string(v.offset) + "LL * P_.stride, P_.stride}; (void)" + v.name + ";\n";
    else s += "    const double* " + v.name + " = P_.shared + " + std::to_string(v.offset) + "; (void)" + v.name + ";\n";
  }
  if (e.n_aux > 0) s += "    const nnhip_ctx::AuxRef aux{P_.aux, P_.stride}; (void)aux;\n";
  return s;
}

std::string make_source(const UserRhsEntry& e) {
  std::string s;
  s += "#include \"ode_kernels.hpp\"\n#include \"quad_kernels.hpp\"\n";
  s += "namespace nnhip_ctx {\n"
       "struct IvpVec { const double* b; long long s; __device__ __forceinline__ double operator()(long long j) const { return b[j * s]; }\n"
       "                __device__ __forceinline__ double operator[](long long j) const { return b[j * s]; } };\n"
       "struct AuxRef { double* b; long long s; __device__ __forceinline__ double& operator()(long long j) const { return b[j * s]; }\n"
       "                __device__ __forceinline__ double& operator[](long long j) const { return b[j * s]; } };\n}\n";
  s += "namespace nnhip {\nstruct UserRhs {\n  static constexpr int dim = " + std::to_string(padded_dim(e)) + ";\n";
  s += "  static constexpr int size = " + std::to_string(e.dim) + ";\n";
  if (e.n_aux > 0) s += "  static constexpr bool mutates = true;\n";  // the drivers then make every evaluation the reference makes (RhsMutates)
  if (!e.perComponent) {
    s += "  NNHIP_DEV static void eval(double t, const double (&y)[dim], double (&dy)[dim], const Params& P_) {\n";
    s += ctx_preamble(e) + "    (void)t;\n";
    s += "    {\n" + e.body + "\n    }\n  }\n";
    s += "  NNHIP_DEV static double comp(double t, int c, const double* ys, const Params& P_) {\n";
    s += "    double y[dim], dy[dim];\n    for (int k = 0; k < dim; ++k) y[k] = ys[k];\n    eval(t, y, dy, P_);\n";
    s += "    double r = dy[0];\n    for (int k = 1; k < dim; ++k) if (c == k) r = dy[k];\n    return r;\n  }\n};\n}\n";
  } else {
    s += "  NNHIP_DEV static double comp(double t, int c, const double* y, const Params& P_) {\n";
    s += ctx_preamble(e) + "    (void)t; (void)c;\n";
    s += "    if (c >= size) return 0.0;\n";
    s += "    constexpr int dim = size; (void)dim;  // inside the body `dim` is the real number of components\n";
    s += "    {\n" + e.body + "\n    }\n  }\n";
    s += "  NNHIP_DEV static void eval(double t, const double (&y)[dim], double (&dy)[dim], const Params& P_) {\n";
    s += "#pragma unroll\n    for (int c = 0; c < dim; ++c) dy[c] = comp(t, c, &y[0], P_);\n  }\n";
    if (e.haloLo >= 0 && e.haloHi >= 0) {
      // the SAME body over a window of the components it declared it reads: `y[j]` resolves into the window (cyclically), so the
      // expression — and its bits — are those of comp()
      const std::string lo = std::to_string(e.haloLo), hi = std::to_string(e.haloHi);
      s += "  static constexpr int halo_lo = " + lo + ", halo_hi = " + hi + ";\n";
      s += "  struct WinRef_ { const double* w; int c;\n"
           "    NNHIP_DEV double operator[](int j) const { int k = j - c; if (k < -" + lo + ") k += size; else if (k > " + hi + ") k -= size;\n"
           "      k = k < -" + lo + " ? -" + lo + " : (k > " + hi + " ? " + hi + " : k);  // outside the declared window: clamped (a wrong declaration gives wrong numbers, not a fault; the Python mirror checks it)\n"
           "      return w[k + " + lo + "]; } };\n";
      s += "  NNHIP_DEV static double comp_window(double t, int c, const double* w_, const Params& P_) {\n";
      s += ctx_preamble(e) + "    (void)t; (void)c;\n";
      s += "    constexpr int dim = size; (void)dim;\n    const WinRef_ y{w_, c};\n";
      s += "    {\n" + e.body + "\n    }\n  }\n";
    }
    s += "};\n}\n";
  }
  return s;
}

bool compile_with(const RtcApi& api, const UserRhsEntry& e, int integrator, CodeObject& out) {
  const std::string src = make_source(e);
  hiprtcProgram prog;
  std::vector<const char*> hsrc, hname;
  for (const Header& h : kHeaders) { hsrc.push_back(h.text); hname.push_back(h.name); }
  if (api.createProgram(&prog, src.c_str(), "nnhip_user_rhs.hip", (int)hsrc.size(), hsrc.data(), hname.data()) != HIPRTC_SUCCESS) {
    g_rtc_err = "hiprtcCreateProgram failed";
    return false;
  }
  std::vector<std::string> names;
  if (integrator >= kCallsKey) {  // the fused solve with per-IVP tspan / options (nnhip_ode_solve_batch_calls_f64_dev / _tspans_): its own code object
    const bool lean = integrator >= kLeanKey, grid = !lean && integrator >= kGridKey;
    const int method = integrator - (lean ? kLeanKey : (grid ? kGridKey : kCallsKey));
    const std::string m = std::to_string(method), mode = lean ? "0" : (grid ? "3" : "2");
    if (uses_lps(e)) {
      int adaptive = 0;
      nnhip_ode_integrator_traits(method, nullptr, nullptr, &adaptive);
      const int cpl = lps_cpl(e, adaptive != 0);
      names.push_back("nnhip::solve_lps_kernel<" + m + ", nnhip::UserRhs, " + std::to_string(cpl) + ", false, " + mode + ">");
      out.ivpsPerBlockSolve = kBlock / (padded_dim(e) / cpl);
    } else {
      names.push_back("nnhip::solve_tpi_kernel<" + m + ", nnhip::UserRhs, " + mode + ">");
    }
  } else if (integrator >= kDenseKey) {  // the dense-output form of the adaptive streaming kernel: its own code object, compiled when first asked for
    const std::string m = std::to_string(integrator - kDenseKey);
    if (uses_lps(e)) {
      const int scpl = lps_step_cpl(e);
      names.push_back("nnhip::advance_dense_lps_kernel<" + m + ", nnhip::UserRhs, " + std::to_string(scpl) + ">");
      out.ivpsPerBlockAdvance = kBlock / (padded_dim(e) / scpl);
    } else {
      names.push_back("nnhip::advance_dense_tpi_kernel<" + m + ", nnhip::UserRhs>");
    }
  } else if (integrator >= kMultiKey) {  // K loop iterations per launch: its own code object, compiled when first asked for
    const std::string m = std::to_string(integrator - kMultiKey);
    if (uses_lps(e)) {
      const int acpl = lps_adv_cpl(e);
      names.push_back("nnhip::advance_lps_kernel<" + m + ", nnhip::UserRhs, " + std::to_string(acpl) + ", true>");
      out.ivpsPerBlockAdvance = kBlock / (padded_dim(e) / acpl) * adv_lps_spg(acpl);
    } else {
      names.push_back("nnhip::advance_tpi_kernel<" + m + ", nnhip::UserRhs, false, true>");
    }
  } else if (integrator >= 0) {
    const std::string m = std::to_string(integrator);
    if (uses_lps(e)) {
      int adaptive = 0;
      nnhip_ode_integrator_traits(integrator, nullptr, nullptr, &adaptive);
      const int cpl = lps_cpl(e, adaptive != 0);
      names.push_back("nnhip::solve_lps_kernel<" + m + ", nnhip::UserRhs, " + std::to_string(cpl) + ", false>");
      const int scpl = lps_step_cpl(e);
      names.push_back("nnhip::step_lps_kernel<" + m + ", nnhip::UserRhs, false, " + std::to_string(scpl) + ">");
      names.push_back("nnhip::step_lps_kernel<" + m + ", nnhip::UserRhs, true, " + std::to_string(scpl) + ">");
      const int acpl = lps_adv_cpl(e);
      if (adaptive) names.push_back("nnhip::advance_lps_kernel<" + m + ", nnhip::UserRhs, " + std::to_string(acpl) + ">");  // adaptive streaming
      out.ivpsPerBlockSolve = kBlock / (padded_dim(e) / cpl);
      out.ivpsPerBlockStep = kBlock / (padded_dim(e) / scpl);
      out.ivpsPerBlockAdvance = kBlock / (padded_dim(e) / acpl) * adv_lps_spg(acpl);
    } else {
      names.push_back("nnhip::solve_tpi_kernel<" + m + ", nnhip::UserRhs>");
      names.push_back("nnhip::step_tpi_kernel<" + m + ", nnhip::UserRhs, false>");
      names.push_back("nnhip::step_tpi_kernel<" + m + ", nnhip::UserRhs, true>");
      int adaptive = 0;
      nnhip_ode_integrator_traits(integrator, nullptr, nullptr, &adaptive);
      if (adaptive) names.push_back("nnhip::advance_tpi_kernel<" + m + ", nnhip::UserRhs>");  // 4th kernel: adaptive streaming
    }
  } else if (integrator == -2) {
    names.push_back("nnhip::cumtrapz_fn_kernel<nnhip::UserRhs>");
    names.push_back("nnhip::cumsimpson_fn_kernel<nnhip::UserRhs>");
  } else {
    names.push_back("nnhip::rhs_batch_kernel<nnhip::UserRhs>");
  }
  for (auto& n : names) api.addNameExpression(prog, n.c_str());
  // same numerical contract as the ahead-of-time kernels
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"};
  const hiprtcResult rc = api.compileProgram(prog, 4, opts);
  if (rc != HIPRTC_SUCCESS) {
    size_t n = 0;
    api.getProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) api.getProgramLog(prog, log.data());
    g_rtc_err = "hiprtc compilation of user RHS '" + e.name + "' failed:\n" + log;
    api.destroyProgram(&prog);
    return false;
  }
  size_t codeSize = 0;
  api.getCodeSize(prog, &codeSize);
  out.code.resize(codeSize);
  api.getCode(prog, out.code.data());
  if (const char* dump = std::getenv("NNHIP_RTC_DUMP")) {  // debugging aid: the code object of every run-time compilation, as DIR/<name>_<key>.co
    const std::string path = std::string(dump) + "/" + e.name + "_" + std::to_string(integrator) + ".co";
    if (FILE* fdump = std::fopen(path.c_str(), "wb")) { std::fwrite(out.code.data(), 1, out.code.size(), fdump); std::fclose(fdump); }
  }
  out.lowered.clear();
  for (auto& n : names) {
    const char* ln = nullptr;
    if (api.getLoweredName(prog, n.c_str(), &ln) != HIPRTC_SUCCESS || !ln) {
      g_rtc_err = "hiprtcGetLoweredName failed for " + n;
      api.destroyProgram(&prog);
      return false;
    }
    out.lowered.push_back(ln);
  }
  api.destroyProgram(&prog);
  return true;
}

// With the build's compiler in its private namespace first; if it fails where the process's own succeeds, the failure was not the user's
// source (comgr / device libraries of that ROCm not found, a code generator problem): the process's own takes over for good.
bool compile(const UserRhsEntry& e, int integrator, CodeObject& out) {
  const RtcApi& api = rtc_api();
  const bool injected = rtc_is_private(api) && rtc_inject("compile");
  if (injected) g_rtc_err = "hiprtc compilation failed (injected by NNHIP_HIPRTC_INJECT=compile)\n";
  if (!injected && compile_with(api, e, integrator, out)) { out.fromPrivate = rtc_is_private(api); return true; }
  if (!rtc_is_private(api)) return false;
  const std::string first = g_rtc_err;
  CodeObject alt;
  if (compile_with(rtc_state().process, e, integrator, alt)) {
    rtc_disable_private("a compilation failed with " + api.origin + " and succeeded with the process's own: " + first.substr(0, first.find('\n', first.find('\n') + 1)));
    out = std::move(alt);
    out.fromPrivate = false;
    return true;
  }
  g_rtc_err = first;  // both refuse it: the user's source
  return false;
}

bool load(const CodeObject& co, int integrator, Program& out) {
  out.ivpsPerBlockSolve = co.ivpsPerBlockSolve;
  out.ivpsPerBlockStep = co.ivpsPerBlockStep;
  out.ivpsPerBlockAdvance = co.ivpsPerBlockAdvance;
  if ((co.fromPrivate && rtc_inject("load")) || hipModuleLoadData(&out.module, co.code.data()) != hipSuccess) {
    out.module = nullptr;
    g_rtc_err = "hipModuleLoadData failed (no HIP device?)";
    return false;
  }
  hipFunction_t* slots[4] = {&out.solve, &out.stepPos, &out.stepNeg, &out.advance};
  if (integrator == -1) slots[0] = &out.rhs;
  if (integrator == -2) { slots[0] = &out.quad[0]; slots[1] = &out.quad[1]; }
  if (integrator >= kMultiKey) slots[0] = &out.advance;
  if (integrator >= kDenseKey) slots[0] = &out.advanceDense;
  if (integrator >= kCallsKey) slots[0] = &out.solve;
  for (size_t i = 0; i < co.lowered.size(); ++i)
    if (hipModuleGetFunction(slots[i], out.module, co.lowered[i].c_str()) != hipSuccess) {
      g_rtc_err = "hipModuleGetFunction failed for " + co.lowered[i];
      (void)hipModuleUnload(out.module);
      out.module = nullptr;
      return false;
    }
  return true;
}

std::shared_ptr<Program> get_program(int rhs_kind, int integrator) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) { g_rtc_err = "hipGetDevice failed (no HIP device?)"; return nullptr; }
  UserRhsEntry snapshot;  // what the compiler needs, copied out so that hiprtc runs WITHOUT the registry lock
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx < 0 || idx >= (int)g_user.size() || !g_user[idx].alive) { g_rtc_err = "unknown user rhs_kind"; return nullptr; }
    auto it = g_user[idx].programs.find(integrator);
    if (it != g_user[idx].programs.end()) {
      auto ld = it->second.loaded.find(device);
      if (ld != it->second.loaded.end()) return ld->second;
    } else {
      snapshot.name = g_user[idx].name; snapshot.body = g_user[idx].body; snapshot.dim = g_user[idx].dim;
      snapshot.n_params = g_user[idx].n_params; snapshot.perComponent = g_user[idx].perComponent; snapshot.alive = true;
      snapshot.hasCtx = g_user[idx].hasCtx; snapshot.vecs = g_user[idx].vecs; snapshot.n_aux = g_user[idx].n_aux;
      snapshot.sharedLen = g_user[idx].sharedLen; snapshot.ivpRows = g_user[idx].ivpRows;
      snapshot.haloLo = g_user[idx].haloLo; snapshot.haloHi = g_user[idx].haloHi;
    }
  }
  CodeObject fresh;
  bool compiled = false;
  if (snapshot.alive) {  // not compiled yet: do it outside the lock (seconds); a concurrent caller may do the same, the first to insert wins
    if (!compile(snapshot, integrator, fresh)) return nullptr;
    compiled = true;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx >= (int)g_user.size() || !g_user[idx].alive) { g_rtc_err = "user rhs_kind was released during compilation"; return nullptr; }
  auto it = g_user[idx].programs.find(integrator);
  if (it == g_user[idx].programs.end()) {
    if (!compiled) { g_rtc_err = "user rhs_kind was released and re-registered concurrently"; return nullptr; }
    it = g_user[idx].programs.emplace(integrator, std::move(fresh)).first;
  }
  auto ld = it->second.loaded.find(device);
  if (ld != it->second.loaded.end()) return ld->second;
  auto p = std::make_shared<Program>();
  p->device = device;
  if (!load(it->second, integrator, *p)) {
    if (!it->second.fromPrivate || !it->second.loaded.empty()) return nullptr;
    // the process's runtime refuses a code object of the build's compiler: recompile with the process's own (rare: once per process)
    const std::string first = g_rtc_err;
    rtc_disable_private("the process's HIP runtime could not load a code object compiled by " + rtc_state().priv.origin + " (" + first + ")");
    CodeObject again;
    if (!compile_with(rtc_state().process, g_user[idx], integrator, again)) return nullptr;
    it->second = std::move(again);
    p = std::make_shared<Program>();
    p->device = device;
    if (!load(it->second, integrator, *p)) return nullptr;
  }
  it->second.loaded[device] = p;
  return p;
}

}  // namespace


const char* rtc_last_error() { return g_rtc_err.c_str(); }
const char* rtc_compiler_origin() {  // which libhiprtc builds the user's right-hand sides, and why not the other
  static thread_local std::string text;
  RtcState& s = rtc_state();
  std::lock_guard<std::mutex> lk(s.mu);
  text = rtc_api().origin;
  if (!s.why.empty()) text += "; the build's (" NNHIP_BUILD_ROCM_PATH ") is not used: " + s.why;
  return text.c_str();
}

int rtc_register(const char* name, int dim, int n_params, const char* body, bool per_component, bool check_compiles, const RtcCtxLayout* ctx) {
  UserRhsEntry e;
  e.name = name ? name : "user";
  e.body = body;
  e.dim = dim;
  e.n_params = n_params;
  e.perComponent = per_component;
  e.alive = true;
  if (ctx) {
    e.hasCtx = true;
    e.n_aux = ctx->n_aux;
    e.sharedLen = n_params > kMaxParams ? n_params : 0;  // scalars that do not fit the kernel arguments lead the shared block
    for (int k = 0; k < ctx->n_vectors; ++k) {
      CtxVec v;
      v.name = ctx->names[k];
      v.len = ctx->lens[k];
      v.perIvp = ctx->per_ivp[k] != 0;
      if (v.perIvp) { v.offset = e.ivpRows; e.ivpRows += v.len; }
      else { v.offset = e.sharedLen; e.sharedLen += v.len; }
      e.vecs.push_back(v);
    }
  }
  if (check_compiles) {  // syntax check now (device-independent), so errors surface at registration
    CodeObject scratch;
    if (!compile(e, -1, scratch)) return -1;  // the rhs_batch kernel only; same private-then-process order as every other compilation
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_user.push_back(std::move(e));
  return NNHIP_RHS_USER_BASE + (int)g_user.size() - 1;
}

namespace {
thread_local std::map<int, std::shared_ptr<CtxBinding>> t_bound;  // this thread's own binding per rhs_kind
struct ShardView { const double* shared; const double* ivp; double* aux; int64_t stride; };
thread_local std::map<int, ShardView> t_shard;                     // a multi-GPU worker thread's column range (rtc_ctx_shard_enter)
}  // namespace

int rtc_release(int rhs_kind) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::shared_ptr<CtxBinding> mine, latest;  // destroyed after g_mu is dropped: ~CtxBinding talks to the devices
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx < 0 || idx >= (int)g_user.size() || !g_user[idx].alive) return -1;
  g_user[idx].programs.clear();  // drops the registry's references; a module is unloaded when its last launch in flight lets go (~Program)
  auto it = t_bound.find(rhs_kind);
  if (it != t_bound.end()) { mine = std::move(it->second); t_bound.erase(it); }  // the releasing thread's own binding goes with the kind
  latest = std::move(g_user[idx].bound);  // (another thread that bound this kind keeps its reference until it binds again or exits; `alive` below makes it unreachable)
  g_user[idx].bound.reset();
  g_user[idx].alive = false;
  return 0;
}

bool rtc_is_thread_per_ivp(int rhs_kind) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::lock_guard<std::mutex> lk(g_mu);
  return idx >= 0 && idx < (int)g_user.size() && g_user[idx].alive && !uses_lps(g_user[idx]);
}

bool rtc_info(int rhs_kind, int* dim, int* n_params) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx < 0 || idx >= (int)g_user.size() || !g_user[idx].alive) return false;
  if (dim) *dim = g_user[idx].dim;
  if (n_params) *n_params = g_user[idx].n_params;
  return true;
}

// ---- context bindings -------------------------------------------------------------------------------------------------------------
namespace {

// the binding the calling thread's calls read: its own if it made one, else the latest of any thread.  Caller holds g_mu.
std::shared_ptr<CtxBinding> binding_of(int idx) {
  if (idx < 0 || idx >= (int)g_user.size() || !g_user[idx].alive) return nullptr;
  auto it = t_bound.find(idx + NNHIP_RHS_USER_BASE);
  if (it != t_bound.end() && it->second) return it->second;
  return g_user[idx].bound;
}
bool check_layout(const UserRhsEntry& e, const void* shared, int64_t shared_len, const void* per_ivp, int64_t per_ivp_rows, const void* aux, int n_aux, int64_t stride) {
  if (!e.hasCtx) { g_rtc_err = "this right-hand side was compiled without a context layout (nnhip_ode_rhs_compile_ctx)"; return false; }
  if (shared_len != e.sharedLen || per_ivp_rows != e.ivpRows || n_aux != e.n_aux) {
    g_rtc_err = "context block does not match the compiled layout: shared " + std::to_string(e.sharedLen) + " doubles, per-IVP " + std::to_string(e.ivpRows) +
                " rows, " + std::to_string(e.n_aux) + " aux slots";
    return false;
  }
  if ((e.sharedLen > 0 && !shared) || (e.ivpRows > 0 && !per_ivp) || (e.n_aux > 0 && !aux) || ((e.ivpRows > 0 || e.n_aux > 0) && stride < 1)) {
    g_rtc_err = "context block: a declared part is NULL (or the stride is not positive)";
    return false;
  }
  return true;
}
// Caller holds g_mu and destroys `dropped` AFTER releasing it: the last reference to a binding the library uploaded itself runs ~CtxBinding — hipSetDevice,
// hipDeviceSynchronize, hipFree per shard and owned block — which must not happen with every other thread's rtc_ctx_fill / get_program / launch waiting on g_mu.
void install(int rhs_kind, const std::shared_ptr<CtxBinding>& b, std::vector<std::shared_ptr<CtxBinding>>& dropped) {
  std::shared_ptr<CtxBinding>& slot = g_user[rhs_kind - NNHIP_RHS_USER_BASE].bound;
  dropped.push_back(std::move(slot));
  slot = b;
  std::shared_ptr<CtxBinding>& mine = t_bound[rhs_kind];
  dropped.push_back(std::move(mine));
  mine = b;
  // this thread's bindings of kinds that have been released since (rtc_release by another thread cannot reach them): let go of their device copies now
  for (auto it = t_bound.begin(); it != t_bound.end();) {
    const int idx = it->first - NNHIP_RHS_USER_BASE;
    if (idx < 0 || idx >= (int)g_user.size() || !g_user[idx].alive) { dropped.push_back(std::move(it->second)); it = t_bound.erase(it); }
    else ++it;
  }
}
// the mutable slots of a binding whose last writer was a sharded solve: device shards -> host copy -> the single-device block
int collect_aux(CtxBinding& b) {
  std::lock_guard<std::mutex> lk(b.mu);
  if (!b.auxInShards) return 0;
  int prev = 0;
  (void)hipGetDevice(&prev);
  bool ok = true;
  if (b.nAux > 0) {
    for (const CtxShard& sh : b.shards) {
      if (sh.n == 0 || !sh.d[2]) continue;
      ok = ok && hipSetDevice(sh.device) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
           hipMemcpy2D(b.hAux.data() + sh.lo, (size_t)b.stride * 8, sh.d[2], (size_t)sh.n * 8, (size_t)sh.n * 8, (size_t)b.nAux, hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (ok && b.ownedDevice >= 0 && b.owned[2])
      ok = hipSetDevice(b.ownedDevice) == hipSuccess && hipMemcpy(b.owned[2], b.hAux.data(), b.hAux.size() * 8, hipMemcpyHostToDevice) == hipSuccess;
  }
  (void)hipSetDevice(prev);
  if (!ok) { g_rtc_err = "gathering the mutable slots of a sharded solve failed"; return -1; }
  b.auxInShards = false;
  return 0;
}
}  // namespace

int rtc_bind_ctx_host(int rhs_kind, const double* shared, int64_t shared_len, const double* per_ivp, int64_t per_ivp_rows, const double* aux_init, int n_aux,
                      int64_t stride, int device) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx < 0 || idx >= (int)g_user.size() || !g_user[idx].alive) { g_rtc_err = "unknown user rhs_kind"; return -1; }
    if (!check_layout(g_user[idx], shared, shared_len, per_ivp, per_ivp_rows, aux_init, n_aux, stride)) return -1;
  }
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) { g_rtc_err = "bad device ordinal"; return -1; }
  auto b = std::make_shared<CtxBinding>();
  const size_t bytes[3] = {(size_t)shared_len * 8, (size_t)per_ivp_rows * (size_t)stride * 8, (size_t)n_aux * (size_t)stride * 8};
  const void* src[3] = {shared, per_ivp, aux_init};
  bool ok = true;
  b->ownedDevice = device;
  for (int k = 0; k < 3 && ok; ++k) {
    if (!bytes[k]) continue;
    ok = src[k] != nullptr && hipMalloc(&b->owned[k], bytes[k]) == hipSuccess && hipMemcpy(b->owned[k], src[k], bytes[k], hipMemcpyHostToDevice) == hipSuccess;
  }
  (void)hipSetDevice(prev);
  if (!ok) { g_rtc_err = "context block: a declared part is NULL or the device allocation failed"; return -1; }  // (~CtxBinding frees what was allocated)
  b->shared = (const double*)b->owned[0]; b->ivp = (const double*)b->owned[1]; b->aux = (double*)b->owned[2]; b->stride = stride;
  b->sharedLen = shared_len; b->ivpRows = per_ivp_rows; b->nAux = n_aux;
  b->haveHost = true;
  if (bytes[0]) b->hShared.assign(shared, shared + shared_len);
  if (bytes[1]) b->hIvp.assign(per_ivp, per_ivp + (size_t)per_ivp_rows * (size_t)stride);
  if (bytes[2]) b->hAux.assign(aux_init, aux_init + (size_t)n_aux * (size_t)stride);
  std::vector<std::shared_ptr<CtxBinding>> dropped;  // destroyed after g_mu is released (declared before the guard)
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_user[idx].alive) { g_rtc_err = "unknown user rhs_kind"; return -1; }
  install(rhs_kind, b, dropped);
  return 0;
}

int rtc_read_aux(int rhs_kind, double* aux_out) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::shared_ptr<CtxBinding> b;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    b = binding_of(idx);
    if (!b || !g_user[idx].hasCtx || g_user[idx].n_aux == 0 || !b->aux) { g_rtc_err = "no mutable slots bound to this rhs_kind"; return -1; }
  }
  if (!aux_out) { g_rtc_err = "aux_out is NULL"; return -1; }
  if (collect_aux(*b) != 0) return -1;
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (b->ownedDevice >= 0) (void)hipSetDevice(b->ownedDevice);
  const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(aux_out, b->aux, (size_t)b->nAux * (size_t)b->stride * 8, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipSetDevice(prev);
  if (!ok) { g_rtc_err = "copying the mutable slots back failed"; return -1; }
  return 0;
}

int rtc_bind_ctx(int rhs_kind, const double* shared, int64_t shared_len, const double* per_ivp, int64_t per_ivp_rows, double* aux, int n_aux, int64_t stride) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::vector<std::shared_ptr<CtxBinding>> dropped;  // a binding the library had uploaded itself is released with its last reference — after g_mu (declared before the guard)
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx < 0 || idx >= (int)g_user.size() || !g_user[idx].alive) { g_rtc_err = "unknown user rhs_kind"; return -1; }
  if (!check_layout(g_user[idx], shared, shared_len, per_ivp, per_ivp_rows, aux, n_aux, stride)) return -1;
  auto b = std::make_shared<CtxBinding>();  // the caller's own device memory: nothing owned, nothing to shard from
  b->shared = shared; b->ivp = per_ivp; b->aux = aux; b->stride = stride;
  b->sharedLen = shared_len; b->ivpRows = per_ivp_rows; b->nAux = n_aux;
  install(rhs_kind, b, dropped);
  return 0;
}
// Declares that the per-component body of `rhs_kind` reads components c - lo .. c + hi (cyclically) only.  Code objects compiled before are dropped.
int rtc_set_halo(int rhs_kind, int lo, int hi) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::lock_guard<std::mutex> lk(g_mu);
  if (idx < 0 || idx >= (int)g_user.size() || !g_user[idx].alive) { g_rtc_err = "unknown user rhs_kind"; return -1; }
  UserRhsEntry& e = g_user[idx];
  if (!e.perComponent) { g_rtc_err = "a halo can only be declared for a per-component body"; return -1; }
  if (lo < 0 || hi < 0 || lo > 4 || hi > 4 || lo + hi + 1 > e.dim) { g_rtc_err = "halo must be 0..4 components on each side and narrower than the system"; return -1; }
  if (e.haloLo == lo && e.haloHi == hi) return 0;
  e.haloLo = lo; e.haloHi = hi;
  e.programs.clear();  // shared_ptr-owned modules: launches in flight keep theirs alive
  return 0;
}
void rtc_drop_owned_ctx(int) {}  // (owned device copies now live and die with their binding)

// 0: no context layout (P untouched apart from zeroed pointers); 1: filled; -1: declared but not bound / N beyond the bound batch
int rtc_ctx_fill(int rhs_kind, int64_t N, Params& P, int* n_scalars_in_block) {
  P.shared = nullptr; P.ivp = nullptr; P.aux = nullptr; P.stride = 0;
  if (n_scalars_in_block) *n_scalars_in_block = 0;
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  if (idx < 0) return 0;
  std::shared_ptr<CtxBinding> b;
  int64_t ivpRows = 0;
  int nAux = 0;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx >= (int)g_user.size() || !g_user[idx].alive) return 0;
    const UserRhsEntry& e = g_user[idx];
    if (!e.hasCtx) return 0;
    if (n_scalars_in_block) *n_scalars_in_block = e.n_params > kMaxParams ? e.n_params : 0;
    ivpRows = e.ivpRows; nAux = e.n_aux;
    auto sv = t_shard.find(rhs_kind);
    if (sv != t_shard.end()) {  // a worker thread of a multi-GPU entry: its device's column range
      if ((ivpRows > 0 || nAux > 0) && N > sv->second.stride) { g_rtc_err = "N exceeds the shard the context block was cut for"; return -1; }
      P.shared = sv->second.shared; P.ivp = sv->second.ivp; P.aux = sv->second.aux; P.stride = sv->second.stride;
      return 1;
    }
    const bool needsBlock = e.sharedLen > 0 || e.ivpRows > 0 || e.n_aux > 0;
    b = binding_of(idx);
    if (needsBlock && !b) { g_rtc_err = "the right-hand side declares a context block but none is bound (nnhip_ode_rhs_bind_ctx_f64_dev)"; return -1; }
    if (!b) return 1;
  }
  if (collect_aux(*b) != 0) return -1;  // (outside g_mu: it talks to the devices)
  if ((ivpRows > 0 || nAux > 0) && N > b->stride) { g_rtc_err = "N exceeds the batch the context block was bound for"; return -1; }
  P.shared = b->shared; P.ivp = b->ivp; P.aux = b->aux; P.stride = b->stride;
  return 1;
}
bool rtc_has_aux(int rhs_kind) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::lock_guard<std::mutex> lk(g_mu);
  return idx >= 0 && idx < (int)g_user.size() && g_user[idx].alive && g_user[idx].hasCtx && g_user[idx].n_aux > 0;
}
bool rtc_has_per_ivp_ctx(int rhs_kind) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::lock_guard<std::mutex> lk(g_mu);
  return idx >= 0 && idx < (int)g_user.size() && g_user[idx].alive && g_user[idx].hasCtx && (g_user[idx].ivpRows > 0 || g_user[idx].n_aux > 0 || g_user[idx].sharedLen > 0);
}

// ---- a bound context block across several devices (the multi-GPU entries) -------------------------------------------------------------
// The calling thread's binding of `rhs_kind`, cut into `n_shards` column ranges [lo[r], lo[r] + n[r]) of the per-IVP rows and the mutable slots,
// shard r uploaded to devices[r] (the shared block whole).  Only a binding made from HOST arrays can be cut (the library has the values); one made
// of the caller's device pointers belongs to that device.  -> 0, or -1 with the reason in rtc_last_error().  *handle keeps the binding alive.
int rtc_ctx_shards_prepare(int rhs_kind, int n_shards, const int* devices, const int64_t* lo, const int64_t* n, std::shared_ptr<void>* handle) {
  const int idx = rhs_kind - NNHIP_RHS_USER_BASE;
  std::shared_ptr<CtxBinding> b;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    b = binding_of(idx);
  }
  if (!b) { g_rtc_err = "the right-hand side declares a context block but none is bound"; return -1; }
  if (!b->haveHost) {
    g_rtc_err = "the context block was bound as device pointers of ONE device (nnhip_ode_rhs_bind_ctx_f64_dev): bind it from host arrays "
                "(nnhip_ode_rhs_bind_ctx_f64) to have it sharded with the batch, or bind and solve per device";
    return -1;
  }
  if (collect_aux(*b) != 0) return -1;
  std::lock_guard<std::mutex> lk(b->mu);
  for (int r = 0; r < n_shards; ++r)
    if (lo[r] < 0 || n[r] < 0 || ((b->ivpRows > 0 || b->nAux > 0) && lo[r] + n[r] > b->stride)) { g_rtc_err = "the batch is larger than the bound context block"; return -1; }
  bool same = (int)b->shards.size() == n_shards;
  for (int r = 0; same && r < n_shards; ++r) same = b->shards[r].device == devices[r] && b->shards[r].lo == lo[r] && b->shards[r].n == n[r];
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (!same) {
    b->free_shards();
    b->shards.resize(n_shards);
    for (int r = 0; r < n_shards; ++r) { b->shards[r].device = devices[r]; b->shards[r].lo = lo[r]; b->shards[r].n = n[r]; }
  }
  bool ok = true;
  if (b->nAux > 0 && b->ownedDevice >= 0 && b->owned[2])  // a single-device solve may have written the slots since the bind: the device block is the truth
    ok = hipSetDevice(b->ownedDevice) == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
         hipMemcpy(b->hAux.data(), b->owned[2], b->hAux.size() * 8, hipMemcpyDeviceToHost) == hipSuccess;
  for (int r = 0; r < n_shards && ok; ++r) {
    CtxShard& sh = b->shards[r];
    if (sh.n == 0) continue;
    ok = hipSetDevice(sh.device) == hipSuccess;
    const size_t bytes[3] = {(size_t)b->sharedLen * 8, (size_t)b->ivpRows * (size_t)sh.n * 8, (size_t)b->nAux * (size_t)sh.n * 8};
    for (int k = 0; k < 3 && ok; ++k)
      if (bytes[k] && !sh.d[k]) ok = hipMalloc(&sh.d[k], bytes[k]) == hipSuccess;
    // the shared block whole (uploaded once per shard set); the column range of every per-IVP row; the CURRENT mutable slots
    if (ok && bytes[0] && !same) ok = hipMemcpy(sh.d[0], b->hShared.data(), bytes[0], hipMemcpyHostToDevice) == hipSuccess;
    if (ok && bytes[1] && !same)
      ok = hipMemcpy2D(sh.d[1], (size_t)sh.n * 8, b->hIvp.data() + sh.lo, (size_t)b->stride * 8, (size_t)sh.n * 8, (size_t)b->ivpRows, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && bytes[2])  // the CURRENT mutable slots, every time
      ok = hipMemcpy2D(sh.d[2], (size_t)sh.n * 8, b->hAux.data() + sh.lo, (size_t)b->stride * 8, (size_t)sh.n * 8, (size_t)b->nAux, hipMemcpyHostToDevice) == hipSuccess;
  }
  (void)hipSetDevice(prev);
  if (!ok) { b->free_shards(); g_rtc_err = "uploading a column range of the context block to its device failed"; return -1; }
  b->auxInShards = b->nAux > 0;
  if (handle) *handle = b;
  return 0;
}
// on the worker thread of shard r: every call of this thread that takes `rhs_kind` reads the shard, until rtc_ctx_shard_leave
int rtc_ctx_shard_enter(int rhs_kind, const std::shared_ptr<void>& handle, int r) {
  CtxBinding* b = static_cast<CtxBinding*>(handle.get());
  if (!b || r < 0 || r >= (int)b->shards.size()) { g_rtc_err = "no such shard"; return -1; }
  const CtxShard& sh = b->shards[r];
  t_shard[rhs_kind] = ShardView{(const double*)sh.d[0], (const double*)sh.d[1], (double*)sh.d[2], sh.n};
  return 0;
}
void rtc_ctx_shard_leave(int rhs_kind) { t_shard.erase(rhs_kind); }
// after a SYNCHRONOUS sharded solve: the mutable slots back into the host copy and the single-device block (asynchronous entries leave that
// to the next reader: rtc_read_aux / the next single-device call / the next sharded one)
int rtc_ctx_shards_collect(const std::shared_ptr<void>& handle) {
  CtxBinding* b = static_cast<CtxBinding*>(handle.get());
  return b ? collect_aux(*b) : 0;
}

static hipError_t launch(hipFunction_t f, int64_t n, int perBlock, void* arg, hipStream_t s) {
  const int64_t grid = (n + perBlock - 1) / perBlock;
  if (grid <= 0) return hipSuccess;
  void* params[] = {arg};
  return hipModuleLaunchKernel(f, (unsigned)grid, 1, 1, kBlock, 1, 1, 0, s, params, nullptr);
}

hipError_t rtc_launch_solve(int rhs_kind, int integrator, const SolveArgs& a, hipStream_t s) {
  // one code object per mode of the solve kernel, compiled when first asked for: per-IVP grids / per-IVP spans / no dense output (the lean instantiation the
  // compiled-in systems use for 2-point tspans) / dense output
  const std::shared_ptr<Program> p = get_program(rhs_kind, a.perCall.tGrid ? kGridKey + integrator : (a.perCall.tEnd ? kCallsKey + integrator : (!a.useDense ? kLeanKey + integrator : integrator)));
  if (!p) return hipErrorInvalidValue;
  SolveArgs copy = a;
  return launch(p->solve, a.N, p->ivpsPerBlockSolve, &copy, s);
}
hipError_t rtc_launch_step(int rhs_kind, int integrator, const StepArgs& a, int negate, hipStream_t s) {
  const std::shared_ptr<Program> p = get_program(rhs_kind, integrator);
  if (!p) return hipErrorInvalidValue;
  StepArgs copy = a;
  return launch(negate ? p->stepNeg : p->stepPos, a.N, p->ivpsPerBlockStep, &copy, s);
}
hipError_t rtc_launch_advance(int rhs_kind, int integrator, const StepArgs& a, hipStream_t s) {
  const std::shared_ptr<Program> p = get_program(rhs_kind, a.stepsPerLaunch > 1 ? kMultiKey + integrator : integrator);
  if (!p) return hipErrorInvalidValue;
  if (!p->advance) { g_rtc_err = "no advance kernel: fixed-step integrator"; return hipErrorInvalidValue; }
  StepArgs copy = a;
  return launch(p->advance, a.N, p->ivpsPerBlockAdvance, &copy, s);
}
hipError_t rtc_launch_advance_dense(int rhs_kind, int integrator, const StepArgs& a, hipStream_t s) {
  const std::shared_ptr<Program> p = get_program(rhs_kind, kDenseKey + integrator);
  if (!p) return hipErrorInvalidValue;
  StepArgs copy = a;
  return launch(p->advanceDense, a.N, p->ivpsPerBlockAdvance, &copy, s);
}
hipError_t rtc_launch_rhs(int rhs_kind, int64_t N, int64_t is, int64_t cs, double t, const double* y, double* dy, const Params& P,
                          hipStream_t s) {
  const std::shared_ptr<Program> p = get_program(rhs_kind, -1);
  if (!p) return hipErrorInvalidValue;
  const int64_t grid = (N + kBlock - 1) / kBlock;
  if (grid <= 0) return hipSuccess;
  Params Pc = P;
  void* params[] = {&N, &is, &cs, &t, &y, &dy, &Pc};
  return hipModuleLaunchKernel(p->rhs, (unsigned)grid, 1, 1, kBlock, 1, 1, 0, s, params, nullptr);
}

// The reference's Vector[float] state has any length; the ahead-of-time kernels cover the sizes of the BASELINE configs and the
// reference's tests (NNHIP_FOR_EACH_TPI_RHS / _LPS_RHS).  Every other size of the size-generic built-in right-hand sides is
// instantiated at run time from the same per-component expressions (hiprtc, ode_rtc.hip; one entry per (kind, dim), compiled on
// first use, cached for the life of the process): dims 1..16 thread-per-IVP, 8 / 16 / 32 lanes-per-system.
bool rtc_builtin_available(int rhs_kind, int dim) {
  if (!(rhs_kind == NNHIP_RHS_NEG_Y || rhs_kind == NNHIP_RHS_LINEAR || rhs_kind == NNHIP_RHS_AFFINE_T || rhs_kind == NNHIP_RHS_RING)) return false;
  return dim >= 1 && dim <= 256;
}
static std::mutex g_synth_mu;
static std::map<std::pair<int, int>, int> g_synth;  // (built-in kind, dim) -> user rhs_kind
int rtc_builtin_kind(int rhs_kind, int dim) {
  if (!rtc_builtin_available(rhs_kind, dim)) return -1;
  std::lock_guard<std::mutex> lk(g_synth_mu);
  auto it = g_synth.find({rhs_kind, dim});
  if (it != g_synth.end()) return it->second;
  const char* body = nullptr;  // the comp() bodies of RhsNegY / RhsLinear / RhsAffineT / RhsRing (ode_device.hpp), verbatim
  int np = 0;
  switch (rhs_kind) {
    case NNHIP_RHS_NEG_Y: body = "return -y[c];"; np = 0; break;
    case NNHIP_RHS_LINEAR: body = "return y[c] * p[0];"; np = 1; break;
    case NNHIP_RHS_AFFINE_T: body = "return p[0] * y[c] + p[1] * t;"; np = 2; break;
    default: body = "return -((double)(c + 1) / (double)dim) * y[c] + p[0] * y[(c + 1) % dim];"; np = 1; break;
  }
  const std::string name = "builtin" + std::to_string(rhs_kind) + "_dim" + std::to_string(dim);
  const int k = rtc_register(name.c_str(), dim, np, body, true, false);
  if (k >= 0) g_synth[{rhs_kind, dim}] = k;
  return k;
}


hipError_t rtc_launch_quad(int rhs_kind, int rule, const QuadArgs& a, hipStream_t s) {
  const std::shared_ptr<Program> p = get_program(rhs_kind, -2);
  if (!p) return hipErrorInvalidValue;
  QuadArgs copy = a;
  return launch(p->quad[rule ? 1 : 0], a.N, kBlock, &copy, s);
}

}  // namespace nnhip

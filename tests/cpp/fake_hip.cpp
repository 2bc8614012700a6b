// fake_hip.cpp — TEST INFRASTRUCTURE: a stand-in HIP runtime (and RCCL) for exercising the library's HOST logic on a machine without a GPU.  Built as a shared
// object and put in front of libamdhip64 with LD_PRELOAD (and found as "librccl.so.1" through LD_LIBRARY_PATH) by tests/test_host_logic_under_fake_hip.py, in a
// subprocess of its own.  It is NOT a way to run the product on a CPU: kernel launches do nothing (device buffers keep whatever the host copied into them), so no
// solve produces a result.  What it gives the CPU test suite:
//   * FAKE_HIP_DEVICES=N devices, so the multi-device branches of the one-call entries run (hipSetDevice(r), one worker thread and stream per device);
//   * device memory = tracked host allocations: every hipMemcpy / hipMemcpy2D / hipMemset range must lie inside ONE live allocation on the device side, on the
//     device the allocation was made on or any other (peer access is not modelled) — a copy that runs over the end of a shard, uses a freed buffer or a wrong pitch
//     aborts the process with a message; double frees and frees of unknown pointers likewise;
//   * RCCL's collectives with their real data movement (ncclAllGather, grouped ncclBroadcast between the fake devices), so the reassembly of sharded state tensors
//     — SoA plane by plane, ragged shards, an empty shard — is checked for placement;
//   * counters (launches, copies, live allocations) for leak checks.
// Nothing in the library or the package knows about it.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Alloc { size_t size; int device; bool host; };
std::mutex g_mu;
std::map<uintptr_t, Alloc> g_allocs;  // by base address
thread_local int t_device = 0;
std::atomic<long> g_launches{0}, g_copies{0}, g_bytes{0};

int n_devices() {
  static const int n = [] { const char* e = std::getenv("FAKE_HIP_DEVICES"); const int v = e ? std::atoi(e) : 1; return v > 0 ? v : 1; }();
  return n;
}
[[noreturn]] void die(const char* what, const void* p, size_t n) {
  std::fprintf(stderr, "fake_hip: %s (pointer %p, %zu bytes)\n", what, p, n);
  std::fflush(stderr);
  std::abort();
}
// the live allocation that holds [p, p + n), or nullptr
const Alloc* find(const void* p, size_t n) {
  const uintptr_t a = (uintptr_t)p;
  auto it = g_allocs.upper_bound(a);
  if (it == g_allocs.begin()) return nullptr;
  --it;
  if (a >= it->first && a + n <= it->first + it->second.size) return &it->second;
  return nullptr;
}
bool touches_tracked(const void* p) {
  const uintptr_t a = (uintptr_t)p;
  auto it = g_allocs.upper_bound(a);
  if (it == g_allocs.begin()) return false;
  --it;
  return a < it->first + it->second.size;
}
void check_side(const void* p, size_t n, bool mustBeDevice, const char* what) {
  if (n == 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  const Alloc* a = find(p, n);
  if (a) return;
  if (mustBeDevice || touches_tracked(p)) die(what, p, n);  // a device pointer outside every live allocation, or a range that starts inside one and runs over its end
}
hipError_t do_alloc(void** p, size_t n, bool host) {
  if (!p) return hipErrorInvalidValue;
  void* q = std::calloc(1, n ? n + 64 : 64);  // (+64: a distinct address for zero-size requests; the tracked size is the requested one)
  if (!q) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lk(g_mu);
  g_allocs[(uintptr_t)q] = Alloc{n, t_device, host};
  *p = q;
  return hipSuccess;
}
hipError_t do_free(void* p, bool host) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_allocs.find((uintptr_t)p);
  if (it == g_allocs.end() || it->second.host != host) die(host ? "hipHostFree of a pointer that is not a live pinned allocation" : "hipFree of a pointer that is not a live device allocation (double free?)", p, 0);
  std::memset(p, 0xdd, it->second.size);  // use after free shows
  g_allocs.erase(it);
  std::free(p);
  return hipSuccess;
}
hipError_t copy(void* dst, const void* src, size_t n, hipMemcpyKind kind) {
  if (n == 0) return hipSuccess;
  const bool dDev = kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice, sDev = kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice;
  check_side(dst, n, dDev, "copy destination is not inside one live device allocation");
  check_side(src, n, sDev, "copy source is not inside one live device allocation");
  std::memmove(dst, src, n);
  g_copies++;
  g_bytes += (long)n;
  return hipSuccess;
}
}  // namespace

// ---- optional: a hook that EXECUTES launches (tests/isa_backed_node.py: the gfx950 interpreter) instead of dropping them ----
typedef int (*fake_launch_hook_t)(const char* name, const void* image, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, void** args);
namespace {
fake_launch_hook_t g_hook = nullptr;
// host stub -> mangled device name (recorded from __hipRegisterFunction).  Constructed on first use and never destroyed: a program LINKED against the library runs the
// library's registration constructors before this preloaded object's own static constructors (the loader initialises a preloaded object last).
std::map<const void*, std::string>& kernel_names() { static auto* m = new std::map<const void*, std::string>(); return *m; }
struct FakeModule { const void* image; };
struct FakeFunction { FakeModule* module; std::string name; };
}  // namespace

extern "C" {
void fake_hip_set_launch_hook(fake_launch_hook_t h) { g_hook = h; }
// the compiler-generated registration of every kernel of a translation unit: remembered here, then handed on to the real runtime (which needs no device for it)
void __hipRegisterFunction(void** modules, const void* hostFunction, char* deviceFunction, const char* deviceName, unsigned int threadLimit, void* tid, void* bid, void* blockDim,
                           void* gridDim, int* wSize) {
  { std::lock_guard<std::mutex> lk(g_mu); kernel_names()[hostFunction] = deviceName; }
  using Fn = void (*)(void**, const void*, char*, const char*, unsigned int, void*, void*, void*, void*, int*);
  static Fn real = (Fn)dlsym(RTLD_NEXT, "__hipRegisterFunction");
  if (real) real(modules, hostFunction, deviceFunction, deviceName, threadLimit, tid, bid, blockDim, gridDim, wSize);
}
// ---- counters for the tests ----
long fake_hip_launches() { return g_launches.load(); }
long fake_hip_copies() { return g_copies.load(); }
int fake_hip_owns(const void* p, size_t n) { std::lock_guard<std::mutex> lk(g_mu); return find(p, n ? n : 1) != nullptr; }
void fake_hip_dump_live() {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_allocs) if (!kv.second.host) std::fprintf(stderr, "fake_hip: live device allocation %p, %zu bytes, device %d\n", (void*)kv.first, kv.second.size, kv.second.device);
}
long fake_hip_live_device_allocations() { std::lock_guard<std::mutex> lk(g_mu); long n = 0; for (auto& kv : g_allocs) n += kv.second.host ? 0 : 1; return n; }

// ---- devices ----
hipError_t hipGetDeviceCount(int* n) { *n = n_devices(); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= n_devices()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake_hip error"; }
hipError_t hipRuntimeGetVersion(int* v) { *v = 70200000; return hipSuccess; }

// ---- memory ----
hipError_t hipMalloc(void** p, size_t n) { return do_alloc(p, n, false); }
hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return do_alloc(p, n, false); }
hipError_t hipFree(void* p) { return do_free(p, false); }
hipError_t hipFreeAsync(void* p, hipStream_t) { return do_free(p, false); }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { return do_alloc(p, n, true); }
hipError_t hipHostFree(void* p) { return do_free(p, true); }
hipError_t hipHostRegister(void*, size_t, unsigned int) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { return copy(d, s, n, k); }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return copy(d, s, n, k); }
hipError_t hipMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k) {
  if (w > dp || w > sp) die("hipMemcpy2D: width exceeds a pitch", d, w);
  for (size_t r = 0; r < h; ++r) copy((char*)d + r * dp, (const char*)s + r * sp, w, k);
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t) { return hipMemcpy2D(d, dp, s, sp, w, h, k); }
hipError_t hipMemset(void* d, int v, size_t n) { check_side(d, n, true, "hipMemset range is not inside one live device allocation"); std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  std::lock_guard<std::mutex> lk(g_mu);
  const Alloc* al = find(p, 1);
  if (!al) return hipErrorInvalidValue;
  std::memset(a, 0, sizeof(*a));
  a->type = al->host ? hipMemoryTypeHost : hipMemoryTypeDevice;
  a->device = al->device;
  a->devicePointer = const_cast<void*>(p);
  a->hostPointer = al->host ? const_cast<void*>(p) : nullptr;
  return hipSuccess;
}

// ---- streams, events, graphs ----
struct FakeStream { int device; bool capturing; long captured; };
struct FakeGraph { long launches; };
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = (hipStream_t) new FakeStream{t_device, false, 0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete (FakeStream*)s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
// graphs: work issued during a capture executes at once (copies move data, launches do nothing) AND is counted; replaying the graph only counts its launches
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st) { *st = (s && ((FakeStream*)s)->capturing) ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) {
  if (g_hook) return hipErrorNotSupported;  // (executing launches: a replay would have to re-execute them; the callers use the eager paths)
  if (!s || ((FakeStream*)s)->capturing) return hipErrorIllegalState;
  ((FakeStream*)s)->capturing = true; ((FakeStream*)s)->captured = 0;
  return hipSuccess;
}
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) {
  if (!s || !((FakeStream*)s)->capturing) return hipErrorIllegalState;
  ((FakeStream*)s)->capturing = false;
  *g = (hipGraph_t) new FakeGraph{((FakeStream*)s)->captured};
  return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, hipGraphNode_t*, char*, size_t) { *e = (hipGraphExec_t) new FakeGraph{((FakeGraph*)g)->launches}; return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { g_launches += ((FakeGraph*)e)->launches; return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete (FakeGraph*)g; return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete (FakeGraph*)e; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t) new int(0); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned int) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete (int*)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }

// ---- kernels: nothing runs ----
hipError_t hipLaunchKernel(const void* fn, dim3 g, dim3 b, void** args, size_t, hipStream_t s) {
  g_launches++;
  if (s && ((FakeStream*)s)->capturing) ((FakeStream*)s)->captured++;
  if (g_hook) {
    std::string name;
    { std::lock_guard<std::mutex> lk(g_mu); auto it = kernel_names().find(fn); if (it != kernel_names().end()) name = it->second; }
    if (name.empty()) die("launch of a kernel that was never registered", fn, 0);
    return g_hook(name.c_str(), nullptr, g.x, g.y, g.z, b.x, b.y, b.z, args) == 0 ? hipSuccess : hipErrorLaunchFailure;
  }
  return hipSuccess;
}
hipError_t hipModuleLoadData(hipModule_t* m, const void* image) { *m = (hipModule_t) new FakeModule{image}; return hipSuccess; }
hipError_t hipModuleUnload(hipModule_t m) { delete (FakeModule*)m; return hipSuccess; }
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char* name) { *f = (hipFunction_t) new FakeFunction{(FakeModule*)m, name}; return hipSuccess; }  // (leaked: a few per program)
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned, hipStream_t s, void** params, void**) {
  g_launches++;
  if (s && ((FakeStream*)s)->capturing) ((FakeStream*)s)->captured++;
  if (g_hook) {
    FakeFunction* ff = (FakeFunction*)f;
    return g_hook(ff->name.c_str(), ff->module->image, gx, gy, gz, bx, by, bz, params) == 0 ? hipSuccess : hipErrorLaunchFailure;
  }
  return hipSuccess;
}

// ---- RCCL: one process, one communicator per fake device; collectives execute when every rank of the clique has issued its call (inside a group: at ncclGroupEnd) ----
struct FakeComm { int rank, n; std::vector<FakeComm*>* clique; };
struct Op { int kind; const void* send; void* recv; size_t count; int root; };
}  // extern "C"
namespace {
std::mutex g_nccl_mu;
std::map<FakeComm*, std::vector<Op>> g_pending;
int g_group = 0;
ncclResult_t flush() {
  // process the queues of every clique in lockstep: the k-th call of every rank must be the same collective
  std::vector<std::vector<FakeComm*>*> cliques;
  for (auto& kv : g_pending) { bool seen = false; for (auto* c : cliques) seen = seen || c == kv.first->clique; if (!seen) cliques.push_back(kv.first->clique); }
  for (auto* cl : cliques) {
    const size_t nOps = g_pending[(*cl)[0]].size();
    for (FakeComm* c : *cl) if (g_pending[c].size() != nOps) { std::fprintf(stderr, "fake_hip: ranks of one communicator clique issued different numbers of collectives\n"); return ncclInvalidUsage; }
    for (size_t k = 0; k < nOps; ++k) {
      const Op& o0 = g_pending[(*cl)[0]][k];
      for (FakeComm* c : *cl) { const Op& o = g_pending[c][k]; if (o.kind != o0.kind || o.count != o0.count || o.root != o0.root) { std::fprintf(stderr, "fake_hip: mismatched collectives across ranks\n"); return ncclInvalidUsage; } }
      const size_t bytes = o0.count * 8;
      if (o0.kind == 0) {  // all-gather: rank r's send buffer lands at offset r * count of every receive buffer
        for (FakeComm* dst : *cl)
          for (FakeComm* src : *cl) copy((char*)g_pending[dst][k].recv + (size_t)src->rank * bytes, g_pending[src][k].send, bytes, hipMemcpyDeviceToDevice);
      } else {             // broadcast from root
        const void* s = g_pending[(*cl)[o0.root]][k].send;
        if (!s) { std::fprintf(stderr, "fake_hip: ncclBroadcast root passed a null send buffer\n"); return ncclInvalidArgument; }
        for (FakeComm* dst : *cl) copy(g_pending[dst][k].recv, s, bytes, hipMemcpyDeviceToDevice);
      }
    }
    for (FakeComm* c : *cl) g_pending[c].clear();
  }
  return ncclSuccess;
}
ncclResult_t enqueue(ncclComm_t comm, Op op) {
  std::lock_guard<std::mutex> lk(g_nccl_mu);
  FakeComm* c = (FakeComm*)comm;
  g_pending[c].push_back(op);
  if (g_group > 0) return ncclSuccess;
  for (FakeComm* o : *c->clique) if (g_pending[o].size() < g_pending[c].size()) return ncclSuccess;  // wait for the other ranks' calls
  return flush();
}
}  // namespace
extern "C" {
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
  if (n < 1 || n > n_devices()) return ncclInvalidArgument;
  auto* clique = new std::vector<FakeComm*>();
  for (int r = 0; r < n; ++r) {
    if (devs && (devs[r] < 0 || devs[r] >= n_devices())) return ncclInvalidArgument;
    auto* c = new FakeComm{r, n, clique};
    clique->push_back(c);
    comms[r] = (ncclComm_t)c;
  }
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  std::lock_guard<std::mutex> lk(g_nccl_mu);
  FakeComm* c = (FakeComm*)comm;
  g_pending.erase(c);
  delete c;  // (the clique vector is leaked: a handful of pointers per communicator set)
  return ncclSuccess;
}
ncclResult_t ncclGroupStart() { std::lock_guard<std::mutex> lk(g_nccl_mu); ++g_group; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { std::lock_guard<std::mutex> lk(g_nccl_mu); if (--g_group > 0) return ncclSuccess; return flush(); }
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t) {
  if (t != ncclDouble) return ncclInvalidArgument;
  return enqueue(comm, Op{0, send, recv, count, 0});
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t) {
  if (t != ncclDouble) return ncclInvalidArgument;
  return enqueue(comm, Op{1, send, recv, count, root});
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error"; }
}  // extern "C"

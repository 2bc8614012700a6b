// ode_multigpu.hip — reassembling a sharded state tensor on every GPU of one node with RCCL over xGMI (one process, G
// devices).  The batch of independent IVPs shards as contiguous index ranges (nothing couples trajectories: one
// solveODE call per IVP in the reference, ode.nim:589), so this all-gather of the per-GPU results is the only collective
// the path ever needs.  RCCL is bound lazily with dlopen("librccl.so.1"): the library has no hard dependency on it, and
// inside a PyTorch process the already-loaded RCCL is reused (same SONAME) instead of a second copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <mutex>
#include <vector>

#include "../../include/nnhip_ode.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
};

std::mutex g_mu;
Rccl g_rccl;
std::vector<ncclComm_t> g_comms;  // one communicator per device, for g_comm_gpus devices
int g_comm_gpus = 0;
thread_local char g_mg_err[256] = "";

bool load_rccl() {
  if (g_rccl.ok) return true;
  g_rccl.handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.handle) g_rccl.handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!g_rccl.handle) { snprintf(g_mg_err, sizeof(g_mg_err), "cannot load RCCL: %s", dlerror()); return false; }
#define BIND(name)                                                                                         \
  g_rccl.name = reinterpret_cast<decltype(g_rccl.name)>(dlsym(g_rccl.handle, "nccl" #name));                \
  if (!g_rccl.name) { snprintf(g_mg_err, sizeof(g_mg_err), "RCCL lacks nccl" #name); return false; }
  BIND(CommInitAll) BIND(CommDestroy) BIND(GroupStart) BIND(GroupEnd) BIND(AllGather) BIND(Broadcast) BIND(GetErrorString)
#undef BIND
  g_rccl.ok = true;
  return true;
}

}  // namespace

namespace nnhip {
void multigpu_release() {  // nnhip_release(): destroy the cached RCCL communicators (rebuilt on the next collective)
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_rccl.ok) for (ncclComm_t c : g_comms) if (c) g_rccl.CommDestroy(c);
  g_comms.clear();
  g_comm_gpus = 0;
}
}  // namespace nnhip

extern "C" const char* nnhip_multigpu_last_error(void) { return g_mg_err; }

// shard[r] (on device r): this device's IVP index range [lo_r, lo_r + counts[r]) of a state tensor in `layout`;
// full[r] (on device r): the whole tensor [dim][N] (SoA) / [N][dim] (AoS), N = sum(counts).  streams[r] (nullable array)
// is the stream on device r whose prior work produced shard[r]; the collective is enqueued on it.
extern "C" int nnhip_allgather_states_f64_dev(int n_gpus, const double* const* shard, const int64_t* counts, int dim, int layout,
                                              double* const* full, void* const* streams) {
  if (n_gpus < 1 || !shard || !counts || !full || dim < 1 || (layout != NNHIP_LAYOUT_SOA && layout != NNHIP_LAYOUT_AOS)) {
    snprintf(g_mg_err, sizeof(g_mg_err), "allgather_states: need n_gpus >= 1, dim >= 1, a layout of NNHIP_LAYOUT_SOA / _AOS and non-NULL shard / counts / full");
    return NNHIP_EVALUE;
  }
  for (int r = 0; r < n_gpus; ++r)
    if (counts[r] < 0 || !full[r] || (counts[r] > 0 && !shard[r])) { snprintf(g_mg_err, sizeof(g_mg_err), "allgather_states: counts[%d] < 0, or shard[%d] / full[%d] is NULL", r, r, r); return NNHIP_EVALUE; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < n_gpus) { snprintf(g_mg_err, sizeof(g_mg_err), "need %d HIP devices, have %d", n_gpus, ndev); return NNHIP_EHIP; }
  std::lock_guard<std::mutex> lk(g_mu);
  if (!load_rccl()) return NNHIP_EHIP;
  if (g_comm_gpus != n_gpus) {
    for (ncclComm_t c : g_comms) g_rccl.CommDestroy(c);
    g_comms.assign((size_t)n_gpus, nullptr);
    std::vector<int> devs((size_t)n_gpus);
    for (int r = 0; r < n_gpus; ++r) devs[r] = r;
    const ncclResult_t rc = g_rccl.CommInitAll(g_comms.data(), n_gpus, devs.data());
    if (rc != ncclSuccess) { snprintf(g_mg_err, sizeof(g_mg_err), "ncclCommInitAll: %s", g_rccl.GetErrorString(rc)); g_comms.clear(); g_comm_gpus = 0; return NNHIP_EHIP; }
    g_comm_gpus = n_gpus;
  }
  int64_t N = 0;
  bool equal = true;
  std::vector<int64_t> lo((size_t)n_gpus);
  for (int r = 0; r < n_gpus; ++r) { lo[r] = N; N += counts[r]; equal = equal && counts[r] == counts[0]; }
  int prev = 0;
  (void)hipGetDevice(&prev);
  // AoS: every shard is one contiguous block of the full tensor -> one all-gather (equal shards) or one broadcast per
  // shard.  SoA: the tensor is `dim` planes of N; each plane is gathered on its own (plane-by-plane).
  const int planes = layout == NNHIP_LAYOUT_SOA ? dim : 1;
  const int64_t width = layout == NNHIP_LAYOUT_SOA ? 1 : dim;
  ncclResult_t rc = g_rccl.GroupStart();
  for (int p = 0; p < planes && rc == ncclSuccess; ++p) {
    for (int r = 0; r < n_gpus && rc == ncclSuccess; ++r) {  // calls of device r
      hipStream_t s = streams ? (hipStream_t)streams[r] : nullptr;
      const double* sendPlane = shard[r] + (int64_t)p * counts[r] * width;
      double* recvPlane = full[r] + (int64_t)p * N * width;
      if (equal) {
        rc = g_rccl.AllGather(sendPlane, recvPlane, (size_t)(counts[r] * width), ncclDouble, g_comms[r], s);
      } else {
        for (int root = 0; root < n_gpus && rc == ncclSuccess; ++root) {
          if (counts[root] == 0) continue;  // an empty shard (fewer IVPs than devices) contributes nothing: no zero-length collective is issued
          rc = g_rccl.Broadcast(root == r ? sendPlane : nullptr, recvPlane + lo[root] * width, (size_t)(counts[root] * width), ncclDouble, root,
                                g_comms[r], s);
        }
      }
    }
  }
  const ncclResult_t rcEnd = g_rccl.GroupEnd();
  (void)hipSetDevice(prev);
  if (rc != ncclSuccess || rcEnd != ncclSuccess) {
    snprintf(g_mg_err, sizeof(g_mg_err), "RCCL collective failed: %s", g_rccl.GetErrorString(rc != ncclSuccess ? rc : rcEnd));
    return NNHIP_EHIP;
  }
  return NNHIP_OK;
}

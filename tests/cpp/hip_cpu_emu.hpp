// hip_cpu_emu.hpp — TEST INFRASTRUCTURE ONLY.  The device vocabulary of numericalnim_amd/csrc/ode_device.hpp / ode_kernels.hpp on the host, so that
// the kernel BODIES can be executed by the CPU test suite (tests/test_kernel_bodies_on_cpu.py): every lane of a workgroup is a coroutine (or, in the sanitizer
// builds, a host thread) that runs concurrently with the others, workgroups run one after the other.  Nothing here is linked into libnnhip_ode.so, nothing in the product includes it (the headers reach it only under
// -DNNHIP_CPU_EMU, which no product build defines); the library still has no CPU path and fails without a HIP device.
// NNHIP_CPU_EMU is a COMPILE-TIME TEST HOOK: no library build defines it, no entry of the C ABI dispatches on it, it must never become a selectable path — and
// tests/test_abi_and_host_logic.py::test_no_emulation_symbol_in_the_product_library checks that libnnhip_ode.so's symbol tables hold no `hipemu` symbol and that neither the library's Makefile nor anything the package imports names the macro.
//
// Why it exists: a round without GPU access (round 5) still had to show that new kernels compute the reference's bits.  What it shows: the indexing,
// masking, control flow and arithmetic of a kernel body as written (compiled -ffp-contract=off: one IEEE rounding per operation, like the device build).
// What it does not: anything about the gfx950 code the real compiler emits (scheduling, register allocation, memory model) — that is the GPU suite's.
//
// Cross-lane operations (DPP moves, ds_bpermute, shuffles): each lane publishes its operand in its own slot sequence and reads the source lane's slot of
// the same sequence number — lanes that exchange data execute the same sequence of cross-lane operations (they belong to one system of the
// lanes-per-system kernels), lanes that do not never wait for each other, so divergent systems inside a wavefront need no EXEC-mask model.
// LDS shared by the lanes of one system (wave_lds_sync<L>: the stage vector of non-banded right-hand sides, the ordered LDS error sum of systems wider than
// four lanes) is a function-scope static here, and the wavefront's lock-step a rendezvous of those L lanes (group_sync).
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __shared__ static  // function-scope arrays: one per kernel instantiation; workgroups run one after the other
#define __launch_bounds__(...)
#define amdgpu_waves_per_eu(...)  // __attribute__((amdgpu_waves_per_eu(f<A, B>()))) -> __attribute__(()): an occupancy hint of the target, and g++ cannot parse its argument

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct alignas(16) double2 { double x, y; };
inline double2 make_double2(double x, double y) { return double2{x, y}; }

namespace hipemu {
constexpr int kSlots = 1 << 14;  // cross-lane operations one lane may execute per launch
struct Block;
struct Tl {
  dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
  Block* block = nullptr;
};

// ---- two engines for "every lane of a workgroup runs concurrently" ----------------------------------------------------------------------------------
// default: FIBERS (ucontext) — the lanes of a workgroup are coroutines of the calling thread, switched round-robin wherever a lane has to wait for another
// (a cross-lane exchange whose source has not published yet, a barrier).  No OS threads, no futexes: a launch costs microseconds, the schedule is
// deterministic, and a workgroup in which every live lane waits is reported as a deadlock at once instead of hanging.
// -DNNHIP_EMU_THREADS: one host thread per lane (a pool reused across launches) — what the sanitizer builds use (ASan does not follow swapcontext).
#ifndef NNHIP_EMU_THREADS
}  // namespace hipemu
#include <ucontext.h>
namespace hipemu {
struct Fiber {
  ucontext_t ctx;
  Tl tl;
  bool done = true;
};
struct Sched {
  static constexpr size_t kStack = 512 * 1024;
  ucontext_t main;
  std::vector<std::unique_ptr<Fiber>> fibers;
  std::vector<std::unique_ptr<char[]>> stacks;
  std::function<void(unsigned)> job;
  unsigned cur = 0;
  size_t progress = 0;  // bumped by everything another lane may be waiting for
  static Sched& get() { static Sched s; return s; }
  static void entry() {
    Sched& s = get();
    const unsigned me = s.cur;
    s.job(me);
    s.fibers[me]->done = true;
    ++s.progress;
    // returning switches to uc_link (= main)
  }
  template <class Job>
  void run(unsigned n, Job&& j) {
    while (fibers.size() < n) { fibers.emplace_back(new Fiber); stacks.emplace_back(new char[kStack]); }
    job = [&j](unsigned t) { j(t); };
    for (unsigned t = 0; t < n; ++t) {
      Fiber& f = *fibers[t];
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = stacks[t].get();
      f.ctx.uc_stack.ss_size = kStack;
      f.ctx.uc_link = &main;
      makecontext(&f.ctx, &Sched::entry, 0);
      f.done = false;
    }
    unsigned live = n;
    while (live) {
      const size_t before = progress;
      live = 0;
      for (unsigned t = 0; t < n; ++t) {
        if (fibers[t]->done) continue;
        cur = t;
        swapcontext(&main, &fibers[t]->ctx);
        if (!fibers[t]->done) ++live;
      }
      if (live && progress == before) {
        fprintf(stderr, "hip_cpu_emu: deadlock — %u lane(s) of a workgroup wait for lanes that never arrive (divergent barrier or exchange)\n", live);
        abort();
      }
    }
  }
  void yield() { swapcontext(&fibers[cur]->ctx, &main); }
};
inline Tl& cur() { Sched& s = Sched::get(); return s.fibers[s.cur]->tl; }
inline void wait_hint() { Sched::get().yield(); }
inline void made_progress() { ++Sched::get().progress; }
template <class Job>
inline void run_lanes(unsigned n, Job&& job) { Sched::get().run(n, job); }
inline void set_lane(unsigned t, const Tl& v) { Sched::get().fibers[t]->tl = v; }
#else
inline thread_local Tl tl_;
inline Tl& cur() { return tl_; }
inline void wait_hint() { std::this_thread::yield(); }
inline void made_progress() {}
// The host threads that play the lanes: started once per process and reused by every workgroup of every launch.  Every worker sleeps on its own word (one
// futex wake per lane and workgroup, no shared lock to fight over), the launcher on the count of lanes still running.
class LanePool {
 public:
  static LanePool& get() { static LanePool p; return p; }
  template <class Job>
  void run(unsigned n, Job&& job) {
    while (workers_.size() < n) {
      workers_.emplace_back(new Worker);
      Worker* w = workers_.back().get();
      const unsigned id = (unsigned)workers_.size() - 1;
      w->th = std::thread([this, w, id]() { loop(w, id); });
    }
    job_ = [&job](unsigned t) { job(t); };
    pending_.store(n, std::memory_order_release);
    for (unsigned t = 0; t < n; ++t) {
      workers_[t]->go.fetch_add(1, std::memory_order_release);
      workers_[t]->go.notify_one();
    }
    for (unsigned left = pending_.load(std::memory_order_acquire); left != 0; left = pending_.load(std::memory_order_acquire)) pending_.wait(left, std::memory_order_acquire);
  }
  ~LanePool() {
    stop_.store(true);
    for (auto& w : workers_) { w->go.fetch_add(1); w->go.notify_one(); }
    for (auto& w : workers_) w->th.join();
  }

 private:
  struct Worker {
    std::thread th;
    std::atomic<uint32_t> go{0};
  };
  void loop(Worker* w, unsigned id) {
    uint32_t seen = 0;
    for (;;) {
      w->go.wait(seen, std::memory_order_acquire);
      seen = w->go.load(std::memory_order_acquire);
      if (stop_.load()) return;
      job_(id);
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) pending_.notify_all();
    }
  }
  std::vector<std::unique_ptr<Worker>> workers_;
  std::function<void(unsigned)> job_;
  std::atomic<unsigned> pending_{0};
  std::atomic<bool> stop_{false};
};
template <class Job>
inline void run_lanes(unsigned n, Job&& job) { LanePool::get().run(n, job); }
#endif

struct Lane {
  std::atomic<size_t> count{0};
  std::unique_ptr<uint64_t[]> slot{new uint64_t[kSlots]};
};
struct Block {
  unsigned nThreads = 0;
  std::vector<Lane> lanes;
  std::mutex mu;
  std::condition_variable cv;
  unsigned arrived = 0, generation = 0;
  int orAcc = 0, orResult = 0;
  explicit Block(unsigned n) : nThreads(n), lanes(n) {}
  int barrier_or(int v) {  // every thread of the workgroup calls it
#ifndef NNHIP_EMU_THREADS
    orAcc |= v;
    const unsigned gen = generation;
    made_progress();
    if (++arrived == nThreads) { orResult = orAcc; orAcc = 0; arrived = 0; ++generation; }
    else while (generation == gen) wait_hint();
    return orResult;
#else
    std::unique_lock<std::mutex> lk(mu);
    orAcc |= v;
    const unsigned gen = generation;
    if (++arrived == nThreads) {
      orResult = orAcc; orAcc = 0; arrived = 0; ++generation;
      cv.notify_all();
    } else {
      cv.wait(lk, [&] { return generation != gen; });
    }
    return orResult;
#endif
  }
};

// publish `v`, return what lane `src` (index inside the workgroup) published at the same sequence number
inline uint64_t exchange(uint64_t v, unsigned src) {
  Block& b = *cur().block;
  Lane& me = b.lanes[cur().threadIdx_.x];
  const size_t k = me.count.load(std::memory_order_relaxed);
  if (k >= (size_t)kSlots) { fprintf(stderr, "hip_cpu_emu: too many cross-lane operations in one launch\n"); abort(); }
  me.slot[k] = v;
  me.count.store(k + 1, std::memory_order_release);
  made_progress();
  Lane& s = b.lanes[src];
  while (s.count.load(std::memory_order_acquire) <= k) wait_hint();
  return s.slot[k];
}
// rendezvous of the L consecutive lanes (L a power of two) this lane belongs to: everything they wrote before is visible to all of them afterwards
inline void group_sync(int L) {
  Block& b = *cur().block;
  const unsigned me = cur().threadIdx_.x, first = me & ~(unsigned)(L - 1);
  Lane& mine = b.lanes[me];
  const size_t k = mine.count.load(std::memory_order_relaxed);
  if (k >= (size_t)kSlots) { fprintf(stderr, "hip_cpu_emu: too many cross-lane operations in one launch\n"); abort(); }
  mine.slot[k] = 0;
  mine.count.store(k + 1, std::memory_order_release);
  made_progress();
  for (unsigned l = first; l < first + (unsigned)L; ++l)
    while (b.lanes[l].count.load(std::memory_order_acquire) <= k) wait_hint();
}
inline unsigned lane_in_block(unsigned laneInWave) { return (cur().threadIdx_.x & ~63u) | (laneInWave & 63u); }
inline int update_dpp(int /*old*/, int src, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
  const unsigned lane = cur().threadIdx_.x & 63u;
  unsigned from;
  if (ctrl >= 0 && ctrl <= 0xFF) from = (lane & ~3u) | ((unsigned)(ctrl >> (2 * (lane & 3u))) & 3u);     // quad_perm
  else if (ctrl >= 0x121 && ctrl <= 0x12F) from = (lane & ~15u) | ((lane - (unsigned)(ctrl - 0x120)) & 15u);  // row_ror:n — lane i reads lane (i - n) mod 16
  else { fprintf(stderr, "hip_cpu_emu: DPP control 0x%x is not modelled\n", ctrl); abort(); }
  return (int)(uint32_t)exchange((uint32_t)src, lane_in_block(from));
}
inline int ds_bpermute(int addr, int v) { return (int)(uint32_t)exchange((uint32_t)v, lane_in_block((unsigned)(addr >> 2))); }
template <class T>
inline T shfl_from(T v, unsigned laneInWave) {
  static_assert(sizeof(T) <= 8, "");
  uint64_t u = 0;
  memcpy(&u, &v, sizeof(T));
  u = exchange(u, lane_in_block(laneInWave));
  T r;
  memcpy(&r, &u, sizeof(T));
  return r;
}
inline std::mutex& atomics_mu() { static std::mutex m; return m; }

// one launch: workgroups one after the other, every lane of a workgroup live at the same time
template <class Kernel, class... Args>
void launch(Kernel kernel, dim3 grid, dim3 block, Args... args) {
  for (unsigned by = 0; by < grid.y; ++by)
    for (unsigned b = 0; b < grid.x; ++b) {
      Block ctx(block.x);
      run_lanes(block.x, [&](unsigned t) {
        Tl& me = cur();
        me.threadIdx_ = dim3(t); me.blockIdx_ = dim3(b, by); me.blockDim_ = block; me.gridDim_ = grid; me.block = &ctx;
        kernel(args...);
      });
    }
}
}  // namespace hipemu

#define threadIdx (hipemu::cur().threadIdx_)
#define blockIdx (hipemu::cur().blockIdx_)
#define blockDim (hipemu::cur().blockDim_)
#define gridDim (hipemu::cur().gridDim_)

#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu::update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_ds_bpermute(addr, v) hipemu::ds_bpermute((addr), (v))
#define __builtin_amdgcn_readfirstlane(x) (x)  // uniform by construction wherever the kernels use it
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_logf(x) log2f(x)
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
inline int __all(int p) { return p; }  // (with readfirstlane(x) = x every lane is its own "uniform" wavefront: the per-lane path of the callers)
inline int __syncthreads_or(int v) { return hipemu::cur().block->barrier_or(v); }
inline void __syncthreads() { (void)hipemu::cur().block->barrier_or(0); }
template <class T> inline T __shfl_down(T v, unsigned off, int = 64) { const unsigned l = threadIdx.x & 63u; return hipemu::shfl_from(v, l + off < 64 ? l + off : l); }
template <class T> inline T __shfl_up(T v, unsigned off, int = 64) { const unsigned l = threadIdx.x & 63u; return hipemu::shfl_from(v, l >= off ? l - off : l); }
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return hipemu::shfl_from(v, (threadIdx.x & 63u) ^ (unsigned)mask); }
inline int __clzll(long long x) { return x == 0 ? 64 : __builtin_clzll((unsigned long long)x); }
inline int __double2loint(double v) { uint64_t u; memcpy(&u, &v, 8); return (int)(uint32_t)u; }
inline int __double2hiint(double v) { uint64_t u; memcpy(&u, &v, 8); return (int)(uint32_t)(u >> 32); }
inline double __hiloint2double(int hi, int lo) { const uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double v; memcpy(&v, &u, 8); return v; }
inline double __longlong_as_double(long long x) { double v; memcpy(&v, &x, 8); return v; }
inline long long __double_as_longlong(double v) { long long x; memcpy(&x, &v, 8); return x; }
template <class T> inline T atomicAdd(T* p, T v) { std::lock_guard<std::mutex> lk(hipemu::atomics_mu()); const T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMax(T* p, T v) { std::lock_guard<std::mutex> lk(hipemu::atomics_mu()); const T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { std::lock_guard<std::mutex> lk(hipemu::atomics_mu()); const T o = *p; if (v < o) *p = v; return o; }

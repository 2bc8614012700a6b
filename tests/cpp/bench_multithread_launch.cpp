// bench_multithread_launch.cpp — how many kernel launches per second do G host threads sustain through ONE HIP runtime?
//
// nnhip_ode_fixed_stream_multi_gpu_f64_dev drives each device from its own std::thread; at config C5 every thread issues 1000 launches of a
// ~23 us kernel per solve, i.e. 8 threads x 44 k launches/s.  Whether the runtime serialises them can be measured on ONE GPU: G threads, each
// with its own stream and its own shard on device 0 — the very worker of the multi-GPU entry (nnhip_ode_fixed_stream_f64_dev), eager and
// replayed from a hipGraph.  Two shard sizes:
//   c5      1e7 / G IVPs per thread: C5's launch count with the GPU-side work of ONE device shared by the G streams
//   tiny    1024 IVPs per thread: the kernel is a few microseconds, so the host side is what is measured
// Prints one JSON object.   usage: bench_multithread_launch [--rk4-steps S] [--reps R]
// Built by tests/test_gpu_cpp_host.py (g++ + libamdhip64 + libnnhip_ode.so; no PyTorch in the process).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "nnhip_ode.h"

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Result { double wall_s = 0, worst_thread_s = 0; int ok = 1; };

// G threads, each: `reps` solves of S RK4 steps over its own n-IVP shard on its own stream (device 0)
static Result run(int G, int64_t n, int S, int reps, int graph) {
  Result res;
  const double dt = std::ldexp(1.0, -10);
  nnhip_ode_options opt;
  nnhip_ode_new_options(&opt, dt, 1e-4, 1e-4, 1e-2, 1e-4, 4.0, 0.1, 0.0);
  nnhip_tune_set("stream_graph", graph ? 1 : 0);
  std::vector<double*> y(G), sc(G);
  std::vector<hipStream_t> st(G);
  std::vector<double> h((size_t)n, 1.0);
  for (int g = 0; g < G; ++g) {
    if (hipMalloc((void**)&y[g], (size_t)n * 8) != hipSuccess || hipMalloc((void**)&sc[g], (size_t)n * 8) != hipSuccess ||
        hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking) != hipSuccess || hipMemcpy(y[g], h.data(), (size_t)n * 8, hipMemcpyHostToDevice) != hipSuccess) { res.ok = 0; return res; }
  }
  std::atomic<int> ready{0}, go{0}, bad{0};
  std::vector<double> tsec(G, 0.0);
  auto work = [&](int g) {
    (void)hipSetDevice(0);
    int64_t nst = 0;
    double* fin = nullptr;
    // warm-up: first sight / capture of this (buffer, stream) pair
    for (int k = 0; k < 2; ++k)
      if (nnhip_ode_fixed_stream_f64_dev(&opt, NNHIP_RK4, NNHIP_RHS_NEG_Y, nullptr, 0, n, 1, NNHIP_LAYOUT_SOA, 0.0, S * dt, y[g], sc[g], &nst, &fin, st[g]) != NNHIP_OK) bad++;
    (void)hipStreamSynchronize(st[g]);
    ready++;
    while (!go.load()) std::this_thread::yield();
    const double t0 = now_s();
    for (int k = 0; k < reps; ++k)
      if (nnhip_ode_fixed_stream_f64_dev(&opt, NNHIP_RK4, NNHIP_RHS_NEG_Y, nullptr, 0, n, 1, NNHIP_LAYOUT_SOA, 0.0, S * dt, y[g], sc[g], &nst, &fin, st[g]) != NNHIP_OK || nst != S) bad++;
    (void)hipStreamSynchronize(st[g]);
    tsec[g] = now_s() - t0;
  };
  std::vector<std::thread> th;
  for (int g = 0; g < G; ++g) th.emplace_back(work, g);
  while (ready.load() < G) std::this_thread::yield();
  const double t0 = now_s();
  go = 1;
  for (auto& t : th) t.join();
  res.wall_s = now_s() - t0;
  for (double t : tsec) res.worst_thread_s = t > res.worst_thread_s ? t : res.worst_thread_s;
  res.ok = bad.load() == 0;
  nnhip_release();  // drops the cached graphs of these buffers before they are freed
  for (int g = 0; g < G; ++g) { (void)hipFree(y[g]); (void)hipFree(sc[g]); (void)hipStreamDestroy(st[g]); }
  return res;
}

int main(int argc, char** argv) {
  int S = 1000, reps = 5;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--rk4-steps") && i + 1 < argc) S = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--reps") && i + 1 < argc) reps = std::atoi(argv[++i]);
  }
  if (hipSetDevice(0) != hipSuccess) { std::fprintf(stderr, "no HIP device\n"); return 2; }
  std::string out = "{";
  int ok = 1;
  const char* shapes[2] = {"c5", "tiny"};
  for (int shape = 0; shape < 2; ++shape) {
    for (int graph = 0; graph < 2; ++graph) {
      double base = 0;
      for (int G : {1, 2, 4, 8}) {
        const int64_t n = shape == 0 ? 10000000 / G : 1024;
        const Result r = run(G, n, S, reps, graph);
        ok = ok && r.ok;
        const double launches = (double)G * reps * S;            // kernel launches (eager) / kernel nodes (graph)
        const double solve_ms = r.wall_s * 1e3 / reps;            // one solve of every thread's shard, all threads concurrently
        if (G == 1) base = solve_ms;
        char buf[512];
        std::snprintf(buf, sizeof buf,
                      "%s\"%s_%s_G%d\": {\"ivps_per_thread\": %lld, \"solve_ms\": %.4f, \"vs_G1\": %.3f, \"launches_per_s_aggregate\": %.0f, \"launches_per_s_per_thread\": %.0f, "
                      "\"us_per_launch_per_thread\": %.3f}",
                      out.size() > 1 ? ", " : "", shapes[shape], graph ? "graph" : "eager", G, (long long)n, solve_ms, solve_ms / base, launches / r.wall_s,
                      launches / r.wall_s / G, r.worst_thread_s * 1e6 / ((double)reps * S));
        out += buf;
      }
    }
  }
  out += ", \"rk4_steps_per_solve\": " + std::to_string(S) + ", \"reps\": " + std::to_string(reps) + ", \"c5_needs_per_thread\": 44000, \"ok\": " + (ok ? "true" : "false") + "}";
  std::printf("%s\n", out.c_str());
  return ok ? 0 : 1;
}

// bench_c5.cpp — BASELINE.json config C5 from a process WITHOUT PyTorch: one process, G devices, every device a resident shard of
// dy/dt = -y states, RK4 step-streaming solve per device and the RCCL reassembly of the final states on every device, all behind ONE
// C call (nnhip_ode_fixed_stream_multi_gpu_f64_dev) — what a compiled or Nim host runs.  Prints one JSON line with the keys of
// bench.py (whole-job trajectory-steps per second, weak scaling: n_per_gpu IVPs per device).
//   usage: bench_c5 [--gpus G] [--steps K] [--warmup W] [--n-per-gpu N] [--rk4-steps S] [--verify]
// Built by tests/test_gpu_cpp_host.py (g++ + libamdhip64 + libnnhip_ode.so); the gather of solve k runs on its own stream under solve k+1
// (three state buffers rotate per device so that a shard being gathered is never overwritten), as in bench.py.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nnhip_ode.h"

#define CK(x) do { if ((x) != hipSuccess) { std::fprintf(stderr, "HIP error at line %d\n", __LINE__); return 2; } } while (0)
#define NK(x) do { if ((x) != NNHIP_OK) { std::fprintf(stderr, "nnhip error at line %d: %s\n", __LINE__, nnhip_last_error()); return 3; } } while (0)

int main(int argc, char** argv) {
  int G = 1, K = 5, W = 2, S = 1000, verify = 0;
  int64_t n = 10000000;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--gpus") && i + 1 < argc) G = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--steps") && i + 1 < argc) K = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--warmup") && i + 1 < argc) W = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--n-per-gpu") && i + 1 < argc) n = std::atoll(argv[++i]);
    else if (!std::strcmp(argv[i], "--rk4-steps") && i + 1 < argc) S = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--verify")) verify = 1;
  }
  int ndev = 0;
  CK(hipGetDeviceCount(&ndev));
  if (G < 1 || G > ndev) { std::fprintf(stderr, "--gpus %d but %d device(s)\n", G, ndev); return 1; }
  const int64_t N = n * G;
  const double dt = std::ldexp(1.0, -10);  // C2's dt = 2^-10: t accumulates exactly, S steps end at S * dt
  nnhip_ode_options opt;
  NK(nnhip_ode_new_options(&opt, dt, 1e-4, 1e-4, 1e-2, 1e-4, 4.0, 0.1, 0.0));
  std::vector<int64_t> counts(G, n);
  std::vector<double*> buf[3], full(G);
  std::vector<void*> streams(G), gstreams(G);
  for (int b = 0; b < 3; ++b) buf[b].resize(G);
  std::vector<double> h((size_t)n);
  for (int r = 0; r < G; ++r) {
    CK(hipSetDevice(r));
    for (int b = 0; b < 3; ++b) CK(hipMalloc((void**)&buf[b][r], (size_t)n * 8));
    CK(hipMalloc((void**)&full[r], (size_t)N * 8));
    hipStream_t s, g;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&g, hipStreamNonBlocking));
    streams[r] = s; gstreams[r] = g;
    for (int64_t i = 0; i < n; ++i) h[(size_t)i] = 1.0 + (double)((r * n + i) % (1 << 20)) * std::ldexp(1.0, -20);  // SURVEY 8d, C2's y0
    CK(hipMemcpy(buf[0][r], h.data(), (size_t)n * 8, hipMemcpyHostToDevice));
  }
  std::vector<hipEvent_t> gathered[3];  // gathered[b][r]: the last gather that read buf[b][r] has finished
  for (int b = 0; b < 3; ++b) {
    gathered[b].resize(G);
    for (int r = 0; r < G; ++r) { CK(hipSetDevice(r)); CK(hipEventCreateWithFlags(&gathered[b][r], hipEventDisableTiming)); CK(hipEventRecord(gathered[b][r], (hipStream_t)gstreams[r])); }
  }
  NK(nnhip_tune_set("stream_graph", 1));  // the S launches of a solve replay as one graph per device
  auto sync_all = [&]() -> int {
    for (int r = 0; r < G; ++r) { if (hipSetDevice(r) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 1; }
    return 0;
  };
  // solve k integrates buf[k % 3] in place... the state keeps marching (as bench.py): solve k reads/writes `cur`, its final shard is
  // gathered on the gather stream while solve k + 1 works on a COPY in the next buffer
  int cur = 0;
  auto one_solve = [&](int k) -> int {
    (void)k;
    const int nxt = (cur + 1) % 3;
    std::vector<double*> y(G), fin(G);
    for (int r = 0; r < G; ++r) y[r] = buf[cur][r];
    int64_t nst = 0;
    const int rc = nnhip_ode_fixed_stream_multi_gpu_f64_dev(&opt, NNHIP_RK4, NNHIP_RHS_NEG_Y, nullptr, 0, G, counts.data(), 1, NNHIP_LAYOUT_SOA, 0.0, S * dt, y.data(),
                                                            nullptr, full.data(), streams.data(), gstreams.data(), &nst, fin.data());
    if (rc != NNHIP_OK || nst != S) return 1;
    for (int r = 0; r < G; ++r) {  // the next solve starts from this one's result, in another buffer (device-to-device, on the solve stream)
      if (hipSetDevice(r) != hipSuccess || hipEventRecord(gathered[cur][r], (hipStream_t)gstreams[r]) != hipSuccess ||   // this solve's gather reads buf[cur]
          hipStreamWaitEvent((hipStream_t)streams[r], gathered[nxt][r], 0) != hipSuccess ||                              // buf[nxt] must have been gathered
          hipMemcpyAsync(buf[nxt][r], fin[r], (size_t)n * 8, hipMemcpyDeviceToDevice, (hipStream_t)streams[r]) != hipSuccess) return 1;
    }
    cur = nxt;
    return 0;
  };
  for (int k = 0; k < W; ++k) if (one_solve(k)) { std::fprintf(stderr, "solve failed: %s\n", nnhip_last_error()); return 4; }
  if (sync_all()) return 5;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < K; ++k) if (one_solve(k)) { std::fprintf(stderr, "solve failed: %s\n", nnhip_last_error()); return 4; }
  if (sync_all()) return 5;
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int ok = 1;
  if (verify) {  // every device's full tensor = the shards of the LAST solve, which are the states after (W + K) * S RK4 steps of y' = -y
    std::vector<double> f0((size_t)N), fr((size_t)N);
    CK(hipSetDevice(0));
    CK(hipMemcpy(f0.data(), full[0], (size_t)N * 8, hipMemcpyDeviceToHost));
    for (int r = 1; r < G; ++r) {
      CK(hipSetDevice(r));
      CK(hipMemcpy(fr.data(), full[r], (size_t)N * 8, hipMemcpyDeviceToHost));
      if (std::memcmp(f0.data(), fr.data(), (size_t)N * 8) != 0) ok = 0;
    }
    const double T = (double)(W + K) * S * dt;
    for (int64_t i = 0; i < N; i += 9973) {
      const double y0 = 1.0 + (double)(i % (1 << 20)) * std::ldexp(1.0, -20);
      if (!(std::fabs(f0[(size_t)i] - y0 * std::exp(-T)) <= 1e-9 * y0)) ok = 0;
    }
  }
  const double value = (double)N * (double)S * (double)K / sec;
  std::printf("{\"metric\": \"RK4 trajectory-steps/sec on 1e7 float64 IVPs\", \"value\": %.6e, \"unit\": \"trajectory-steps/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, "
              "\"ms_per_step\": %.4f, \"higher_is_better\": true, \"scaling\": \"weak\", \"vs_baseline\": null, \"dtype\": \"f64\", \"data\": \"synthetic\", "
              "\"config\": {\"workload\": \"C5 via one C call (nnhip_ode_fixed_stream_multi_gpu_f64_dev): dy/dt=-y, RK4 dt=2^-10, %d steps per solve, %lld IVPs per GPU, "
              "RCCL all-gather of the final states on a second stream\", \"n_per_gpu\": %lld, \"host\": \"C++ (no PyTorch)\"}, \"verified\": %s}\n",
              value, G, K, W, sec * 1e3 / K, S, (long long)n, (long long)n, verify ? (ok ? "true" : "false") : "null");
  return ok ? 0 : 6;
}

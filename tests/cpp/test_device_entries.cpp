// Stand-alone (no PyTorch in the process) exercise of the device-pointer entries a compiled / Nim host would call with its own
// hipMalloc'ed buffers: fused solve, step-streaming loop (eager and hipGraph replay), adaptive streaming, the RCCL reassembly
// with the system's librccl, and a run-time compiled RHS.  Built with g++ against libamdhip64 by tests/test_gpu_cpp_host.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "nnhip_ode.h"

static int failures = 0;
#define CHECK(cond) do { if (!(cond)) { ++failures; std::printf("CHECK failed: %s (line %d): %s\n", #cond, __LINE__, nnhip_last_error()); } } while (0)

int main() {
  const int64_t N = 100000;
  std::vector<double> y0(N), out(2 * N), fin(N);
  for (int64_t i = 0; i < N; ++i) y0[i] = 1.0 + 1e-5 * (double)i;
  double *dY0, *dOut, *dY, *dScratch, *dFull;
  void* dWs;
  hipStream_t s;
  CHECK(hipStreamCreate(&s) == hipSuccess);
  CHECK(hipMalloc((void**)&dY0, N * 8) == hipSuccess && hipMalloc((void**)&dOut, 2 * N * 8) == hipSuccess && hipMalloc((void**)&dY, N * 8) == hipSuccess);
  CHECK(hipMalloc((void**)&dScratch, N * 8) == hipSuccess && hipMalloc((void**)&dFull, N * 8) == hipSuccess);
  CHECK(hipMalloc(&dWs, (size_t)nnhip_ode_adaptive_stream_workspace_bytes(N, 1) + (size_t)nnhip_ode_solve_workspace_bytes(2)) == hipSuccess);
  CHECK(hipMemcpy(dY0, y0.data(), N * 8, hipMemcpyHostToDevice) == hipSuccess);
  nnhip_ode_options opt;
  CHECK(nnhip_ode_new_options(&opt, 1.0 / 256, 1e-9, 1e-9, 1e-1, 1e-7, 4.0, 0.1, 0.0) == NNHIP_OK);
  const double tspan[2] = {0.0, 1.0};
  double tOut[2];
  // fused RK4 solve
  CHECK(nnhip_ode_solve_batch_f64_dev(&opt, NNHIP_RK4, NNHIP_RHS_NEG_Y, nullptr, 0, dY0, N, 1, NNHIP_LAYOUT_SOA, tspan, 2, tOut, dOut, nullptr, nullptr,
                                      nullptr, 0, dWs, nnhip_ode_solve_workspace_bytes(2), s) == NNHIP_OK);
  CHECK(hipMemcpyAsync(out.data(), dOut, 2 * N * 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess);
  CHECK(std::fabs(out[N] - std::exp(-1.0)) < 1e-9);
  // step-streaming loop: eager, then captured in a hipGraph and replayed — all three must agree bit for bit
  for (int graph = 0; graph < 2; ++graph) {
    CHECK(nnhip_tune_set("stream_graph", graph) == NNHIP_OK);
    for (int rep = 0; rep < 2; ++rep) {
      int64_t nSteps = 0;
      double* yFinal = nullptr;
      CHECK(hipMemcpyAsync(dY, dY0, N * 8, hipMemcpyDeviceToDevice, s) == hipSuccess);
      CHECK(nnhip_ode_fixed_stream_f64_dev(&opt, NNHIP_RK4, NNHIP_RHS_NEG_Y, nullptr, 0, N, 1, NNHIP_LAYOUT_SOA, 0.0, 1.0, dY, dScratch, &nSteps, &yFinal, s) == NNHIP_OK);
      CHECK(nSteps == 256 && yFinal == dY);
      CHECK(hipMemcpyAsync(fin.data(), yFinal, N * 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess);
      for (int64_t i = 0; i < N; i += 997) CHECK(fin[i] == out[N + i]);
    }
  }
  CHECK(nnhip_tune_set("stream_graph", 2) == NNHIP_OK);
  // adaptive solve streamed through HBM == fused adaptive solve
  CHECK(nnhip_ode_solve_batch_f64_dev(&opt, NNHIP_TSIT54, NNHIP_RHS_NEG_Y, nullptr, 0, dY0, N, 1, NNHIP_LAYOUT_SOA, tspan, 2, tOut, dOut, nullptr, nullptr,
                                      nullptr, 0, dWs, nnhip_ode_solve_workspace_bytes(2), s) == NNHIP_OK);
  CHECK(hipMemcpyAsync(out.data(), dOut, 2 * N * 8, hipMemcpyDeviceToHost, s) == hipSuccess);
  CHECK(hipMemcpyAsync(dY, dY0, N * 8, hipMemcpyDeviceToDevice, s) == hipSuccess);
  int64_t launches = 0;
  CHECK(nnhip_ode_adaptive_stream_f64_dev(&opt, NNHIP_TSIT54, NNHIP_RHS_NEG_Y, nullptr, 0, N, 1, NNHIP_LAYOUT_SOA, 0.0, 1.0, dY, dWs,
                                          nnhip_ode_adaptive_stream_workspace_bytes(N, 1), 4, 0, &launches, s) == NNHIP_OK);
  CHECK(hipMemcpyAsync(fin.data(), dY, N * 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess);
  CHECK(launches > 0);
  for (int64_t i = 0; i < N; i += 997) CHECK(fin[i] == out[N + i]);
  // RCCL reassembly with one device (librccl.so.1 from the system ROCm, loaded lazily)
  const double* shard[1] = {dY};
  double* full[1] = {dFull};
  const int64_t counts[1] = {N};
  void* streams[1] = {s};
  const int rc = nnhip_allgather_states_f64_dev(1, shard, counts, 1, NNHIP_LAYOUT_SOA, full, streams);
  if (rc != NNHIP_OK) std::printf("allgather: %s\n", nnhip_multigpu_last_error());
  CHECK(rc == NNHIP_OK);
  std::vector<double> g(N);
  CHECK(hipStreamSynchronize(s) == hipSuccess && hipMemcpy(g.data(), dFull, N * 8, hipMemcpyDeviceToHost) == hipSuccess);
  for (int64_t i = 0; i < N; i += 997) CHECK(g[i] == fin[i]);
  // host-pointer entry with page-locked buffers from the library's own allocator (chunked, overlapped transfers) == pageable buffers
  {
    const int64_t NB = 3000000;
    double *py0 = nullptr, *pout = nullptr;
    CHECK(nnhip_host_alloc((void**)&py0, NB * 8) == NNHIP_OK && nnhip_host_alloc((void**)&pout, 2 * NB * 8) == NNHIP_OK);
    std::vector<double> hy0(NB), hout(2 * NB);
    for (int64_t i = 0; i < NB; ++i) hy0[i] = py0[i] = 1.0 + 1e-7 * (double)i;
    nnhip_ode_stats st;
    CHECK(nnhip_ode_solve_batch_f64(&opt, NNHIP_RK4, NNHIP_RHS_NEG_Y, nullptr, 0, py0, NB, 1, NNHIP_LAYOUT_SOA, tspan, 2, tOut, pout, nullptr, nullptr, nullptr,
                                    0, &st, 0) == NNHIP_OK);
    CHECK(nnhip_ode_solve_batch_f64(&opt, NNHIP_RK4, NNHIP_RHS_NEG_Y, nullptr, 0, hy0.data(), NB, 1, NNHIP_LAYOUT_SOA, tspan, 2, tOut, hout.data(), nullptr,
                                    nullptr, nullptr, 0, &st, 0) == NNHIP_OK);
    int bad = 0;
    for (int64_t i = 0; i < 2 * NB; ++i) bad += pout[i] != hout[i];
    CHECK(bad == 0 && st.steps_total == NB * 256);
    CHECK(nnhip_host_free(py0) == NNHIP_OK && nnhip_host_free(pout) == NNHIP_OK && nnhip_host_free(nullptr) == NNHIP_OK);
  }
  // run-time compiled RHS on the device entry
  int kind = 0;
  CHECK(nnhip_ode_rhs_compile("negy_user", 1, 0, "dy[0] = -y[0];", &kind) == NNHIP_OK);
  CHECK(nnhip_ode_solve_batch_f64_dev(&opt, NNHIP_TSIT54, kind, nullptr, 0, dY0, N, 1, NNHIP_LAYOUT_SOA, tspan, 2, tOut, dOut, nullptr, nullptr, nullptr, 0,
                                      dWs, nnhip_ode_solve_workspace_bytes(2), s) == NNHIP_OK);
  CHECK(hipMemcpyAsync(g.data(), dOut + N, N * 8, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess);
  for (int64_t i = 0; i < N; i += 997) CHECK(g[i] == out[N + i]);
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}

// ode_capi_aux.hip — the smaller C-ABI entries: batched RHS evaluation, batched hermiteSpline,
// and the one-process multi-GPU solve (contiguous shards of the IVP index range, no exchange).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "ode_kernels.hpp"
#include "ode_rtc.hpp"

namespace nnhip {

template <class RHS>
hipError_t launch_rhs_batch(int64_t N, int64_t is, int64_t cs, double t, const double* y, double* dy, const Params& P, hipStream_t s) {
  const int64_t grid = (N + kBlock - 1) / kBlock;
  if (grid <= 0) return hipSuccess;
  return launch_kernel(rhs_batch_kernel<RHS>, dim3((unsigned)grid), dim3(kBlock), s, N, is, cs, t, y, dy, P);
}

// hermiteSpline (utils.nim:273-279) over a flat batch
__global__ __launch_bounds__(kBlock) void hermite_kernel(double x, double x1, double x2, const double* __restrict__ y1,
                                                         const double* __restrict__ y2, const double* __restrict__ dy1,
                                                         const double* __restrict__ dy2, double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const HermiteW w = hermite_weights(x, x1, x2);
  out[i] = hermite_apply(w, y1[i], y2[i], dy1[i], dy2[i]);
}

// the controller's step-size factor (ode.nim:71,537) over an array of error norms
template <int ORDER>
__global__ __launch_bounds__(kBlock) void controller_factor_kernel(const double* __restrict__ error, double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = shrink_factor<ORDER>(error[i]);
}

}  // namespace nnhip

extern "C" {

int nnhip_hermite_spline_f64_dev(double x, double x1, double x2, const double* y1, const double* y2, const double* dy1,
                                 const double* dy2, double* out, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!y1 || !y2 || !dy1 || !dy2 || !out))) return NNHIP_EVALUE;
  if (n == 0) return NNHIP_OK;
  const int64_t grid = (n + nnhip::kBlock - 1) / nnhip::kBlock;
  return nnhip::launch_kernel(nnhip::hermite_kernel, dim3((unsigned)grid), dim3(nnhip::kBlock), (hipStream_t)stream, x, x1, x2, y1,
                              y2, dy1, dy2, out, n) == hipSuccess ? NNHIP_OK : NNHIP_EHIP;
}

int nnhip_ode_controller_factor_f64_dev(int order, const double* error, double* out, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!error || !out))) return NNHIP_EVALUE;
  if (n == 0) return NNHIP_OK;
  const dim3 grid((unsigned)((n + nnhip::kBlock - 1) / nnhip::kBlock)), block(nnhip::kBlock);
  hipError_t e;
  switch (order) {
    case 2: e = nnhip::launch_kernel(nnhip::controller_factor_kernel<2>, grid, block, (hipStream_t)stream, error, out, n); break;
    case 3: e = nnhip::launch_kernel(nnhip::controller_factor_kernel<3>, grid, block, (hipStream_t)stream, error, out, n); break;
    case 5: e = nnhip::launch_kernel(nnhip::controller_factor_kernel<5>, grid, block, (hipStream_t)stream, error, out, n); break;
    case 6: e = nnhip::launch_kernel(nnhip::controller_factor_kernel<6>, grid, block, (hipStream_t)stream, error, out, n); break;
    default: return NNHIP_EVALUE;
  }
  return e == hipSuccess ? NNHIP_OK : NNHIP_EHIP;
}

int nnhip_ode_rhs_batch_f64_dev(int rhs_kind, const double* rhs_params, int n_params, int64_t N, int dim, int layout, double t,
                                const double* y, double* dy, void* stream) {
  if (N < 0 || dim < 1 || n_params < 0 || n_params > nnhip::kMaxParams) return NNHIP_EVALUE;
  if (N == 0) return NNHIP_OK;
  nnhip::Params P;
  for (int k = 0; k < nnhip::kMaxParams; ++k) P.p[k] = k < n_params ? rhs_params[k] : 0.0;
  const int64_t is = layout == NNHIP_LAYOUT_SOA ? 1 : dim, cs = layout == NNHIP_LAYOUT_SOA ? N : 1;
  if (rhs_kind >= NNHIP_RHS_USER_BASE) {
    int d = 0;
    if (!nnhip::rtc_info(rhs_kind, &d, nullptr) || d != dim) return NNHIP_EVALUE;
    return nnhip::rtc_launch_rhs(rhs_kind, N, is, cs, t, y, dy, P, (hipStream_t)stream) == hipSuccess ? NNHIP_OK : NNHIP_EHIP;
  }
  hipError_t e = hipErrorInvalidValue;
  bool found = false;
#define X(kind, d, T)                                                                    \
  if (!found && rhs_kind == kind && dim == d) {                                          \
    found = true;                                                                        \
    e = nnhip::launch_rhs_batch<nnhip::T>(N, is, cs, t, y, dy, P, (hipStream_t)stream);  \
  }
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
  if (!found) return NNHIP_EUNSUPPORTED;
  return e == hipSuccess ? NNHIP_OK : NNHIP_EHIP;
}

int nnhip_ode_solve_batch_multi_gpu_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind,
                                        const double* rhs_params, int n_params, const double* y0, int64_t N, int dim,
                                        int layout, const double* tspan, int n_t, double* t_out, double* y_out,
                                        int32_t* ny_out, int64_t max_steps, nnhip_ode_stats* stats, int n_gpus) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NNHIP_EHIP;
  if (n_gpus <= 0 || n_gpus > ndev || N < 0 || dim < 1 || n_t < 0) return NNHIP_EVALUE;
  std::vector<int> rcs(n_gpus, NNHIP_OK);
  std::vector<nnhip_ode_stats> sts(n_gpus);
  std::vector<std::thread> th;
  // contiguous index ranges [r*N/G, (r+1)*N/G): nothing couples trajectories (ode.nim:589: one solveODE call per IVP)
  for (int r = 0; r < n_gpus; ++r) {
    th.emplace_back([&, r]() {
      const int64_t lo = N * r / n_gpus, hi = N * (r + 1) / n_gpus, n = hi - lo;
      std::memset(&sts[r], 0, sizeof(nnhip_ode_stats));
      if (n == 0) { sts[r].ny_min = 0x7fffffff; return; }
      if (layout == NNHIP_LAYOUT_AOS && n_t <= 1) {  // shard is contiguous in both arrays
        rcs[r] = nnhip_ode_solve_batch_f64(opt, integrator, rhs_kind, rhs_params, n_params, y0 + lo * dim, n, dim, layout, tspan,
                                           n_t, r == 0 ? t_out : nullptr, y_out + lo * dim, ny_out ? ny_out + lo : nullptr,
                                           nullptr, nullptr, max_steps, &sts[r], r);
        return;
      }
      std::vector<double> y0s((size_t)n * dim), outs((size_t)n * dim * n_t);
      if (layout == NNHIP_LAYOUT_SOA) for (int c = 0; c < dim; ++c) std::memcpy(&y0s[(size_t)c * n], y0 + (size_t)c * N + lo, (size_t)n * sizeof(double));
      else std::memcpy(y0s.data(), y0 + (size_t)lo * dim, (size_t)n * dim * sizeof(double));
      rcs[r] = nnhip_ode_solve_batch_f64(opt, integrator, rhs_kind, rhs_params, n_params, y0s.data(), n, dim, layout, tspan, n_t,
                                         r == 0 ? t_out : nullptr, outs.data(), ny_out ? ny_out + lo : nullptr, nullptr, nullptr,
                                         max_steps, &sts[r], r);
      if (rcs[r]) return;
      if (layout == NNHIP_LAYOUT_SOA) {
        for (int64_t p = 0; p < (int64_t)n_t * dim; ++p) std::memcpy(y_out + (size_t)p * N + lo, &outs[(size_t)p * n], (size_t)n * sizeof(double));
      } else {
        for (int j = 0; j < n_t; ++j) std::memcpy(y_out + ((size_t)j * N + lo) * dim, &outs[(size_t)j * n * dim], (size_t)n * dim * sizeof(double));
      }
    });
  }
  for (auto& t : th) t.join();
  for (int r = 0; r < n_gpus; ++r) if (rcs[r]) return rcs[r];
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->ny_min = 0x7fffffff;
    for (int r = 0; r < n_gpus; ++r) {
      stats->steps_total += sts[r].steps_total; stats->rejected_total += sts[r].rejected_total;
      if (sts[r].steps_max > stats->steps_max) stats->steps_max = sts[r].steps_max;
      if (sts[r].ny_min < stats->ny_min) stats->ny_min = sts[r].ny_min;
      stats->nan_aborts += sts[r].nan_aborts; stats->truncated += sts[r].truncated;
      if (sts[r].kernel_ms > stats->kernel_ms) stats->kernel_ms = sts[r].kernel_ms;
      if (r == 0) stats->n_t_out = sts[r].n_t_out;
    }
  }
  return NNHIP_OK;
}

}  // extern "C"

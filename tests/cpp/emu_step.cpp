// emu_step.cpp — TEST INFRASTRUCTURE: one IntegratorProc call (ode.nim:38) through the BODY of step_tpi_kernel, executed on the host
// (tests/cpp/hip_cpu_emu.hpp), for every integrator — what nnhip_ode_step_batch_f64_dev launches for a thread-per-IVP system.  Reads cases from stdin,
//   <method id> <t> <dt> <absTol> <relTol> <dtMax> <dtMin> <y0> <y1> <y2> <f0> <f1> <f2>        (hex floats; Lorenz, sigma = 10, rho = 28, beta = 8/3)
// runs each on 5 lanes of a 64-thread workgroup and prints  yNew[3] fsalOut[3] dtUsed error  of the first and the last of them.
#include "ode_kernels.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace nnhip;

template <int METHOD>
static void one(const StepArgs& a) { hipemu::launch(step_tpi_kernel<METHOD, RhsLorenz, false>, dim3(1), dim3(kBlock), a); }

int main() {
  char name[64];
  int method;
  double t, dt, absTol, relTol, dtMax, dtMin, y[3], f[3];
  while (std::scanf("%d %la %la %la %la %la %la %la %la %la %la %la %la", &method, &t, &dt, &absTol, &relTol, &dtMax, &dtMin, &y[0], &y[1], &y[2], &f[0], &f[1], &f[2]) == 13) {
    (void)name;
    const int64_t N = 5;
    std::vector<double> yin(3 * N), fin(3 * N), yout(3 * N, -1.0), fout(3 * N, -1.0), dtu(N, -1.0), err(N, -1.0);
    for (int c = 0; c < 3; ++c)
      for (int64_t i = 0; i < N; ++i) { yin[c * N + i] = y[c]; fin[c * N + i] = f[c]; }
    StepArgs a{};
    a.N = N; a.ivpStride = 1; a.compStride = N;
    a.t_uniform = t; a.dt_uniform = dt;
    a.y_in = yin.data(); a.fsal_in = fin.data(); a.y_out = yout.data(); a.fsal_out = fout.data(); a.dt_used = dtu.data(); a.error = err.data();
    a.ctl.absTol = absTol; a.ctl.relTol = relTol; a.ctl.dtMax = dtMax; a.ctl.dtMin = dtMin;
    a.P.p[0] = 10.0; a.P.p[1] = 28.0; a.P.p[2] = 8.0 / 3.0;
    switch (method) {
      case NNHIP_HEUN2: one<NNHIP_HEUN2>(a); break;
      case NNHIP_RALSTON2: one<NNHIP_RALSTON2>(a); break;
      case NNHIP_KUTTA3: one<NNHIP_KUTTA3>(a); break;
      case NNHIP_HEUN3: one<NNHIP_HEUN3>(a); break;
      case NNHIP_RALSTON3: one<NNHIP_RALSTON3>(a); break;
      case NNHIP_SSPRK3: one<NNHIP_SSPRK3>(a); break;
      case NNHIP_RALSTON4: one<NNHIP_RALSTON4>(a); break;
      case NNHIP_KUTTA4: one<NNHIP_KUTTA4>(a); break;
      case NNHIP_RK4: one<NNHIP_RK4>(a); break;
      case NNHIP_RK21: one<NNHIP_RK21>(a); break;
      case NNHIP_BS32: one<NNHIP_BS32>(a); break;
      case NNHIP_DOPRI54: one<NNHIP_DOPRI54>(a); break;
      case NNHIP_TSIT54: one<NNHIP_TSIT54>(a); break;
      case NNHIP_VERN65: one<NNHIP_VERN65>(a); break;
      default: return 2;
    }
    for (int64_t i : {(int64_t)0, N - 1}) {
      for (int c = 0; c < 3; ++c) std::printf("%a ", yout[c * N + i]);
      for (int c = 0; c < 3; ++c) std::printf("%a ", fout[c * N + i]);
      std::printf("%a %a\n", dtu[i], err[i]);
    }
  }
  return 0;
}

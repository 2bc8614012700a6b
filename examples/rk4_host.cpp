// Minimal compiled-host example over include/numericalnim_hip.hpp: RK4, dy/dt = -y, one million IVPs, host buffers.
//   g++ -O2 -std=c++17 -I include examples/rk4_host.cpp -L numericalnim_amd/csrc -lnnhip_ode \
//       -Wl,-rpath,$PWD/numericalnim_amd/csrc -Wl,-rpath,/opt/rocm/lib -o rk4_host && ./rk4_host
#include <cmath>
#include <cstdio>

#include "numericalnim_hip.hpp"

int main() {
  using namespace numericalnim;
  OdeBatch y0 = OdeBatch::zeros(1000000);
  for (int64_t i = 0; i < y0.N; ++i) y0.at(i, 0) = 1.0 + 1e-6 * (double)i;
  const ODEoptions opt = newODEoptions(/*dt=*/1e-3);
  const OdeSolution s = solveODE(rhsNegY(), y0, {0.0, 1.0}, opt, static_cast<const NumContext<double>*>(nullptr), "rk4");
  std::printf("y(1) of IVP 0 = %.15f  (exp(-1) = %.15f), kernel %.3f ms, %lld steps in total\n", s.y[1].at(0, 0), std::exp(-1.0),
              s.stats.kernel_ms, (long long)s.stats.steps_total);
  return 0;
}

// emu_advance.cpp — TEST INFRASTRUCTURE: executes the BODIES of the adaptive streaming kernels on the host (tests/cpp/hip_cpu_emu.hpp: lanes as threads) —
// the lean kernels of round 5 (advance_lps_lean_kernel, advance_tpi_lean_kernel) side by side with the general ones they stand in for — through the whole
// loop of nnhip_ode_adaptive_stream_f64_dev as the host drives it (t = t0, dt = sqrt(dtMax * dtMin), one launch per iteration until no workgroup reports
// work left), compares the two states bit for bit after EVERY launch, and prints the final states as hex floats for the caller to compare with the oracle.
// Built and run by tests/test_kernel_bodies_on_cpu.py:  g++ -std=c++20 -O1 -ffp-contract=off -DNNHIP_CPU_EMU -I tests/cpp -I numericalnim_amd/csrc ...
//   emu_advance <case> <N> <method: 1 dopri54 | 2 tsit54> <absTol> <relTol> <dtMin> <dtMax> <tEnd>
//   case: ring16 (16-component ring, AoS, 4 lanes per system x 4 components per lane) | ring8 (8 components, 4 lanes x 2) | ring32 (8 lanes x 4) | linear16 (dy = a y, stage vector through LDS) | lorenz (thread per IVP, SoA)
#include "ode_kernels.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace nnhip;

struct Run {
  int64_t N;
  int dim;
  bool aos;
  std::vector<double> y, td, fsal;
  std::vector<unsigned int> active = std::vector<unsigned int>(kAggSlots, 0u);
};

static StepArgs general_args(Run& r, const StepCtl& ctl, const Params& P, double tEnd) {
  StepArgs a{};
  a.N = r.N;
  if (r.aos) { a.ivpStride = r.dim; a.compStride = 1; } else { a.ivpStride = 1; a.compStride = r.N; }
  a.y_in = r.y.data(); a.y_out = r.y.data(); a.fsal_in = r.fsal.data(); a.fsal_out = r.fsal.data();
  a.ctl = ctl; a.P = P; a.tEnd = tEnd;
  a.t_io = r.td.data(); a.dt_io = nullptr;  // (t, dt) side by side
  a.active = r.active.data();
  a.stepsPerLaunch = 1; a.recomputeFsal = 1;
  return a;
}
static AdvLeanArgs lean_args(Run& r, const StepCtl& ctl, const Params& P, double tEnd) {
  AdvLeanArgs l{};
  l.y = r.y.data(); l.td = reinterpret_cast<double2*>(r.td.data()); l.N = r.N; l.tEnd = tEnd; l.ctl = ctl; l.P = P; l.active = r.active.data();
  return l;
}
static bool any(std::vector<unsigned int>& f) {
  bool a = false;
  for (auto& x : f) { a = a || x != 0; x = 0; }
  return a;
}
static bool same_bits(const std::vector<double>& a, const std::vector<double>& b) { return a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * 8) == 0; }

template <int METHOD, class RHS, int CPL>
static int run_lps(int64_t N, const StepCtl& ctl, const Params& P, double tEnd, std::vector<double> y0, int& launches, std::vector<double>& yOut) {
  constexpr int DIM = RHS::dim, perBlock = kBlock / (DIM / CPL);
  Run g{N, DIM, true, y0, {}, std::vector<double>((size_t)N * DIM, 0.0)}, l = g;
  g.td.resize(2 * N); l.td.resize(2 * N);
  for (int64_t i = 0; i < N; ++i) { g.td[2 * i] = l.td[2 * i] = 0.0; g.td[2 * i + 1] = l.td[2 * i + 1] = std::sqrt(ctl.dtMax * ctl.dtMin); }
  const unsigned grid = (unsigned)((N + perBlock - 1) / perBlock);
  for (launches = 0; launches < 100000;) {
    hipemu::launch(advance_lps_kernel<METHOD, RHS, CPL, false>, dim3(grid), dim3(kBlock), general_args(g, ctl, P, tEnd));
    hipemu::launch(advance_lps_lean_kernel<METHOD, RHS, CPL>, dim3(grid), dim3(kBlock), lean_args(l, ctl, P, tEnd));
    ++launches;
    if (!same_bits(g.y, l.y) || !same_bits(g.td, l.td)) { std::fprintf(stderr, "lean and general kernels differ after launch %d\n", launches); return 2; }
    const bool ga = any(g.active), la = any(l.active);
    if (ga != la) { std::fprintf(stderr, "work-left flags differ after launch %d\n", launches); return 3; }
    if (!la) break;
  }
  yOut = l.y;
  return 0;
}
template <int METHOD, class RHS>
static int run_tpi(int64_t N, int block, const StepCtl& ctl, const Params& P, double tEnd, std::vector<double> y0, int& launches, std::vector<double>& yOut) {
  constexpr int DIM = RHS::dim;
  Run g{N, DIM, false, y0, {}, std::vector<double>((size_t)N * DIM, 0.0)}, l = g;
  g.td.resize(2 * N); l.td.resize(2 * N);
  for (int64_t i = 0; i < N; ++i) { g.td[2 * i] = l.td[2 * i] = 0.0; g.td[2 * i + 1] = l.td[2 * i + 1] = std::sqrt(ctl.dtMax * ctl.dtMin); }
  const unsigned grid = (unsigned)((N + block - 1) / block);
  for (launches = 0; launches < 100000;) {
    hipemu::launch(advance_tpi_kernel<METHOD, RHS, false, false>, dim3(grid), dim3(block), general_args(g, ctl, P, tEnd));
    hipemu::launch(advance_tpi_lean_kernel<METHOD, RHS>, dim3(grid), dim3(block), lean_args(l, ctl, P, tEnd));
    ++launches;
    if (!same_bits(g.y, l.y) || !same_bits(g.td, l.td)) { std::fprintf(stderr, "lean and general kernels differ after launch %d\n", launches); return 2; }
    const bool ga = any(g.active), la = any(l.active);
    if (ga != la) { std::fprintf(stderr, "work-left flags differ after launch %d\n", launches); return 3; }
    if (!la) break;
  }
  yOut = l.y;
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 9) { std::fprintf(stderr, "usage: emu_advance <ring16|ring8|lorenz> N method absTol relTol dtMin dtMax tEnd\n"); return 64; }
  const std::string c = argv[1];
  const int64_t N = std::atoll(argv[2]);
  const int method = std::atoi(argv[3]);
  StepCtl ctl{};
  ctl.absTol = std::atof(argv[4]); ctl.relTol = std::atof(argv[5]); ctl.dtMin = std::atof(argv[6]); ctl.dtMax = std::atof(argv[7]);
  const double tEnd = std::atof(argv[8]);
  Params P{};
  std::vector<double> y0, y;
  int launches = 0, rc = 64;
  if (c == "ring16" || c == "ring8" || c == "ring32") {
    const int d = c == "ring16" ? 16 : c == "ring32" ? 32 : 8;
    P.p[0] = 0.1;
    y0.resize((size_t)N * d);
    for (int64_t s = 0; s < N; ++s)
      for (int i = 0; i < d; ++i) y0[s * d + i] = (1.0 + (double)i / d + (double)(s % 1024) * 0x1p-20) * (double)(1 << (3 * (s % 5)));  // (magnitudes 1 ... 4096: under an absolute tolerance the systems take different step sequences)
    if (d == 16) rc = method == 1 ? run_lps<NNHIP_DOPRI54, RhsRing<16>, 4>(N, ctl, P, tEnd, y0, launches, y) : run_lps<NNHIP_TSIT54, RhsRing<16>, 4>(N, ctl, P, tEnd, y0, launches, y);
    else if (d == 32) rc = method == 1 ? run_lps<NNHIP_DOPRI54, RhsRing<32>, 4>(N, ctl, P, tEnd, y0, launches, y) : run_lps<NNHIP_TSIT54, RhsRing<32>, 4>(N, ctl, P, tEnd, y0, launches, y);  // 8 lanes per system: neighbours by ds_bpermute, the ordered LDS-free chain is for <= 4 lanes
    else rc = method == 1 ? run_lps<NNHIP_DOPRI54, RhsRing<8>, 2>(N, ctl, P, tEnd, y0, launches, y) : run_lps<NNHIP_TSIT54, RhsRing<8>, 2>(N, ctl, P, tEnd, y0, launches, y);
  } else if (c == "linear16") {  // dy = a y on 16 components: not banded — the stage vector goes through LDS (wave_lds_sync<4>), 4 lanes x 4 components
    P.p[0] = -0.8;
    y0.resize((size_t)N * 16);
    for (int64_t s = 0; s < N; ++s)
      for (int i = 0; i < 16; ++i) y0[s * 16 + i] = (1.0 + (double)i / 16 + (double)(s % 1024) * 0x1p-20) * (double)(1 << (3 * (s % 5)));
    rc = method == 1 ? run_lps<NNHIP_DOPRI54, RhsLinear<16>, 4>(N, ctl, P, tEnd, y0, launches, y) : run_lps<NNHIP_TSIT54, RhsLinear<16>, 4>(N, ctl, P, tEnd, y0, launches, y);
  } else if (c == "lorenz") {
    P.p[0] = 10.0; P.p[1] = 28.0; P.p[2] = 8.0 / 3.0;
    y0.resize((size_t)N * 3);
    for (int64_t i = 0; i < N; ++i) { y0[i] = 1.0 + (double)(i % 1024) * 0x1p-20 + (double)(i % 5); y0[N + i] = 1.0; y0[2 * N + i] = 1.0 + (double)(i % 3) * 7.0; }
    rc = method == 1 ? run_tpi<NNHIP_DOPRI54, RhsLorenz>(N, 64, ctl, P, tEnd, y0, launches, y) : run_tpi<NNHIP_TSIT54, RhsLorenz>(N, 64, ctl, P, tEnd, y0, launches, y);
  }
  if (rc) return rc;
  std::printf("launches %d\ny0", launches);
  for (double v : y0) std::printf(" %a", v);
  std::printf("\ny");
  for (double v : y) std::printf(" %a", v);
  std::printf("\n");
  return 0;
}

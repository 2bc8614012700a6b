// emu_stream.cpp — TEST INFRASTRUCTURE: the BODIES of the step-streaming kernels executed on the host (tests/cpp/hip_cpu_emu.hpp: lanes as threads) —
// the headline kernel rk4_stream_vec_kernel (every VEC / MODE the library instantiates; full tiles, the ragged tail tile, the persistent grid-stride form
// with fewer workgroups than tiles, in place and ping-pong), fixed_stream_vec_kernel (any fixed-step method, two IVPs per lane, both layouts, uniform and
// per-IVP (t, dt)), step_lps_kernel (one IntegratorProc call of a 16-component system spread over 16 lanes) and dense_rows_kernel (the rows due at the
// head of an iteration, ode.nim:512-524, of the fixed-step dense streaming driver).  The caller (tests/test_kernel_bodies_on_cpu.py) compares the printed
// hex floats with the oracle, bit for bit.  Built:  g++ -std=c++20 -O1 -ffp-contract=off -DNNHIP_CPU_EMU -I tests/cpp -I numericalnim_amd/csrc ...
//   emu_stream rk4 <N> <steps> <VEC 1|2|4> <MODE 0..3> <inplace 0|1> <neg 0|1>        dy = -y (neg: g(t, y) = -f(-t, y)), y0[i] = 1 + (i mod 2^10) 2^-10, dt = 2^-10
//   emu_stream fixed <method id> <N> <steps> <aos 0|1> <perIvpTime 0|1>               Lorenz, y0 = (1 + (i mod 1024) 2^-20, 1, 1), dt = 2^-8
//   emu_stream steplps <method id> <N> <neg 0|1>                                      16-component ring, one step from t = 0.25 with dt = 2^-6
//   emu_stream rows <N> <neg 0|1>                                                     Lorenz, three rows between two states
#include "ode_kernels.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace nnhip;

static void print_row(const char* tag, const std::vector<double>& v) {
  std::printf("%s", tag);
  for (double x : v) std::printf(" %a", x);
  std::printf("\n");
}

template <int VEC, int MODE, bool NEG>
static void rk4_launch(const double* yin, double* yout, int64_t n, double t, double dt, const Params& P) {
  const int64_t per = (int64_t)kRk4Block * 2 * VEC;
  int64_t grid = (n + per - 1) / per;
  if (MODE & 2) grid = grid > 2 ? 2 : grid;  // fewer workgroups than tiles: the grid-stride loop runs more than once
  const Rk4Dt h{dt, 0.5 * dt, dt / 6.0};     // as launch_rk4_stream_vec
  hipemu::launch(rk4_stream_vec_kernel<RhsNegY<1>, NEG, VEC, MODE>, dim3((unsigned)grid), dim3(kRk4Block), yin, yout, n, t, h.dt, h.hdt, h.dt6, P);
}
template <int VEC, bool NEG>
static void rk4_mode(int mode, const double* yin, double* yout, int64_t n, double t, double dt, const Params& P) {
  switch (mode) {
    case 0: rk4_launch<VEC, 0, NEG>(yin, yout, n, t, dt, P); break;
    case 1: rk4_launch<VEC, 1, NEG>(yin, yout, n, t, dt, P); break;
    case 2: rk4_launch<VEC, 2, NEG>(yin, yout, n, t, dt, P); break;
    default: rk4_launch<VEC, 3, NEG>(yin, yout, n, t, dt, P); break;
  }
}
static int run_rk4(int64_t N, int steps, int vec, int mode, int inplace, int neg) {
  std::vector<double> a((size_t)N), b((size_t)N, -7.0);
  for (int64_t i = 0; i < N; ++i) a[(size_t)i] = 1.0 + (double)(i % 1024) * 0x1p-10;
  print_row("y0", a);
  const double dt = 0x1p-10;
  Params P{};
  double t = 0.0;
  double *cur = a.data(), *nxt = inplace ? a.data() : b.data();
  for (int k = 0; k < steps; ++k) {
    auto go = [&](auto V) {
      constexpr int VEC = decltype(V)::value;
      if (neg) rk4_mode<VEC, true>(mode, cur, nxt, N, t, dt, P); else rk4_mode<VEC, false>(mode, cur, nxt, N, t, dt, P);
    };
    if (vec == 1) go(std::integral_constant<int, 1>{}); else if (vec == 2) go(std::integral_constant<int, 2>{}); else go(std::integral_constant<int, 4>{});
    t += dt;
    if (!inplace) std::swap(cur, nxt);
  }
  print_row("y", std::vector<double>(cur, cur + N));
  return 0;
}

template <int METHOD>
static void fixed_launch(const FixedVecArgs& a) {
  const int64_t per = (int64_t)kBlock * 2;
  hipemu::launch(fixed_stream_vec_kernel<METHOD, RhsLorenz, false, 2, false>, dim3((unsigned)((a.N + per - 1) / per)), dim3(kBlock), a);
}
static int run_fixed(int method, int64_t N, int steps, int aos, int perIvp) {
  std::vector<double> y((size_t)3 * N), out((size_t)3 * N, -7.0), fs((size_t)3 * N, -7.0), tv((size_t)N, 0.0), dv((size_t)N, 0x1p-8);
  for (int64_t i = 0; i < N; ++i) {
    const double c[3] = {1.0 + (double)(i % 1024) * 0x1p-20, 1.0, 1.0};
    for (int k = 0; k < 3; ++k) y[(size_t)(aos ? i * 3 + k : k * N + i)] = c[k];
  }
  print_row("y0", y);
  const double dt = 0x1p-8;
  double t = 0.0;
  double *cur = y.data(), *nxt = out.data();
  for (int s = 0; s < steps; ++s) {
    FixedVecArgs a{};
    a.yin = cur; a.yout = nxt; a.fsalOut = fs.data(); a.N = N; a.aos = aos; a.t = t; a.dt = dt;
    if (perIvp) { a.tDev = tv.data(); a.dtDev = dv.data(); a.t = -1.0; a.dt = -1.0; }
    a.P.p[0] = 10.0; a.P.p[1] = 28.0; a.P.p[2] = 8.0 / 3.0;
    switch (method) {
      case NNHIP_HEUN2: fixed_launch<NNHIP_HEUN2>(a); break;
      case NNHIP_RALSTON2: fixed_launch<NNHIP_RALSTON2>(a); break;
      case NNHIP_KUTTA3: fixed_launch<NNHIP_KUTTA3>(a); break;
      case NNHIP_HEUN3: fixed_launch<NNHIP_HEUN3>(a); break;
      case NNHIP_RALSTON3: fixed_launch<NNHIP_RALSTON3>(a); break;
      case NNHIP_SSPRK3: fixed_launch<NNHIP_SSPRK3>(a); break;
      case NNHIP_RALSTON4: fixed_launch<NNHIP_RALSTON4>(a); break;
      case NNHIP_KUTTA4: fixed_launch<NNHIP_KUTTA4>(a); break;
      case NNHIP_RK4: fixed_launch<NNHIP_RK4>(a); break;
      default: return 2;
    }
    t += dt;
    for (auto& x : tv) x += dt;
    std::swap(cur, nxt);
  }
  print_row("y", std::vector<double>(cur, cur + 3 * N));
  print_row("fsal", fs);  // the FSAL slot of a fixed-step IntegratorProc is yNew (ode.nim:189)
  return 0;
}

template <int METHOD, bool NEG>
static void steplps_launch(const StepArgs& a) {
  constexpr int perBlock = kBlock / 16;
  hipemu::launch(step_lps_kernel<METHOD, RhsRing<16>, NEG>, dim3((unsigned)((a.N + perBlock - 1) / perBlock)), dim3(kBlock), a);
}
static int run_steplps(int method, int64_t N, int neg) {
  const int D = 16;
  std::vector<double> y((size_t)N * D), f((size_t)N * D), yo((size_t)N * D, -7.0), fo((size_t)N * D, -7.0), dtu((size_t)N, -7.0), err((size_t)N, -7.0);
  for (int64_t i = 0; i < N; ++i)
    for (int c = 0; c < D; ++c) {
      y[(size_t)(i * D + c)] = 1.0 + (double)c / 16.0 + (double)(i % 1024) * 0x1p-20;
      f[(size_t)(i * D + c)] = 0.0;  // filled by the caller's convention below: FSAL = f(t, y) of the same direction
    }
  StepArgs a{};
  a.N = N; a.ivpStride = D; a.compStride = 1;
  a.t_uniform = 0.25; a.dt_uniform = 0x1p-6;
  a.P.p[0] = 0.1;
  // FSAL in: g(t, y) with g = f or -f(-t, .) (the ring has no t): per component -((c+1)/16) y_c + 0.1 y_{c+1}
  for (int64_t i = 0; i < N; ++i)
    for (int c = 0; c < D; ++c) {
      const double v = -((double)(c + 1) / 16.0) * y[(size_t)(i * D + c)] + 0.1 * y[(size_t)(i * D + (c + 1) % D)];
      f[(size_t)(i * D + c)] = neg ? -v : v;
    }
  a.y_in = y.data(); a.fsal_in = f.data(); a.y_out = yo.data(); a.fsal_out = fo.data(); a.dt_used = dtu.data(); a.error = err.data();
  a.ctl.absTol = 1e-9; a.ctl.relTol = 1e-9; a.ctl.dtMax = 1.0; a.ctl.dtMin = 1e-6;
  print_row("y0", y);
  print_row("f0", f);
#define CASE(M) case M: if (neg) steplps_launch<M, true>(a); else steplps_launch<M, false>(a); break;
  switch (method) {
    CASE(NNHIP_RK4) CASE(NNHIP_KUTTA3) CASE(NNHIP_RK21) CASE(NNHIP_BS32) CASE(NNHIP_DOPRI54) CASE(NNHIP_TSIT54) CASE(NNHIP_VERN65)
    default: return 2;
  }
#undef CASE
  print_row("y", yo);
  print_row("fsal", fo);
  print_row("dt", dtu);
  print_row("err", err);
  return 0;
}

static int run_rows(int64_t N, int neg) {
  std::vector<double> ya((size_t)3 * N), yb((size_t)3 * N), r0((size_t)3 * N, -7.0), r1 = r0, r2 = r0;
  for (int64_t i = 0; i < N; ++i) {
    const double a[3] = {1.0 + (double)(i % 1024) * 0x1p-20, 1.0, 1.0}, b[3] = {1.0625 + (double)(i % 1024) * 0x1p-20, 1.3125, 0.96875};
    for (int k = 0; k < 3; ++k) { ya[(size_t)(k * N + i)] = a[k]; yb[(size_t)(k * N + i)] = b[k]; }
  }
  DenseRows r{};
  r.n = 3;
  const double tA = 0.125, tB = 0.15625;
  r.treq[0] = 0.125; r.treq[1] = 0.140625; r.treq[2] = 0.15;
  r.out[0] = r0.data(); r.out[1] = r1.data(); r.out[2] = r2.data();
  Params P{};
  P.p[0] = 10.0; P.p[1] = 28.0; P.p[2] = 8.0 / 3.0;
  hipemu::launch(dense_rows_kernel<RhsLorenz>, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), N, (int64_t)1, N, tA, tB, neg, (const double*)ya.data(),
                 (const double*)yb.data(), r, P);
  print_row("ya", ya);
  print_row("yb", yb);
  print_row("r0", r0);
  print_row("r1", r1);
  print_row("r2", r2);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string what = argv[1];
  auto I = [&](int k) { return k < argc ? std::atoll(argv[k]) : 0LL; };
  if (what == "rk4") return run_rk4(I(2), (int)I(3), (int)I(4), (int)I(5), (int)I(6), (int)I(7));
  if (what == "fixed") return run_fixed((int)I(2), I(3), (int)I(4), (int)I(5), (int)I(6));
  if (what == "steplps") return run_steplps((int)I(2), I(3), (int)I(4));
  if (what == "rows") return run_rows(I(2), (int)I(3));
  return 2;
}

// mb_adv.hip — A/B micro-harness for the HBM-resident adaptive loop kernels ("advance": one iteration of ODESolver's loop body per
// launch, ode.nim:525-541).  Stand-alone (no torch, no Python): instantiates candidate kernels from ode_kernels.hpp next to the
// production one, runs the same number of loop iterations with each on identical initial data, times them with HIP events and
// demands bitwise equality of every state array.  Build:  make -C tools/microbench     Run on the GPU box:  tools/microbench/mb_adv
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <functional>

#include "adv_experiments.hpp"

using namespace nnhip;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

struct Problem {
  int64_t N; int dim; int layout;  // 0 SoA, 1 AoS
  std::vector<double> y0;          // host, in the device layout
  Params P; StepCtl ctl; double t0, tEnd, dt0;
};
struct DevState {
  double *y, *fsal, *t, *dt, *err, *td; unsigned int* active;
  int64_t N; int dim;
  void alloc(int64_t n, int d) {
    N = n; dim = d;
    CK(hipMalloc(&y, sizeof(double) * n * d)); CK(hipMalloc(&fsal, sizeof(double) * n * d));
    CK(hipMalloc(&t, sizeof(double) * n)); CK(hipMalloc(&dt, sizeof(double) * n)); CK(hipMalloc(&err, sizeof(double) * n)); CK(hipMalloc(&td, sizeof(double) * 2 * n));
    CK(hipMalloc(&active, sizeof(unsigned) * kAggSlots));
  }
  void release() { (void)hipFree(y); (void)hipFree(fsal); (void)hipFree(t); (void)hipFree(dt); (void)hipFree(err); (void)hipFree(td); (void)hipFree(active); }
};

template <class RHS>
static void init_state(const Problem& p, DevState& s) {
  CK(hipMemcpy(s.y, p.y0.data(), sizeof(double) * p.N * p.dim, hipMemcpyHostToDevice));
  const int64_t is = p.layout ? p.dim : 1, cs = p.layout ? 1 : p.N;
  CK(launch_kernel(rhs_batch_kernel<RHS>, dim3((unsigned)((p.N + kBlock - 1) / kBlock)), dim3(kBlock), nullptr, p.N, is, cs, p.t0, (const double*)s.y, s.fsal, p.P));
  CK(launch_kernel(fill_t_dt_kernel<0>, dim3((unsigned)((p.N + kBlock - 1) / kBlock)), dim3(kBlock), nullptr, s.t, s.dt, p.N, p.t0, p.dt0));
  CK(launch_kernel(fill_td_kernel<0>, dim3((unsigned)((p.N + kBlock - 1) / kBlock)), dim3(kBlock), nullptr, (double2*)s.td, p.N, p.t0, p.dt0));
  CK(hipMemset(s.err, 0, sizeof(double) * p.N));
  CK(hipMemset(s.active, 0, sizeof(unsigned) * kAggSlots));
  CK(hipDeviceSynchronize());
}
// mode 0: t, dt, error as three columns (the layout up to round 3's first half); 1: (t, dt) side by side, error not stored (production
// since); 2: columns, error not stored
static StepArgs make_args(const Problem& p, DevState& s, int mode = 0) {
  StepArgs a{};
  a.N = p.N;
  if (p.layout == 0) { a.ivpStride = 1; a.compStride = p.N; } else { a.ivpStride = p.dim; a.compStride = 1; }
  a.y_in = s.y; a.y_out = s.y; a.fsal_in = s.fsal; a.fsal_out = s.fsal; a.error = s.err;
  a.ctl = p.ctl; a.P = p.P; a.tEnd = p.tEnd; a.t_io = s.t; a.dt_io = s.dt; a.active = nullptr; a.steps_io = nullptr;
  if (mode) a.error = nullptr;
  if (mode == 1 || mode == 3) { a.t_io = s.td; a.dt_io = nullptr; }
  if (mode == 3) a.recomputeFsal = 1;  // mode 3: mode 1 + FSAL re-evaluated at the start of the launch instead of carried through HBM
  return a;
}
using Launch = std::function<hipError_t(const StepArgs&, hipStream_t)>;
struct Candidate { std::string name; Launch launch; int K = 1; int mode = 0; };  // K: loop iterations per launch (the harness then issues 1/K of the launches)

static std::vector<double> fetch(const DevState& s, int mode) {  // y, FSAL, t, dt (the error column is not compared: modes 1 and 2 do not store it)
  std::vector<double> h((size_t)s.N * (2 * s.dim + 2));
  double* q = h.data();
  CK(hipMemcpy(q, s.y, sizeof(double) * s.N * s.dim, hipMemcpyDeviceToHost)); q += s.N * s.dim;
  CK(hipMemcpy(q, s.fsal, sizeof(double) * s.N * s.dim, hipMemcpyDeviceToHost)); q += s.N * s.dim;
  if (mode == 1 || mode == 3) {
    std::vector<double> td((size_t)s.N * 2);
    CK(hipMemcpy(td.data(), s.td, sizeof(double) * 2 * s.N, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < s.N; ++i) { q[i] = td[2 * i]; q[s.N + i] = td[2 * i + 1]; }
  } else {
    CK(hipMemcpy(q, s.t, sizeof(double) * s.N, hipMemcpyDeviceToHost)); q += s.N;
    CK(hipMemcpy(q, s.dt, sizeof(double) * s.N, hipMemcpyDeviceToHost));
  }
  return h;
}

// runs `warm + iters` loop iterations from the initial state (all IVPs still short of tEnd in that window), timing the last `iters`
template <class RHS>
static void run_all(const char* title, const Problem& p, std::vector<Candidate>& cands, int warm, int iters, int reps, double bytesPerIvpStep) {
  DevState s; s.alloc(p.N, p.dim);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<double> ref;
  printf("== %s: N=%lld dim=%d, %d timed iterations after %d, best of %d; %.0f B per IVP-step\n", title, (long long)p.N, p.dim, iters, warm, reps, bytesPerIvpStep);
  for (auto& c : cands) {
    float best = 1e30f;
    std::vector<double> out;
    for (int r = 0; r < reps; ++r) {
      init_state<RHS>(p, s);
      StepArgs a = make_args(p, s, c.mode);
      for (int k = 0; k < warm / c.K; ++k) CK(c.launch(a, st));
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < iters / c.K; ++k) CK(c.launch(a, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
      if (r == 0) {
        if (c.mode == 3) {  // the FSAL array was not touched: fill it with f(t, y) of the final state (the right-hand sides here are autonomous), so that
                            // the comparison also checks that the carried FSAL of the other candidates IS f of the state they end in
          const int64_t is = p.layout ? p.dim : 1, cs = p.layout ? 1 : p.N;
          CK(launch_kernel(rhs_batch_kernel<RHS>, dim3((unsigned)((p.N + kBlock - 1) / kBlock)), dim3(kBlock), st, p.N, is, cs, p.t0, (const double*)s.y, s.fsal, p.P));
          CK(hipStreamSynchronize(st));
        }
        out = fetch(s, c.mode);
      }
    }
#ifdef NNHIP_ADV_TIMING
    {
      unsigned long long z[16] = {0}, h[16];
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_adv_timing), z, sizeof(z)));
      StepArgs a = make_args(p, s);
      CK(c.launch(a, st)); CK(hipStreamSynchronize(st));
      CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_adv_timing), sizeof(h)));
      if (h[8] && h[11]) printf("    [%s] mover loop: %.0f shader cycles in %.2f us -> %.2f GHz\n", c.name.c_str(), (double)h[10] / h[8], (double)h[11] / h[8] / 100.0, (double)h[10] / h[11] * 0.1);
      if (h[8]) printf("    [%s] mover per wave: spin %.0f  stores %.0f  loads %.0f  wait+publish %.0f cycles (%llu waves); consumer per wave: spin %.0f  work %.0f  (%llu waves)\n", c.name.c_str(),
                       (double)h[0] / h[8], (double)h[1] / h[8], (double)h[2] / h[8], (double)h[3] / h[8], h[8], (double)h[4] / h[9], (double)h[5] / h[9], h[9]);
      if (h[5] && !h[8]) printf("    per wave: %.0f tiles; cycles per tile: wait %.0f  issue %.0f  arithmetic %.0f; whole kernel %.0f cycles per wave (%llu waves)\n",
                       (double)h[3] / h[5], (double)h[0] / h[3], (double)h[1] / h[3], (double)h[2] / h[3], (double)h[4] / h[5], h[5]);
    }
#endif
    bool same = true;
    if (ref.empty()) ref = out;
    else same = out.size() == ref.size() && std::memcmp(out.data(), ref.data(), out.size() * sizeof(double)) == 0;
    const double us = best * 1e3 / iters;
    printf("%-34s %8.2f us/iter  %7.1f GB/s  frac8T %.3f  %s%s\n", c.name.c_str(), us, bytesPerIvpStep / c.K * p.N / us * 1e-3, bytesPerIvpStep / c.K * p.N / us * 1e-3 / 8000.0,
           same ? "bit-identical" : "** DIFFERS **", c.K > 1 ? "  [K iterations per launch: bytes = 8(4d+5)/K per step]" : "");
    fflush(stdout);
  }
  s.release();
}

template <int METHOD, class RHS, int CPL>
static Launch lps_base(int K = 1) {
  return [K](const StepArgs& a0, hipStream_t st) { StepArgs a = a0; a.stepsPerLaunch = K; return launch_advance_lps<METHOD, RHS, CPL>(a, 0, st); };
}
// the general kernel where the lean one would apply (StepArgs::noLean: what tuning knob "adv_lean" 0 sets in the library)
template <int METHOD, class RHS, int CPL>
static Launch lps_general() {
  return [](const StepArgs& a0, hipStream_t st) { StepArgs a = a0; a.noLean = 1; return launch_advance_lps<METHOD, RHS, CPL>(a, 0, st); };
}
template <int METHOD, class RHS>
static Launch tpi_general(int block) {
  return [block](const StepArgs& a0, hipStream_t st) { StepArgs a = a0; a.noLean = 1; return launch_advance_tpi<METHOD, RHS>(a, block, st); };
}
template <int METHOD, class RHS, int CPL, bool PP = false, int PROBE = 0>
static Launch lps_persist(int blocksPerCU) {
  return [blocksPerCU](const StepArgs& a, hipStream_t st) {
    constexpr int perTile = kBlock / (RHS::dim / CPL);
    const int64_t nTiles = (a.N + perTile - 1) / perTile;
    int64_t grid = 256LL * blocksPerCU;
    if (grid > nTiles) grid = nTiles;
    return launch_kernel(advance_lps_persist_kernel<METHOD, RHS, CPL, PP, PROBE>, dim3((unsigned)grid), dim3(kBlock), st, a);
  };
}
template <int METHOD, class RHS, int CPL, int PROBE = 0, int PRIO = 0, int SUB = 1>
static Launch lps_mc(int blocksPerCU) {
  return [blocksPerCU](const StepArgs& a, hipStream_t st) {
    constexpr int T = (64 / (RHS::dim / CPL)) * 3;
    const int64_t nTiles = (a.N + T - 1) / T;
    int64_t grid = 256LL * blocksPerCU;
    if (grid * SUB > nTiles) grid = (nTiles + SUB - 1) / SUB;
    return launch_kernel(advance_lps_mc_kernel<METHOD, RHS, CPL, PROBE, PRIO, SUB>, dim3((unsigned)grid), dim3(kBlock * SUB), st, a);
  };
}
template <int METHOD, class RHS>
static Launch tpi_base(int block, int nt, int K = 1) {
  return [block, nt, K](const StepArgs& a0, hipStream_t st) { StepArgs a = a0; a.nontemporal = nt; a.stepsPerLaunch = K; return launch_advance_tpi<METHOD, RHS>(a, block, st); };
}
template <int METHOD, class RHS, bool NT>
static Launch tpi_persist(int block, int blocksPerCU) {
  return [block, blocksPerCU](const StepArgs& a, hipStream_t st) {
    const int64_t nTiles = (a.N + block - 1) / block;
    int64_t grid = 256LL * blocksPerCU;
    if (grid > nTiles) grid = nTiles;
    return launch_kernel(advance_tpi_persist_kernel<METHOD, RHS, NT>, dim3((unsigned)grid), dim3(block), st, a);
  };
}

template <int METHOD, class RHS, bool NT>
static Launch tpi_pair(int block) {
  return [block](const StepArgs& a, hipStream_t st) {
    const int64_t pairs = (a.N + 1) / 2;
    return launch_kernel(advance_tpi_pair_kernel<METHOD, RHS, NT>, dim3((unsigned)((pairs + block - 1) / block)), dim3(block), st, a);
  };
}

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : "";
  int dev = 0; CK(hipSetDevice(dev));
  const double dtMax = 1e-2, dtMin = 1e-4;
  if (strstr(only, "lean")) {  // round 6: the general kernels against the lean ones (mode 3 = the streaming driver's own layout, where lps_base / tpi_base dispatch to the
                               // lean kernels), BASELINE's streamed C4 and C3 at 1e6, interleaved and repeated.  Other builds of this file, same argument:
                               //   make mb_adv_contract   both columns FMA-contracted (the lean column = the library's "fp_contract" kernels; bits differ from the exact build's)
                               //   make mb_adv_report0    round 5's flag store (every active lane, no barrier) — only matters in polled launches, which this harness never issues
                               //   make mb_adv EXTRA=-DNNHIP_ADV_LEAN_WPE=4   contracted C4 at 128 VGPRs (9 spilled) = 4 waves per SIMD
    {
      Problem p; p.N = 1000000; p.dim = 16; p.layout = 1; p.y0.resize((size_t)p.N * 16);
      for (int64_t i = 0; i < p.N; ++i) for (int c = 0; c < 16; ++c) p.y0[i * 16 + c] = 1.0 + c / 16.0 + (double)(i % 1024) * 0x1p-20;
      std::memset(&p.P, 0, sizeof(p.P)); p.P.p[0] = 0.1;
      p.ctl = StepCtl{1e-4, 1e-4, dtMax, dtMin}; p.t0 = 0; p.tEnd = 1.0; p.dt0 = std::sqrt(dtMax * dtMin);
      using R = RhsRing<16>;
      std::vector<Candidate> c;
      for (int rep = 0; rep < 2; ++rep) {
        c.push_back({"tsit54 general CPL=4", lps_general<NNHIP_TSIT54, R, 4>(), 1, 3});
        c.push_back({"tsit54 LEAN CPL=4", lps_base<NNHIP_TSIT54, R, 4>(), 1, 3});
        c.push_back({"dopri54 general CPL=4", lps_general<NNHIP_DOPRI54, R, 4>(), 1, 3});
        c.push_back({"dopri54 LEAN CPL=4", lps_base<NNHIP_DOPRI54, R, 4>(), 1, 3});
      }
      run_all<R>("C4 streamed, general vs lean (288 B per system-step)", p, c, 10, 60, 5, 8.0 * (2 * 16 + 4));
    }
    {
      Problem p; p.N = 1000000; p.dim = 3; p.layout = 0; p.y0.resize((size_t)p.N * 3);
      for (int64_t i = 0; i < p.N; ++i) { p.y0[i] = 1.0 + (double)(i % 1024) * 0x1p-20; p.y0[p.N + i] = 1.0; p.y0[2 * p.N + i] = 1.0; }
      std::memset(&p.P, 0, sizeof(p.P)); p.P.p[0] = 10.0; p.P.p[1] = 28.0; p.P.p[2] = 8.0 / 3.0;
      p.ctl = StepCtl{1e-4, 1e-4, dtMax, dtMin}; p.t0 = 0; p.tEnd = 1.0; p.dt0 = std::sqrt(dtMax * dtMin);
      using R = RhsLorenz;
      std::vector<Candidate> c;
      for (int rep = 0; rep < 2; ++rep) {
        c.push_back({"dopri54 general b64", tpi_general<NNHIP_DOPRI54, R>(64), 1, 3});
        c.push_back({"dopri54 LEAN b64", tpi_base<NNHIP_DOPRI54, R>(64, 0), 1, 3});
        c.push_back({"tsit54 general b64", tpi_general<NNHIP_TSIT54, R>(64), 1, 3});
        c.push_back({"tsit54 LEAN b64", tpi_base<NNHIP_TSIT54, R>(64, 0), 1, 3});
      }
      run_all<R>("C3 streamed at 1e6, general vs lean (80 B per IVP-step)", p, c, 10, 60, 5, 8.0 * (2 * 3 + 4));
    }
    return 0;
  }
  if (!*only || strstr(only, "c4")) {  // C4: 1e6 x 16 ring, AoS, default options
    Problem p; p.N = 1000000; p.dim = 16; p.layout = 1; p.y0.resize((size_t)p.N * 16);
    for (int64_t i = 0; i < p.N; ++i) for (int c = 0; c < 16; ++c) p.y0[i * 16 + c] = 1.0 + c / 16.0 + (double)(i % 1024) * 0x1p-20;
    std::memset(&p.P, 0, sizeof(p.P)); p.P.p[0] = 0.1;
    p.ctl = StepCtl{1e-4, 1e-4, dtMax, dtMin}; p.t0 = 0; p.tEnd = 1.0; p.dt0 = std::sqrt(dtMax * dtMin);
    using R = RhsRing<16>;
    std::vector<Candidate> c;
    c.push_back({"tsit54 base (2 tiles/block)", lps_base<NNHIP_TSIT54, R, 2>()});
    c.push_back({"tsit54 base, error not stored", lps_base<NNHIP_TSIT54, R, 2>(), 1, 2});
    c.push_back({"tsit54 base, (t,dt) packed", lps_base<NNHIP_TSIT54, R, 2>(), 1, 1});
    c.push_back({"tsit54 base again", lps_base<NNHIP_TSIT54, R, 2>()});
    c.push_back({"tsit54 base, (t,dt) packed again", lps_base<NNHIP_TSIT54, R, 2>(), 1, 1});
    c.push_back({"tsit54 packed, FSAL recomputed", lps_base<NNHIP_TSIT54, R, 2>(), 1, 3});
    c.push_back({"tsit54 packed, FSAL recomputed CPL=4", lps_base<NNHIP_TSIT54, R, 4>(), 1, 3});
    if (strstr(only, "quick")) {
      c.push_back({"dopri54 base", lps_base<NNHIP_DOPRI54, R, 2>()});
      c.push_back({"dopri54 base, (t,dt) packed", lps_base<NNHIP_DOPRI54, R, 2>(), 1, 1});
      c.push_back({"dopri54 packed, FSAL recomputed", lps_base<NNHIP_DOPRI54, R, 2>(), 1, 3});
      c.push_back({"dopri54 packed, FSAL recomputed CPL=4", lps_base<NNHIP_DOPRI54, R, 4>(), 1, 3});
      c.push_back({"dopri54 packed, FSAL recomputed CPL=8", lps_base<NNHIP_DOPRI54, R, 8>(), 1, 3});
      c.push_back({"tsit54 packed, FSAL recomputed CPL=8", lps_base<NNHIP_TSIT54, R, 8>(), 1, 3});
      c.push_back({"tsit54 packed CPL=4 (FSAL carried)", lps_base<NNHIP_TSIT54, R, 4>(), 1, 1});
      c.push_back({"tsit54 packed, FSAL recomputed CPL=4 again", lps_base<NNHIP_TSIT54, R, 4>(), 1, 3});
      run_all<R>("C4 streamed (quick list)", p, c, 10, 60, 5, 8.0 * (4 * 16 + 5));
      std::vector<Candidate> b;   // the methods that never read the FSAL slot
      b.push_back({"bs32 CPL=2", lps_base<NNHIP_BS32, R, 2>(), 1, 1});
      b.push_back({"bs32 CPL=4", lps_base<NNHIP_BS32, R, 4>(), 1, 1});
      b.push_back({"bs32 CPL=1", lps_base<NNHIP_BS32, R, 1>(), 1, 1});
      b.push_back({"bs32 CPL=2 again", lps_base<NNHIP_BS32, R, 2>(), 1, 1});
      run_all<R>("C4 streamed, BS32 (quick list)", p, b, 10, 60, 5, 8.0 * (2 * 16 + 4));
      std::vector<Candidate> k;
      k.push_back({"rk21 CPL=2", lps_base<NNHIP_RK21, R, 2>(), 1, 1});
      k.push_back({"rk21 CPL=4", lps_base<NNHIP_RK21, R, 4>(), 1, 1});
      k.push_back({"rk21 CPL=1", lps_base<NNHIP_RK21, R, 1>(), 1, 1});
      run_all<R>("C4 streamed, RK21 (quick list)", p, k, 10, 60, 5, 8.0 * (2 * 16 + 4));
      std::vector<Candidate> v;
      v.push_back({"vern65 CPL=2", lps_base<NNHIP_VERN65, R, 2>(), 1, 1});
      v.push_back({"vern65 CPL=4", lps_base<NNHIP_VERN65, R, 4>(), 1, 1});
      run_all<R>("C4 streamed, Vern65 (quick list)", p, v, 10, 60, 5, 8.0 * (4 * 16 + 4));
    } else {
    for (int b : {3, 4}) c.push_back({"tsit54 persist grid=256x" + std::to_string(b), lps_persist<NNHIP_TSIT54, R, 2>(b)});
    c.push_back({"tsit54 pingpong grid=256x4", lps_persist<NNHIP_TSIT54, R, 2, true>(4)});
    c.push_back({"tsit54 base again", lps_base<NNHIP_TSIT54, R, 2>()});
    for (int K : {2, 5, 10}) c.push_back({"tsit54 base K=" + std::to_string(K), lps_base<NNHIP_TSIT54, R, 2>(K), K});
    for (int b : {2, 3}) c.push_back({"tsit54 mover/consumer CPL=4 grid=256x" + std::to_string(b), lps_mc<NNHIP_TSIT54, R, 4>(b)});
    c.push_back({"tsit54 mc 3 groups per WG grid=256x1", lps_mc<NNHIP_TSIT54, R, 4, 0, 0, 3>(1)});
    c.push_back({"tsit54 mc 3 groups per WG prio3", lps_mc<NNHIP_TSIT54, R, 4, 0, 3, 3>(1)});
    c.push_back({"PROBE mc 3 groups memory only", lps_mc<NNHIP_TSIT54, R, 4, 2, 0, 3>(1)});
    c.push_back({"PROBE mc 3 groups + 500 fma64", lps_mc<NNHIP_TSIT54, R, 4, 3, 0, 3>(1)});
    c.push_back({"PROBE mc memory path only grid=256x3", lps_mc<NNHIP_TSIT54, R, 4, 2, 0>(3)});
    c.push_back({"PROBE mc + 500 FP64 fma grid=256x3", lps_mc<NNHIP_TSIT54, R, 4, 3, 0>(3)});
    c.push_back({"PROBE mc + sleep 1k grid=256x3", lps_mc<NNHIP_TSIT54, R, 4, 10, 0>(3)});
    c.push_back({"PROBE mc + sleep 3k grid=256x3", lps_mc<NNHIP_TSIT54, R, 4, 12, 0>(3)});
    c.push_back({"PROBE mc + sleep 6k grid=256x3", lps_mc<NNHIP_TSIT54, R, 4, 15, 0>(3)});
    c.push_back({"PROBE mc + 250 fma64 grid=256x3", lps_mc<NNHIP_TSIT54, R, 4, 20, 0>(3)});
    c.push_back({"PROBE mc + 1000 fma64 grid=256x3", lps_mc<NNHIP_TSIT54, R, 4, 23, 0>(3)});
    c.push_back({"PROBE mc + 2000 fma64 grid=256x3", lps_mc<NNHIP_TSIT54, R, 4, 27, 0>(3)});
    c.push_back({"tsit54 base CPL=4", lps_base<NNHIP_TSIT54, R, 4>()});
    for (int b : {2, 3}) c.push_back({"tsit54 persist CPL=4 grid=256x" + std::to_string(b), lps_persist<NNHIP_TSIT54, R, 4>(b)});
    for (int b : {1, 2}) c.push_back({"tsit54 persist CPL=8 grid=256x" + std::to_string(b), lps_persist<NNHIP_TSIT54, R, 8>(b)});
    if (getenv("MB_PROBE")) {
      for (int b : {1, 2, 3, 4}) c.push_back({"PROBE arithmetic only grid=256x" + std::to_string(b), lps_persist<NNHIP_TSIT54, R, 2, false, 1>(b)});
      for (int b : {1, 2, 3, 4}) c.push_back({"PROBE memory only grid=256x" + std::to_string(b), lps_persist<NNHIP_TSIT54, R, 2, false, 2>(b)});
    }
    run_all<R>("C4 streamed, Tsit54", p, c, 10, 60, 3, 8.0 * (4 * 16 + 5));
    std::vector<Candidate> d;
    d.push_back({"dopri54 base", lps_base<NNHIP_DOPRI54, R, 2>()});
    d.push_back({"dopri54 persist grid=256x4", lps_persist<NNHIP_DOPRI54, R, 2>(4)});
    d.push_back({"dopri54 pingpong grid=256x4", lps_persist<NNHIP_DOPRI54, R, 2, true>(4)});
    run_all<R>("C4 streamed, DOPRI54", p, d, 10, 60, 3, 8.0 * (4 * 16 + 5));
    }
  }
  for (int64_t n : {(int64_t)1000000, (int64_t)10000000}) {  // C3: Lorenz, SoA, default options
    const std::string tag = n == 1000000 ? "c3a" : "c3b";
    if (*only && !strstr(only, tag.c_str())) continue;
    Problem p; p.N = n; p.dim = 3; p.layout = 0; p.y0.resize((size_t)n * 3);
    for (int64_t i = 0; i < n; ++i) { p.y0[i] = 1.0 + (double)(i % 1024) * 0x1p-20; p.y0[n + i] = 1.0; p.y0[2 * n + i] = 1.0; }
    std::memset(&p.P, 0, sizeof(p.P)); p.P.p[0] = 10.0; p.P.p[1] = 28.0; p.P.p[2] = 8.0 / 3.0;
    p.ctl = StepCtl{1e-4, 1e-4, dtMax, dtMin}; p.t0 = 0; p.tEnd = 1.0; p.dt0 = std::sqrt(dtMax * dtMin);
    using R = RhsLorenz;
    constexpr int M = NNHIP_DOPRI54;
    std::vector<Candidate> c;
    c.push_back({"dopri54 base b256", tpi_base<M, R>(256, 0)});
    c.push_back({"dopri54 base b64", tpi_base<M, R>(64, 0)});
    c.push_back({"dopri54 base b64 nt", tpi_base<M, R>(64, 1)});
    c.push_back({"dopri54 base b256 err not stored", tpi_base<M, R>(256, 0), 1, 2});
    c.push_back({"dopri54 base b256 packed", tpi_base<M, R>(256, 0), 1, 1});
    c.push_back({"dopri54 base b64 packed", tpi_base<M, R>(64, 0), 1, 1});
    c.push_back({"dopri54 base b64 nt packed", tpi_base<M, R>(64, 1), 1, 1});
    c.push_back({"dopri54 base b256 again", tpi_base<M, R>(256, 0)});
    c.push_back({"dopri54 base b256 packed again", tpi_base<M, R>(256, 0), 1, 1});
    c.push_back({"tsit54 base b256", tpi_base<NNHIP_TSIT54, R>(256, 0)});
    c.push_back({"tsit54 base b256 packed", tpi_base<NNHIP_TSIT54, R>(256, 0), 1, 1});
    c.push_back({"dopri54 b256 packed FSAL recomputed", tpi_base<M, R>(256, 0), 1, 3});
    c.push_back({"dopri54 b64 packed FSAL recomputed", tpi_base<M, R>(64, 0), 1, 3});
    c.push_back({"dopri54 b64 nt packed FSAL recomputed", tpi_base<M, R>(64, 1), 1, 3});
    c.push_back({"tsit54 b64 packed FSAL recomputed", tpi_base<NNHIP_TSIT54, R>(64, 0), 1, 3});
    c.push_back({"dopri54 PAIRS b64", tpi_pair<M, R, false>(64), 1, 3});
    c.push_back({"dopri54 PAIRS b64 nt", tpi_pair<M, R, true>(64), 1, 3});
    c.push_back({"dopri54 PAIRS b256", tpi_pair<M, R, false>(256), 1, 3});
    c.push_back({"dopri54 b64 nt packed FSAL recomputed again", tpi_base<M, R>(64, 1), 1, 3});
    if (strstr(only, "quick")) { run_all<R>(("C3 streamed (quick list) " + tag).c_str(), p, c, 10, 60, 5, 8.0 * (4 * 3 + 5)); continue; }
    for (int b : {3, 4, 5, 8}) c.push_back({"dopri54 persist b256 grid=256x" + std::to_string(b), tpi_persist<M, R, false>(256, b)});
    for (int b : {12, 16, 20}) c.push_back({"dopri54 persist b64 grid=256x" + std::to_string(b), tpi_persist<M, R, false>(64, b)});
    c.push_back({"dopri54 persist nt b256 grid=256x4", tpi_persist<M, R, true>(256, 4)});
    c.push_back({"dopri54 persist nt b64 grid=256x16", tpi_persist<M, R, true>(64, 16)});
    c.push_back({"dopri54 base b256 again", tpi_base<M, R>(256, 0)});
    for (int K : {2, 5, 10}) c.push_back({"dopri54 base b256 K=" + std::to_string(K), tpi_base<M, R>(256, 0, K), K});
    for (int K : {2, 5, 10}) c.push_back({"dopri54 base b64 nt K=" + std::to_string(K), tpi_base<M, R>(64, 1, K), K});
    run_all<R>(("C3 streamed, DOPRI54 " + tag).c_str(), p, c, 10, 60, 3, 8.0 * (4 * 3 + 5));
    std::vector<Candidate> d;
    d.push_back({"tsit54 base b256", tpi_base<NNHIP_TSIT54, R>(256, 0)});
    d.push_back({"tsit54 base b64 nt", tpi_base<NNHIP_TSIT54, R>(64, 1)});
    d.push_back({"tsit54 persist b256 grid=256x4", tpi_persist<NNHIP_TSIT54, R, false>(256, 4)});
    d.push_back({"tsit54 persist nt b256 grid=256x4", tpi_persist<NNHIP_TSIT54, R, true>(256, 4)});
    run_all<R>(("C3 streamed, Tsit54 " + tag).c_str(), p, d, 10, 60, 3, 8.0 * (4 * 3 + 5));
  }
  return 0;
}

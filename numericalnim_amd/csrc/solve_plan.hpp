// solve_plan.hpp — everything of a fused solve's launch record that follows from (options, tspan, integrator) alone: ODESolver's bookkeeping before the
// loops (ode.nim:476-510, 549, 585), the first step size (:491-496) and, for fixed-step methods, the host replay of the reference's time loop (:511-532) that
// yields the step schedule and — with dense output — the step and the four Hermite weights of every requested row.  Plain C++ on top of the argument
// structs: included by ode_capi.hip (the library) and, as it stands, by tests/cpp/emu_solve.cpp, which feeds the kernel BODIES with it on the host.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

#include "ode_kernels.hpp"

namespace nnhip_capi {
using nnhip::SolveArgs;
using nnhip::HermiteW;
using nnhip::hermite_weights;
inline double nmin_h(double x, double y) { return (x <= y) ? x : y; }  // system.min / max of Nim: compare + select, not fmin / fmax
inline double nmax_h(double x, double y) { return (y <= x) ? x : y; }
inline nnhip::StepCtl ctl_of(const nnhip_ode_options* o) { return nnhip::StepCtl{o->absTol, o->relTol, o->dtMax, o->dtMin}; }

struct TimeGrid {
  std::vector<double> sorted, tPos, tNeg /*descending*/, tOut;
  int nZero = 0;
  double tEndPos = 0, tEndNeg = 0;
};
// ODESolver's bookkeeping before the loops (ode.nim:476-487, 510, 549, 585)
inline void make_grid(const nnhip_ode_options* opt, const double* tspan, int n_t, TimeGrid& g) {
  g.sorted.assign(tspan, tspan + n_t);
  std::sort(g.sorted.begin(), g.sorted.end());  // tspan.sorted() (ode.nim:609)
  const double t0 = opt->tStart;
  for (double x : g.sorted) if (x > t0) g.tPos.push_back(x);  // :479
  for (double x : g.sorted) if (x < t0) g.tNeg.push_back(x);  // :480
  std::reverse(g.tNeg.begin(), g.tNeg.end());
  g.nZero = std::find(g.sorted.begin(), g.sorted.end(), t0) != g.sorted.end() ? 1 : 0;  // `t0 in tspan` (:485)
  if (!g.tPos.empty()) { g.tEndPos = g.tPos[0]; for (double x : g.tPos) g.tEndPos = nmax_h(g.tEndPos, x); }
  if (!g.tNeg.empty()) { double mn = g.tNeg[0]; for (double x : g.tNeg) mn = nmin_h(mn, x); g.tEndNeg = -mn; }
  for (auto it = g.tNeg.rbegin(); it != g.tNeg.rend(); ++it) g.tOut.push_back(*it);  // :585
  if (g.nZero) g.tOut.push_back(t0);
  for (double x : g.tPos) g.tOut.push_back(x);
}

// fills a.n_t, nPos / nNeg / nZero, t0, tEndPos / tEndNeg, dtInit, useDense, maxSteps, ctl, uniformFull / nTail / tailDt / nEmit and the emission tables
// (emitW / emitStep: the caller places them where the kernel can read them and points a.emitW / a.emitStep, a.tPos / a.tNeg there)
inline void plan_solve(const nnhip_ode_options* opt, bool adaptive, const double* tspan, int n_t, int64_t max_steps, SolveArgs& a, TimeGrid& g,
                       std::vector<double> (&emitW)[2], std::vector<int64_t> (&emitStep)[2]) {
  make_grid(opt, tspan, n_t, g);
  a.n_t = n_t;
  a.nPos = (int)g.tPos.size(); a.nNeg = (int)g.tNeg.size(); a.nZero = g.nZero;
  a.t0 = opt->tStart; a.tEndPos = g.tEndPos; a.tEndNeg = g.tEndNeg;
  a.dtInit = adaptive ? std::sqrt(opt->dtMax * opt->dtMin) : opt->dt;  // ode.nim:491-496
  a.useDense = (n_t != 2) ? 1 : 0;                                      // ode.nim:499-502
  a.maxSteps = max_steps;
  a.ctl = ctl_of(opt);
  a.uniformFull[0] = a.uniformFull[1] = -1;
  a.nTail[0] = a.nTail[1] = 0;
  a.emitW[0] = a.emitW[1] = nullptr; a.emitStep[0] = a.emitStep[1] = nullptr; a.nEmit[0] = a.nEmit[1] = 0;
  if (!adaptive) {
    // Replay ODESolver's fixed-step time loop on the host (same IEEE double operations, ode.nim:511-532): it does not depend on the
    // state, so the device loop needs no `tEnd - t` / compare / select per step — and, with dense output, no `tReq <= t` test, no
    // per-step lastIter copy and no per-lane Hermite weights either: the step at whose start each requested row is interpolated and the
    // four weights of utils.nim:273-279 come out of the same replay (DriveIn::emitStep / emitW).
    const double tS[2] = {opt->tStart, -opt->tStart}, tE[2] = {g.tEndPos, g.tEndNeg};
    const bool have[2] = {a.nPos > 0, a.nNeg > 0};
    for (int dir = 0; dir < 2; ++dir) {
      if (!have[dir]) { a.uniformFull[dir] = 0; continue; }
      if (!((tE[dir] - tS[dir]) / opt->dt < 5e7)) continue;  // keep the replay itself negligible; generic path otherwise
      const std::vector<double>& req = dir == 0 ? g.tPos : g.tNeg;
      const int high = (int)req.size() - 1;
      double t = tS[dir], dt = opt->dt, lastT = tS[dir];
      int64_t full = 0, total = 0;
      int nTail = 0, denseIndex = 0;
      bool ok = true;
      while (t < tE[dir]) {  // :511
        if (max_steps > 0 && total >= max_steps) break;
        if (a.useDense) {      // :512-524
          if (high < denseIndex) break;
          while ((dir == 0 ? req[denseIndex] : -req[denseIndex]) <= t) {
            if (total == 0) { ok = false; break; }  // (a requested time at or before the start of the first step: cannot happen, tPositive > t0)
            const HermiteW w = hermite_weights(dir == 0 ? req[denseIndex] : -req[denseIndex], lastT, t);
            emitW[dir].insert(emitW[dir].end(), {w.h00, w.h10w, w.h01, w.h11w});
            emitStep[dir].push_back(total);
            denseIndex += 1;
            if (high < denseIndex) break;
          }
          if (!ok) break;
        }
        const double dtc = nmin_h(dt, tE[dir] - t);  // :525
        lastT = t;                                   // :526-530
        if (nTail == 0 && dtc == opt->dt) ++full;
        else if (nTail < 4) a.tailDt[dir][nTail++] = dtc;
        else { ok = false; break; }
        dt = dtc;  // fixed-step steppers hand their input dt back (ode.nim:189): a clipped dt persists
        t += dtc;  // :532
        ++total;
      }
      if (ok) { a.uniformFull[dir] = full; a.nTail[dir] = nTail; a.nEmit[dir] = (int)emitStep[dir].size(); }
      else { emitW[dir].clear(); emitStep[dir].clear(); }
    }
  }
}
}  // namespace nnhip_capi

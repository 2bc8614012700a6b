// quad_plan.hpp — everything cumtrapz(f, X, ctx, dx) / cumsimpson(f, X, ctx, dx) (integrate.nim:138-175, 377-400) decide from X and dx alone, replayed on the
// host with the reference's own IEEE operations: the size of the sampling grid, which grid interval every result row is taken from and its Hermite basis
// weights (hermiteInterpolate, utils.nim:282-312), the non-uniform Simpson weights.  Plain C++ (no HIP call): ode_capi_quad.hip uploads the tables and launches;
// the CPU test suite feeds the same plan to the kernel bodies executed on the host (tests/cpp/emu_consumers.cpp).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "quad_kernels.hpp"

namespace nnhip {

constexpr int64_t kMaxGrid = (int64_t)1 << 31;        // cumtrapz: marched on the fly, nothing stored per grid point
constexpr int64_t kMaxSimpsonGrid = (int64_t)1 << 24; // cumsimpson: 11 doubles of X-only tables per interval pair live in HBM

// hermiteInterpolate(x, t, y, dy) (utils.nim:282-312) restricted to what depends on x and t: which grid interval every result row
// is taken from.  `gridAt(i)` yields t[i] for i in [0, nGrid); it is called with non-decreasing i only, so the caller may
// generate the grid on the fly.  Returns false for the reference's ValueError (unsorted branch, x outside the grid).
struct RowPlan {
  std::vector<QuadEmit> emits;     // rows taken from an interval
  std::vector<int32_t> lastRows;   // rows that are y[y.high]
  int nRows = 0;
};
template <class GridAt>
bool plan_rows(const double* x, int n_x, int64_t nGrid, GridAt gridAt, RowPlan& plan) {
  const bool sorted = std::is_sorted(x, x + n_x);  // isSorted: non-decreasing
  plan = RowPlan();
  if (sorted) {  // :290-300 — one pass over the intervals, the queries consumed in order
    int xIndex = 0;
    bool done = false;
    double lo = gridAt(0);
    double last = lo;
    for (int64_t i = 0; i + 1 < nGrid; ++i) {
      const double hi = gridAt(i + 1);
      last = hi;
      if (!done) {
        while (lo <= x[xIndex] && x[xIndex] < hi) {
          QuadEmit e;
          e.interval = (int32_t)i;
          e.row = plan.nRows++;
          hermite_spline_weights(x[xIndex], lo, hi, e.w);
          plan.emits.push_back(e);
          xIndex += 1;
          if (n_x - 1 < xIndex) { done = true; break; }
        }
      }
      lo = hi;
    }
    if (x[n_x - 1] == last) plan.lastRows.push_back(plan.nRows++);
    return true;
  }
  // :302-311 — every a looks for the first interval with t[i] <= a < t[i+1]; the intervals are disjoint and ascending, so one pass
  // over the queries in ascending order finds the same interval; rows keep the order of x.
  std::vector<int> order((size_t)n_x);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return x[a] < x[b]; });
  std::vector<int64_t> interval((size_t)n_x, -1);
  std::vector<double> lo_of((size_t)n_x, 0.0), hi_of((size_t)n_x, 0.0);
  size_t q = 0;
  double lo = gridAt(0), last = lo;
  for (int64_t i = 0; i + 1 < nGrid; ++i) {
    const double hi = gridAt(i + 1);
    last = hi;
    while (q < order.size() && x[order[q]] < lo) ++q;  // below the grid (or NaN-free gap): no interval
    while (q < order.size() && lo <= x[order[q]] && x[order[q]] < hi) {
      interval[(size_t)order[q]] = i; lo_of[(size_t)order[q]] = lo; hi_of[(size_t)order[q]] = hi;
      ++q;
    }
    lo = hi;
  }
  for (int j = 0; j < n_x; ++j) {
    if (interval[(size_t)j] >= 0) {
      QuadEmit e;
      e.interval = (int32_t)interval[(size_t)j];
      e.row = plan.nRows++;
      hermite_spline_weights(x[j], lo_of[(size_t)j], hi_of[(size_t)j], e.w);
      plan.emits.push_back(e);
    } else if (x[j] == last) {
      plan.lastRows.push_back(plan.nRows++);
    } else {
      return false;  // ValueError "{a} not in interval" (:311)
    }
  }
  std::stable_sort(plan.emits.begin(), plan.emits.end(), [](const QuadEmit& a, const QuadEmit& b) { return a.interval < b.interval; });
  return true;
}

// The plan of one call: `a` holds the grid (x0, dx, xLast, nGrid, nPairs, evenN) and the point the march may stop at (nPoints); the caller adds the
// output layout, the parameters and the device addresses of the tables.
struct CumquadPlan {
  RowPlan rows;
  std::vector<SimpsonPair> pairs;
  std::vector<SimpsonPoint> pts;
  QuadArgs a;
};
// rule 0: cumtrapz, 1: cumsimpson.  -> NNHIP_OK, or NNHIP_EVALUE / NNHIP_EUNSUPPORTED with the reason in `err`.  X: n_x >= 1 finite values, dx > 0 finite.
inline int plan_cumquad(int rule, const double* X, int n_x, double dx, CumquadPlan& pl, std::string& err) {
  const double lo = *std::min_element(X, X + n_x), hi = *std::max_element(X, X + n_x);
  RowPlan& plan = pl.rows;
  QuadArgs& a = pl.a;
  std::memset(&a, 0, sizeof(a));
  pl.pairs.clear();
  pl.pts.clear();
  char buf[256];
  if (rule == 0) {
    // replay `t = min(X); t += dx; while t <= max(X) + 1.0` (integrate.nim:160-174) to learn the grid size
    const double tEnd = hi + 1.0;
    if (!(lo + dx > lo) || !(tEnd + dx > tEnd)) {
      std::snprintf(buf, sizeof(buf), "cumtrapz(f, X): dx = %g does not advance t near %g (the reference would loop forever)", dx, tEnd);
      err = buf;
      return NNHIP_EVALUE;
    }
    if ((tEnd - lo) / dx > (double)kMaxGrid) { err = "cumtrapz(f, X): more than 2^31 grid points"; return NNHIP_EUNSUPPORTED; }
    int64_t nGrid = 1;
    for (double t = lo + dx; t <= tEnd; t += dx) ++nGrid;
    double tCur = lo;
    int64_t iCur = 0;
    auto gridAt = [&](int64_t i) {  // non-decreasing i
      while (iCur < i) { tCur += dx; ++iCur; }
      return tCur;
    };
    if (!plan_rows(X, n_x, nGrid, gridAt, plan)) { err = "cumtrapz(f, X): a value of X lies outside the integration grid (ValueError, utils.nim:311)"; return NNHIP_EVALUE; }
    a.x0 = lo; a.dx = dx; a.xLast = 0.0; a.nGrid = nGrid;
  } else {
    // t = linspace(min(X), max(X), ((max(X) - min(X)) / dx).toInt + 2)   (integrate.nim:395; toInt rounds half away from zero)
    const double cnt = std::round((hi - lo) / dx);
    if (cnt + 2.0 > (double)kMaxSimpsonGrid) { err = "cumsimpson(f, X): more than 2^24 grid points"; return NNHIP_EUNSUPPORTED; }
    const int64_t nGrid = (int64_t)cnt + 2;
    if (nGrid < 3) { err = "X and Y must have at least 3 elements to perform Simpson, use cumtrapz instead"; return NNHIP_EVALUE; }  // :345-346
    const double step = (hi - lo) / (double)(nGrid - 1);
    std::vector<double> t((size_t)nGrid);
    t[0] = lo;
    for (int64_t i = 1; i <= nGrid - 2; ++i) t[(size_t)i] = lo + step * (double)i;
    t[(size_t)nGrid - 1] = hi;
    for (int64_t i = 1; i < nGrid; ++i)
      if (!(t[(size_t)i - 1] < t[(size_t)i])) {  // cumsimpson(dy, t) would sort / trim the grid (sortAndTrimDataset): not a linspace any more
        err = "cumsimpson(f, X): the sampling grid is not strictly increasing (max(X) == min(X) or dx below the spacing of doubles)";
        return NNHIP_EUNSUPPORTED;
      }
    if (!plan_rows(X, n_x, nGrid, [&](int64_t i) { return t[(size_t)i]; }, plan)) {
      err = "cumsimpson(f, X): a value of X lies outside the integration grid (ValueError, utils.nim:311)";
      return NNHIP_EVALUE;
    }
    bool evenN = false;
    simpson_tables(t.data(), nGrid, pl.pairs, pl.pts, a.nPairs, evenN);
    a.evenN = evenN ? 1 : 0;
    a.x0 = lo; a.dx = step; a.xLast = hi; a.nGrid = nGrid;
  }
  // the march stops after the last grid point any row needs
  a.nPoints = plan.lastRows.empty() ? (plan.emits.empty() ? 1 : (int64_t)plan.emits.back().interval + 2) : a.nGrid;
  a.nEmit = (int)plan.emits.size();
  a.nLast = (int)plan.lastRows.size();
  if (rule == 1 && a.nPoints < a.nGrid) {  // only the pairs the march reaches are needed on the device
    const size_t needPairs = std::min((size_t)a.nPairs + 1, (size_t)(a.nPoints / 2 + 2));
    const size_t needPts = std::min(pl.pts.size(), (size_t)a.nPoints + 3);
    pl.pairs.resize(std::max(needPairs, (size_t)1));
    pl.pts.resize(needPts);
  }
  return NNHIP_OK;
}

}  // namespace nnhip

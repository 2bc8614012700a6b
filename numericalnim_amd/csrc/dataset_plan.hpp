// dataset_plan.hpp — sortAndTrimDataset (utils.nim:404-413 = sortDataset :384-402 + removeDuplicates :360-381), the front end of the discrete consumers
// cumtrapz(Y, X) (integrate.nim:131), cumsimpson(Y, X) (:340) and newHermiteSpline (interpolate.nim:231, 244), decided from the caller's X alone on the host:
// which of the caller's rows supplies each row of the sorted, duplicate-free dataset, which rows must hold equal values (a "pure" duplicate; an impure one
// raises ValueError in the reference), and — for cumsimpson, which interpolates back to the caller's abscissae (hermiteInterpolate, utils.nim:282-312) — which
// sorted row each result row is read from.  Plain C++ (no HIP): used by ode_capi_aux.hip and, as it stands, by tests/cpp/emu_consumers.cpp.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <numeric>
#include <string>
#include <vector>

namespace nnhip {
struct DatasetPlan {
  bool identity = true;          // X strictly ascending: the dataset is the caller's, nothing moves
  bool inputSorted = true;       // isSorted(X) — non-decreasing: the branch hermiteInterpolate takes (utils.nim:289)
  std::vector<double> x;         // sorted, one entry per distinct abscissa (the first of each run of equal values in sorted order)
  std::vector<int32_t> src;      // row k of the sorted dataset = the caller's row src[k]
  std::vector<int32_t> dupKeep, dupDrop;  // caller's rows (kept, deleted) with the same abscissa: every y must be equal (removeDuplicates :367-372)
  std::vector<int32_t> rank;     // [n] the caller's row j has abscissa x[rank[j]]
};

// -> 0, or -1 with the reason in `err` (NaN in X: the reference's comparison sort has no defined order for it).
inline int dataset_plan(const double* X, int n, DatasetPlan& pl, std::string& err) {
  pl = DatasetPlan{};
  for (int i = 0; i < n; ++i)
    if (X[i] != X[i]) { err = "X[" + std::to_string(i) + "] is NaN: sortAndTrimDataset has no defined order for it"; return -1; }
  for (int i = 1; i < n; ++i) {
    if (!(X[i - 1] < X[i])) pl.identity = false;
    if (X[i - 1] > X[i]) pl.inputSorted = false;     // isSorted: cmp(a[i], a[i+1]) > 0 -> false
  }
  if (pl.identity) {
    pl.x.assign(X, X + n);
    pl.src.resize((size_t)n);
    std::iota(pl.src.begin(), pl.src.end(), 0);
    pl.rank = pl.src;
    return 0;
  }
  // sortDataset: zip(x, 0 .. n-1) sorted ascending as tuples (:392-393) — by value, equal values (-0.0 == 0.0 included) by original index
  std::vector<int32_t> order((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return X[a] < X[b]; });
  // removeDuplicates: of every run of equal abscissae the first stays (idxDelete.add dups[1 .. ^1], :376), the others must be pure duplicates of it
  pl.rank.assign((size_t)n, 0);
  for (int k = 0; k < n; ++k) {
    const int32_t j = order[(size_t)k];
    if (!pl.x.empty() && X[j] == pl.x.back()) {
      pl.dupKeep.push_back(pl.src.back());
      pl.dupDrop.push_back(j);
    } else {
      pl.x.push_back(X[j]);
      pl.src.push_back(j);
    }
    pl.rank[(size_t)j] = (int32_t)pl.x.size() - 1;
  }
  return 0;
}

// Rows of cumsimpson(Y, X)'s result = hermiteInterpolate(X, xs, y, dy) (integrate.nim:375; utils.nim:282-312), as indices into the per-abscissa values of the
// sorted dataset: unsorted X — every caller's row, in the caller's order (:303-311); sorted X — the rows below the maximum in order (:290-299), then the maximum
// ONCE (`if x[x.high] == t[t.high]: result.add(y[y.high])`, :300-301), however often the caller repeated it.
inline void simpson_result_rows(const DatasetPlan& pl, const double* X, int n, std::vector<int32_t>& rows) {
  rows.clear();
  const int32_t last = (int32_t)pl.x.size() - 1;
  if (!pl.inputSorted) {
    rows.assign(pl.rank.begin(), pl.rank.end());
    return;
  }
  for (int j = 0; j < n; ++j)
    if (X[j] < pl.x.back()) rows.push_back(pl.rank[(size_t)j]);
  rows.push_back(last);
}
}  // namespace nnhip

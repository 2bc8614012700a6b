// consumer_kernels.hpp — the kernels of the output consumers and the host arithmetic that feeds them:
//   hermiteSpline over a flat batch (utils.nim:273-279), newHermiteSpline(X, Y[, dY]).eval / .derivEval with every ExtrapolateKind
//   (interpolate.nim:186-253, 299-390), cumtrapz(Y, X) / cumsimpson(Y, X) (integrate.nim:120-135, 329-375).
// Everything that depends on X and the query points only — interval search, basis weights, extrapolation branch, interval weights — is computed on the host in
// the reference's expression order (plain C++ below, no HIP call) and shipped as kernel arguments; the kernels do the per-series part.
// Included by ONE translation unit of the library (ode_capi_aux.hip: the kernels are ordinary, non-inline __global__ functions) and by the CPU test suite,
// which runs the kernel bodies on the host (tests/cpp/emu_consumers.cpp, -DNNHIP_CPU_EMU).
#pragma once
#include <algorithm>
#include <cstring>

#include "quad_kernels.hpp"

namespace nnhip {

// hermiteSpline (utils.nim:273-279) over a flat batch
// negate_dy: the slopes are those of g(t, y) = -f(-t, y) (backward branch, ode.nim:545) while dy1 / dy2 hold f: use their negatives
__global__ __launch_bounds__(kBlock) void hermite_kernel(double x, double x1, double x2, const double* __restrict__ y1,
                                                         const double* __restrict__ y2, const double* __restrict__ dy1,
                                                         const double* __restrict__ dy2, double* __restrict__ out, int64_t n, int negate_dy) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const HermiteW w = hermite_weights(x, x1, x2);
  const double d1 = negate_dy ? -dy1[i] : dy1[i], d2 = negate_dy ? -dy2[i] : dy2[i];
  out[i] = hermite_apply(w, y1[i], y2[i], d1, d2);
}

// newHermiteSpline(X, Y, dY).eval / .derivEval over M independent series (interpolate.nim:186-217, 299-390).
// Everything that depends only on the query point (interval, basis weights, extrapolation branch) is computed on
// the host in the reference's expression order and shipped as a descriptor; the kernel does the per-series part.
struct HermSet {
  int row;           // knots (row, row+1)
  double w[4];       // h00, h10*xDiff, h01, h11*xDiff
  double xDiff;
};
struct HermQuery {
  int mode;          // 0 eval, 1 derivEval, 2 constant, 3 copy row A, 4 linear between rows A/B, 5 linear between derivEval sets A/B
  HermSet a, b;
  double k, value;
};
constexpr int kHermChunk = 24;
struct HermChunk {
  HermQuery q[kHermChunk];
};

NNHIP_DEV double herm_apply(const HermSet& s, const double* __restrict__ Y, const double* __restrict__ dY, int64_t M, int64_t m, bool deriv) {
  const double p1 = Y[(int64_t)s.row * M + m], p2 = Y[(int64_t)(s.row + 1) * M + m];
  const double m1 = dY[(int64_t)s.row * M + m], m2 = dY[(int64_t)(s.row + 1) * M + m];
  const double v = s.w[0] * p1 + s.w[1] * m1 + s.w[2] * p2 + s.w[3] * m2;  // h00*p1 + h10*xDiff*m1 + h01*p2 + h11*xDiff*m2
  return deriv ? v / s.xDiff : v;
}

__global__ __launch_bounds__(kBlock) void hermite_interp_kernel(const HermChunk c, int nq, const double* __restrict__ Y,
                                                                const double* __restrict__ dY, int64_t M, double* __restrict__ out) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int qi = blockIdx.y;
  if (m >= M || qi >= nq) return;
  const HermQuery& q = c.q[qi];
  double r;
  switch (q.mode) {
    case 0: r = herm_apply(q.a, Y, dY, M, m, false); break;
    case 1: r = herm_apply(q.a, Y, dY, M, m, true); break;
    case 2: r = q.value; break;
    case 3: r = Y[(int64_t)q.a.row * M + m]; break;
    case 4: { const double y0 = Y[(int64_t)q.a.row * M + m], y1 = Y[(int64_t)q.b.row * M + m]; r = y0 + q.k * (y1 - y0); break; }
    default: { const double y0 = herm_apply(q.a, Y, dY, M, m, true), y1 = herm_apply(q.b, Y, dY, M, m, true); r = y0 + q.k * (y1 - y0); break; }
  }
  out[(int64_t)qi * M + m] = r;
}

// The slopes newHermiteSpline(X, Y) estimates when no derivatives are given (interpolate.nim:241-253): one-sided differences at
// the ends, the mean of the two adjacent difference quotients inside.  Thread per (knot, series); invDx-free: the reference divides.
__global__ __launch_bounds__(kBlock) void hermite_slopes_kernel(const double* __restrict__ X, int n, const double* __restrict__ Y,
                                                                int64_t M, double* __restrict__ dY) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int i = blockIdx.y;
  if (m >= M || i >= n) return;
  const int64_t r = (int64_t)i * M + m;
  if (i == 0) dY[r] = (Y[r + M] - Y[r]) / (X[1] - X[0]);
  else if (i == n - 1) dY[r] = (Y[r] - Y[r - M]) / (X[n - 1] - X[n - 2]);
  else dY[r] = 0.5 * ((Y[r + M] - Y[r]) / (X[i + 1] - X[i]) + (Y[r] - Y[r - M]) / (X[i] - X[i - 1]));
}

// cumtrapz(Y, X) over M series (integrate.nim:120-135): thread per series marches down the time axis;
// the interval weights 0.5*(x_{i+1}-x_i) are computed on the host (same IEEE ops) and arrive as arguments.
constexpr int kTrapzChunk = 384;
struct TrapzWeights {
  double w[kTrapzChunk];
};
__global__ __launch_bounds__(kBlock) void cumtrapz_kernel(const TrapzWeights W, int nw, int first, const double* __restrict__ Y,
                                                          double* __restrict__ out, int64_t M) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  // rows [first, first + nw] of Y / out; out[first] is already final unless first == 0
  double yPrev = Y[(int64_t)first * M + m];
  double integral;
  if (first == 0) {
    integral = yPrev - yPrev;  // "the right kind of zero" (:131-132): NaN/Inf states stay NaN
    out[m] = integral;
  } else {
    integral = out[(int64_t)first * M + m];
  }
  for (int i = 0; i < nw; ++i) {
    const double yNext = Y[(int64_t)(first + i + 1) * M + m];
    integral += W.w[i] * (yNext + yPrev);  // 0.5 * (x[i+1] - x[i]) * (y[i+1] + y[i])  (:134)
    out[(int64_t)(first + i + 1) * M + m] = integral;
    yPrev = yNext;
  }
}

// cumsimpson(Y, X) over M series (integrate.nim:329-375): composite Simpson on interval pairs + hermiteInterpolate
// (utils.nim:282-312) with dy = Y.  Per-point weights depend only on X: computed on the host in reference order.
__global__ __launch_bounds__(kBlock) void cumsimpson_kernel(const SimpsonPair* __restrict__ pairs, int nPairs, int evenN,
                                                            const SimpsonPoint* __restrict__ pts, const double* __restrict__ Y,
                                                            double* __restrict__ out, int64_t M, int n) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  auto herm = [](const SimpsonPoint& p, double y1, double y2, double dy1, double dy2) {
    return p.w[0] * y1 + p.w[1] * dy1 + p.w[2] * y2 + p.w[3] * dy2;  // h00*y1 + h10*(x2-x1)*dy1 + h01*y2 + h11*(x2-x1)*dy2
  };
  double y0 = Y[m];
  double integral = y0 - y0;  // the right kind of zero (:350)
  for (int i = 0; i < nPairs; ++i) {
    const double y1 = Y[(int64_t)(2 * i + 1) * M + m], y2 = Y[(int64_t)(2 * i + 2) * M + m];
    const SimpsonPair w = pairs[i];
    const double next = integral + (w.alpha * y2 + w.beta * y1 + w.eta * y0);
    out[(int64_t)(2 * i) * M + m] = herm(pts[2 * i], integral, next, y0, y2);
    out[(int64_t)(2 * i + 1) * M + m] = herm(pts[2 * i + 1], integral, next, y0, y2);
    integral = next;
    y0 = y2;
  }
  if (evenN) {  // odd number of intervals: the last one is closed with the three-point rule of :363-373
    const int l = n - 1;
    const double ym2 = Y[(int64_t)(l - 2) * M + m], ym1 = y0, yl = Y[(int64_t)l * M + m];
    const SimpsonPair w = pairs[nPairs];
    const double next = integral + (w.eta * ym2 + w.beta * ym1 + w.alpha * yl);
    out[(int64_t)(l - 1) * M + m] = herm(pts[l - 1], integral, next, ym1, yl);
    integral = next;
  }
  out[(int64_t)(n - 1) * M + m] = integral;  // `if x[x.high] == t[t.high]: result.add(y[y.high])` (utils.nim:300-301)
}

// sortAndTrimDataset on the device side (utils.nim:360-407; the host decides the permutation from X: dataset_plan.hpp).
// Rows of the sorted, duplicate-free dataset gathered from the caller's (sortDataset :399-402 + the deletions of removeDuplicates :377-381); also the
// inverse step of cumsimpson(Y, X), whose result is read back at the caller's abscissae (hermiteInterpolate, integrate.nim:375).  src[k] < 0: a NaN row
// (rows the reference's result does not have).  Thread per (row, series).
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const int32_t* __restrict__ src, int nRows, const double* __restrict__ Y, double* __restrict__ out, int64_t M) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  for (int k = blockIdx.y; k < nRows; k += gridDim.y) {  // (gridDim.y <= 65535: long series take several rows per block row)
    const int32_t j = src[k];
    out[(int64_t)k * M + m] = j >= 0 ? Y[(int64_t)j * M + m] : __longlong_as_double(0x7ff8000000000000LL);
  }
}
// removeDuplicates' check (:367-372): rows with the same abscissa must hold the same values (`y[iy][i] != ys[iy]` raises; NaN != NaN, so a NaN
// duplicate is impure there and here).  *flag becomes 1 if any pair differs anywhere in the batch.
__global__ __launch_bounds__(kBlock) void dup_rows_differ_kernel(const int32_t* __restrict__ keep, const int32_t* __restrict__ drop, int nPairs, const double* __restrict__ Y,
                                                                 int64_t M, unsigned int* __restrict__ flag) {
  const int64_t m = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  for (int p = blockIdx.y; p < nPairs; p += gridDim.y)
    if (Y[(int64_t)keep[p] * M + m] != Y[(int64_t)drop[p] * M + m]) *flag = 1u;
}

inline HermSet herm_set(const double* X, int n, double x, bool deriv) {
  // findInterval (interpolate.nim:114-115): clamp(lowerbound(X, x) - 1, 0, high - 1)
  int k = (int)(std::lower_bound(X, X + n, x) - X) - 1;
  if (k < 0) k = 0;
  if (k > n - 2) k = n - 2;
  HermSet s;
  s.row = k;
  const double xDiff = X[k + 1] - X[k];
  const double t = (x - X[k]) / xDiff;
  const double t2 = t * t;
  s.xDiff = xDiff;
  if (!deriv) {  // interpolate.nim:190-195
    const double t3 = t2 * t;
    const double h00 = 2 * t3 - 3 * t2 + 1, h10 = t3 - 2 * t2 + t, h01 = -2 * t3 + 3 * t2, h11 = t3 - t2;
    s.w[0] = h00; s.w[1] = h10 * xDiff; s.w[2] = h01; s.w[3] = h11 * xDiff;
  } else {  // :207-211
    const double h00 = 6 * t2 - 6 * t, h10 = 3 * t2 - 4 * t + 1, h01 = -6 * t2 + 6 * t, h11 = 3 * t2 - 2 * t;
    s.w[0] = h00; s.w[1] = h10 * xDiff; s.w[2] = h01; s.w[3] = h11 * xDiff;
  }
  return s;
}

// the descriptors of queries xq[0 .. nq) (nq <= kHermChunk): eval / derivEval inside the knots, the ExtrapolateKind's branch outside (0 Constant, 1 Edge, 2 Linear,
// 3 Native, 4 Error — refused by the caller)
inline void herm_chunk_fill(const double* X, int n_knots, const double* xq, int nq, bool deriv, int extrap, double extrap_value, HermChunk& c) {
  std::memset(&c, 0, sizeof(c));
  for (int j = 0; j < nq; ++j) {
    const double x = xq[j];
    HermQuery& hq = c.q[j];
    const bool xLeft = x < X[0], xRight = x > X[n_knots - 1];
    hq.mode = deriv ? 1 : 0;
    hq.a = herm_set(X, n_knots, x, deriv);
    if (xLeft || xRight) {  // interpolate.nim:317-341 / 364-388
      if (extrap == 0) { hq.mode = 2; hq.value = extrap_value; }
      else if (extrap == 1) {
        if (!deriv) { hq.mode = 3; hq.a.row = xLeft ? 0 : n_knots - 1; }
        else hq.a = herm_set(X, n_knots, xLeft ? X[0] : X[n_knots - 1], true);
      } else if (extrap == 2) {
        const int r0 = xLeft ? 0 : n_knots - 2, r1 = r0 + 1;
        hq.k = (x - X[r0]) / (X[r1] - X[r0]);
        if (!deriv) { hq.mode = 4; hq.a.row = r0; hq.b.row = r1; }
        else { hq.mode = 5; hq.a = herm_set(X, n_knots, X[r0], true); hq.b = herm_set(X, n_knots, X[r1], true); }
      }
    }
  }
}
// interval weights 0.5 * (x[i+1] - x[i]) (integrate.nim:134) of rows first .. first + nw of an n-point grid; -> nw (<= kTrapzChunk)
inline int trapz_weights_fill(const double* X, int n, int first, TrapzWeights& W) {
  const int nw = std::min(kTrapzChunk, n - 1 - first);
  for (int i = 0; i < nw; ++i) W.w[i] = 0.5 * (X[first + i + 1] - X[first + i]);
  return nw;
}

}  // namespace nnhip

// quad_kernels.hpp — cumulative quadrature of a *function* of x on the device ("march + Hermite resample"):
//   cumtrapz(f, X, ctx, dx)    src/numericalnim/integrate.nim:138-175
//   cumsimpson(f, X, ctx, dx)  src/numericalnim/integrate.nim:377-400  (-> cumsimpson(Y, X) :329-375)
//   hermiteInterpolate         src/numericalnim/utils.nim:282-312, hermiteSpline :273-279
// The reference samples a host closure f(x, ctx) on a fine grid, accumulates the rule left to right and resamples the running
// integral at X with cubic Hermite pieces.  Here f(x) := rhs(x, y = 0, params) for any registered right-hand side (the
// integrand is a function of x alone: NumContextProc, integrate.nim:9), one thread marches one parameter set (the batch axis
// is a parameter sweep), nothing but the requested rows ever reaches HBM.  Everything that depends on X / dx only (which grid
// interval a query falls into, its Hermite basis weights, the non-uniform Simpson weights of each interval pair) is replayed on
// the host with the reference's own IEEE operations and shipped as tables; a wavefront reads them at uniform addresses.
#pragma once
#include "ode_kernels.hpp"
#if !NNHIP_RTC || defined(NNHIP_CPU_EMU)  // (the host helpers below: the library, and the CPU test suite that runs the kernel bodies on the host)
#include <vector>
#endif

namespace nnhip_abi {

struct QuadEmit {   // one row of the result: hermiteSpline over grid interval [t_interval, t_interval+1)
  int32_t interval;
  int32_t row;
  double w[4];      // h00, h10*(x2-x1), h01, h11*(x2-x1)   (utils.nim:274-279, left to right)
};
struct SimpsonPair {  // integral += alpha*y2 + beta*y1 + eta*y0  (integrate.nim:355-359; tail rule :366-370)
  double alpha, beta, eta;
};
struct SimpsonPoint {  // hermiteSpline weights of a grid point inside its pair interval (cumsimpson's own resampling, :375)
  double w[4];
};

struct QuadArgs {
  double* out;                // row r, component c, item i at out[r*rowStride + c*compStride + i*ivpStride]
  int64_t N;
  int64_t ivpStride, compStride, rowStride;
  Params P;
  const double* perIvpParams;  // nullable [nPerIvp][N]: parameter k of item i overrides P.p[k] (as SolveArgs)
  int nPerIvp;
  int64_t perIvpStride;
  double x0;                  // first grid point = min(X)
  double dx;                  // cumtrapz: the caller's dx (t += dx); cumsimpson: the linspace spacing (hi - lo) / (nGrid - 1)
  double xLast;               // cumsimpson: the last grid point, exactly max(X) (linspace appends x2 itself, utils.nim:507)
  int64_t nGrid;              // number of grid points of the reference's march
  int64_t nPoints;            // the march may stop once grid point nPoints-1 has been produced (later points reach no row)
  const QuadEmit* emits;      // device, sorted by interval
  int nEmit;
  const int32_t* lastRows;    // device: rows that receive the integral at the last grid point (`x[x.high] == t[t.high]`, utils.nim:299)
  int nLast;
  const SimpsonPair* pairs;   // device, nPairs (+1 with the tail rule)
  const SimpsonPoint* pts;    // device, one per grid point
  int64_t nPairs;
  int evenN;
};

#if !NNHIP_RTC || defined(NNHIP_CPU_EMU)
// hermiteSpline's basis weights for x in [x1, x2] (utils.nim:273-279), in the reference's expression order
inline void hermite_spline_weights(double x, double x1, double x2, double (&w)[4]) {
  const double t = (x - x1) / (x2 - x1);
  const double omt = 1.0 - t;
  w[0] = (1.0 + 2.0 * t) * (omt * omt);
  w[1] = (t * (omt * omt)) * (x2 - x1);
  w[2] = (t * t) * (3.0 - 2.0 * t);
  w[3] = ((t * t * t) - (t * t)) * (x2 - x1);
}
// The X-only part of cumsimpson(Y, X) (integrate.nim:347-375): Simpson weights of every interval pair (+ the tail rule when the
// number of points is even, stored at index nPairs) and the Hermite weights that resample the pair integrals at every point.
inline void simpson_tables(const double* X, int64_t n, std::vector<SimpsonPair>& pairs, std::vector<SimpsonPoint>& pts, int64_t& nPairs,
                           bool& evenN) {
  evenN = (n % 2) == 0;
  const int64_t N = evenN ? n - 1 : n;
  nPairs = (N - 1) / 2;
  auto cube = [](double v) { return v * v * v; };  // Nim's `^` with a literal 3 / 2: repeated multiplication
  auto sq = [](double v) { return v * v; };
  pairs.assign((size_t)nPairs + 1, SimpsonPair{0.0, 0.0, 0.0});
  pts.assign((size_t)n, SimpsonPoint{{0.0, 0.0, 0.0, 0.0}});
  for (int64_t i = 0; i < nPairs; ++i) {  // integrate.nim:354-358
    const double h1 = X[2 * i + 1] - X[2 * i], h2 = X[2 * i + 2] - X[2 * i + 1];
    pairs[i].alpha = (2.0 * cube(h2) - cube(h1) + 3.0 * h1 * sq(h2)) / (6.0 * h2 * (h2 + h1));
    pairs[i].beta = (cube(h2) + cube(h1) + 3.0 * h1 * h2 * (h2 + h1)) / (6.0 * h2 * h1);
    pairs[i].eta = (2.0 * cube(h1) - cube(h2) + 3.0 * h2 * sq(h1)) / (6.0 * h1 * (h2 + h1));
    hermite_spline_weights(X[2 * i], X[2 * i], X[2 * i + 2], pts[2 * i].w);
    hermite_spline_weights(X[2 * i + 1], X[2 * i], X[2 * i + 2], pts[2 * i + 1].w);
  }
  if (evenN) {  // :363-370
    const int64_t l = n - 1;
    const double h1 = X[l - 1] - X[l - 2], h2 = X[l] - X[l - 1];
    pairs[nPairs].alpha = (2.0 * sq(h2) + 3.0 * h1 * h2) / (6.0 * (h1 + h2));
    pairs[nPairs].beta = (sq(h2) + 3.0 * h1 * h2) / (6.0 * h1);
    pairs[nPairs].eta = -(cube(h2)) / (6.0 * h1 * (h1 + h2));
    hermite_spline_weights(X[l - 1], X[l - 1], X[l], pts[l - 1].w);
  }
}
#endif  // !NNHIP_RTC || NNHIP_CPU_EMU

}  // namespace nnhip_abi

namespace NNHIP_NS {

template <int D>
NNHIP_DEV void quad_emit(const QuadArgs& a, int64_t i, int& k, int64_t upTo, const double (&y1)[D], const double (&dy1)[D],
                         const double (&y2)[D], const double (&dy2)[D]) {
  // every requested x inside grid interval `upTo - 1`, whose two end points (y1, dy1), (y2, dy2) are now known
  while (k < a.nEmit && (int64_t)a.emits[k].interval < upTo) {
    const QuadEmit e = a.emits[k];
#pragma unroll
    for (int c = 0; c < D; ++c)  // h00*y1 + h10*(x2-x1)*dy1 + h01*y2 + h11*(x2-x1)*dy2  (utils.nim:279)
      a.out[(int64_t)e.row * a.rowStride + c * a.compStride + i * a.ivpStride] =
          e.w[0] * y1[c] + e.w[1] * dy1[c] + e.w[2] * y2[c] + e.w[3] * dy2[c];
    ++k;
  }
}

template <int D>
NNHIP_DEV void quad_last_rows(const QuadArgs& a, int64_t i, const double (&integral)[D]) {
  for (int r = 0; r < a.nLast; ++r)
#pragma unroll
    for (int c = 0; c < D; ++c) a.out[(int64_t)a.lastRows[r] * a.rowStride + c * a.compStride + i * a.ivpStride] = integral[c];
}

// cumtrapz(f, X, ctx, dx): t = min(X); integral += 0.5*dx*(f(t_prev) + f(t)) while t <= max(X) + 1  (integrate.nim:160-174)
template <class RHS>
__global__ __launch_bounds__(kBlock) void cumtrapz_fn_kernel(const QuadArgs a) {
  constexpr int D = RHS::dim;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.N) return;
  const Params P = params_of(a, i);
  double zero[D], dyPrev[D], dyNext[D], integral[D], next[D];
#pragma unroll
  for (int c = 0; c < D; ++c) zero[c] = 0.0;
  double t = a.x0;
  RHS::eval(t, zero, dyPrev, P);
#pragma unroll
  for (int c = 0; c < D; ++c) integral[c] = dyPrev[c] - dyPrev[c];  // "the right kind of zero" (:162): NaN/Inf stay NaN
  const double hdx = 0.5 * a.dx;                                    // 0.5 * dx * (...) associates left (:170)
  int k = 0;
  for (int64_t p = 1; p < a.nPoints; ++p) {
    t += a.dx;                                                      // :166 / :174
    RHS::eval(t, zero, dyNext, P);
#pragma unroll
    for (int c = 0; c < D; ++c) next[c] = integral[c] + hdx * (dyPrev[c] + dyNext[c]);
    quad_emit<D>(a, i, k, p, integral, dyPrev, next, dyNext);
#pragma unroll
    for (int c = 0; c < D; ++c) { integral[c] = next[c]; dyPrev[c] = dyNext[c]; }
  }
  quad_last_rows<D>(a, i, integral);  // only non-empty when the march ran to the reference's last grid point
}

// cumsimpson(f, X, ctx, dx): grid t = linspace(min X, max X, N), dy = f(t), ys = cumsimpson(dy, t) — composite Simpson on
// interval pairs, resampled at every grid point by hermiteInterpolate — and result = hermiteInterpolate(X, t, ys, dy).
template <class RHS>
__global__ __launch_bounds__(kBlock) void cumsimpson_fn_kernel(const QuadArgs a) {
  constexpr int D = RHS::dim;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.N) return;
  const Params P = params_of(a, i);
  auto gridPoint = [&](int64_t p) { return p == a.nGrid - 1 ? a.xLast : a.x0 + a.dx * (double)p; };  // utils.nim:503-507
  double zero[D], f0[D], f1[D], f2[D], integral[D], next[D];
  double ysPrev[D], dyPrevPt[D], ys[D];  // the previous grid point of the outer resampling: (ys, dy) at index `have - 1`
#pragma unroll
  for (int c = 0; c < D; ++c) zero[c] = 0.0;
  RHS::eval(a.x0, zero, f0, P);
#pragma unroll
  for (int c = 0; c < D; ++c) { integral[c] = f0[c] - f0[c]; ysPrev[c] = integral[c]; dyPrevPt[c] = f0[c]; }
  int k = 0;
  int64_t have = 0;  // grid points pushed to the outer resampling so far
  // push grid point `have` with value v and slope s: rows inside interval have-1 can now be written
  auto push = [&](const double (&v)[D], const double (&s)[D]) {
    if (have > 0) quad_emit<D>(a, i, k, have, ysPrev, dyPrevPt, v, s);
#pragma unroll
    for (int c = 0; c < D; ++c) { ysPrev[c] = v[c]; dyPrevPt[c] = s[c]; }
    ++have;
  };
  auto herm = [&](const SimpsonPoint& w, const double (&y1)[D], const double (&y2)[D], const double (&d1)[D], const double (&d2)[D]) {
#pragma unroll
    for (int c = 0; c < D; ++c) ys[c] = w.w[0] * y1[c] + w.w[1] * d1[c] + w.w[2] * y2[c] + w.w[3] * d2[c];
  };
  bool stopped = false;
  for (int64_t j = 0; j < a.nPairs; ++j) {
    RHS::eval(gridPoint(2 * j + 1), zero, f1, P);
    RHS::eval(gridPoint(2 * j + 2), zero, f2, P);
    const SimpsonPair w = a.pairs[j];
#pragma unroll
    for (int c = 0; c < D; ++c) next[c] = integral[c] + (w.alpha * f2[c] + w.beta * f1[c] + w.eta * f0[c]);  // :359
    herm(a.pts[2 * j], integral, next, f0, f2);      // ys[2j]   = hermiteSpline(t[2j],   xs[j], xs[j+1], ...)
    push(ys, f0);
    herm(a.pts[2 * j + 1], integral, next, f0, f2);  // ys[2j+1]
    push(ys, f1);
#pragma unroll
    for (int c = 0; c < D; ++c) { integral[c] = next[c]; f0[c] = f2[c]; }
    if (have >= a.nPoints) { stopped = true; break; }
  }
  if (!stopped) {
    if (a.evenN) {  // odd number of intervals: the last one is closed with the three-point rule (:363-373); f1 = Y[last-2]
      const int64_t l = a.nGrid - 1;
      RHS::eval(gridPoint(l), zero, f2, P);
      const SimpsonPair w = a.pairs[a.nPairs];
#pragma unroll
      for (int c = 0; c < D; ++c) next[c] = integral[c] + (w.eta * f1[c] + w.beta * f0[c] + w.alpha * f2[c]);
      herm(a.pts[l - 1], integral, next, f0, f2);
      push(ys, f0);
#pragma unroll
      for (int c = 0; c < D; ++c) { integral[c] = next[c]; f0[c] = f2[c]; }
    }
    push(integral, f0);                 // the last grid point takes y[y.high] itself (utils.nim:299-300)
    quad_last_rows<D>(a, i, integral);  // and so does a requested x equal to it
  }
}

}  // namespace NNHIP_NS

// ode_capi_quad.hip — C-ABI entries for the function-argument forms of cumtrapz / cumsimpson (integrate.nim:138-175, 377-400):
// host replay of everything that depends on X and dx only, table upload, dispatch over the registered right-hand sides.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "ode_rtc.hpp"
#include "quad_kernels.hpp"

namespace nnhip {

int fail_msg(int code, const char* fmt, ...);  // ode_capi.hip: sets nnhip_last_error()

namespace {

constexpr int64_t kMaxGrid = (int64_t)1 << 31;        // cumtrapz: marched on the fly, nothing stored per grid point
constexpr int64_t kMaxSimpsonGrid = (int64_t)1 << 24; // cumsimpson: 11 doubles of X-only tables per interval pair live in HBM

template <class RHS>
hipError_t launch_quad(int rule, const QuadArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.N + kBlock - 1) / kBlock)), block(kBlock);
  return rule == 0 ? launch_kernel(cumtrapz_fn_kernel<RHS>, grid, block, s, a) : launch_kernel(cumsimpson_fn_kernel<RHS>, grid, block, s, a);
}

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

struct DeviceTables {  // freed on scope exit (after the stream has been drained by the caller)
  std::vector<void*> ptrs;
  ~DeviceTables() { for (void* p : ptrs) (void)hipFree(p); }
  template <class T>
  const T* upload(const std::vector<T>& v, hipStream_t s, int& rc) {
    if (v.empty() || rc != NNHIP_OK) return nullptr;
    void* d = nullptr;
    const hipError_t me = hipMalloc(&d, v.size() * sizeof(T));
    if (me != hipSuccess) {  // no device / no driver is a HIP failure, not an allocation failure
      rc = fail_msg(me == hipErrorOutOfMemory ? NNHIP_ENOMEM : NNHIP_EHIP, "hipMalloc of a %zu-byte quadrature table failed: %s", v.size() * sizeof(T), hipGetErrorString(me));
      return nullptr;
    }
    ptrs.push_back(d);
    if (hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) { rc = fail_msg(NNHIP_EHIP, "hipMemcpyAsync of a quadrature table failed"); return nullptr; }
    return (const T*)d;
  }
};

int cumquad_fn(int rule, int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params, int n_per_item, int64_t N,
               int dim, int layout, const double* X, int n_x, double dx, double* out, int* n_rows_out, void* stream) {
  const char* what = rule == 0 ? "cumtrapz" : "cumsimpson";
  if (N < 0 || dim < 1 || n_params < 0 || n_params > kMaxParams || n_per_item < 0 || n_per_item > kMaxParams || n_x < 1 || !X ||
      (n_params > 0 && !rhs_params) || (n_per_item > 0 && !per_item_params) || (layout != NNHIP_LAYOUT_SOA && layout != NNHIP_LAYOUT_AOS))
    return fail_msg(NNHIP_EVALUE, "%s(f, X): bad argument", what);
  for (int j = 0; j < n_x; ++j)
    if (!std::isfinite(X[j])) return fail_msg(NNHIP_EVALUE, "%s(f, X): X[%d] is not finite", what, j);
  if (dim > 16) return fail_msg(NNHIP_EUNSUPPORTED, "%s(f, X): integrands of more than 16 components are not supported (thread-per-item kernels)", what);
  if (!(dx > 0.0) || !std::isfinite(dx)) return fail_msg(NNHIP_EVALUE, "%s(f, X): dx must be a positive finite number (the reference's march would not end)", what);
  const double lo = *std::min_element(X, X + n_x), hi = *std::max_element(X, X + n_x);
  hipStream_t s = (hipStream_t)stream;
  RowPlan plan;
  QuadArgs a;
  std::memset(&a, 0, sizeof(a));
  std::vector<SimpsonPair> pairs;
  std::vector<SimpsonPoint> pts;
  if (rule == 0) {
    // replay `t = min(X); t += dx; while t <= max(X) + 1.0` (integrate.nim:160-174) to learn the grid size
    const double tEnd = hi + 1.0;
    if (!(lo + dx > lo) || !(tEnd + dx > tEnd)) return fail_msg(NNHIP_EVALUE, "cumtrapz(f, X): dx = %g does not advance t near %g (the reference would loop forever)", dx, tEnd);
    if ((tEnd - lo) / dx > (double)kMaxGrid) return fail_msg(NNHIP_EUNSUPPORTED, "cumtrapz(f, X): more than 2^31 grid points");
    int64_t nGrid = 1;
    for (double t = lo + dx; t <= tEnd; t += dx) ++nGrid;
    double tCur = lo;
    int64_t iCur = 0;
    auto gridAt = [&](int64_t i) {  // non-decreasing i
      while (iCur < i) { tCur += dx; ++iCur; }
      return tCur;
    };
    if (!plan_rows(X, n_x, nGrid, gridAt, plan)) return fail_msg(NNHIP_EVALUE, "cumtrapz(f, X): a value of X lies outside the integration grid (ValueError, utils.nim:311)");
    a.x0 = lo; a.dx = dx; a.xLast = 0.0; a.nGrid = nGrid;
  } else {
    // t = linspace(min(X), max(X), ((max(X) - min(X)) / dx).toInt + 2)   (integrate.nim:395; toInt rounds half away from zero)
    const double cnt = std::round((hi - lo) / dx);
    if (cnt + 2.0 > (double)kMaxSimpsonGrid) return fail_msg(NNHIP_EUNSUPPORTED, "cumsimpson(f, X): more than 2^24 grid points");
    const int64_t nGrid = (int64_t)cnt + 2;
    if (nGrid < 3) return fail_msg(NNHIP_EVALUE, "X and Y must have at least 3 elements to perform Simpson, use cumtrapz instead");  // :345-346
    const double step = (hi - lo) / (double)(nGrid - 1);
    std::vector<double> t((size_t)nGrid);
    t[0] = lo;
    for (int64_t i = 1; i <= nGrid - 2; ++i) t[(size_t)i] = lo + step * (double)i;
    t[(size_t)nGrid - 1] = hi;
    for (int64_t i = 1; i < nGrid; ++i)
      if (!(t[(size_t)i - 1] < t[(size_t)i]))  // cumsimpson(dy, t) would sort / trim the grid (sortAndTrimDataset): not a linspace any more
        return fail_msg(NNHIP_EUNSUPPORTED, "cumsimpson(f, X): the sampling grid is not strictly increasing (max(X) == min(X) or dx below the spacing of doubles)");
    if (!plan_rows(X, n_x, nGrid, [&](int64_t i) { return t[(size_t)i]; }, plan))
      return fail_msg(NNHIP_EVALUE, "cumsimpson(f, X): a value of X lies outside the integration grid (ValueError, utils.nim:311)");
    bool evenN = false;
    simpson_tables(t.data(), nGrid, pairs, pts, a.nPairs, evenN);
    a.evenN = evenN ? 1 : 0;
    a.x0 = lo; a.dx = step; a.xLast = hi; a.nGrid = nGrid;
  }
  if (n_rows_out) *n_rows_out = plan.nRows;
  if (N == 0 || plan.nRows == 0) return NNHIP_OK;
  if (!out) return fail_msg(NNHIP_EVALUE, "%s(f, X): out is null", what);
  // the march stops after the last grid point any row needs
  a.nPoints = plan.lastRows.empty() ? (plan.emits.empty() ? 1 : (int64_t)plan.emits.back().interval + 2) : a.nGrid;
  a.out = out;
  a.N = N;
  a.ivpStride = layout == NNHIP_LAYOUT_SOA ? 1 : dim;
  a.compStride = layout == NNHIP_LAYOUT_SOA ? N : 1;
  a.rowStride = N * dim;
  for (int k = 0; k < kMaxParams; ++k) a.P.p[k] = k < n_params ? rhs_params[k] : 0.0;
  if (rtc_ctx_fill(rhs_kind, N, a.P, nullptr) < 0) return fail_msg(NNHIP_EVALUE, "%s(f, X): rhs_kind %d: %s", what, rhs_kind, rtc_last_error());
  a.perIvpParams = n_per_item > 0 ? per_item_params : nullptr;
  a.nPerIvp = n_per_item;
  a.perIvpStride = N;
  int rc = NNHIP_OK;
  DeviceTables tabs;
  a.emits = tabs.upload(plan.emits, s, rc);
  a.nEmit = (int)plan.emits.size();
  a.lastRows = tabs.upload(plan.lastRows, s, rc);
  a.nLast = (int)plan.lastRows.size();
  if (rule == 1) {
    // only the pairs the march reaches are needed on the device
    const size_t needPairs = std::min((size_t)a.nPairs + 1, (size_t)(a.nPoints / 2 + 2));
    const size_t needPts = std::min(pts.size(), (size_t)a.nPoints + 3);
    if (a.nPoints < a.nGrid) { pairs.resize(std::max(needPairs, (size_t)1)); pts.resize(needPts); }
    a.pairs = tabs.upload(pairs, s, rc);
    a.pts = tabs.upload(pts, s, rc);
  }
  if (rc != NNHIP_OK) { (void)hipStreamSynchronize(s); return rc; }
  hipError_t e = hipErrorInvalidValue;
  bool found = false;
  if (rhs_kind >= NNHIP_RHS_USER_BASE) {
    int d = 0;
    if (!rtc_info(rhs_kind, &d, nullptr) || d != dim) { (void)hipStreamSynchronize(s); return fail_msg(NNHIP_EVALUE, "%s(f, X): unknown user rhs_kind %d or dim mismatch", what, rhs_kind); }
    found = true;
    e = rtc_launch_quad(rhs_kind, rule, a, s);
    if (e != hipSuccess) { (void)hipStreamSynchronize(s); return fail_msg(NNHIP_EHIP, "%s(f, X): user integrand: %s", what, rtc_last_error()); }
  }
#define X(kind, d, T)                                  \
  if (!found && rhs_kind == kind && dim == d) {        \
    found = true;                                      \
    e = launch_quad<T>(rule, a, s);                    \
  }
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
  (void)hipStreamSynchronize(s);  // the tables are freed when `tabs` goes out of scope
  if (!found) return fail_msg(NNHIP_EUNSUPPORTED, "%s(f, X): no thread-per-item kernel for rhs_kind %d with dim %d", what, rhs_kind, dim);
  if (e != hipSuccess) return fail_msg(NNHIP_EHIP, "%s(f, X): kernel launch failed: %s", what, hipGetErrorString(e));
  return NNHIP_OK;
}

}  // namespace
}  // namespace nnhip

extern "C" {

int nnhip_cumtrapz_fn_batch_f64_dev(int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params, int n_per_item,
                                    int64_t N, int dim, int layout, const double* X, int n_x, double dx, double* out, int* n_rows_out,
                                    void* stream) {
  return nnhip::cumquad_fn(0, rhs_kind, rhs_params, n_params, per_item_params, n_per_item, N, dim, layout, X, n_x, dx, out, n_rows_out, stream);
}

int nnhip_cumsimpson_fn_batch_f64_dev(int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params, int n_per_item,
                                      int64_t N, int dim, int layout, const double* X, int n_x, double dx, double* out, int* n_rows_out,
                                      void* stream) {
  return nnhip::cumquad_fn(1, rhs_kind, rhs_params, n_params, per_item_params, n_per_item, N, dim, layout, X, n_x, dx, out, n_rows_out, stream);
}

}  // extern "C"

// ---- host-pointer forms of the consumers (what a host language without device-memory management calls): stage, run the
// device-pointer entry on a private stream, copy back.  Same results bit for bit. ------------------------------------------------
namespace {

struct HostStage {
  std::vector<void*> bufs;
  hipStream_t s = nullptr;
  ~HostStage() {
    if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (void* b : bufs) (void)hipFree(b);
  }
  int begin(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return nnhip::fail_msg(NNHIP_EHIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return nnhip::fail_msg(NNHIP_EVALUE, "device %d out of range [0,%d)", device, ndev);
    if (hipSetDevice(device) != hipSuccess) return nnhip::fail_msg(NNHIP_EHIP, "hipSetDevice(%d) failed", device);
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nnhip::fail_msg(NNHIP_EHIP, "hipStreamCreate failed");
    return NNHIP_OK;
  }
  int upload(const void* host, size_t bytes, void** dev) {  // host may be null: allocate only
    *dev = nullptr;
    if (bytes == 0) return NNHIP_OK;
    const hipError_t e = hipMalloc(dev, bytes);
    if (e != hipSuccess) return nnhip::fail_msg(e == hipErrorOutOfMemory ? NNHIP_ENOMEM : NNHIP_EHIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    bufs.push_back(*dev);
    if (host && hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, s) != hipSuccess) return nnhip::fail_msg(NNHIP_EHIP, "host-to-device copy failed");
    return NNHIP_OK;
  }
  int download(void* host, const void* dev, size_t bytes) {
    if (bytes == 0) return NNHIP_OK;
    if (hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return nnhip::fail_msg(NNHIP_EHIP, "device-to-host copy failed");
    return NNHIP_OK;
  }
};

int cumquad_fn_host(int rule, int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params, int n_per_item, int64_t N, int dim,
                    int layout, const double* X, int n_x, double dx, double* out, int* n_rows_out, int device) {
  if (N < 0 || dim < 1 || n_x < 1 || n_per_item < 0) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes");
  HostStage st;
  int rc = st.begin(device);
  if (rc) return rc;
  void *dPer = nullptr, *dOut = nullptr;
  const size_t outBytes = (size_t)n_x * (size_t)dim * (size_t)N * sizeof(double);
  if ((rc = st.upload(n_per_item > 0 ? per_item_params : nullptr, (size_t)n_per_item * (size_t)N * sizeof(double), &dPer))) return rc;
  if ((rc = st.upload(nullptr, outBytes, &dOut))) return rc;
  int rows = 0;
  rc = nnhip::cumquad_fn(rule, rhs_kind, rhs_params, n_params, (const double*)dPer, n_per_item, N, dim, layout, X, n_x, dx, (double*)dOut, &rows, st.s);
  if (n_rows_out) *n_rows_out = rows;
  if (rc) return rc;
  if (N > 0 && rows > 0 && !out) return nnhip::fail_msg(NNHIP_EVALUE, "out is NULL");
  return st.download(out, dOut, (size_t)rows * (size_t)dim * (size_t)N * sizeof(double));
}

}  // namespace

extern "C" {

int nnhip_cumtrapz_fn_batch_f64(int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params, int n_per_item, int64_t N,
                                int dim, int layout, const double* X, int n_x, double dx, double* out, int* n_rows_out, int device) {
  return cumquad_fn_host(0, rhs_kind, rhs_params, n_params, per_item_params, n_per_item, N, dim, layout, X, n_x, dx, out, n_rows_out, device);
}

int nnhip_cumsimpson_fn_batch_f64(int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params, int n_per_item, int64_t N,
                                  int dim, int layout, const double* X, int n_x, double dx, double* out, int* n_rows_out, int device) {
  return cumquad_fn_host(1, rhs_kind, rhs_params, n_params, per_item_params, n_per_item, N, dim, layout, X, n_x, dx, out, n_rows_out, device);
}

int nnhip_cumtrapz_batch_f64(const double* X, int n, const double* Y, int64_t M, double* out, int device) {
  if (n < 1 || M < 0) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes");
  HostStage st;
  int rc = st.begin(device);
  if (rc) return rc;
  void *dY = nullptr, *dOut = nullptr;
  const size_t bytes = (size_t)n * (size_t)M * sizeof(double);
  if ((rc = st.upload(Y, bytes, &dY)) || (rc = st.upload(nullptr, bytes, &dOut))) return rc;
  rc = nnhip_cumtrapz_batch_f64_dev(X, n, (const double*)dY, M, (double*)dOut, st.s);
  if (rc) return nnhip::fail_msg(rc, "cumtrapz(Y, X): bad arguments (X must be strictly ascending) or launch failure");
  return st.download(out, dOut, bytes);
}

int nnhip_cumsimpson_batch_f64(const double* X, int n, const double* Y, int64_t M, double* out, int device) {
  if (n < 1 || M < 0) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes");
  HostStage st;
  int rc = st.begin(device);
  if (rc) return rc;
  void *dY = nullptr, *dOut = nullptr;
  const size_t bytes = (size_t)n * (size_t)M * sizeof(double);
  if ((rc = st.upload(Y, bytes, &dY)) || (rc = st.upload(nullptr, bytes, &dOut))) return rc;
  rc = nnhip_cumsimpson_batch_f64_dev(X, n, (const double*)dY, M, (double*)dOut, st.s);
  if (rc) return nnhip::fail_msg(rc, "cumsimpson(Y, X): needs >= 3 strictly ascending points (ValueError, integrate.nim:345-346) or launch failure");
  return st.download(out, dOut, bytes);
}

int nnhip_hermite_spline_eval_batch_f64(const double* X, int n_knots, const double* Y, const double* dY, int64_t M, const double* xq, int n_q,
                                        int deriv, int extrap, double extrap_value, double* out, int device) {
  if (n_knots < 2 || M < 0 || n_q < 0) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes");
  HostStage st;
  int rc = st.begin(device);
  if (rc) return rc;
  void *dYd = nullptr, *ddY = nullptr, *dOut = nullptr;
  const size_t bytes = (size_t)n_knots * (size_t)M * sizeof(double);
  if ((rc = st.upload(Y, bytes, &dYd)) || (rc = st.upload(dY, bytes, &ddY)) || (rc = st.upload(nullptr, (size_t)n_q * (size_t)M * sizeof(double), &dOut))) return rc;
  if (!dY && M > 0) {  // newHermiteSpline(X, Y): estimate the slopes (interpolate.nim:241-253)
    rc = nnhip_hermite_spline_slopes_f64_dev(X, n_knots, (const double*)dYd, M, (double*)ddY, st.s);
    if (rc) return nnhip::fail_msg(rc, "newHermiteSpline(X, Y): X must be strictly ascending");
  }
  rc = nnhip_hermite_spline_eval_batch_f64_dev(X, n_knots, (const double*)dYd, (const double*)ddY, M, xq, n_q, deriv, extrap, extrap_value, (double*)dOut, st.s);
  if (rc) return nnhip::fail_msg(rc, "HermiteSpline eval: bad arguments (X strictly ascending, extrap in 0..4; Error extrapolation raises outside the knots)");
  return st.download(out, dOut, (size_t)n_q * (size_t)M * sizeof(double));
}

}  // extern "C"

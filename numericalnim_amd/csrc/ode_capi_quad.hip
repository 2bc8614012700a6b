// ode_capi_quad.hip — C-ABI entries for the function-argument forms of cumtrapz / cumsimpson (integrate.nim:138-175, 377-400):
// host replay of everything that depends on X and dx only, table upload, dispatch over the registered right-hand sides.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "ode_rtc.hpp"
#include "quad_kernels.hpp"
#include "quad_plan.hpp"

namespace nnhip {

int fail_msg(int code, const char* fmt, ...);  // ode_capi.hip: sets nnhip_last_error()

namespace {

template <class RHS>
hipError_t launch_quad(int rule, const QuadArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)((a.N + kBlock - 1) / kBlock)), block(kBlock);
  return rule == 0 ? launch_kernel(cumtrapz_fn_kernel<RHS>, grid, block, s, a) : launch_kernel(cumsimpson_fn_kernel<RHS>, grid, block, s, a);
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
  hipStream_t s = (hipStream_t)stream;
  CumquadPlan pl;  // everything X and dx decide (quad_plan.hpp)
  std::string why;
  const int prc = plan_cumquad(rule, X, n_x, dx, pl, why);
  if (prc != NNHIP_OK) return fail_msg(prc, "%s", why.c_str());
  RowPlan& plan = pl.rows;
  QuadArgs& a = pl.a;
  std::vector<SimpsonPair>& pairs = pl.pairs;
  std::vector<SimpsonPoint>& pts = pl.pts;
  if (n_rows_out) *n_rows_out = plan.nRows;
  if (N == 0 || plan.nRows == 0) return NNHIP_OK;
  if (!out) return fail_msg(NNHIP_EVALUE, "%s(f, X): out is null", what);
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

// N items x `per` doubles each beyond 2^44 doubles is no batch: refused before the byte counts below can overflow
inline bool sizes_addressable(int64_t N, int64_t per) { return N >= 0 && per >= 0 && (per == 0 || N <= ((int64_t)1 << 44) / per); }

int cumquad_fn_host(int rule, int rhs_kind, const double* rhs_params, int n_params, const double* per_item_params, int n_per_item, int64_t N, int dim,
                    int layout, const double* X, int n_x, double dx, double* out, int* n_rows_out, int device) {
  if (N < 0 || dim < 1 || n_x < 1 || n_per_item < 0) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes");
  if (!sizes_addressable(N, (int64_t)dim * n_x + n_per_item)) return nnhip::fail_msg(NNHIP_EVALUE, "N x dim x n_x is beyond any device");
  if (!X || (N > 0 && n_per_item > 0 && !per_item_params) || (n_params > 0 && !rhs_params)) return nnhip::fail_msg(NNHIP_EVALUE, "X / per_item_params / rhs_params is NULL");  // before anything is allocated
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
  if (n < 1 || M < 0 || !sizes_addressable(M, n)) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes");
  if (!X || (M > 0 && (!Y || !out))) return nnhip::fail_msg(NNHIP_EVALUE, "cumtrapz: X / Y / out is NULL");  // before anything is allocated for them
  HostStage st;
  int rc = st.begin(device);
  if (rc) return rc;
  void *dY = nullptr, *dOut = nullptr;
  const size_t bytes = (size_t)n * (size_t)M * sizeof(double);
  if ((rc = st.upload(Y, bytes, &dY)) || (rc = st.upload(nullptr, bytes, &dOut))) return rc;
  rc = nnhip_cumtrapz_batch_f64_dev(X, n, (const double*)dY, M, (double*)dOut, st.s);
  if (rc) return rc;  // (the device entry's message stands: NaN in X, impure duplicates, ...)
  return st.download(out, dOut, bytes);
}

int nnhip_cumsimpson_batch_f64(const double* X, int n, const double* Y, int64_t M, double* out, int device) {
  if (n < 1 || M < 0 || !sizes_addressable(M, n)) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes");
  if (!X || (M > 0 && (!Y || !out))) return nnhip::fail_msg(NNHIP_EVALUE, "cumsimpson: X / Y / out is NULL");  // before anything is allocated for them
  HostStage st;
  int rc = st.begin(device);
  if (rc) return rc;
  void *dY = nullptr, *dOut = nullptr;
  const size_t bytes = (size_t)n * (size_t)M * sizeof(double);
  if ((rc = st.upload(Y, bytes, &dY)) || (rc = st.upload(nullptr, bytes, &dOut))) return rc;
  rc = nnhip_cumsimpson_batch_f64_dev(X, n, (const double*)dY, M, (double*)dOut, st.s);
  if (rc) return rc;  // (the device entry's message stands: fewer than 3 distinct abscissae, NaN in X, impure duplicates, ...)
  return st.download(out, dOut, bytes);
}

int nnhip_hermite_spline_eval_batch_f64(const double* X, int n_knots, const double* Y, const double* dY, int64_t M, const double* xq, int n_q,
                                        int deriv, int extrap, double extrap_value, double* out, int device) {
  if (n_knots < 2 || M < 0 || n_q < 0 || !sizes_addressable(M, (int64_t)n_knots + n_q)) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes");
  if (!X || (n_q > 0 && !xq) || (M > 0 && (!Y || (n_q > 0 && !out)))) return nnhip::fail_msg(NNHIP_EVALUE, "hermite spline eval: X / xq / Y / out is NULL");  // before anything is allocated for them
  HostStage st;
  int rc = st.begin(device);
  if (rc) return rc;
  void *dYd = nullptr, *ddY = nullptr, *dOut = nullptr;
  const size_t bytes = (size_t)n_knots * (size_t)M * sizeof(double);
  if ((rc = st.upload(Y, bytes, &dYd)) || (rc = st.upload(dY, bytes, &ddY)) || (rc = st.upload(nullptr, (size_t)n_q * (size_t)M * sizeof(double), &dOut))) return rc;
  std::vector<double> xs;
  if (!dY && M > 0) {  // newHermiteSpline(X, Y): sort and trim (interpolate.nim:244), then estimate the slopes on the sorted data (:245-251)
    void* dYs = nullptr;
    if ((rc = st.upload(nullptr, bytes, &dYs))) return rc;
    xs.resize((size_t)n_knots);
    const double* in1[1] = {(const double*)dYd};
    double* out1[1] = {(double*)dYs};
    int ns = 0;
    if ((rc = nnhip_sort_and_trim_dataset_f64_dev(X, n_knots, in1, 1, M, xs.data(), out1, &ns, st.s))) return rc;
    if (ns < 2) return nnhip::fail_msg(NNHIP_EVALUE, "newHermiteSpline(X, Y): fewer than 2 distinct knots");
    X = xs.data(); n_knots = ns; dYd = dYs;
    if ((rc = nnhip_hermite_spline_slopes_f64_dev(X, n_knots, (const double*)dYd, M, (double*)ddY, st.s))) return rc;
  }
  rc = nnhip_hermite_spline_eval_batch_f64_dev(X, n_knots, (const double*)dYd, (const double*)ddY, M, xq, n_q, deriv, extrap, extrap_value, (double*)dOut, st.s);
  if (rc) return rc;  // (the device entry's message stands: extrap outside 0..4, Error extrapolation outside the knots, impure duplicates, ...)
  return st.download(out, dOut, (size_t)n_q * (size_t)M * sizeof(double));
}

}  // extern "C"

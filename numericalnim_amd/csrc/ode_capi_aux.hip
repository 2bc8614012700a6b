// ode_capi_aux.hip — the smaller C-ABI entries: batched RHS evaluation, batched hermiteSpline,
// and the one-process multi-GPU solve (contiguous shards of the IVP index range, no exchange).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "ode_kernels.hpp"
#include "ode_rtc.hpp"
#include "quad_kernels.hpp"
#include "consumer_kernels.hpp"  // (this translation unit only: the kernels in it are ordinary, non-inline __global__ functions)
#include "dataset_plan.hpp"

namespace nnhip {
// ode_capi.hip
int fail_msg(int code, const char* fmt, ...);
const char* thread_error();
bool multi_gpu_oversubscribe();
void release_thread_staging();
int solve_host_range(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                     const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t NFull, int64_t lo0, int64_t N, int dim,
                     int layout, const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                     int64_t* rejected_out, int64_t max_steps, nnhip_ode_stats* stats, int device);

template <class RHS>
hipError_t launch_rhs_batch(int64_t N, int64_t is, int64_t cs, double t, const double* y, double* dy, const Params& P, hipStream_t s) {
  const int64_t grid = (N + kBlock - 1) / kBlock;
  if (grid <= 0) return hipSuccess;
  return launch_kernel(rhs_batch_kernel<RHS>, dim3((unsigned)grid), dim3(kBlock), s, N, is, cs, t, y, dy, P);
}

// dense_rows_kernel of a compiled-in right-hand side; returns false when (rhs_kind, dim) has no ahead-of-time instantiation (the caller then
// evaluates f and hermiteSpline in separate launches, as for run-time compiled kinds)
bool launch_dense_rows_kind(int rhs_kind, int dim, int64_t N, int64_t is, int64_t cs, double tA, double tB, int neg, const double* yA, const double* yB,
                            const DenseRows& r, const Params& P, hipStream_t s, hipError_t* err) {
#define X(kind, d, T) \
  if (rhs_kind == kind && dim == d) { *err = launch_dense_rows<T>(N, is, cs, tA, tB, neg, yA, yB, r, P, s); return true; }
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
#define X(kind, d, T, CA, CF) \
  if (rhs_kind == kind && dim == d) { *err = launch_dense_rows<T>(N, is, cs, tA, tB, neg, yA, yB, r, P, s); return true; }
  NNHIP_FOR_EACH_LPS_RHS(X)
#undef X
  return false;
}

__global__ __launch_bounds__(kBlock) void fill_f64_kernel(double* __restrict__ p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) p[i] = v;
}
hipError_t launch_hermite(double x, double x1, double x2, const double* y1, const double* y2, const double* dy1, const double* dy2, double* out, int64_t n,
                          int negate_dy, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  return launch_kernel(hermite_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), s, x, x1, x2, y1, y2, dy1, dy2, out, n, negate_dy);
}
hipError_t launch_fill_f64(double* p, int64_t n, double v, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  return launch_kernel(fill_f64_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), s, p, n, v);
}

// ---- the tableaux the kernels are compiled with, for verification against the reference's text ----------------------------------
// [S, NB, c_1..c_S, a_21, a_31, a_32, ..., a_S,S-1, b_1..b_NB, bHat_1..bHat_S] through Tableau<M>'s own accessors, on either side.
template <int METHOD>
__host__ __device__ inline int tableau_flatten(double* out) {
  using T = Tableau<METHOD>;
  int k = 0;
  out[k++] = (double)T::S;
  out[k++] = (double)T::NB;
  for (int s = 0; s < T::S; ++s) out[k++] = T::c(s);
  for (int s = 1; s < T::S; ++s)
    for (int j = 0; j < s; ++j) out[k++] = T::a(s, j);
  for (int j = 0; j < T::NB; ++j) out[k++] = T::b(j);
  for (int j = 0; j < T::S; ++j) out[k++] = T::bhat(j);
  return k;
}
template <int METHOD>
constexpr int tableau_count() { return 2 + Tableau<METHOD>::S + Tableau<METHOD>::S * (Tableau<METHOD>::S - 1) / 2 + Tableau<METHOD>::NB + Tableau<METHOD>::S; }
template <int METHOD>
__global__ void tableau_dump_kernel(double* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) (void)tableau_flatten<METHOD>(out);
}
template <int METHOD>
int tableau_read(int device, double* out, int cap) {
  constexpr int n = tableau_count<METHOD>();
  if (cap < n) return fail_msg(NNHIP_EVALUE, "nnhip_ode_tableau_f64: out has room for %d values, %d needed", cap, n);
  if (device < 0) return tableau_flatten<METHOD>(out);
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return fail_msg(NNHIP_EHIP, "nnhip_ode_tableau_f64: no HIP device %d", device);
  double* d = nullptr;
  hipError_t e = hipMalloc((void**)&d, sizeof(double) * n);
  if (e == hipSuccess) e = hipMemset(d, 0xff, sizeof(double) * n);
  if (e == hipSuccess) e = launch_kernel(tableau_dump_kernel<METHOD>, dim3(1), dim3(64), (hipStream_t) nullptr, d);
  if (e == hipSuccess) e = hipMemcpy(out, d, sizeof(double) * n, hipMemcpyDeviceToHost);
  if (d) (void)hipFree(d);
  (void)hipSetDevice(prev);
  if (e != hipSuccess) return fail_msg(NNHIP_EHIP, "nnhip_ode_tableau_f64: %s", hipGetErrorString(e));
  return n;
}

// the controller's step-size factor (ode.nim:71,537) over an array of error norms
template <int ORDER>
__global__ __launch_bounds__(kBlock) void controller_factor_kernel(const double* __restrict__ error, double* __restrict__ out, int64_t n) {
  controller_prologue();
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) out[i] = shrink_factor<ORDER>(error[i]);
}

// sortAndTrimDataset (utils.nim:404-407) in front of a discrete consumer.  The permutation, the trimmed abscissae and the pairs of rows that must be pure
// duplicates come from the host (dataset_plan.hpp); the caller's series are gathered into sorted order on the device, after the purity check (an impure
// duplicate anywhere in the batch: NNHIP_EVALUE, the reference's ValueError :372).  Strictly ascending X — the solver's own output grid — moves nothing.
struct SortedDataset {
  DatasetPlan pl;
  std::vector<double*> owned;  // gathered series [n'][M] (device)
  int32_t* dIdx = nullptr;     // src | keep | drop
  unsigned int* dFlag = nullptr;
  bool touchedStream = false;
  ~SortedDataset() {
    for (double* p : owned) (void)hipFree(p);
    if (dIdx) (void)hipFree(dIdx);
    if (dFlag) (void)hipFree(dFlag);
  }
  int n() const { return (int)pl.x.size(); }
  const double* X() const { return pl.x.data(); }
};
int sorted_dataset(const char* who, const double* X, int n, const double* const* Y, int nY, int64_t M, hipStream_t s, SortedDataset& sd, const double** sorted) {
  std::string why;
  if (dataset_plan(X, n, sd.pl, why) != 0) return fail_msg(NNHIP_EVALUE, "%s: %s", who, why.c_str());
  for (int k = 0; k < nY; ++k) sorted[k] = Y[k];
  if (sd.pl.identity || M == 0) return NNHIP_OK;
  for (int k = 0; k < nY; ++k) if (!Y[k]) return fail_msg(NNHIP_EVALUE, "%s: a series pointer is NULL", who);
  const DatasetPlan& pl = sd.pl;
  const size_t nS = pl.src.size(), nD = pl.dupKeep.size();
  std::vector<int32_t> idx(pl.src);
  idx.insert(idx.end(), pl.dupKeep.begin(), pl.dupKeep.end());
  idx.insert(idx.end(), pl.dupDrop.begin(), pl.dupDrop.end());
  if (hipMalloc((void**)&sd.dIdx, idx.size() * sizeof(int32_t)) != hipSuccess) return fail_msg(NNHIP_ENOMEM, "%s: hipMalloc failed", who);
  sd.touchedStream = true;
  if (hipMemcpyAsync(sd.dIdx, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)  // (idx is a local)
    return fail_msg(NNHIP_EHIP, "%s: upload of the sort permutation failed", who);
  const unsigned gx = (unsigned)((M + kBlock - 1) / kBlock);
  if (nD > 0) {  // removeDuplicates :367-372
    unsigned int flag = 0;
    if (hipMalloc((void**)&sd.dFlag, sizeof(unsigned int)) != hipSuccess) return fail_msg(NNHIP_ENOMEM, "%s: hipMalloc failed", who);
    if (hipMemsetAsync(sd.dFlag, 0, sizeof(unsigned int), s) != hipSuccess) return fail_msg(NNHIP_EHIP, "%s: memset failed", who);
    for (int k = 0; k < nY; ++k)
      if (launch_kernel(dup_rows_differ_kernel, dim3(gx, (unsigned)std::min<size_t>(nD, 65535)), dim3(kBlock), s, (const int32_t*)(sd.dIdx + nS), (const int32_t*)(sd.dIdx + nS + nD), (int)nD, Y[k], M,
                        sd.dFlag) != hipSuccess)
        return fail_msg(NNHIP_EHIP, "%s: kernel launch failed", who);
    if (hipMemcpyAsync(&flag, sd.dFlag, sizeof(flag), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return fail_msg(NNHIP_EHIP, "%s: reading the duplicate check back failed", who);
    if (flag) return fail_msg(NNHIP_EVALUE, "%s: impure y-duplicates were found (the same x with different y; ValueError, utils.nim:372)", who);
  }
  for (int k = 0; k < nY; ++k) {
    double* g = nullptr;
    if (hipMalloc((void**)&g, nS * (size_t)M * sizeof(double)) != hipSuccess) return fail_msg(NNHIP_ENOMEM, "%s: hipMalloc failed", who);
    sd.owned.push_back(g);
    if (launch_kernel(gather_rows_kernel, dim3(gx, (unsigned)std::min<size_t>(nS, 65535)), dim3(kBlock), s, (const int32_t*)sd.dIdx, (int)nS, Y[k], g, M) != hipSuccess)
      return fail_msg(NNHIP_EHIP, "%s: kernel launch failed", who);
    sorted[k] = g;
  }
  return NNHIP_OK;
}
// rows [first, n) of out [n][M] := NaN — rows the reference's result does not have
int nan_rows(double* out, int first, int n, int64_t M, hipStream_t s) {
  if (first >= n || M <= 0) return NNHIP_OK;
  return launch_fill_f64(out + (int64_t)first * M, (int64_t)(n - first) * M, __builtin_nan(""), s) == hipSuccess ? NNHIP_OK : fail_msg(NNHIP_EHIP, "fill failed");
}

}  // namespace nnhip

extern "C" {

int nnhip_hermite_spline_f64_dev(double x, double x1, double x2, const double* y1, const double* y2, const double* dy1,
                                 const double* dy2, double* out, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!y1 || !y2 || !dy1 || !dy2 || !out))) return nnhip::fail_msg(NNHIP_EVALUE, "hermite_spline: n < 0 or a NULL array");
  if (n == 0) return NNHIP_OK;
  return nnhip::launch_hermite(x, x1, x2, y1, y2, dy1, dy2, out, n, 0, (hipStream_t)stream) == hipSuccess
             ? NNHIP_OK : nnhip::fail_msg(NNHIP_EHIP, "hermite_spline: kernel launch failed");
}

int nnhip_hermite_spline_eval_batch_f64_dev(const double* X, int n_knots, const double* Y, const double* dY, int64_t M,
                                            const double* xq, int n_q, int deriv, int extrap, double extrap_value, double* out,
                                            void* stream) {
  if (n_knots < 2 || M < 0 || n_q < 0 || !X || (n_q > 0 && !xq) || extrap < 0 || extrap > 4) return nnhip::fail_msg(NNHIP_EVALUE, "hermite spline eval: need >= 2 knots, M >= 0, n_q >= 0, non-NULL X / xq and an ExtrapolateKind in 0..4");
  if (M > 0 && n_q > 0 && (!Y || !dY || !out)) return nnhip::fail_msg(NNHIP_EVALUE, "hermite spline eval: Y / dY / out is NULL");
  // newHermiteSpline(X, Y, dY) sorts and trims its data (interpolate.nim:231): knots in any order are the constructor's business, done here per call
  // (a caller that evaluates one spline often sorts once: nnhip_sort_and_trim_dataset_f64_dev)
  nnhip::SortedDataset sd;
  const double* ser[2] = {Y, dY};
  const double* srt[2];
  if (int rcs = nnhip::sorted_dataset("newHermiteSpline(X, Y, dY)", X, n_knots, ser, 2, (n_q > 0 ? M : 0), (hipStream_t)stream, sd, srt)) return rcs;
  if (sd.n() < 2) return nnhip::fail_msg(NNHIP_EVALUE, "hermite spline eval: fewer than 2 distinct knots");
  struct SyncAtExit { hipStream_t s; bool on; ~SyncAtExit() { if (on) (void)hipStreamSynchronize(s); } } syncAtExit{(hipStream_t)stream, !sd.pl.identity};  // the gathered series are freed on return
  X = sd.X(); n_knots = sd.n(); Y = srt[0]; dY = srt[1];
  if (M == 0 || n_q == 0) return NNHIP_OK;
  if (extrap == 4) for (int q = 0; q < n_q; ++q) if (xq[q] < X[0] || xq[q] > X[n_knots - 1]) return nnhip::fail_msg(NNHIP_EVALUE, "x = %g is outside the interpolation range [%g, %g] (ExtrapolateKind.Error)", xq[q], X[0], X[n_knots - 1]);  // ValueError :340-341
  for (int q0 = 0; q0 < n_q; q0 += nnhip::kHermChunk) {
    nnhip::HermChunk c;
    const int nq = std::min(nnhip::kHermChunk, n_q - q0);
    nnhip::herm_chunk_fill(X, n_knots, xq + q0, nq, deriv != 0, extrap, extrap_value, c);
    const dim3 grid((unsigned)((M + nnhip::kBlock - 1) / nnhip::kBlock), (unsigned)nq), block(nnhip::kBlock);
    if (nnhip::launch_kernel(nnhip::hermite_interp_kernel, grid, block, (hipStream_t)stream, c, nq, Y, dY, M, out + (int64_t)q0 * M) != hipSuccess)
      return nnhip::fail_msg(NNHIP_EHIP, "hermite spline eval: kernel launch failed");
  }
  return NNHIP_OK;
}

int nnhip_hermite_spline_slopes_f64_dev(const double* X, int n_knots, const double* Y, int64_t M, double* dY, void* stream) {
  if (n_knots < 2 || M < 0 || !X) return nnhip::fail_msg(NNHIP_EVALUE, "hermite slopes: need >= 2 knots, M >= 0 and a non-NULL X");
  if (n_knots > 65535) return nnhip::fail_msg(NNHIP_EUNSUPPORTED, "hermite slopes: at most 65535 knots (got %d)", n_knots);
  if (M > 0 && (!Y || !dY)) return nnhip::fail_msg(NNHIP_EVALUE, "hermite slopes: Y / dY is NULL");
  hipStream_t s = (hipStream_t)stream;
  // newHermiteSpline(X, Y) estimates its slopes on the sorted, trimmed data (interpolate.nim:244-251): dY row k belongs to sortAndTrimDataset's row k
  nnhip::SortedDataset sd;
  const double* srt[1];
  if (int rcs = nnhip::sorted_dataset("newHermiteSpline(X, Y)", X, n_knots, &Y, 1, M, s, sd, srt)) return rcs;
  if (sd.n() < 2) return nnhip::fail_msg(NNHIP_EVALUE, "hermite slopes: fewer than 2 distinct knots");
  if (int rcn = nnhip::nan_rows(dY, sd.n(), n_knots, M, s)) return rcn;
  X = sd.X(); n_knots = sd.n(); Y = srt[0];
  if (M == 0) return NNHIP_OK;
  double* dX = nullptr;
  if (hipMalloc((void**)&dX, (size_t)n_knots * sizeof(double)) != hipSuccess) return nnhip::fail_msg(NNHIP_ENOMEM, "hermite slopes: hipMalloc failed");
  int rc = NNHIP_OK;
  if (hipMemcpyAsync(dX, X, (size_t)n_knots * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess) rc = NNHIP_EHIP;
  const dim3 grid((unsigned)((M + nnhip::kBlock - 1) / nnhip::kBlock), (unsigned)n_knots), block(nnhip::kBlock);
  if (!rc && nnhip::launch_kernel(nnhip::hermite_slopes_kernel, grid, block, s, (const double*)dX, n_knots, Y, M, dY) != hipSuccess) rc = NNHIP_EHIP;
  (void)hipStreamSynchronize(s);  // X was staged from pageable memory and is freed below
  (void)hipFree(dX);
  return rc;
}

int nnhip_cumtrapz_batch_f64_dev(const double* X, int n, const double* Y, int64_t M, double* out, void* stream) {
  if (n < 1 || M < 0 || !X) return nnhip::fail_msg(NNHIP_EVALUE, "cumtrapz: need n >= 1, M >= 0 and a non-NULL X");
  if (M > 0 && (!Y || !out)) return nnhip::fail_msg(NNHIP_EVALUE, "cumulative quadrature: Y / out is NULL");
  nnhip::SortedDataset sd;   // `let (xSorted, ySorted) = sortAndTrimDataset(@X, @Y)` (integrate.nim:131)
  const double* srt[1];
  if (int rcs = nnhip::sorted_dataset("cumtrapz(Y, X)", X, n, &Y, 1, M, (hipStream_t)stream, sd, srt)) return rcs;
  if (M == 0) return NNHIP_OK;
  if (int rcn = nnhip::nan_rows(out, sd.n(), n, M, (hipStream_t)stream)) return rcn;
  const int ns = sd.n();
  const dim3 grid((unsigned)((M + nnhip::kBlock - 1) / nnhip::kBlock)), block(nnhip::kBlock);
  int first = 0;
  do {
    nnhip::TrapzWeights W;
    const int nw = nnhip::trapz_weights_fill(sd.X(), ns, first, W);
    if (nnhip::launch_kernel(nnhip::cumtrapz_kernel, grid, block, (hipStream_t)stream, W, nw, first, srt[0], out, M) != hipSuccess) return nnhip::fail_msg(NNHIP_EHIP, "cumtrapz: kernel launch failed");
    first += nw;
  } while (first < ns - 1);
  if (!sd.pl.identity && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return nnhip::fail_msg(NNHIP_EHIP, "cumtrapz: stream failed");  // the gathered series are freed on return
  return NNHIP_OK;
}

int nnhip_cumsimpson_batch_f64_dev(const double* X, int n, const double* Y, int64_t M, double* out, void* stream) {
  if (M < 0 || !X) return nnhip::fail_msg(NNHIP_EVALUE, "cumsimpson: need M >= 0 and a non-NULL X");
  if (n < 1) return nnhip::fail_msg(NNHIP_EVALUE, "cumsimpson: n must be >= 1");
  if (M > 0 && (!Y || !out)) return nnhip::fail_msg(NNHIP_EVALUE, "cumulative quadrature: Y / out is NULL");
  hipStream_t s = (hipStream_t)stream;
  nnhip::SortedDataset sd;   // `var (xSorted, ySorted) = sortAndTrimDataset(@X, @Y)` (integrate.nim:340)
  const double* srt[1];
  if (int rcs = nnhip::sorted_dataset("cumsimpson(Y, X)", X, n, &Y, 1, M, s, sd, srt)) return rcs;
  if (sd.n() < 3) return nnhip::fail_msg(NNHIP_EVALUE, "cumsimpson: X and Y must have at least 3 elements (got %d distinct abscissae)", sd.n());  // ValueError (integrate.nim:345-346)
  if (M == 0) return NNHIP_OK;
  // the rule runs on the sorted data; the result is read back at the CALLER's abscissae (hermiteInterpolate(X, xs, y, dy), :375): value at x = row rank(x)
  std::vector<int32_t> resultRows;
  nnhip::simpson_result_rows(sd.pl, X, n, resultRows);
  const bool direct = sd.pl.identity;  // then row j of the result is sorted row j
  const int nCaller = n;
  double* outCaller = out;
  double* sortedOut = nullptr;
  int32_t* dRows = nullptr;
  struct Scratch { double*& a; int32_t*& b; ~Scratch() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); } } scratch{sortedOut, dRows};
  if (!direct) {
    if (hipMalloc((void**)&sortedOut, (size_t)sd.n() * (size_t)M * sizeof(double)) != hipSuccess || hipMalloc((void**)&dRows, (size_t)nCaller * sizeof(int32_t)) != hipSuccess)
      return nnhip::fail_msg(NNHIP_ENOMEM, "cumsimpson: hipMalloc failed");
    out = sortedOut;
  }
  X = sd.X(); n = sd.n(); Y = srt[0];
  std::vector<nnhip::SimpsonPair> pairs;
  std::vector<nnhip::SimpsonPoint> pts;
  int64_t nPairs64 = 0;
  bool evenN = false;
  nnhip::simpson_tables(X, n, pairs, pts, nPairs64, evenN);
  const int nPairs = (int)nPairs64;
  nnhip::SimpsonPair* dPairs = nullptr;
  nnhip::SimpsonPoint* dPts = nullptr;
  if (hipMalloc((void**)&dPairs, pairs.size() * sizeof(pairs[0])) != hipSuccess) return nnhip::fail_msg(NNHIP_ENOMEM, "cumsimpson: hipMalloc failed");
  if (hipMalloc((void**)&dPts, pts.size() * sizeof(pts[0])) != hipSuccess) { (void)hipFree(dPairs); return nnhip::fail_msg(NNHIP_ENOMEM, "cumsimpson: hipMalloc failed"); }
  int rc = NNHIP_OK;
  if (hipMemcpyAsync(dPairs, pairs.data(), pairs.size() * sizeof(pairs[0]), hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(dPts, pts.data(), pts.size() * sizeof(pts[0]), hipMemcpyHostToDevice, s) != hipSuccess)
    rc = NNHIP_EHIP;
  const dim3 grid((unsigned)((M + nnhip::kBlock - 1) / nnhip::kBlock)), block(nnhip::kBlock);
  if (!rc && nnhip::launch_kernel(nnhip::cumsimpson_kernel, grid, block, s, (const nnhip::SimpsonPair*)dPairs, nPairs, evenN ? 1 : 0,
                                  (const nnhip::SimpsonPoint*)dPts, Y, out, M, n) != hipSuccess)
    rc = NNHIP_EHIP;
  if (!rc && !direct) {
    resultRows.resize((size_t)nCaller, -1);  // rows the reference's result does not have (a repeated maximum of a sorted X): NaN
    if (hipMemcpyAsync(dRows, resultRows.data(), (size_t)nCaller * sizeof(int32_t), hipMemcpyHostToDevice, s) != hipSuccess ||
        nnhip::launch_kernel(nnhip::gather_rows_kernel, dim3(grid.x, (unsigned)std::min(nCaller, 65535)), block, s, (const int32_t*)dRows, nCaller, (const double*)sortedOut, outCaller, M) != hipSuccess)
      rc = NNHIP_EHIP;
  }
  (void)hipStreamSynchronize(s);  // the weight tables are freed below
  (void)hipFree(dPairs);
  (void)hipFree(dPts);
  return rc;
}

int nnhip_sort_and_trim_dataset_f64_dev(const double* X, int n, const double* const* Y, int n_y, int64_t M, double* X_out, double* const* Y_out, int* n_out,
                                        void* stream) {
  if (n < 1 || n_y < 0 || M < 0 || !X || !X_out || !n_out || (n_y > 0 && (!Y || !Y_out))) return nnhip::fail_msg(NNHIP_EVALUE, "sortAndTrimDataset: need n >= 1, n_y >= 0, M >= 0 and non-NULL X / X_out / n_out / Y / Y_out");
  if (n_y > 8) return nnhip::fail_msg(NNHIP_EUNSUPPORTED, "sortAndTrimDataset: at most 8 series arrays (got %d)", n_y);
  for (int k = 0; k < n_y; ++k) if (M > 0 && (!Y[k] || !Y_out[k] || Y[k] == Y_out[k])) return nnhip::fail_msg(NNHIP_EVALUE, "sortAndTrimDataset: series %d is NULL or Y_out aliases Y", k);
  hipStream_t s = (hipStream_t)stream;
  nnhip::SortedDataset sd;
  const double* srt[8];
  if (int rcs = nnhip::sorted_dataset("sortAndTrimDataset", X, n, Y, n_y, M, s, sd, srt)) return rcs;
  const int ns = sd.n();
  for (int k = 0; k < n_y && M > 0; ++k) {
    if (hipMemcpyAsync(Y_out[k], srt[k], (size_t)ns * (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess) return nnhip::fail_msg(NNHIP_EHIP, "sortAndTrimDataset: copy failed");
    if (int rcn = nnhip::nan_rows(Y_out[k], ns, n, M, s)) return rcn;
  }
  if (!sd.pl.identity && hipStreamSynchronize(s) != hipSuccess) return nnhip::fail_msg(NNHIP_EHIP, "sortAndTrimDataset: stream failed");  // the gathered series are freed on return
  std::copy(sd.pl.x.begin(), sd.pl.x.end(), X_out);
  for (int i = ns; i < n; ++i) X_out[i] = __builtin_nan("");
  *n_out = ns;
  return NNHIP_OK;
}

int nnhip_dataset_rows_f64(const double* X, int n, int* n_sorted_trimmed, int* n_cumsimpson_rows) {
  if (n < 1 || !X) return nnhip::fail_msg(NNHIP_EVALUE, "dataset rows: need n >= 1 and a non-NULL X");
  nnhip::DatasetPlan pl;
  std::string why;
  if (nnhip::dataset_plan(X, n, pl, why) != 0) return nnhip::fail_msg(NNHIP_EVALUE, "sortAndTrimDataset: %s", why.c_str());
  if (n_sorted_trimmed) *n_sorted_trimmed = (int)pl.x.size();
  if (n_cumsimpson_rows) {
    std::vector<int32_t> rows;
    nnhip::simpson_result_rows(pl, X, n, rows);
    *n_cumsimpson_rows = (int)rows.size();
  }
  return NNHIP_OK;
}

int nnhip_ode_controller_factor_f64_dev(int order, const double* error, double* out, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!error || !out))) return nnhip::fail_msg(NNHIP_EVALUE, "controller_factor: n < 0 or a NULL array");
  if (n == 0) return NNHIP_OK;
  const dim3 grid((unsigned)((n + nnhip::kBlock - 1) / nnhip::kBlock)), block(nnhip::kBlock);
  hipError_t e;
  switch (order) {
    case 2: e = nnhip::launch_kernel(nnhip::controller_factor_kernel<2>, grid, block, (hipStream_t)stream, error, out, n); break;
    case 3: e = nnhip::launch_kernel(nnhip::controller_factor_kernel<3>, grid, block, (hipStream_t)stream, error, out, n); break;
    case 5: e = nnhip::launch_kernel(nnhip::controller_factor_kernel<5>, grid, block, (hipStream_t)stream, error, out, n); break;
    case 6: e = nnhip::launch_kernel(nnhip::controller_factor_kernel<6>, grid, block, (hipStream_t)stream, error, out, n); break;
    default: return nnhip::fail_msg(NNHIP_EVALUE, "controller_factor: order must be 2, 3, 5 or 6 (got %d)", order);
  }
  return e == hipSuccess ? NNHIP_OK : nnhip::fail_msg(NNHIP_EHIP, "controller_factor: kernel launch failed: %s", hipGetErrorString(e));
}

int nnhip_ode_tableau_f64(int integrator, int device, double* out, int cap) {
  if (!out) return nnhip::fail_msg(NNHIP_EVALUE, "nnhip_ode_tableau_f64: out is null");
  switch (integrator) {
    case NNHIP_DOPRI54: return nnhip::tableau_read<NNHIP_DOPRI54>(device, out, cap);
    case NNHIP_TSIT54: return nnhip::tableau_read<NNHIP_TSIT54>(device, out, cap);
    case NNHIP_VERN65: return nnhip::tableau_read<NNHIP_VERN65>(device, out, cap);
    default: return nnhip::fail_msg(NNHIP_EINTEGRATOR, "nnhip_ode_tableau_f64: integrator %d has no tableau (its coefficients are literals of its step expression)", integrator);
  }
}

int nnhip_ode_rhs_batch_f64_dev(int rhs_kind, const double* rhs_params, int n_params, int64_t N, int dim, int layout, double t,
                                const double* y, double* dy, void* stream) {
  if (N < 0 || dim < 1 || n_params < 0 || n_params > nnhip::kMaxParams) return nnhip::fail_msg(NNHIP_EVALUE, "rhs_batch: bad N / dim / n_params");
  if (n_params > 0 && !rhs_params) return nnhip::fail_msg(NNHIP_EVALUE, "rhs_batch: rhs_params is NULL");
  if (layout != NNHIP_LAYOUT_SOA && layout != NNHIP_LAYOUT_AOS) return nnhip::fail_msg(NNHIP_EVALUE, "rhs_batch: layout must be NNHIP_LAYOUT_SOA or NNHIP_LAYOUT_AOS");
  if (N == 0) return NNHIP_OK;
  if (!y || !dy) return nnhip::fail_msg(NNHIP_EVALUE, "rhs_batch: y / dy is NULL");
  nnhip::Params P;
  for (int k = 0; k < nnhip::kMaxParams; ++k) P.p[k] = k < n_params ? rhs_params[k] : 0.0;
  if (nnhip::rtc_ctx_fill(rhs_kind, N, P, nullptr) < 0) return nnhip::fail_msg(NNHIP_EVALUE, "rhs_batch: rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  const int64_t is = layout == NNHIP_LAYOUT_SOA ? 1 : dim, cs = layout == NNHIP_LAYOUT_SOA ? N : 1;
  if (rhs_kind >= NNHIP_RHS_USER_BASE) {
    int d = 0;
    if (!nnhip::rtc_info(rhs_kind, &d, nullptr) || d != dim) return nnhip::fail_msg(NNHIP_EVALUE, "rhs_batch: unknown user rhs_kind %d or dim mismatch", rhs_kind);
    return nnhip::rtc_launch_rhs(rhs_kind, N, is, cs, t, y, dy, P, (hipStream_t)stream) == hipSuccess ? NNHIP_OK : nnhip::fail_msg(NNHIP_EHIP, "rhs_batch: %s", nnhip::rtc_last_error());
  }
  hipError_t e = hipErrorInvalidValue;
  bool found = false;
#define X(kind, d, T)                                                                    \
  if (!found && rhs_kind == kind && dim == d) {                                          \
    found = true;                                                                        \
    e = nnhip::launch_rhs_batch<nnhip::T>(N, is, cs, t, y, dy, P, (hipStream_t)stream);  \
  }
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
#define X(kind, d, T, CA, CF)                                                            \
  if (!found && rhs_kind == kind && dim == d) {                                          \
    found = true;                                                                        \
    e = nnhip::launch_rhs_batch<nnhip::T>(N, is, cs, t, y, dy, P, (hipStream_t)stream);  \
  }
  NNHIP_FOR_EACH_LPS_RHS(X)
#undef X
  if (!found) {  // size-generic built-in kind at a size without an ahead-of-time kernel: run-time instantiation
    const int k = nnhip::rtc_builtin_kind(rhs_kind, dim);
    if (k < 0) return nnhip::fail_msg(NNHIP_EUNSUPPORTED, "rhs_batch: no kernel for rhs_kind=%d dim=%d", rhs_kind, dim);
    return nnhip::rtc_launch_rhs(k, N, is, cs, t, y, dy, P, (hipStream_t)stream) == hipSuccess ? NNHIP_OK : nnhip::fail_msg(NNHIP_EHIP, "rhs_batch: %s", nnhip::rtc_last_error());
  }
  return e == hipSuccess ? NNHIP_OK : nnhip::fail_msg(NNHIP_EHIP, "rhs_batch: kernel launch failed: %s", hipGetErrorString(e));
}

int nnhip_ode_solve_batch_multi_gpu_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind,
                                        const double* rhs_params, int n_params, const double* y0, int64_t N, int dim,
                                        int layout, const double* tspan, int n_t, double* t_out, double* y_out,
                                        int32_t* ny_out, int64_t max_steps, nnhip_ode_stats* stats, int n_gpus) {
  return nnhip_ode_solve_batch_multi_gpu_sweep_f64(opt, integrator, rhs_kind, rhs_params, n_params, nullptr, 0, y0, N, dim, layout, tspan, n_t, t_out, y_out,
                                                   ny_out, nullptr, nullptr, max_steps, stats, n_gpus);
}

// The same with per-IVP right-hand-side parameters (every IVP its own ctx, ode.nim:589-591, 599) and the per-IVP step counters: a
// parameter sweep sharded over the devices.  Shard r reads columns [lo_r, hi_r) of the caller's [n_per_ivp][N] table in place.
int nnhip_ode_solve_batch_multi_gpu_sweep_f64(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                              const double* per_ivp_params, int n_per_ivp, const double* y0, int64_t N, int dim, int layout,
                                              const double* tspan, int n_t, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                                              int64_t* rejected_out, int64_t max_steps, nnhip_ode_stats* stats, int n_gpus) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nnhip::fail_msg(NNHIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  if (n_gpus <= 0 || (n_gpus > ndev && !nnhip::multi_gpu_oversubscribe()) || n_gpus > 64)
    return nnhip::fail_msg(NNHIP_EVALUE, "n_gpus = %d, but this node has %d HIP device(s)", n_gpus, ndev);
  if (N < 0 || dim < 1 || n_t < 0 || !opt || (n_t > 0 && !tspan)) return nnhip::fail_msg(NNHIP_EVALUE, "bad sizes / NULL options or tspan");
  // A context block (NumContext beyond eight scalars: every IVP its own vectors / mutable slots, commonTypes.nim:4-27, ode.nim:599) travels with
  // the batch: device r gets the columns [lo_r, hi_r) of the per-IVP rows and slots, the shared block whole
  std::shared_ptr<void> ctxShards;
  if (n_gpus > 1 && nnhip::rtc_has_per_ivp_ctx(rhs_kind)) {
    std::vector<int> devs(n_gpus);
    std::vector<int64_t> los(n_gpus), ns(n_gpus);
    for (int r = 0; r < n_gpus; ++r) { devs[r] = r % ndev; los[r] = N * r / n_gpus; ns[r] = N * (r + 1) / n_gpus - los[r]; }
    if (nnhip::rtc_ctx_shards_prepare(rhs_kind, n_gpus, devs.data(), los.data(), ns.data(), &ctxShards) != 0)
      return nnhip::fail_msg(NNHIP_EUNSUPPORTED, "rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  }
  // The output time grid depends on (options, tspan) only: assembled once, on the calling thread (ode.nim:476-487, 585) — also
  // when some (or all) shards are empty.
  int nTOut = 0;
  int rc = nnhip_ode_time_grid(opt, tspan, n_t, t_out, &nTOut);
  if (rc) return rc;
  std::vector<int> rcs(n_gpus, NNHIP_OK);
  std::vector<std::string> errs(n_gpus);
  std::vector<nnhip_ode_stats> sts(n_gpus);
  std::vector<std::thread> th;
  // Contiguous index ranges [r*N/G, (r+1)*N/G): nothing couples trajectories (ode.nim:589: one solveODE call per IVP).  Every
  // device works on its range of the caller's own arrays (nnhip::solve_host_range: strided copies straight between the caller's
  // buffers and the device, chunked and overlapped with the kernel when the buffers are page-locked) — no intermediate copies.
  auto shard = [&](int r, bool worker) {
    const int64_t lo = N * r / n_gpus, hi = N * (r + 1) / n_gpus, n = hi - lo;
    std::memset(&sts[r], 0, sizeof(nnhip_ode_stats));
    sts[r].ny_min = 0x7fffffff;
    if (n == 0) return;
    if (ctxShards) (void)nnhip::rtc_ctx_shard_enter(rhs_kind, ctxShards, r);  // this thread's calls read shard r of the context block
    rcs[r] = nnhip::solve_host_range(opt, integrator, rhs_kind, rhs_params, n_params, per_ivp_params, n_per_ivp, y0, N, lo, n, dim, layout, tspan, n_t,
                                     nullptr, y_out, ny_out, steps_out, rejected_out, max_steps, &sts[r], r % ndev);
    if (rcs[r]) errs[r] = nnhip::thread_error();  // the message lives in THIS thread's buffer: hand it to the caller
    if (ctxShards) nnhip::rtc_ctx_shard_leave(rhs_kind);
    if (worker) nnhip::release_thread_staging();  // pinned staging + event of a (short-lived) worker thread
  };
  // One device: the whole batch on the CALLING thread — a context block is then read through this thread's own binding (the per-thread contract of
  // nnhip_ode_rhs_bind_ctx_f64: a fresh worker thread has no binding of its own and would read the latest bind of ANY thread).  Several devices: one worker
  // thread each, reading the column range cut above from the calling thread's binding.
  if (n_gpus == 1) shard(0, false);
  else {
    for (int r = 0; r < n_gpus; ++r) th.emplace_back(shard, r, true);
    for (auto& t : th) t.join();
  }
  if (ctxShards && nnhip::rtc_ctx_shards_collect(ctxShards) != 0)  // the mutable slots back where nnhip_ode_rhs_read_aux_f64 and single-device calls read them
    return nnhip::fail_msg(NNHIP_EHIP, "rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  for (int r = 0; r < n_gpus; ++r)
    if (rcs[r]) return nnhip::fail_msg(rcs[r], "device %d (IVPs %lld..%lld): %s", r, (long long)(N * r / n_gpus), (long long)(N * (r + 1) / n_gpus), errs[r].c_str());
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    stats->ny_min = 0x7fffffff;
    for (int r = 0; r < n_gpus; ++r) {
      stats->steps_total += sts[r].steps_total; stats->rejected_total += sts[r].rejected_total;
      if (sts[r].steps_max > stats->steps_max) stats->steps_max = sts[r].steps_max;
      if (sts[r].ny_min < stats->ny_min) stats->ny_min = sts[r].ny_min;
      stats->nan_aborts += sts[r].nan_aborts; stats->truncated += sts[r].truncated;
      if (sts[r].kernel_ms > stats->kernel_ms) stats->kernel_ms = sts[r].kernel_ms;
    }
    if (N == 0) stats->ny_min = 0;
    stats->n_t_out = nTOut;
  }
  return NNHIP_OK;
}

// ---- C5 behind one call (BASELINE.json config 5) ------------------------------------------------------------------
// One process, n_gpus devices.  Device r holds its contiguous shard (counts[r] IVPs) resident in its own memory; every shard is
// integrated on its device's stream (one worker thread per device enqueues it: the reference is one solveODE call per IVP,
// ode.nim:589-591, nothing couples trajectories), then the shards are reassembled ON EVERY DEVICE with RCCL over xGMI
// (ncclAllGather for equal shards, grouped ncclBroadcasts for ragged ones) into device-resident full tensors.  No host copy anywhere.
// The collective is enqueued behind the solve — on the solve's own stream, or on gather_streams[r] behind an event, so that the
// caller's next solve on streams[r] overlaps the gather (as bench.py does).  Returns after enqueueing; the caller synchronises.
namespace {
int mg_check(int n_gpus, const int64_t* counts, bool gather, int* ndev_out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return nnhip::fail_msg(NNHIP_EHIP, "no HIP device available (this library has no CPU fallback)");
  // (knob "multi_gpu_oversubscribe": more shards than devices, shard r on device r mod #devices — exercises the sharding on a small box; RCCL itself
  // wants one device per rank, so the gather is refused then)
  if (n_gpus <= 0 || n_gpus > 64 || (n_gpus > ndev && !(nnhip::multi_gpu_oversubscribe() && !gather)))
    return nnhip::fail_msg(NNHIP_EVALUE, "n_gpus = %d, but this node has %d HIP device(s)", n_gpus, ndev);
  if (!counts) return nnhip::fail_msg(NNHIP_EVALUE, "counts is NULL");
  for (int r = 0; r < n_gpus; ++r) if (counts[r] < 0) return nnhip::fail_msg(NNHIP_EVALUE, "counts[%d] < 0", r);
  if (ndev_out) *ndev_out = ndev;
  return NNHIP_OK;
}
// the context block of `rhs_kind` cut along `counts` (shard r on device r mod ndev); *out stays empty when there is nothing to cut
int mg_ctx_shards(int rhs_kind, int n_gpus, const int64_t* counts, int ndev, std::shared_ptr<void>* out) {
  if (n_gpus <= 1 || !nnhip::rtc_has_per_ivp_ctx(rhs_kind)) return NNHIP_OK;
  std::vector<int> devs(n_gpus);
  std::vector<int64_t> los(n_gpus), ns(n_gpus);
  int64_t lo = 0;
  for (int r = 0; r < n_gpus; ++r) { devs[r] = r % ndev; los[r] = lo; ns[r] = counts[r]; lo += counts[r]; }
  if (nnhip::rtc_ctx_shards_prepare(rhs_kind, n_gpus, devs.data(), los.data(), ns.data(), out) != 0)
    return nnhip::fail_msg(NNHIP_EUNSUPPORTED, "rhs_kind %d: %s", rhs_kind, nnhip::rtc_last_error());
  return NNHIP_OK;
}
// gather on gather_streams[r] (nullable array) behind whatever streams[r] holds now
int mg_gather(int n_gpus, const double* const* shard, const int64_t* counts, int planesOrDim, int layout, double* const* full, void* const* streams,
              void* const* gather_streams) {
  if (gather_streams) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int r = 0; r < n_gpus; ++r) {
      if (gather_streams[r] == (streams ? streams[r] : nullptr)) continue;
      hipEvent_t ev = nullptr;
      if (hipSetDevice(r) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
          hipEventRecord(ev, streams ? (hipStream_t)streams[r] : nullptr) != hipSuccess ||
          hipStreamWaitEvent((hipStream_t)gather_streams[r], ev, 0) != hipSuccess) {
        if (ev) (void)hipEventDestroy(ev);
        (void)hipSetDevice(prev);
        return nnhip::fail_msg(NNHIP_EHIP, "device %d: ordering the gather behind the solve failed", r);
      }
      (void)hipEventDestroy(ev);  // released by the runtime once the recorded work has completed
    }
    (void)hipSetDevice(prev);
  }
  const int rc = nnhip_allgather_states_f64_dev(n_gpus, shard, counts, planesOrDim, layout, full, gather_streams ? gather_streams : streams);
  if (rc) return nnhip::fail_msg(rc, "reassembling the shards: %s", nnhip_multigpu_last_error());
  return NNHIP_OK;
}
}  // namespace

int nnhip_ode_fixed_stream_multi_gpu_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                             int n_gpus, const int64_t* counts, int dim, int layout, double t0, double tEnd, double* const* y,
                                             double* const* scratch, double* const* full, void* const* streams, void* const* gather_streams,
                                             int64_t* n_steps_out, double** y_final) {
  int ndev = 1;
  int rc = mg_check(n_gpus, counts, full != nullptr, &ndev);
  if (rc) return rc;
  if (!y || !streams) return nnhip::fail_msg(NNHIP_EVALUE, "y / streams is NULL (one non-default stream per device)");
  std::shared_ptr<void> ctxShards;  // a context block travels with the batch: shard r's columns on shard r's device (asynchronous entry: the mutable
  rc = mg_ctx_shards(rhs_kind, n_gpus, counts, ndev, &ctxShards);  // slots are gathered by their next reader, e.g. nnhip_ode_rhs_read_aux_f64)
  if (rc) return rc;
  std::vector<int> rcs(n_gpus, NNHIP_OK);
  std::vector<std::string> errs(n_gpus);
  std::vector<int64_t> steps(n_gpus, 0);
  std::vector<double*> fin(n_gpus, nullptr);
  auto work = [&](int r) {
    if (hipSetDevice(r % ndev) != hipSuccess) { rcs[r] = NNHIP_EHIP; errs[r] = "hipSetDevice failed"; return; }
    fin[r] = y[r];
    if (counts[r] == 0) return;
    if (ctxShards) (void)nnhip::rtc_ctx_shard_enter(rhs_kind, ctxShards, r);
    rcs[r] = nnhip_ode_fixed_stream_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, counts[r], dim, layout, t0, tEnd, y[r],
                                            scratch ? scratch[r] : nullptr, &steps[r], &fin[r], streams[r]);
    if (rcs[r]) errs[r] = nnhip::thread_error();
    if (ctxShards) nnhip::rtc_ctx_shard_leave(rhs_kind);
  };
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (n_gpus == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int r = 0; r < n_gpus; ++r) th.emplace_back(work, r);
    for (auto& t : th) t.join();
  }
  (void)hipSetDevice(prev);
  for (int r = 0; r < n_gpus; ++r) if (rcs[r]) return nnhip::fail_msg(rcs[r], "device %d: %s", r, errs[r].c_str());
  if (n_steps_out) { *n_steps_out = 0; for (int r = 0; r < n_gpus; ++r) if (steps[r] > *n_steps_out) *n_steps_out = steps[r]; }
  if (y_final) for (int r = 0; r < n_gpus; ++r) y_final[r] = fin[r];
  if (!full) return NNHIP_OK;
  return mg_gather(n_gpus, fin.data(), counts, dim, layout, full, streams, gather_streams);
}

// The fused solve (any integrator, any tspan) per device-resident shard, then the trajectory tensor reassembled on every device:
// y0[r] / y_out[r] on device r ([dim][counts[r]] / [n_t][dim][counts[r]] for SoA), full[r] = [n_t][dim][N] (SoA) or [n_t][N][dim] (AoS)
// on device r (nullable array: no gather).  ny_out[r] (nullable array, int32 [counts[r]] on device r) as nnhip_ode_solve_batch_f64_dev.
// ws[r]: device scratch of nnhip_ode_solve_workspace_bytes(n_t) on device r.
int nnhip_ode_solve_batch_multi_gpu_f64_dev(const nnhip_ode_options* opt, int integrator, int rhs_kind, const double* rhs_params, int n_params,
                                            int n_gpus, const int64_t* counts, int dim, int layout, const double* tspan, int n_t, double* t_out,
                                            const double* const* y0, double* const* y_out, int32_t* const* ny_out, int64_t max_steps,
                                            void* const* ws, int64_t ws_bytes, double* const* full, void* const* streams, void* const* gather_streams) {
  int ndev = 1;
  int rc = mg_check(n_gpus, counts, full != nullptr, &ndev);
  if (rc) return rc;
  if (!y0 || !y_out || !streams || !ws) return nnhip::fail_msg(NNHIP_EVALUE, "y0 / y_out / ws / streams is NULL");
  std::shared_ptr<void> ctxShards;
  rc = mg_ctx_shards(rhs_kind, n_gpus, counts, ndev, &ctxShards);
  if (rc) return rc;
  int nTOut = 0;
  rc = nnhip_ode_time_grid(opt, tspan, n_t, t_out, &nTOut);  // (options, tspan) only: once, also when shards are empty
  if (rc) return rc;
  std::vector<int> rcs(n_gpus, NNHIP_OK);
  std::vector<std::string> errs(n_gpus);
  auto work = [&](int r) {
    if (hipSetDevice(r % ndev) != hipSuccess) { rcs[r] = NNHIP_EHIP; errs[r] = "hipSetDevice failed"; return; }
    if (counts[r] == 0) return;
    if (ctxShards) (void)nnhip::rtc_ctx_shard_enter(rhs_kind, ctxShards, r);
    rcs[r] = nnhip_ode_solve_batch_f64_dev(opt, integrator, rhs_kind, rhs_params, n_params, y0[r], counts[r], dim, layout, tspan, n_t, nullptr, y_out[r],
                                           ny_out ? ny_out[r] : nullptr, nullptr, nullptr, max_steps, ws[r], ws_bytes, streams[r]);
    if (rcs[r]) errs[r] = nnhip::thread_error();
    if (ctxShards) nnhip::rtc_ctx_shard_leave(rhs_kind);
    nnhip::release_thread_staging();
  };
  int prev = 0;
  (void)hipGetDevice(&prev);
  if (n_gpus == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int r = 0; r < n_gpus; ++r) th.emplace_back(work, r);
    for (auto& t : th) t.join();
  }
  (void)hipSetDevice(prev);
  for (int r = 0; r < n_gpus; ++r) if (rcs[r]) return nnhip::fail_msg(rcs[r], "device %d: %s", r, errs[r].c_str());
  if (!full || n_t == 0) return NNHIP_OK;
  std::vector<const double*> sh(n_gpus);
  std::vector<double*> fu(n_gpus);
  if (layout == NNHIP_LAYOUT_SOA) {  // [n_t][dim][count] -> [n_t][dim][N]: n_t * dim planes, gathered plane by plane in one group
    for (int r = 0; r < n_gpus; ++r) { sh[r] = y_out[r]; fu[r] = full[r]; }
    return mg_gather(n_gpus, sh.data(), counts, n_t * dim, NNHIP_LAYOUT_SOA, fu.data(), streams, gather_streams);
  }
  int64_t N = 0;
  for (int r = 0; r < n_gpus; ++r) N += counts[r];
  for (int j = 0; j < n_t; ++j) {  // [n_t][count][dim] -> [n_t][N][dim]: one contiguous block per shard and row
    for (int r = 0; r < n_gpus; ++r) { sh[r] = y_out[r] + (int64_t)j * counts[r] * dim; fu[r] = full[r] + (int64_t)j * N * dim; }
    rc = mg_gather(n_gpus, sh.data(), counts, dim, NNHIP_LAYOUT_AOS, fu.data(), streams, j == 0 ? gather_streams : nullptr);
    if (rc) return rc;
    if (j == 0 && gather_streams) streams = gather_streams;  // the rows after the first follow on the gather streams
  }
  return NNHIP_OK;
}

}  // extern "C"

// ode_sort.hip — order of integration for divergence binning: one float64 key per IVP counted into bins on the device (hand-written
// counting sort; no library sort).  Used by nnhip_ode_solve_batch_sorted_f64_dev (ode_capi.hip): the fused adaptive
// kernels integrate IVP perm[k] in work item k, so lanes of a wavefront hold IVPs with similar step sequences.
#include <hip/hip_runtime.h>
#include <cstring>

#include <cstdint>

#include "sort_kernels.hpp"

namespace nnhip {

namespace {
// key of the automatic mode: IVPs that got furthest in the probe (largest steps) first; NaN / non-finite progress last
__global__ void negate_kernel(const double* in, double* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double v = in[i]; out[i] = (v == v) ? -v : __longlong_as_double(0x7ff0000000000000LL); }
}
using namespace sortk;  // bin_count_kernel / bin_place_kernel and their helpers: sort_kernels.hpp
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ODESolver's bookkeeping before the loops (ode.nim:476-487, 609) for EVERY IVP its own tspan: one thread sorts its row (insertion
// sort: stable like Nim's `sorted`, rows are short), splits it around its tStart and lays it out as the solve kernels read it —
// tNegative (values < tStart) in descending order at the front, tPositive (values > tStart) ascending at the back — and writes the
// row of output times the reference returns: tNegative.reversed ++ (tStart if it is in tspan) ++ tPositive (:585), NaN beyond.
__global__ void prepare_tspans_kernel(const double* tspans, int n_t, int64_t N, const double* tStart, double t0u, double* grid, int32_t* counts,
                                      double* t_out, double* span_key) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double* in = tspans + i * n_t;
  double* g = grid + i * n_t;
  const double t0 = tStart ? tStart[i] : t0u;
  const double qnan = __longlong_as_double(0x7ff8000000000000LL), inf = __longlong_as_double(0x7ff0000000000000LL);
  bool ok = true;
  for (int j = 0; j < n_t; ++j) {
    const double v = in[j];
    g[j] = v;
    ok = ok && (v == v) && fabs(v) != inf;
  }
  if (!ok) {  // the reference's loop would never end on a non-finite requested time: this call is refused, the others are not
    if (span_key) span_key[i] = inf;  // (binning key: refused calls last)
    counts[3 * i] = -1; counts[3 * i + 1] = 0; counts[3 * i + 2] = 0;
    if (t_out) for (int j = 0; j < n_t; ++j) t_out[i * n_t + j] = qnan;
    return;
  }
  for (int j = 1; j < n_t; ++j) {
    const double v = g[j];
    int k = j - 1;
    while (k >= 0 && g[k] > v) { g[k + 1] = g[k]; --k; }
    g[k + 1] = v;
  }
  int nNeg = 0, nPos = 0;
  for (int j = 0; j < n_t; ++j) { nNeg += g[j] < t0 ? 1 : 0; nPos += g[j] > t0 ? 1 : 0; }   // :479-480
  if (span_key) {  // binning key of the call: minus the time it integrates over (both directions), longest first; NaN t0: last
    const double span = (nPos > 0 ? g[n_t - 1] - t0 : 0.0) + (nNeg > 0 ? t0 - g[0] : 0.0);
    span_key[i] = (span == span) ? -span : inf;
  }
  int nZero = 0;
  for (int j = nNeg; j < n_t - nPos; ++j) nZero |= (g[j] == t0) ? 1 : 0;                      // `t0 in tspan` (:485); false for every j when t0 is NaN
  if (t_out) {
    int w = 0;
    for (int j = 0; j < nNeg; ++j) t_out[i * n_t + w++] = g[j];
    if (nZero) t_out[i * n_t + w++] = t0;
    for (int j = n_t - nPos; j < n_t; ++j) t_out[i * n_t + w++] = g[j];
    for (; w < n_t; ++w) t_out[i * n_t + w] = qnan;
  }
  for (int lo = 0, hi = nNeg - 1; lo < hi; ++lo, --hi) { const double v = g[lo]; g[lo] = g[hi]; g[hi] = v; }  // tNegative as the reference holds it
  counts[3 * i] = nNeg; counts[3 * i + 1] = nZero; counts[3 * i + 2] = nPos;
}
}  // namespace

hipError_t prepare_tspans(const double* tspans, int n_t, int64_t N, const double* tStart, double t0, double* grid, int32_t* counts, double* t_out,
                          hipStream_t s, double* span_key) {
  if (N <= 0) return hipSuccess;
  void* args[] = {(void*)&tspans, (void*)&n_t, (void*)&N, (void*)&tStart, (void*)&t0, (void*)&grid, (void*)&counts, (void*)&t_out, (void*)&span_key};
  return hipLaunchKernel((const void*)prepare_tspans_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), args, 0, s);
}

// min / max of the finite keys as order-preserving 64-bit images ({~0, 0} when no key is finite), reduced in two small launches: one partial pair
// per block, then one block over the partials.  The result goes to the head of the binning workspace (the binning kernels read it there) and, if the
// caller wants to look at it, to two words of page-locked host memory (`pinned2`, valid once the stream has drained; plain stores, no host atomics).
namespace {
constexpr int kKeyRangeBlocks = 1024;
constexpr size_t kRangeBytes = 256 + (size_t)kKeyRangeBlocks * 16;  // [range: 2 words, padded][partials]
template <bool FINAL>
__global__ void key_range_kernel(const double* __restrict__ keys, const unsigned long long* __restrict__ partials, int64_t n, unsigned long long* __restrict__ out,
                                 unsigned long long* __restrict__ pinned2) {
  __shared__ unsigned long long smn[4], smx[4];
  unsigned long long mn = ~0ULL, mx = 0ULL;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (FINAL) {
      const unsigned long long a = partials[2 * i], b = partials[2 * i + 1];
      mn = a < mn ? a : mn;
      mx = b > mx ? b : mx;
    } else {
      const double v = keys[i];
      if (v == v && fabs(v) != __longlong_as_double(0x7ff0000000000000LL)) {
        const unsigned long long o = ordered_img(v);
        mn = o < mn ? o : mn;
        mx = o > mx ? o : mx;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long a = __shfl_down(mn, off, 64), b = __shfl_down(mx, off, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) { mn = smn[w] < mn ? smn[w] : mn; mx = smx[w] > mx ? smx[w] : mx; }
    out[2 * (FINAL ? 0 : blockIdx.x)] = mn;
    out[2 * (FINAL ? 0 : blockIdx.x) + 1] = mx;
    if (FINAL && pinned2) { pinned2[0] = mn; pinned2[1] = mx; }
  }
}
}  // namespace
// host side of the images: {min, max} as doubles
void key_range_decode(const unsigned long long* img, double* mn, double* mx) {
  auto dec = [](unsigned long long o) { const unsigned long long b = (o >> 63) ? (o & 0x7fffffffffffffffULL) : ~o; double v; std::memcpy(&v, &b, 8); return v; };
  *mn = img[0] == ~0ULL ? __builtin_inf() : dec(img[0]);
  *mx = img[1] == 0ULL ? -__builtin_inf() : dec(img[1]);
}
// `ws`: the workspace argsort_f64 is given afterwards (the range stays at its head)
hipError_t key_range_f64(const double* keys, int64_t N, void* ws, unsigned long long* pinned2, hipStream_t s) {
  if (pinned2) { pinned2[0] = ~0ULL; pinned2[1] = 0ULL; }  // host memory: the previous use has been waited for
  if (N <= 0) return hipSuccess;
  char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  unsigned long long* range = (unsigned long long*)base;
  unsigned long long* part = (unsigned long long*)(base + 256);
  int64_t blocks = (N + 255) / 256;
  if (blocks > kKeyRangeBlocks) blocks = kKeyRangeBlocks;
  const unsigned long long* none = nullptr;
  unsigned long long* nohost = nullptr;
  {
    void* args[] = {(void*)&keys, (void*)&none, (void*)&N, (void*)&part, (void*)&nohost};
    const hipError_t e = hipLaunchKernel((const void*)key_range_kernel<false>, dim3((unsigned)blocks), dim3(256), args, 0, s);
    if (e != hipSuccess) return e;
  }
  const double* nokeys = nullptr;
  const unsigned long long* cpart = part;
  void* args[] = {(void*)&nokeys, (void*)&cpart, (void*)&blocks, (void*)&range, (void*)&pinned2};
  return hipLaunchKernel((const void*)key_range_kernel<true>, dim3(1), dim3(256), args, 0, s);
}

// layout of the workspace: [range of the keys: two 64-bit images, + the partials of its reduction][histogram + cursors: 2 x kBins x 4][bins: 2N]
int64_t argsort_workspace_bytes(int64_t N) {
  if (N <= 0) return 0;
  return (int64_t)(align256(kRangeBytes) + align256(2 * kBins * 4) + align256((size_t)N * 2) + 256);
}

// perm_out[k] = index of an IVP of the k-th bin in ascending key order.  N < 2^31.  key_range_f64(keys, N, ws, ..) has been enqueued on the same
// stream before.  Failures come back as the hipError_t; the C entry that called turns them into its NNHIP_* code and thread-local message
// (nothing is printed from here).
hipError_t argsort_f64(const double* keys, int64_t N, uint32_t* perm_out, void* ws, int64_t ws_bytes, hipStream_t s, double min_spread_on_device) {
  if (N <= 0) return hipSuccess;
  if (N >= (int64_t)1 << 31 || ws_bytes < argsort_workspace_bytes(N)) return hipErrorInvalidValue;
  char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  const unsigned long long* range = (const unsigned long long*)base;
  uint32_t* hist = (uint32_t*)(base + align256(kRangeBytes));
  uint32_t* cursor = hist + kBins;
  uint16_t* bins = (uint16_t*)((char*)hist + align256(2 * kBins * 4));
  hipError_t e = hipMemsetAsync(hist, 0, 2 * kBins * 4, s);
  if (e != hipSuccess) return e;
  const unsigned blocks = (unsigned)((N + kBinThreads * kBinItems - 1) / (kBinThreads * kBinItems));
  {  // hipLaunchKernel's own status (hipGetLastError() can hand back a stale error of an unrelated earlier call)
    void* args[] = {(void*)&keys, (void*)&range, (void*)&bins, (void*)&hist, (void*)&N};
    e = hipLaunchKernel((const void*)bin_count_kernel, dim3(blocks), dim3(kBinThreads), args, 0, s);
    if (e != hipSuccess) return e;
  }
  const uint16_t* b = bins;
  const uint32_t* h = hist;
  void* args[] = {(void*)&b, (void*)&h, (void*)&cursor, (void*)&perm_out, (void*)&N, (void*)&range, (void*)&min_spread_on_device};
  return hipLaunchKernel((const void*)bin_place_kernel, dim3(blocks), dim3(kBinThreads), args, 0, s);
}

namespace {
// arrays viewed as [R][N][W]: GATHER dst[r][k][w] = src[r][perm[k]][w];  else (scatter) dst[r][perm[k]][w] = src[r][k][w]
template <class T, bool GATHER>
__global__ void permute_kernel(const T* __restrict__ src, T* __restrict__ dst, const uint32_t* __restrict__ perm, int64_t N, int R, int W) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // element (k, w) of one [N][W] plane
  if (e >= N * W) return;
  const int64_t k = e / W;
  const int w = (int)(e - k * W);
  const int64_t o = (int64_t)perm[k] * W + w;
  for (int r = 0; r < R; ++r) {
    const int64_t plane = (int64_t)r * N * W;
    if (GATHER) dst[plane + e] = src[plane + o];
    else dst[plane + o] = src[plane + e];
  }
}
template <class T, bool GATHER>
hipError_t permute_launch(const T* src, T* dst, const uint32_t* perm, int64_t N, int R, int W, hipStream_t s) {
  if (N <= 0 || R <= 0 || W <= 0) return hipSuccess;
  void* args[] = {(void*)&src, (void*)&dst, (void*)&perm, (void*)&N, (void*)&R, (void*)&W};
  return hipLaunchKernel((const void*)permute_kernel<T, GATHER>, dim3((unsigned)((N * W + 255) / 256)), dim3(256), args, 0, s);
}
}  // namespace
// Physical reordering around a binned solve: the batch is gathered into integration order once (8 bytes per element, one side of the
// copy scattered), solved with plain coalesced accesses, and the results are scattered back — cheaper than the same indirection inside
// the solve kernel, where every lane's first load and last stores waited on it (measured: 10 % of a 1.8 ms solve against ~60 us of copies).
hipError_t gather_f64(const double* src, double* dst, const uint32_t* perm, int64_t N, int R, int W, hipStream_t s) { return permute_launch<double, true>(src, dst, perm, N, R, W, s); }
hipError_t scatter_f64(const double* src, double* dst, const uint32_t* perm, int64_t N, int R, int W, hipStream_t s) { return permute_launch<double, false>(src, dst, perm, N, R, W, s); }
hipError_t scatter_i32(const int32_t* src, int32_t* dst, const uint32_t* perm, int64_t N, hipStream_t s) { return permute_launch<int32_t, false>(src, dst, perm, N, 1, 1, s); }
hipError_t scatter_i64(const int64_t* src, int64_t* dst, const uint32_t* perm, int64_t N, hipStream_t s) { return permute_launch<int64_t, false>(src, dst, perm, N, 1, 1, s); }
hipError_t gather_i32(const int32_t* src, int32_t* dst, const uint32_t* perm, int64_t N, hipStream_t s) { return permute_launch<int32_t, true>(src, dst, perm, N, 1, 1, s); }
hipError_t gather_i64(const int64_t* src, int64_t* dst, const uint32_t* perm, int64_t N, hipStream_t s) { return permute_launch<int64_t, true>(src, dst, perm, N, 1, 1, s); }
// inv[perm[k]] = k.  The way back from integration order goes through the INVERSE order as a gather (scattered 8-byte reads, full-line
// writes): scattered 8-byte WRITES each cost a partial-line read-modify-write at HBM — 7 of them per IVP (two rows of a 2-component state,
// ny, steps, rejected) were 0.25 ms of a 1.7 ms binned solve, the one scatter of 4-byte indices here is the only one left.
namespace {
__global__ void invert_perm_kernel(const uint32_t* __restrict__ perm, uint32_t* __restrict__ inv, int64_t N) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < N) inv[perm[k]] = (uint32_t)k;
}
}  // namespace
hipError_t invert_perm(const uint32_t* perm, uint32_t* inv, int64_t N, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  void* args[] = {(void*)&perm, (void*)&inv, (void*)&N};
  return hipLaunchKernel((const void*)invert_perm_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), args, 0, s);
}

// key of the automatic mode, variant "steps still to take": -(tEnd - t) / dt from where the probe stopped (most work first; anything non-finite last)
namespace {
__global__ void remaining_key_kernel(const double* __restrict__ t, const double* __restrict__ dt, double tEnd, double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double r = (tEnd - t[i]) / dt[i];
    out[i] = (r == r && r >= 0.0 && r != __longlong_as_double(0x7ff0000000000000LL)) ? -r : __longlong_as_double(0x7ff0000000000000LL);
  }
}
}  // namespace
hipError_t remaining_key_f64(const double* t, const double* dt, double tEnd, double* out, int64_t N, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  void* args[] = {(void*)&t, (void*)&dt, (void*)&tEnd, (void*)&out, (void*)&N};
  return hipLaunchKernel((const void*)remaining_key_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), args, 0, s);
}

// key of the per-IVP-call solves: the longest spans first, -|tEnd_i - tStart_i| (t_start NULL: the batch-wide t0); non-finite spans last
namespace {
__global__ void span_key_kernel(const double* __restrict__ tEnd, const double* __restrict__ tStart, double t0, double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double r = fabs(tEnd[i] - (tStart ? tStart[i] : t0));
    out[i] = (r == r && r != __longlong_as_double(0x7ff0000000000000LL)) ? -r : __longlong_as_double(0x7ff0000000000000LL);
  }
}
}  // namespace
hipError_t span_key_f64(const double* tEnd, const double* tStart, double t0, double* out, int64_t N, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  void* args[] = {(void*)&tEnd, (void*)&tStart, (void*)&t0, (void*)&out, (void*)&N};
  return hipLaunchKernel((const void*)span_key_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), args, 0, s);
}

hipError_t negate_f64(const double* in, double* out, int64_t N, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  void* args[] = {(void*)&in, (void*)&out, (void*)&N};
  return hipLaunchKernel((const void*)negate_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), args, 0, s);
}

}  // namespace nnhip

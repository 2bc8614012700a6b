// ode_sort.hip — order of integration for divergence binning: a stable argsort of one float64 key per IVP on the device
// (hipCUB radix sort of (key, index) pairs).  Used by nnhip_ode_solve_batch_sorted_f64_dev (ode_capi.hip): the fused adaptive
// kernels integrate IVP perm[k] in work item k, so lanes of a wavefront hold IVPs with similar step sequences.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>
#include <cstdio>

namespace nnhip {

namespace {
// key of the automatic mode: IVPs that got furthest in the probe (largest steps) first; NaN / non-finite progress last
__global__ void negate_kernel(const double* in, double* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double v = in[i]; out[i] = (v == v) ? -v : __longlong_as_double(0x7ff0000000000000LL); }
}
// Binning needs the ORDER of the keys only coarsely: a float64 key is narrowed to the top 16 bits of its float32 image in
// order-preserving unsigned form (sign, exponent, 7 mantissa bits: 128 bins per octave, so skewed key distributions still spread) —
// 2 radix passes instead of 8 (1e6 keys: 21 sort kernels / 170 us -> measured in profiles/r02_bench_divergence.json).  IVPs whose keys
// share a bin keep the caller's relative order (stable sort), as good a binning as any.
__global__ void narrow_keys_kernel(const double* in, uint16_t* out, uint32_t* iota, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const double v = in[i];
    uint32_t b = __float_as_uint((v == v) ? (float)v : __int_as_float(0x7f800000));  // NaN keys sort last
    b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);                                  // IEEE order -> unsigned order
    out[i] = (uint16_t)(b >> 16);
    iota[i] = (uint32_t)i;
  }
}
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ODESolver's bookkeeping before the loops (ode.nim:476-487, 609) for EVERY IVP its own tspan: one thread sorts its row (insertion
// sort: stable like Nim's `sorted`, rows are short), splits it around its tStart and lays it out as the solve kernels read it —
// tNegative (values < tStart) in descending order at the front, tPositive (values > tStart) ascending at the back — and writes the
// row of output times the reference returns: tNegative.reversed ++ (tStart if it is in tspan) ++ tPositive (:585), NaN beyond.
__global__ void prepare_tspans_kernel(const double* tspans, int n_t, int64_t N, const double* tStart, double t0u, double* grid, int32_t* counts,
                                      double* t_out) {
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
                          hipStream_t s) {
  if (N <= 0) return hipSuccess;
  void* args[] = {(void*)&tspans, (void*)&n_t, (void*)&N, (void*)&tStart, (void*)&t0, (void*)&grid, (void*)&counts, (void*)&t_out};
  return hipLaunchKernel((const void*)prepare_tspans_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), args, 0, s);
}

// layout of the workspace: [keys_in: 2N][keys_out: 2N][iota: 4N][cub temp]
int64_t argsort_workspace_bytes(int64_t N) {
  if (N <= 0) return 0;
  size_t temp = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, temp, (const uint16_t*)nullptr, (uint16_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                           (int)N);
  return (int64_t)(2 * align256((size_t)N * 2) + align256((size_t)N * 4) + align256(temp) + 256);
}

// perm_out[k] = index of the k-th smallest key (stable).  N < 2^31.
hipError_t argsort_f64(const double* keys, int64_t N, uint32_t* perm_out, void* ws, int64_t ws_bytes, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  if (N >= (int64_t)1 << 31 || ws_bytes < argsort_workspace_bytes(N)) {
    fprintf(stderr, "argsort_f64: N=%lld ws_bytes=%lld need=%lld\n", (long long)N, (long long)ws_bytes, (long long)argsort_workspace_bytes(N));
    return hipErrorInvalidValue;
  }
  char* base = (char*)(((uintptr_t)ws + 255) & ~(uintptr_t)255);
  uint16_t* keysIn = (uint16_t*)base;
  uint16_t* keysOut = (uint16_t*)(base + align256((size_t)N * 2));
  uint32_t* iota = (uint32_t*)(base + 2 * align256((size_t)N * 2));
  void* temp = base + 2 * align256((size_t)N * 2) + align256((size_t)N * 4);
  size_t tempBytes = (size_t)ws_bytes - (size_t)((char*)temp - (char*)ws);
  {  // hipLaunchKernel's own status (hipGetLastError() can hand back a stale error of an unrelated earlier call)
    void* args[] = {(void*)&keys, (void*)&keysIn, (void*)&iota, (void*)&N};
    const hipError_t e = hipLaunchKernel((const void*)narrow_keys_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), args, 0, s);
    if (e != hipSuccess) return e;
  }
  const hipError_t e2 = hipcub::DeviceRadixSort::SortPairs(temp, tempBytes, keysIn, keysOut, iota, perm_out, (int)N, 0, 16, s);
  if (e2 != hipSuccess) fprintf(stderr, "argsort_f64: SortPairs failed: %s (N=%lld tempBytes=%zu)\n", hipGetErrorString(e2), (long long)N, tempBytes);
  return e2;
}

hipError_t negate_f64(const double* in, double* out, int64_t N, hipStream_t s) {
  if (N <= 0) return hipSuccess;
  void* args[] = {(void*)&in, (void*)&out, (void*)&N};
  return hipLaunchKernel((const void*)negate_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), args, 0, s);
}

}  // namespace nnhip

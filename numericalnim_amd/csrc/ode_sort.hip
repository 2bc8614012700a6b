// ode_sort.hip — order of integration for divergence binning: one float64 key per IVP counted into bins on the device (hand-written
// counting sort; no library sort).  Used by nnhip_ode_solve_batch_sorted_f64_dev (ode_capi.hip): the fused adaptive
// kernels integrate IVP perm[k] in work item k, so lanes of a wavefront hold IVPs with similar step sequences.
#include <hip/hip_runtime.h>
#include <cstring>

#include <cstdint>

namespace nnhip {

namespace {
// key of the automatic mode: IVPs that got furthest in the probe (largest steps) first; NaN / non-finite progress last
__global__ void negate_kernel(const double* in, double* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const double v = in[i]; out[i] = (v == v) ? -v : __longlong_as_double(0x7ff0000000000000LL); }
}
// The ORDER OF INTEGRATION is a binning, not a sort: what the solve kernels gain from is that the 64 lanes of a wavefront hold IVPs with similar step
// sequences, and IVPs whose keys agree to 1 part in 4096 of the key range are as similar as the key can tell.  So the keys are counted into kBins
// bins and laid out bin after bin (a one-pass counting sort, three small launches, ~20 us at 1e6 keys) — rocPRIM's radix_sort_pairs, which round 3
// called, picks a 10-pass merge sort for 1e6 32-bit keys (168 us; 61 us with 16-bit keys and its one-sweep passes: profiles/r04_bench_divergence.json).
// The bin of a key, when all finite keys have one sign and none is zero: its order-preserving 64-bit image (monotone; logarithmic across binades,
// linear inside one) minus the image of the smallest finite key, shifted so that the largest lands in bin kBins-2.  The bins are spent on the range
// the keys actually cover: a sweep over [100, 101] and a probe whose progress spans six decades both use all of them.  The image is logarithmic
// ALL THE WAY DOWN, though: one key equal to 0 (an IVP that finished inside the probe: steps still to take = 0; a zero-length span) or keys of both
// signs (a centred parameter) stretch it over ~2000 binades, two bins per binade — a uniform sweep over [0, 10] then puts half its IVPs into four
// bins and the speed-up is gone without a trace in the results (round-4 advice).  So when the range touches or straddles zero the bins are LINEAR
// IN VALUE, (key - min) / (max - min): equal slices of the steps still to take / of the span / of the parameter.  Non-finite keys go last (-inf
// first).  Inside a bin the order is whatever the atomics produce — it differs from run to run and changes nothing but which wavefront an IVP
// rides in (the results are per IVP).
constexpr int kBins = 4096, kBinThreads = 1024, kBinItems = 4;
__device__ unsigned long long ordered_img(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ int bin_shift(const unsigned long long* range) {  // smallest shift with (max - min) >> shift <= kBins - 2
  const unsigned long long span = range[1] > range[0] ? range[1] - range[0] : 0ULL;
  int shift = 64 - __clzll((long long)span) - 12;
  shift = shift < 0 ? 0 : shift;
  return ((span >> shift) > (unsigned long long)(kBins - 2)) ? shift + 1 : shift;
}
__device__ double img_to_double(unsigned long long o) {
  const unsigned long long b = (o >> 63) ? (o & 0x7fffffffffffffffULL) : ~o;
  return __longlong_as_double((long long)b);
}
// all finite keys of one sign, none of them zero: the images of +0.0 / -0.0 are 0x8000... / 0x7fff...
__device__ bool bins_by_image(const unsigned long long* range) {
  const unsigned long long posZero = 0x8000000000000000ULL, negZero = 0x7fffffffffffffffULL;
  return (range[0] > posZero && range[1] > posZero) || (range[0] < negZero && range[1] < negZero);
}
// pass 1: the bin of every key (kept as 16 bits for pass 2) and the global histogram; `hist` arrives zeroed
__global__ __launch_bounds__(kBinThreads) void bin_count_kernel(const double* __restrict__ keys, const unsigned long long* __restrict__ range, uint16_t* __restrict__ bins,
                                                                 uint32_t* __restrict__ hist, int64_t n) {
  __shared__ uint32_t lh[kBins];
  for (int b = threadIdx.x; b < kBins; b += kBinThreads) lh[b] = 0;
  __syncthreads();
  const unsigned long long imgMin = range[0];
  const int shift = bin_shift(range);
  const bool byImage = bins_by_image(range);
  const double mnHalf = img_to_double(range[0]) * 0.5, halfSpan = img_to_double(range[1]) * 0.5 - mnHalf;  // (halved: max - min of finite doubles can overflow)
  const double perUnit = halfSpan > 0.0 ? (double)(kBins - 2) / halfSpan : 0.0;
  const int64_t base = (int64_t)blockIdx.x * (kBinThreads * kBinItems);
  for (int k = 0; k < kBinItems; ++k) {
    const int64_t i = base + k * kBinThreads + threadIdx.x;
    if (i < n) {
      const double v = keys[i];
      uint32_t q = kBins - 1;
      if (v == v && v != __longlong_as_double(0x7ff0000000000000LL)) {
        if (byImage) {
          const unsigned long long o = ordered_img(v);
          const unsigned long long d = o > imgMin ? (o - imgMin) >> shift : 0ULL;
          q = d > (unsigned long long)(kBins - 2) ? kBins - 2 : (uint32_t)d;
        } else {  // the range touches or straddles zero: linear in value
          const double r = (v * 0.5 - mnHalf) * perUnit;
          q = r >= (double)(kBins - 2) ? kBins - 2 : (r > 0.0 ? (uint32_t)r : 0u);  // (a NaN from 0 * inf on a denormal span lands in bin 0)
        }
      }
      bins[i] = (uint16_t)q;
      atomicAdd(&lh[q], 1u);
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += kBinThreads) { const uint32_t c = lh[b]; if (c) atomicAdd(&hist[b], c); }
}
// pass 2: every block scans the histogram into bin starts (4096 counters: cheaper than a launch of its own), reserves its share of each bin with one
// atomic per bin it holds keys of, and writes the indices of its keys there; `cursor` arrives zeroed
// `minSpread` > 0: keys within that fraction of their magnitude of each other (or no finite key at all) leave the batch in the caller's order — decided here, on
// the device, for callers that must not wait for the range (the per-IVP-call entries); the sorted entry decides on the host and passes 0
__global__ __launch_bounds__(kBinThreads) void bin_place_kernel(const uint16_t* __restrict__ bins, const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor,
                                                                 uint32_t* __restrict__ perm, int64_t n, const unsigned long long* __restrict__ range, double minSpread) {
  if (minSpread > 0.0) {
    bool narrow = range[0] > range[1];
    if (!narrow) {
      const double mn = img_to_double(range[0]), mx = img_to_double(range[1]);
      const double scale = fabs(mn) > fabs(mx) ? fabs(mn) : fabs(mx);
      narrow = !(scale > 0.0 && (mx - mn) > minSpread * scale);
    }
    if (narrow) {  // uniform: the whole grid takes this branch
      const int64_t base = (int64_t)blockIdx.x * (kBinThreads * kBinItems);
      for (int k = 0; k < kBinItems; ++k) {
        const int64_t i = base + k * kBinThreads + threadIdx.x;
        if (i < n) perm[i] = (uint32_t)i;
      }
      return;
    }
  }
  __shared__ uint32_t start[kBins];  // bin starts, then this block's base inside each bin
  __shared__ uint32_t lh[kBins];
  __shared__ uint32_t waveSum[kBinThreads / 64];
  constexpr int per = kBins / kBinThreads;  // consecutive bins per thread
  uint32_t c[per], sum = 0;
  for (int j = 0; j < per; ++j) { c[j] = hist[threadIdx.x * per + j]; sum += c[j]; lh[threadIdx.x * per + j] = 0; }
  uint32_t incl = sum;
  for (int off = 1; off < 64; off <<= 1) { const uint32_t v = __shfl_up(incl, off, 64); if ((int)(threadIdx.x & 63) >= off) incl += v; }
  if ((threadIdx.x & 63) == 63) waveSum[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t run = incl - sum;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += waveSum[w];
  for (int j = 0; j < per; ++j) { start[threadIdx.x * per + j] = run; run += c[j]; }
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * (kBinThreads * kBinItems);
  uint32_t q[kBinItems], rank[kBinItems];
  for (int k = 0; k < kBinItems; ++k) {
    const int64_t i = base + k * kBinThreads + threadIdx.x;
    if (i < n) { q[k] = bins[i]; rank[k] = atomicAdd(&lh[q[k]], 1u); }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += kBinThreads) { const uint32_t cnt = lh[b]; if (cnt) start[b] += atomicAdd(&cursor[b], cnt); }
  __syncthreads();
  for (int k = 0; k < kBinItems; ++k) {
    const int64_t i = base + k * kBinThreads + threadIdx.x;
    if (i < n) perm[start[q[k]] + rank[k]] = (uint32_t)i;
  }
}
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

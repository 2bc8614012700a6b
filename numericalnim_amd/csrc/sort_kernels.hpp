// sort_kernels.hpp — the two kernels of the order of integration (divergence binning), device code only: included by ode_sort.hip, and — under
// -DNNHIP_CPU_EMU — by the CPU test suite, which runs their bodies on the host (tests/cpp/hip_cpu_emu.hpp; test infrastructure).
#pragma once
#if defined(NNHIP_CPU_EMU)
#include "hip_cpu_emu.hpp"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace nnhip {
namespace sortk {
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
}  // namespace sortk
}  // namespace nnhip

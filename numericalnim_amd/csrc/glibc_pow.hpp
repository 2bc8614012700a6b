// glibc_pow.hpp — pow(x, y) for the step-size controller, bit-identical to the C library the reference runs on.
//
// The reference evaluates pow(1/error, 1/order) (src/numericalnim/ode.nim:71 in commonAdaptiveMethodCode, :537 in
// ODESolver) through Nim's std/math `pow` = <math.h> pow = glibc's table-driven double-precision pow (glibc >= 2.28:
// sysdeps/ieee754/dbl-64/e_pow.c; on every x86-64 CPU with FMA3 + AVX2 — the GPU box's EPYC and this image's Xeon —
// the ifunc resolver picks the build compiled with -mfma -mavx2).  That function is accurate to ~0.52 ulp but NOT
// correctly rounded, and a last-ulp difference in the controller factor can flip an `error <= 1` decision or make
// `t + (tEnd - t)` land one ulp short of tEnd (one more step).  So this file restates the published algorithm
// (log via a 128-entry table and a degree-7 polynomial in double-double, then exp via a 128-entry 2^(i/128) table)
// with every IEEE operation — including which multiply-adds are FUSED in the FMA build and which are not — in the
// order the library executes them; `__builtin_fma` is a correctly rounded FMA on both gfx950 (v_fma_f64) and x86-64
// FMA3, and the plain * + - are compiled without contraction (-ffp-contract=off), so host and device produce the
// library's bits.  tests/test_glibc_pow_port.py compares this against the live libm on >1e7 arguments (exact equality
// required) wherever the test suite runs; the tables come from scripts/extract_glibc_pow_tables.py.
//
// Domain: x >= 0 (any finite value incl. subnormals and 0, +inf, NaN), 0 < y <= 1 with y a normal number —
// all the controller needs (x = 1/error, y = fl(1/order), order in {2,3,5,6}).  Outside it the result is unspecified.
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif
#include "glibc_pow_tables.inc"

#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
#define NNHIP_GPOW_FN __host__ __device__ __forceinline__
#else
#define NNHIP_GPOW_FN inline
#endif

namespace nnhip_gpow {

// One packed table of 768 eight-byte words: rows 0..127 {invc, logc, logctail, 0} (32 B each), then rows 0..127
// {tail, sbits} (16 B each).  A lane's row index depends on its own x, so lookups are per-lane gathers.  Three homes:
//   host    plain array (tests/cpp/test_glibc_pow.cpp, exactness check against the live libm)
//   global  (default on the device) __device__ array, L1/L2-resident: one 16-byte + one 8-byte gather for the log row, one
//           16-byte gather for the exp row
//   LDS     a per-workgroup copy filled by the kernel prologue (lds_fill) and read with ds_read_b128 — A/B only
//           (-DNNHIP_GPOW_LDS): measured slower, see ode_device.hpp
constexpr int kTabWords = 4 * NNHIP_GPOW_N + 2 * NNHIP_GPOW_N;
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
static __device__ const uint64_t d_tab[kTabWords] __attribute__((aligned(16))) = {NNHIP_GPOW_LOG_TABLE, NNHIP_GPOW_EXP_TABLE};
static __shared__ uint64_t s_tab[kTabWords] __attribute__((aligned(16)));
#endif
#if !defined(__HIP_DEVICE_COMPILE__) && !defined(__HIPCC_RTC__)
static const uint64_t h_tab[kTabWords] __attribute__((aligned(16))) = {NNHIP_GPOW_LOG_TABLE, NNHIP_GPOW_EXP_TABLE};
#endif

NNHIP_GPOW_FN double as_f64(uint64_t u) { return __builtin_bit_cast(double, u); }
NNHIP_GPOW_FN uint64_t as_u64(double d) { return __builtin_bit_cast(uint64_t, d); }

struct U64x2 { uint64_t a, b; } __attribute__((aligned(16)));

struct TabHostOrGlobal {  // host: plain array; device: the __device__ array
  NNHIP_GPOW_FN static U64x2 pair(int word) {
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
    return *reinterpret_cast<const U64x2*>(&d_tab[word]);
#else
    return *reinterpret_cast<const U64x2*>(&h_tab[word]);
#endif
  }
  NNHIP_GPOW_FN static uint64_t word(int word) {
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC_RTC__)
    return d_tab[word];
#else
    return h_tab[word];
#endif
  }
};
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
struct TabLds {  // the workgroup's LDS copy; the kernel must have called lds_fill()
  __device__ __forceinline__ static U64x2 pair(int word) { return *reinterpret_cast<const U64x2*>(&s_tab[word]); }
  __device__ __forceinline__ static uint64_t word(int word) { return s_tab[word]; }
};
// Kernel prologue of every kernel that may evaluate the controller: all threads of the workgroup, before any early exit.
__device__ __forceinline__ void lds_fill() {
  for (int k = threadIdx.x; k < kTabWords; k += blockDim.x) s_tab[k] = d_tab[k];
  __syncthreads();
}
#endif

// pow(x, y) on the domain stated above.
template <class Tab>
NNHIP_GPOW_FN double pow_pos_t(double x, double y) {
  uint64_t ix = as_u64(x);
  if (!(x == x)) return x + y;                        // NaN propagates
  if (ix == 0) return 0.0;                            // pow(+0, y>0) = +0
  if (ix == 0x7ff0000000000000ULL) return x;          // pow(+inf, y>0) = +inf
  if (ix == 0x3ff0000000000000ULL) return 1.0;        // pow(1, y) = 1 (the library's main path gives the same)
  if ((ix >> 52) == 0) {                              // subnormal x: normalise (e_pow.c: ix = asuint64(x * 0x1p52) - 52<<52)
    ix = as_u64(x * 0x1p52);
    ix &= 0x7fffffffffffffffULL;
    ix -= 52ULL << 52;
  }
  // ---- log_inline: x = 2^k z, z in [OFF, 2 OFF), log(x) = k ln2 + log(c) + log1p(z/c - 1) as hi + lo ----
  const uint64_t OFF = 0x3fe6955500000000ULL;
  const uint64_t tmp = ix - OFF;
  const int i = (int)((tmp >> (52 - 7)) % NNHIP_GPOW_N);
  const int k = (int)((int64_t)tmp >> 52);
  const uint64_t iz = ix - (tmp & (0xfffULL << 52));
  const double z = as_f64(iz);
  const double kd = (double)k;
  const U64x2 row = Tab::pair(4 * i);
  const double invc = as_f64(row.a), logc = as_f64(row.b), logctail = as_f64(Tab::word(4 * i + 2));
  const double A0 = as_f64(NNHIP_GPOW_A0), A1 = as_f64(NNHIP_GPOW_A1), A2 = as_f64(NNHIP_GPOW_A2), A3 = as_f64(NNHIP_GPOW_A3),
               A4 = as_f64(NNHIP_GPOW_A4), A5 = as_f64(NNHIP_GPOW_A5), A6 = as_f64(NNHIP_GPOW_A6);
  const double t1 = __builtin_fma(kd, as_f64(NNHIP_GPOW_LN2HI), logc);      // fused in the FMA build
  const double r = __builtin_fma(z, invc, -1.0);                           // exact by construction of invc
  const double ar = r * A0;
  const double lo1 = __builtin_fma(kd, as_f64(NNHIP_GPOW_LN2LO), logctail);
  const double a12 = __builtin_fma(r, A2, A1);
  const double a34 = __builtin_fma(r, A4, A3);
  const double t2 = r + t1;
  const double ar2 = r * ar;
  const double ar3 = r * ar2;
  const double lo3 = __builtin_fma(ar, r, -ar2);
  const double lo2 = (t1 - t2) + r;
  const double a56 = __builtin_fma(r, A6, A5);
  const double hi = t2 + ar2;
  const double q = __builtin_fma(a56, ar2, a34);
  const double lo4 = (t2 - hi) + ar2;
  const double pp = __builtin_fma(ar2, q, a12);
  const double lo = __builtin_fma(ar3, pp, ((lo1 + lo2) + lo3) + lo4);
  const double lhi = hi + lo;
  const double llo = (hi - lhi) + lo;
  // ---- y * log(x) in double-double ----
  const double ehi = y * lhi;
  const double elo = __builtin_fma(y, llo, __builtin_fma(lhi, y, -ehi));
  // ---- exp_inline(ehi, elo) ----
  const uint32_t abstop = (uint32_t)(as_u64(ehi) >> 52) & 0x7ff;
  if (abstop - 0x3c9u >= 0x3fu) {
    // |ehi| < 2^-54: the library returns 1.0 + ehi.  (|ehi| >= 512 cannot happen: |log x| <= 745.2 and y <= 1/2 for every
    // order the controller uses; kept for y up to 1 by saturating the way the library's overflow/underflow paths do.)
    if (abstop < 0x3c9u) return 1.0 + ehi;
    if (abstop >= 0x409u) return (as_u64(ehi) >> 63) ? 0.0 : as_f64(0x7ff0000000000000ULL);
    // 512 <= |ehi| < 1024 (only reachable for y > 0.68): the library's special-case scaling is not restated
    return (as_u64(ehi) >> 63) ? 0.0 : as_f64(0x7ff0000000000000ULL);
  }
  const double Shift = as_f64(NNHIP_GPOW_SHIFT);
  double kd2 = __builtin_fma(ehi, as_f64(NNHIP_GPOW_INVLN2N), Shift);     // z + Shift, fused in the FMA build
  const uint64_t ki = as_u64(kd2);
  kd2 = kd2 - Shift;
  double rr = __builtin_fma(kd2, as_f64(NNHIP_GPOW_NEGLN2LON), __builtin_fma(kd2, as_f64(NNHIP_GPOW_NEGLN2HIN), ehi));
  rr = elo + rr;
  const int idx = 2 * (int)(ki % NNHIP_GPOW_N);
  const uint64_t top = ki << (52 - 7);
  const U64x2 erow = Tab::pair(4 * NNHIP_GPOW_N + idx);
  const double tail = as_f64(erow.a);
  const uint64_t sbits = erow.b + top;
  const double c23 = __builtin_fma(rr, as_f64(NNHIP_GPOW_C3), as_f64(NNHIP_GPOW_C2));
  const double tr = rr + tail;
  const double r2 = rr * rr;
  const double c45 = __builtin_fma(rr, as_f64(NNHIP_GPOW_C5), as_f64(NNHIP_GPOW_C4));
  const double u = __builtin_fma(c23, r2, tr);
  const double r4 = r2 * r2;
  const double tm = __builtin_fma(c45, r4, u);
  const double scale = as_f64(sbits);
  return __builtin_fma(tm, scale, scale);
}

// host + device entry reading the plain / global-memory table
NNHIP_GPOW_FN double pow_pos(double x, double y) { return pow_pos_t<TabHostOrGlobal>(x, y); }

}  // namespace nnhip_gpow

// ode_device.hpp — device-side building blocks of the batched ODE backend (gfx950 / CDNA4).
//
// What lives here: the compiled-in RHS library, the Butcher tableaux, the per-IVP steppers and the
// per-IVP time-loop driver, all as __device__ templates over (RHS, DIM) so that the state vector, the
// stage vectors k1..kS and (t, dt) of one IVP are VGPR-resident for a whole solve.  No MFMA anywhere:
// this is AXPY-shaped FP64 VALU work (the kernels built from these pieces are bounded by HBM when the
// state streams through memory once per step, and by the FP64 VALU when a whole solve is fused).
//
// Numerical contract: every floating-point expression below is written in the evaluation order of the
// reference (src/numericalnim/ode.nim; citations inline) and this translation unit is compiled with
// -ffp-contract=off, so +,-,*,/ and sqrt round exactly as the reference's C backend does on x86-64.
// pow(1/error, 1/order) in the step-size controller is the one libm call on the path: it is evaluated by
// glibc_pow.hpp, an operation-for-operation restatement of glibc's table-driven pow (bit-identical to the libm the
// reference links against), so adaptive solves follow the reference's step sequence exactly.
#pragma once
#if defined(NNHIP_CPU_EMU)
#include "hip_cpu_emu.hpp"  // tests/cpp: the device vocabulary on the host, lanes as threads (test infrastructure only)
#include "../../include/nnhip_ode.h"
#elif defined(__HIPCC_RTC__)
// hiprtc pre-includes its built-in HIP runtime header; an explicit <hip/hip_runtime.h> is not found by a stand-alone
// (non-PyTorch) process using /opt/rocm's hiprtc
#include "nnhip_ode.h"  // virtual header handed to hiprtc
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nnhip_ode.h"
#endif
#include "glibc_pow.hpp"  // pow(1/error, 1/order) with the C library's bits (the reference's libm)

// The kernels live in a build-variant namespace: `nnhip` for the default bit-parity build (-ffp-contract=off) and
// `nnhip_fast` for the opt-in FMA-contracted instantiations of the compute-bound fused kernels (Makefile: the same
// sources compiled with -ffp-contract=fast -DNNHIP_NS=nnhip_fast).  Plain argument structs shared by both live in nnhip_abi.
#ifndef NNHIP_NS
#define NNHIP_NS nnhip
#endif

namespace nnhip_abi {
constexpr int kMaxParams = 8;  // scalars that travel as kernel arguments (SGPRs); a run-time compiled right-hand side may declare more
struct Params {
  double p[kMaxParams];
  // NumContext beyond eight scalars (commonTypes.nim:4-27; run-time compiled right-hand sides, nnhip_ode_rhs_compile_ctx).  Device
  // pointers, all nullable; the compiled-in right-hand sides never look at them.
  const double* shared;  // [the scalars, when there are more than kMaxParams][shared vectors, concatenated]   (ctx.fValues / ctx.tValues)
  const double* ivp;     // per-IVP vectors [rows][stride]; inside a kernel already advanced to the IVP: entry j = ivp[j * stride]
  double* aux;           // per-IVP mutable slots [n_aux][stride] ("IT IS MUTABLE", ode.nim:599), advanced like ivp
  long long stride;      // IVPs of the bound batch
};
struct StepCtl {  // the option fields the steppers read (ode.nim:283-286)
  double absTol, relTol, dtMax, dtMin;
};
}  // namespace nnhip_abi

#ifndef NNHIP_PEEL_FUSED
#define NNHIP_PEEL_FUSED 0  // (A/B hook, see ode_kernels.hpp)
#endif

namespace NNHIP_NS {
using namespace nnhip_abi;

#define NNHIP_DEV __device__ __forceinline__
#define NNHIP_HD __host__ __device__ __forceinline__  // the tableau accessors: the host reads the same constexpr tables (nnhip_ode_tableau_f64)

// Nim's system.min/max on floats: `if x <= y: x else: y` / `if y <= x: x else: y` (NaN falls through
// to the second operand exactly as in the reference).
NNHIP_DEV double nmin(double x, double y) { return (x <= y) ? x : y; }
NNHIP_DEV double nmax(double x, double y) { return (y <= x) ? x : y; }

// ------------------------------------------------------------------------------------------------
// RHS library (enum nnhip_rhs_kind).  eval(): thread-per-IVP form, whole state in registers.
// comp(): lanes-per-system form, lane c computes component c reading the stage argument vector from
// LDS (`ys`, DIM doubles).  Both must produce bit-identical values.
// ------------------------------------------------------------------------------------------------
template <int DIM>
struct RhsNegY {  // dy = -y   (ode.nim:16-17)
  static constexpr int dim = DIM;
  NNHIP_DEV static void eval(double, const double (&y)[DIM], double (&dy)[DIM], const Params&) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) dy[c] = -y[c];
  }
  NNHIP_DEV static double comp(double, int c, const double* ys, const Params&) { return -ys[c]; }
};

template <int DIM>
struct RhsLinear {  // dy = p0 * y   (tests/test_ode.nim:5-7: -0.1 * y)
  static constexpr int dim = DIM;
  NNHIP_DEV static void eval(double, const double (&y)[DIM], double (&dy)[DIM], const Params& P) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) dy[c] = y[c] * P.p[0];
  }
  NNHIP_DEV static double comp(double, int c, const double* ys, const Params& P) { return ys[c] * P.p[0]; }
};

template <int DIM>
struct RhsAffineT {  // dy = p0*y + p1*t
  static constexpr int dim = DIM;
  NNHIP_DEV static void eval(double t, const double (&y)[DIM], double (&dy)[DIM], const Params& P) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) dy[c] = P.p[0] * y[c] + P.p[1] * t;
  }
  NNHIP_DEV static double comp(double t, int c, const double* ys, const Params& P) { return P.p[0] * ys[c] + P.p[1] * t; }
};

struct RhsLorenz {  // sigma=p0, rho=p1, beta=p2
  static constexpr int dim = 3;
  NNHIP_DEV static void eval(double, const double (&y)[3], double (&dy)[3], const Params& P) {
    dy[0] = P.p[0] * (y[1] - y[0]);
    dy[1] = y[0] * (P.p[1] - y[2]) - y[1];
    dy[2] = y[0] * y[1] - P.p[2] * y[2];
  }
  NNHIP_DEV static double comp(double, int c, const double* ys, const Params& P) {
    if (c == 0) return P.p[0] * (ys[1] - ys[0]);
    if (c == 1) return ys[0] * (P.p[1] - ys[2]) - ys[1];
    return ys[0] * ys[1] - P.p[2] * ys[2];
  }
};

struct RhsVanDerPol {  // mu = p0
  static constexpr int dim = 2;
  NNHIP_DEV static void eval(double, const double (&y)[2], double (&dy)[2], const Params& P) {
    dy[0] = y[1];
    dy[1] = P.p[0] * ((1.0 - y[0] * y[0]) * y[1]) - y[0];
  }
  NNHIP_DEV static double comp(double, int c, const double* ys, const Params& P) {
    if (c == 0) return ys[1];
    return P.p[0] * ((1.0 - ys[0] * ys[0]) * ys[1]) - ys[0];
  }
};

template <int DIM>
struct RhsRing {  // dy_c = -((c+1)/d)*y_c + p0*y_{(c+1) mod d}
  static constexpr int dim = DIM;
  NNHIP_DEV static void eval(double, const double (&y)[DIM], double (&dy)[DIM], const Params& P) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) dy[c] = -((double)(c + 1) / (double)DIM) * y[c] + P.p[0] * y[(c + 1) % DIM];
  }
  NNHIP_DEV static double comp(double, int c, const double* ys, const Params& P) {
    return -((double)(c + 1) / (double)DIM) * ys[c] + P.p[0] * ys[(c + 1) % DIM];
  }
  // Banded form (see LpsOps::rhs): component c reads components c .. c+1 (cyclic) only.  w[0] = y_c, w[1] = y_{(c+1) mod DIM}.
  static constexpr int halo_lo = 0, halo_hi = 1;
  NNHIP_DEV static double comp_window(double, int c, const double* w, const Params& P) {
    return -((double)(c + 1) / (double)DIM) * w[0] + P.p[0] * w[1];
  }
};

// ------------------------------------------------------------------------------------------------
// "Ops" policies: how one lane sees its IVP.  The steppers and the driver below are written once over
//   Ops::D          number of state components THIS LANE owns
//   ops.rhs(t,y,dy) dy = f(t,y), or g(t,y) = -f(-t,y) when NEG (backward branch, ode.nim:545)
//   ops.norm(...)   scaled RMS error norm of commonAdaptiveMethodCode (ode.nim:61-65)
// TpiOps: thread-per-IVP — the lane owns the whole state (D = dim), everything in VGPRs.
// LpsOps: lanes-per-system — DIM/CPL lanes of ONE wavefront own CPL components each (D = CPL).  The stage
//         argument vector is staged in LDS so any component's RHS can read any other component, and the
//         error norm is reduced across the DIM lanes through LDS in the reference's left-to-right order
//         (every lane of a system gets the bit-identical `error`, so the group stays in lock-step).
// ------------------------------------------------------------------------------------------------
// A right-hand side may declare fewer real components (`size`) than lanes are laid out for (`dim`, a power of two): systems
// of any length then run on the lanes-per-system kernels with the tail components switched off.
template <class R, class = void>
struct RhsSize { static constexpr int value = R::dim; };
template <class R>
struct RhsSize<R, decltype((void)R::size)> { static constexpr int value = R::size; };

// A right-hand side that writes to its context (run-time compiled ones with `aux` slots declare `mutates`): the drivers then make every
// evaluation of f the reference makes, where it makes it — including the ones whose VALUE is only needed if a requested time falls into a
// step (ode.nim:521, :530), which are otherwise evaluated lazily.
template <class R, class = void>
struct RhsMutates { static constexpr bool value = false; };
template <class R>
struct RhsMutates<R, decltype((void)R::mutates)> { static constexpr bool value = R::mutates; };

template <class RHS, bool NEG>
struct TpiOps {
  static constexpr int D = RHS::dim;
  static constexpr bool mutates = RhsMutates<RHS>::value;
  const Params& P;
  NNHIP_DEV static constexpr bool owns(int) { return true; }  // every component slot of the lane is a real component
  NNHIP_DEV void rhs(double t, const double (&y)[D], double (&dy)[D]) const {
    if constexpr (NEG) {
      RHS::eval(-t, y, dy, P);
#pragma unroll
      for (int c = 0; c < D; ++c) dy[c] = -dy[c];
    } else {
      RHS::eval(t, y, dy, P);
    }
  }
  NNHIP_DEV double norm(const double (&yNew)[D], const double (&err_y)[D], const StepCtl& o) const {
    double sum = 0.0;  // std/math sum: left to right from 0.0 (utils.nim:233-235)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      const double totalTol = fabs(yNew[c]) * o.relTol + o.absTol;  // absTol +. relTol * abs(yNew)  (:61)
      const double e = err_y[c] / totalTol;                        // :62
      sum = sum + e * e;                                           // :63, :65
    }
    return sqrt(1.0 / (double)D * sum);  // :65
  }
};

// Orders this wavefront's LDS traffic: DS operations of one wave execute in issue order, so all that is
// needed between a lane's ds_write and another lane's ds_read is that the compiler keeps program order.
template <int L = 64>  // L: the lanes that share the LDS region being synchronised (one system) — only the host-side test build needs to know
NNHIP_DEV void wave_lds_sync() {
#ifdef NNHIP_CPU_EMU
  hipemu::group_sync(L);  // tests/cpp/hip_cpu_emu.hpp: lanes are threads there, a rendezvous of the system's lanes stands in for the wavefront's lock-step
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// A right-hand side is "banded" when component c only reads components c-halo_lo .. c+halo_hi (cyclically; ring couplings,
// stencils of a method-of-lines discretisation).  A lane of the lanes-per-system kernels then needs, besides its own CPL
// components, only the first halo_hi components of the next lane and the last halo_lo of the previous one: they are exchanged
// with DPP / lane permutes between registers instead of staging the whole stage argument vector in LDS (one exposed LDS round
// trip per stage otherwise: 7 per Tsit54 attempt).  Same per-component expression (comp_window), hence the same bits.
template <class R, class = void>
struct RhsBanded { static constexpr bool value = false; };
template <class R>
struct RhsBanded<R, decltype((void)R::halo_hi)> { static constexpr bool value = true; };

template <class R, int CPL>
constexpr bool rhs_banded_applies() {
  if constexpr (RhsBanded<R>::value) return RhsSize<R>::value == R::dim && R::dim / CPL >= 2 && R::halo_lo <= CPL && R::halo_hi <= CPL;
  else return false;
}
// The DPP moves below have a source lane for every lane, so bound_ctrl changes no value — but it tells the compiler that the destination's previous
// content is dead: with bound_ctrl off and `old` = 0 it zeroed the destination first, two v_mov_b32 per exchanged double (24 of the 587 VALU
// instructions of a 16-component Tsit54 attempt; profiles/r04_dpp_bound_ctrl_ab.json).
#ifndef NNHIP_DPP_BOUND_CTRL
#define NNHIP_DPP_BOUND_CTRL 1
#endif
// value of `v` in the next (DIR = +1) / previous (DIR = -1) lane of this lane's group of L consecutive lanes, cyclically
template <int L, int DIR>
NNHIP_DEV double lane_rotate(double v) {
  static_assert(L >= 2 && L <= 64 && (L & (L - 1)) == 0, "group size must be a power of two");
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (L == 2) {         // quad_perm [1,0,3,2]
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
  } else if constexpr (L == 4) {  // quad_perm [1,2,3,0] / [3,0,1,2]
    constexpr int ctrl = DIR > 0 ? 0x39 : 0x93;
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
  } else if constexpr (L == 16) {  // row_ror: lane i <- lane (i - n) mod 16
    constexpr int ctrl = DIR > 0 ? 0x12F : 0x121;
    lo = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
    hi = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
  } else {                         // any other group size: ds_bpermute (no LDS memory involved)
    const int lane = (int)(threadIdx.x & 63);
    const int src = (lane & ~(L - 1)) | ((lane + (DIR > 0 ? 1 : L - 1)) & (L - 1));
    lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
  }
  return __hiloint2double(hi, lo);
}

// value of `v` in the LAST lane of this lane's group of L consecutive lanes
template <int L>
NNHIP_DEV double lane_last(double v) {
  static_assert(L >= 1 && L <= 64 && (L & (L - 1)) == 0, "group size must be a power of two");
  if constexpr (L == 1) return v;
  int lo = __double2loint(v), hi = __double2hiint(v);
  if constexpr (L == 2) {         // quad_perm [1,1,3,3]
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xF5, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xF5, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
  } else if constexpr (L == 4) {  // quad_perm [3,3,3,3]
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xFF, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xFF, 0xf, 0xf, NNHIP_DPP_BOUND_CTRL != 0);
  } else {                        // ds_bpermute (crossbar only, no LDS memory)
    const int src = ((int)(threadIdx.x & 63)) | (L - 1);
    lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
  }
  return __hiloint2double(hi, lo);
}

// The ordered register chain of LpsOps::norm costs L - 1 lane hops on top of the DIM dependent additions every variant has; it replaces an LDS
// write, two wave syncs and DIM LDS reads per lane.  Systems of up to NNHIP_LPS_CHAIN_MAX_L lanes use it (measured: profiles/r04_norm_chain_ab.json).
#ifndef NNHIP_LPS_CHAIN_MAX_L
#define NNHIP_LPS_CHAIN_MAX_L 4
#endif
template <int L>
constexpr bool lps_chain_norm() { return L <= NNHIP_LPS_CHAIN_MAX_L; }

template <class RHS, bool NEG, int CPL = 1, bool SHUFFLE_NORM = false>
struct LpsOps {
  static constexpr bool mutates = RhsMutates<RHS>::value;
  static constexpr int D = CPL;            // components per lane
  static constexpr int DIM = RHS::dim;     // component slots per system; DIM / CPL lanes of one wavefront share a system
  static constexpr int SIZE = RhsSize<RHS>::value;  // real components (<= DIM); slots SIZE..DIM-1 stay 0 and touch no memory
  const Params& P;
  double* ys;  // LDS, DIM doubles: stage argument vector of this lane's system
  double* es;  // LDS, DIM doubles: squared scaled error components
  int c0;      // first component owned by this lane (owns c0 .. c0+CPL-1)
  NNHIP_DEV bool owns(int j) const {
    if constexpr (SIZE == DIM) return true;
    else return c0 + j < SIZE;
  }
  NNHIP_DEV void rhs(double t, const double (&y)[CPL], double (&dy)[CPL]) const {
#ifndef NNHIP_NO_BANDED_RHS
    if constexpr (rhs_banded_applies<RHS, CPL>()) {  // (a halo wider than a lane's share, a padded system, one lane per system: the LDS path below)
      constexpr int LO = RHS::halo_lo, HI = RHS::halo_hi, L = DIM / CPL;
      double w[LO + CPL + HI];
#pragma unroll
      for (int j = 0; j < CPL; ++j) w[LO + j] = y[j];
#pragma unroll
      for (int h = 0; h < HI; ++h) w[LO + CPL + h] = lane_rotate<L, +1>(y[h]);            // next lane's first components
#pragma unroll
      for (int h = 0; h < LO; ++h) w[LO - 1 - h] = lane_rotate<L, -1>(y[CPL - 1 - h]);    // previous lane's last components
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const double v = RHS::comp_window(NEG ? -t : t, c0 + j, &w[j], P);  // window of component c0+j starts at w[j] (= y_{c-LO})
        dy[j] = NEG ? -v : v;
      }
      return;
    }
#endif
#pragma unroll
    for (int j = 0; j < CPL; ++j) ys[c0 + j] = y[j];
    wave_lds_sync<DIM / CPL>();
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const double v = owns(j) ? RHS::comp(NEG ? -t : t, c0 + j, ys, P) : 0.0;
      dy[j] = NEG ? -v : v;
    }
    wave_lds_sync<DIM / CPL>();  // the next stage overwrites ys
  }
  NNHIP_DEV double norm(const double (&yNew)[CPL], const double (&err_y)[CPL], const StepCtl& o) const {
    double e2[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const double totalTol = fabs(yNew[j]) * o.relTol + o.absTol;
      const double e = err_y[j] / totalTol;
      e2[j] = e * e;
    }
    if constexpr (SHUFFLE_NORM) {
      // A/B variant (tuning knob "lps_shuffle_norm"): butterfly all-reduce across the system's lanes with wavefront
      // shuffles (ds_bpermute / DPP), no LDS round trip.  The association order differs from the reference's
      // left-to-right sum, so `error` can differ in the last ulp (all lanes of a system still get identical bits).
      double part = 0.0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) part = part + (owns(j) ? e2[j] : 0.0);
#pragma unroll
      for (int off = 1; off < DIM / CPL; off <<= 1) part = part + __shfl_xor(part, off, 64);
      return sqrt(1.0 / (double)SIZE * part);
    } else if constexpr (lps_chain_norm<DIM / CPL>()) {
      // Ordered chain through registers: the reference's sum is sequential — ((0 + e_0) + e_1) + ... (utils.nim:233-235) — so it is passed
      // from lane to lane of the system: in round r every lane adds its CPL terms, left to right, to the running sum it receives from the
      // previous lane (DPP / lane permute), and lane r's result of round r is the exact prefix sum through its last component (lane r - 1
      // held the exact prefix after round r - 1; what the other lanes compute in that round is never read).  After L rounds the last lane
      // holds the total, which is broadcast.  Same association as the LDS loop below, hence the same bits; no LDS traffic, no wave sync:
      // with a banded right-hand side the kernel touches no LDS at all.  Switched-off tail components add +0.0 (exact: the terms are >= +0).
      constexpr int L = DIM / CPL;
      double acc = 0.0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) acc = acc + (owns(j) ? e2[j] : 0.0);  // round 0: exact in the system's first lane
      if constexpr (L > 1) {
#pragma unroll
        for (int r = 1; r < L; ++r) {
          double run = lane_rotate<L, -1>(acc);
#pragma unroll
          for (int j = 0; j < CPL; ++j) run = run + (owns(j) ? e2[j] : 0.0);
          acc = run;
        }
      }
      return sqrt(1.0 / (double)SIZE * lane_last<L>(acc));
    } else {
#pragma unroll
      for (int j = 0; j < CPL; ++j) es[c0 + j] = e2[j];
      wave_lds_sync<DIM / CPL>();
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < SIZE; ++j) sum = sum + es[j];  // same left-to-right order as utils.nim:233-235
      wave_lds_sync<DIM / CPL>();
      return sqrt(1.0 / (double)SIZE * sum);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Butcher tableaux of the embedded FSAL pairs.  A[s][j] = a_{s+2, j+1} (row of stage s+2).
// Zero entries are kept: the reference multiplies by them (ode.nim:263,277,299,302).
// ------------------------------------------------------------------------------------------------
template <int METHOD>
struct Tableau;

template <>
struct Tableau<NNHIP_DOPRI54> {  // ode.nim:240-282
  static constexpr int S = 7;
  static constexpr int ORDER = 5;
  static constexpr bool DIRECT_ERR = false;  // error_y = yNew - yLow (:303)
  static constexpr int NB = 6;               // terms in yNew
  static constexpr bool B_IS_LAST_ROW = true;  // b_i = a_7i literally (:269-274): yNew == the last stage argument
  NNHIP_HD static double c(int s) {
    constexpr double C[7] = {0.0, 1.0 / 5.0, 3.0 / 10.0, 4.0 / 5.0, 8.0 / 9.0, 1.0, 1.0};
    return C[s];
  }
  NNHIP_HD static double a(int s, int j) {
    constexpr double A[7][6] = {
        {0, 0, 0, 0, 0, 0},
        {1.0 / 5.0, 0, 0, 0, 0, 0},
        {3.0 / 40.0, 9.0 / 40.0, 0, 0, 0, 0},
        {44.0 / 45.0, -56.0 / 15.0, 32.0 / 9.0, 0, 0, 0},
        {19372.0 / 6561.0, -25360.0 / 2187.0, 64448.0 / 6561.0, -212.0 / 729.0, 0, 0},
        {9017.0 / 3168.0, -355.0 / 33.0, 46732.0 / 5247.0, 49.0 / 176.0, -5103.0 / 18656.0, 0},
        {35.0 / 384.0, 0.0, 500.0 / 1113.0, 125.0 / 192.0, -2187.0 / 6784.0, 11.0 / 84.0}};
    return A[s][j];
  }
  NNHIP_HD static double b(int j) { return a(6, j); }  // b_i = a_7i (:269-274)
  NNHIP_HD static double bhat(int j) {
    constexpr double B[7] = {5179.0 / 57600.0, 0.0, 7571.0 / 16695.0, 393.0 / 640.0, -92097.0 / 339200.0, 187.0 / 2100.0, 1.0 / 40.0};
    return B[j];
  }
};

template <>
struct Tableau<NNHIP_TSIT54> {  // ode.nim:310-352
  static constexpr int S = 7;
  static constexpr int ORDER = 5;
  static constexpr bool DIRECT_ERR = true;  // error_y = dt * (sum bHat_i k_i) (:372)
  static constexpr int NB = 6;
  static constexpr bool B_IS_LAST_ROW = true;  // :339-344
  NNHIP_HD static double c(int s) {
    constexpr double C[7] = {0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0};
    return C[s];
  }
  NNHIP_HD static double a(int s, int j) {
    constexpr double A[7][6] = {
        {0, 0, 0, 0, 0, 0},
        {0.161, 0, 0, 0, 0, 0},
        {-0.008480655492356989, 0.335480655492357, 0, 0, 0, 0},
        {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0, 0},
        {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0, 0},
        {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383, 0},
        {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774}};
    return A[s][j];
  }
  NNHIP_HD static double b(int j) { return a(6, j); }
  NNHIP_HD static double bhat(int j) {
    constexpr double B[7] = {-0.001780011052226, -0.000816434459657, 0.007880878010262, -0.144711007173263,
                             0.582357165452555,  -0.458082105929187, 1.0 / 66.0};
    return B[j];
  }
};

template <>
struct Tableau<NNHIP_VERN65> {  // ode.nim:380-443
  static constexpr int S = 9;
  static constexpr int ORDER = 6;
  static constexpr bool DIRECT_ERR = false;     // error_y = yNew - yLow (:466)
  static constexpr int NB = 8;
  static constexpr bool B_IS_LAST_ROW = false;  // b4..b7 are separate literals (:426-433), not a9j
  NNHIP_HD static double c(int s) {
    constexpr double C[9] = {0.0, 0.06, 0.09593333333333333, 0.1439, 0.4973, 0.9725, 0.9995, 1.0, 1.0};
    return C[s];
  }
  NNHIP_HD static double a(int s, int j) {
    constexpr double A[9][8] = {
        {0, 0, 0, 0, 0, 0, 0, 0},
        {0.06, 0, 0, 0, 0, 0, 0, 0},
        {0.019239962962962962, 0.07669337037037037, 0, 0, 0, 0, 0, 0},
        {0.035975, 0.0, 0.107925, 0, 0, 0, 0, 0},
        {1.3186834152331484, 0.0, -5.042058063628562, 4.220674648395414, 0, 0, 0, 0},
        {-41.87259166432751, 0.0, 159.43256216313748, -122.11921356501004, 5.531743066200053, 0, 0, 0},
        {-54.430156935316504, 0.0, 207.06725136501848, -158.61081378459, 6.991816585950242, -0.01859723106220323, 0, 0},
        {-54.66374178728198, 0.0, 207.95280625538936, -159.2889574744995, 7.018743740796944, -0.018338785905045722, -0.0005119484997882099, 0},
        {0.03438957868357036, 0.0, 0.0, 0.25826245556335037, 0.4209371189673537, 4.405396469669310, -176.48311902429865, 172.36413340141507}};
    return A[s][j];
  }
  NNHIP_HD static double b(int j) {
    constexpr double B[8] = {0.03438957868357036, 0.0, 0.0, 0.25826245556335034, 0.42093711896735372, 4.4053964696693102,
                             -176.48311902429866, 172.36413340141507};
    return B[j];
  }
  NNHIP_HD static double bhat(int j) {
    constexpr double B[9] = {0.04909967648382, 0.0, 0.0, 0.22511122295165, 0.46946822530296, 0.80657922499889, 0.0,
                             -0.60711948917780, 0.05686113944048};
    return B[j];
  }
};

// Which methods are adaptive / FSAL / their `order` float — the triple solveODE passes (ode.nim:608-649)
template <int METHOD> struct MethodTraits;
#define NNHIP_TRAITS(M, FSAL, ORDER, ADAPTIVE) \
  template <> struct MethodTraits<M> { static constexpr bool fsal = FSAL, adaptive = ADAPTIVE; static constexpr double order = ORDER; };
NNHIP_TRAITS(NNHIP_RK4, false, 4.0, false)      NNHIP_TRAITS(NNHIP_DOPRI54, true, 5.0, true)   NNHIP_TRAITS(NNHIP_TSIT54, true, 5.0, true)
NNHIP_TRAITS(NNHIP_VERN65, true, 6.0, true)     NNHIP_TRAITS(NNHIP_BS32, true, 3.0, true)      NNHIP_TRAITS(NNHIP_RK21, false, 2.0, true)
NNHIP_TRAITS(NNHIP_HEUN2, false, 2.0, false)    NNHIP_TRAITS(NNHIP_RALSTON2, false, 2.0, false) NNHIP_TRAITS(NNHIP_KUTTA3, false, 3.0, false)
NNHIP_TRAITS(NNHIP_HEUN3, false, 3.0, false)    NNHIP_TRAITS(NNHIP_RALSTON3, false, 3.0, false) NNHIP_TRAITS(NNHIP_SSPRK3, false, 3.0, false)
NNHIP_TRAITS(NNHIP_RALSTON4, false, 4.0, false) NNHIP_TRAITS(NNHIP_KUTTA4, false, 4.0, false)
#undef NNHIP_TRAITS
constexpr bool has_tableau(int m) { return m == NNHIP_DOPRI54 || m == NNHIP_TSIT54 || m == NNHIP_VERN65; }

// ------------------------------------------------------------------------------------------------
// Steppers.  Signature mirrors IntegratorProc (ode.nim:38):
//   in  (t, y, FSAL, dt)   out (yNew, FSAL', dt used, error);  returns status bits.
// ------------------------------------------------------------------------------------------------
constexpr int kStatusNaN = 1;

// dt-derived constants of one RK4 step.  `dt / 6.0` is an IEEE division (~11 VALU instructions on gfx950):
// it is computed once per distinct dt (on the host for the uniform-dt streaming kernel, cached in the fused
// driver) instead of once per step — the value is bit-identical wherever it is computed.
struct Rk4Dt {
  double dt, hdt, dt6;
};
NNHIP_DEV Rk4Dt rk4_dt(double dt) { return Rk4Dt{dt, 0.5 * dt, dt / 6.0}; }

template <class Ops>
NNHIP_DEV void rk4_step(const Ops& ops, double t, const Rk4Dt& h, const double (&y)[Ops::D], double (&yNew)[Ops::D]) {
  constexpr int D = Ops::D;  // ode.nim:180-189
  double k1[D], k2[D], k3[D], k4[D], ya[D];
  const double dt = h.dt;
  ops.rhs(t, y, k1);
  const double hdt = h.hdt;  // 0.5 * dt
#pragma unroll
  for (int c = 0; c < D; ++c) ya[c] = y[c] + hdt * k1[c];          // y + 0.5 * dt * k1
  ops.rhs(t + hdt, ya, k2);  // t + 0.5*dt
#pragma unroll
  for (int c = 0; c < D; ++c) ya[c] = y[c] + hdt * k2[c];
  ops.rhs(t + hdt, ya, k3);
#pragma unroll
  for (int c = 0; c < D; ++c) ya[c] = y[c] + dt * k3[c];
  ops.rhs(t + dt, ya, k4);
  const double dt6 = h.dt6;  // dt / 6.0
#pragma unroll
  for (int c = 0; c < D; ++c) yNew[c] = y[c] + dt6 * (k1[c] + 2.0 * (k2[c] + k3[c]) + k4[c]);  // :188
}

// The other eight fixed-step steppers (ode.nim:107-178), each in the reference's own expression order
// (Nim: `a * b * c` = (a*b)*c, `2/3` is the float 0.666.., `-1/3` is (-1)/3).
template <int METHOD, class Ops>
NNHIP_DEV void fixed_step(const Ops& ops, double t, double dt, const double (&y)[Ops::D], double (&yNew)[Ops::D]) {
  constexpr int D = Ops::D;
  double k1[D], k2[D], k3[D], k4[D], ya[D];
  ops.rhs(t, y, k1);
  if constexpr (METHOD == NNHIP_HEUN2) {  // :107-113
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + dt * k1[c];
    ops.rhs(t + dt, ya, k2);
    const double h = 0.5 * dt;
#pragma unroll
    for (int c = 0; c < D; ++c) yNew[c] = y[c] + h * (k1[c] + k2[c]);
  } else if constexpr (METHOD == NNHIP_RALSTON2) {  // :115-121
    const double h = 2.0 / 3.0 * dt;
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + h * k1[c];
    ops.rhs(t + h, ya, k2);
#pragma unroll
    for (int c = 0; c < D; ++c) yNew[c] = y[c] + dt * (0.25 * k1[c] + 0.75 * k2[c]);
  } else if constexpr (METHOD == NNHIP_KUTTA3) {  // :123-130
    const double h = 0.5 * dt, h2 = 2.0 * dt;
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + h * k1[c];
    ops.rhs(t + h, ya, k2);
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] - dt * k1[c] + h2 * k2[c];
    ops.rhs(t + dt, ya, k3);
#pragma unroll
    for (int c = 0; c < D; ++c) yNew[c] = y[c] + dt * (1.0 / 6.0 * k1[c] + 2.0 / 3.0 * k2[c] + 1.0 / 6.0 * k3[c]);
  } else if constexpr (METHOD == NNHIP_HEUN3) {  // :132-139
    const double h1 = 1.0 / 3.0 * dt, h2 = 2.0 / 3.0 * dt;
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + h1 * k1[c];
    ops.rhs(t + h1, ya, k2);
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + h2 * k2[c];
    ops.rhs(t + h2, ya, k3);
#pragma unroll
    for (int c = 0; c < D; ++c) yNew[c] = y[c] + dt * (0.25 * k1[c] + 0.75 * k3[c]);
  } else if constexpr (METHOD == NNHIP_RALSTON3) {  // :141-148
    const double h1 = 1.0 / 2.0 * dt, h2 = 3.0 / 4.0 * dt;
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + h1 * k1[c];
    ops.rhs(t + h1, ya, k2);
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + h2 * k2[c];
    ops.rhs(t + h2, ya, k3);
#pragma unroll
    for (int c = 0; c < D; ++c) yNew[c] = y[c] + dt * (2.0 / 9.0 * k1[c] + 1.0 / 3.0 * k2[c] + 4.0 / 9.0 * k3[c]);
  } else if constexpr (METHOD == NNHIP_SSPRK3) {  // :150-157
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + dt * k1[c];
    ops.rhs(t + dt, ya, k2);
    const double q = 0.25 * dt;
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + q * (k1[c] + k2[c]);
    ops.rhs(t + 0.5 * dt, ya, k3);
#pragma unroll
    for (int c = 0; c < D; ++c) yNew[c] = y[c] + dt * (1.0 / 6.0 * k1[c] + 1.0 / 6.0 * k2[c] + 2.0 / 3.0 * k3[c]);
  } else if constexpr (METHOD == NNHIP_RALSTON4) {  // :160-168
    const double h = 0.4 * dt;
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + h * k1[c];
    ops.rhs(t + h, ya, k2);
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + dt * (0.29697761 * k1[c] + 0.15875964 * k2[c]);
    ops.rhs(t + 0.45573725 * dt, ya, k3);
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + dt * (0.21810040 * k1[c] - 3.05096516 * k2[c] + 3.83286476 * k3[c]);
    ops.rhs(t + dt, ya, k4);
#pragma unroll
    for (int c = 0; c < D; ++c)
      yNew[c] = y[c] + dt * (0.17476028 * k1[c] - 0.55148066 * k2[c] + 1.20553560 * k3[c] + 0.17118478 * k4[c]);
  } else {  // NNHIP_KUTTA4 :170-178
    static_assert(METHOD == NNHIP_KUTTA4, "unknown fixed-step method");
    const double h1 = 1.0 / 3.0 * dt;
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + h1 * k1[c];
    ops.rhs(t + h1, ya, k2);
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + dt * (-1.0 / 3.0 * k1[c] + k2[c]);
    ops.rhs(t + 2.0 / 3.0 * dt, ya, k3);
#pragma unroll
    for (int c = 0; c < D; ++c) ya[c] = y[c] + dt * (k1[c] - k2[c] + k3[c]);
    ops.rhs(t + dt, ya, k4);
#pragma unroll
    for (int c = 0; c < D; ++c) yNew[c] = y[c] + dt * (1.0 / 8.0 * k1[c] + 3.0 / 8.0 * k2[c] + 3.0 / 8.0 * k3[c] + 1.0 / 8.0 * k4[c]);
  }
}

// x^(1/N) for the step-size controller's pow(1/error, 1/order) (ode.nim:71,537), N = order in {2,3,5,6}.
// ocml's general pow() costs 215 VALU instructions — a third of a whole DOPRI54 Lorenz step — so the root is taken
// directly: an fp32 v_log_f32/v_exp_f32 estimate (rel. error ~1e-6), one plain Newton step in fp64 (-> ~1e-11), then a
// final Newton correction whose residual x - r^N is evaluated in double-double arithmetic (explicit FMAs give the exact
// low parts of the products), so r + delta rounds to the correctly rounded root except in astronomically rare
// half-way cases — of x^fl(1/N), the function the reference actually evaluates.  glibc's pow (what the reference calls)
// is itself within 0.52 ulp, so the two agree bit for bit in all but a few percent of calls — which is what keeps accept/reject decisions and the loop's last-ulp end
// condition (`t < tEnd`) aligned with the reference.  ~65 VALU instructions.  Outside [1e-30, 1e30] the value only
// has to land on the right side of the controller's clamp min(4, max(0.125, 0.9*root)): 0 / 1e30 do; NaN propagates.
template <int N>
NNHIP_DEV double nth_root(double x) {
#ifdef NNHIP_USE_OCML_POW
  return pow(x, 1.0 / (double)N);
#else
  if (!(x == x)) return x;
  if (x < 1e-30) return 0.0;
  if (x > 1e30) return 1e30;
  const float lf = __builtin_amdgcn_logf((float)x);  // log2
  double r = (double)__builtin_amdgcn_exp2f(lf * (1.0f / (float)N));
  {  // Newton step 1, plain fp64
    double rn = r;
#pragma unroll
    for (int k = 1; k < N; ++k) rn = rn * r;
    const double d = x / rn - 1.0;
    r = r + r * (d * (1.0 / (double)N));
  }
  {  // Newton step 2 with a double-double residual: r^N = p + e to ~2^-100 relative
    double p = r, e = 0.0, pm1 = 1.0;  // pm1 = r^(N-1) (plain precision is enough for the slope)
#pragma unroll
    for (int k = 1; k < N; ++k) {
      pm1 = p;
      const double np_ = p * r;
      e = __builtin_fma(p, r, -np_) + e * r;
      p = np_;
    }
    const double resid = (x - p) - e;              // x - r^N; (x - p) is exact (Sterbenz: p is within a few ulp of x)
    // The reference raises to the DOUBLE 1/order, which is not 1/N: pow(x, fl(1/N)) = x^(1/N) * x^(fl(1/N) - 1/N)
    // = root * (1 + dN*ln x), dN = fl(1/N) - 1/N (3: -1.85e-17, 5: +1.11e-17, 6: -9.25e-18, 2: 0) — a shift of up to
    // 0.4 ulp that decides the rounding in a quarter of all calls.  ln x from the fp32 estimate is ample here.
    constexpr double dN = N == 3 ? -1.850371707708594e-17 : N == 5 ? 1.1102230246251566e-17 : N == 6 ? -9.25185853854297e-18 : 0.0;
    const double lnx = (double)lf * 0.6931471805599453;
    r = r + (resid / ((double)N * pm1) + r * (dN * lnx));  // both corrections are far below 1 ulp of r: one final rounding
  }
  return r;
#endif
}

// min(4, max(0.125, 0.9 * pow(1/error, 1/order)))  (:71,:537).  Default (bit-parity) build: glibc's pow, bit for bit
// (glibc_pow.hpp).  The opt-in FMA-contracted build (namespace nnhip_fast, -DNNHIP_FAST_ROOT) keeps the correctly
// rounded nth_root above: it is not bit-exact anyway and the root is ~20 VALU instructions shorter.
// pow's two table lookups are per-lane gathers from a 6 KiB __device__ array (L1/L2-resident).  A per-workgroup LDS copy
// (-DNNHIP_GPOW_LDS, filled by controller_prologue) was measured and rejected: the fill + barrier put one more dependent
// memory round trip in front of every workgroup (advance kernel, 1e7 Lorenz IVPs: 342 us vs 302 us per loop iteration;
// fused kernels unchanged; profiles/r02_pow_tables_ab.txt).
// Early-out on the clamp: min(4, max(0.125, 0.9*p)) with p = pow(1/error, 1/order) is exactly 4 once 0.9*p >= 4 and exactly 0.125
// once 0.9*p <= 0.125.  pow is within 1 ulp of the true value, so thresholds that leave a relative margin of 1e-4 (twelve orders
// of magnitude more than an ulp) decide the clamp without evaluating pow: error < kClampHi<ORDER> -> 4, error > kClampLo<ORDER> -> 0.125
// (also covers error = inf: pow(0, y) = 0).  With loose tolerances (the BASELINE configs' default options) almost every step
// lands on the upper clamp, so whole wavefronts skip the ~90-instruction pow and its two table gathers; NaN takes the pow path
// and propagates as before.  Bit-identical by construction; tests/test_gpu_adaptive_parity.py::test_controller_factor_is_libm_exact
// sweeps both thresholds densely.
template <int ORDER> struct ClampThresholds;
// (0.9/4)^ORDER * (1 - 1e-3)   and   (0.9/0.125)^ORDER * (1 + 1e-3)
template <> struct ClampThresholds<2> { static constexpr double hi = 0.050574375, lo = 51.89184; };
template <> struct ClampThresholds<3> { static constexpr double hi = 0.011379234375, lo = 373.621248; };
template <> struct ClampThresholds<5> { static constexpr double hi = 5.7607374023437e-4, lo = 19368.52; };
template <> struct ClampThresholds<6> { static constexpr double hi = 1.2961659155273e-4, lo = 139453.4; };

template <int ORDER>
NNHIP_DEV double shrink_factor(double error) {
  if (error < ClampThresholds<ORDER>::hi) return 4.0;   // (both builds: the root is within 1 ulp as well, the same margin decides the clamp)
  if (error > ClampThresholds<ORDER>::lo) return 0.125;
#if defined(NNHIP_FAST_ROOT)
  return nmin(4.0, nmax(0.125, 0.9 * nth_root<ORDER>(1.0 / error)));
#else
#if defined(NNHIP_GPOW_LDS)
  return nmin(4.0, nmax(0.125, 0.9 * nnhip_gpow::pow_pos_t<nnhip_gpow::TabLds>(1.0 / error, 1.0 / (double)ORDER)));
#else
  return nmin(4.0, nmax(0.125, 0.9 * nnhip_gpow::pow_pos_t<nnhip_gpow::TabHostOrGlobal>(1.0 / error, 1.0 / (double)ORDER)));
#endif
#endif
}
// Prologue of every kernel whose method has a step-size controller; a no-op unless the A/B build -DNNHIP_GPOW_LDS stages pow's
// tables in LDS.  All threads of the workgroup must reach it (it may contain a barrier) — before any `if (i >= N) return`.
template <bool ADAPTIVE = true>
NNHIP_DEV void controller_prologue() {
#if !defined(NNHIP_FAST_ROOT) && defined(NNHIP_GPOW_LDS)
  if constexpr (ADAPTIVE) nnhip_gpow::lds_fill();
#endif
}

// One adaptive IntegratorProc call = the method's stage block inside commonAdaptiveMethodCode's retry loop
// (ode.nim:57-76).  `fsal` is k1 on entry for the tableau methods and the returned FSAL slot on exit.
// `factor` returns min(4, max(0.125, 0.9*pow(1/error, 1/order))) of the ACCEPTED attempt's error: the post-step controller
// (ode.nim:537) evaluates exactly that expression on the error this call returns, so the one pow per attempt is evaluated at
// one place — here — for both the in-step shrink (:71) and the caller's post-step update (same operands, same bits).
template <int METHOD, bool PEEL = false, class Ops>
NNHIP_DEV int embedded_step(const Ops& ops, double t, double& dt, const double (&y)[Ops::D], const double (&fsal)[Ops::D], double (&fsalOut)[Ops::D],
                            double (&yNew)[Ops::D], double& error, const StepCtl& o, int64_t& rejected, double& factor) {
  constexpr int D = Ops::D;
  double ya[D], err_y[D], fsalNew[D];
  int limitCounter = 0;
  int status = 0;
  constexpr int ORDER = METHOD == NNHIP_RK21 ? 2 : METHOD == NNHIP_BS32 ? 3 : METHOD == NNHIP_VERN65 ? 6 : 5;
  // One attempt = the method's stage block + the error norm + the controller factor (the body of the `while limitCounter < 2` loop, :58-65).
  // PEEL (the step-streaming kernels: one call per launch): the FIRST attempt is peeled out of the retry loop.  Nearly every step is accepted
  // at once, and as straight-line code the attempt carries no loop-back register copies and keeps the loop-invariant constants of the rarely
  // taken pow path out of its way (hoisted out of the loop they were materialised once per call): 612 -> 587 VALU instructions per wave and
  // 82.2 -> 79.2 us per iteration on the streamed 16-component Tsit54 kernel.  The fused solves call this inside their time loop, where the
  // hoisting costs nothing and the second copy of the stage block does (Tsit54 16 components 5.61 -> 5.90 ms): they keep the plain loop.
  auto attempt = [&]() {
    if constexpr (has_tableau(METHOD)) {
      using T = Tableau<METHOD>;
      constexpr int S = T::S;
      double k[S][D];
#pragma unroll
      for (int c = 0; c < D; ++c) k[0][c] = fsal[c];  // k1 = FSAL (:293,:363,:454)
#pragma unroll
      for (int s = 1; s < S; ++s) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double acc = T::a(s, 0) * k[0][c];
#pragma unroll
          for (int j = 1; j < s; ++j) acc = acc + T::a(s, j) * k[j][c];
          ya[c] = y[c] + dt * acc;  // y + dt * (a_s1*k1 + ... )
        }
        ops.rhs(t + dt * T::c(s), ya, k[s]);
      }
      if constexpr (T::B_IS_LAST_ROW) {
        // yNew = y + dt*(b1*k1+...): identical expression to the last stage's argument (:299-301) -> ya holds it
#pragma unroll
        for (int c = 0; c < D; ++c) yNew[c] = ya[c];
      } else {
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double acc = T::b(0) * k[0][c];
#pragma unroll
          for (int j = 1; j < T::NB; ++j) acc = acc + T::b(j) * k[j][c];
          yNew[c] = y[c] + dt * acc;  // :464
        }
      }
#pragma unroll
      for (int c = 0; c < D; ++c) {
        double acc = T::bhat(0) * k[0][c];
#pragma unroll
        for (int j = 1; j < S; ++j) acc = acc + T::bhat(j) * k[j][c];
        if constexpr (T::DIRECT_ERR) {
          err_y[c] = dt * acc;  // :372
        } else {
          const double yLow = y[c] + dt * acc;  // :302,:465
          err_y[c] = yNew[c] - yLow;            // :303,:466
        }
        fsalNew[c] = k[S - 1][c];  // returns the last stage (:305,:374,:468)
      }
    } else if constexpr (METHOD == NNHIP_RK21) {  // :203-210
      double k1[D], k2[D];
      ops.rhs(t, y, k1);
#pragma unroll
      for (int c = 0; c < D; ++c) ya[c] = y[c] + dt * k1[c];
      ops.rhs(t + dt, ya, k2);
      const double h = dt * 0.5;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        yNew[c] = y[c] + h * (k1[c] + k2[c]);   // y + dt * 0.5 * (k1 + k2)
        const double yLow = y[c] + dt * k1[c];
        err_y[c] = yNew[c] - yLow;
        fsalNew[c] = yNew[c];                   // result = (yNew, yNew, dt, error)
      }
    } else {  // NNHIP_BS32 :224-234
      static_assert(METHOD == NNHIP_BS32, "unknown adaptive method");
      double k1[D], k2[D], k3[D], k4[D];
      ops.rhs(t, y, k1);
      const double h1 = 0.5 * dt, h2 = 0.75 * dt;
#pragma unroll
      for (int c = 0; c < D; ++c) ya[c] = y[c] + h1 * k1[c];
      ops.rhs(t + h1, ya, k2);
#pragma unroll
      for (int c = 0; c < D; ++c) ya[c] = y[c] + h2 * k2[c];
      ops.rhs(t + h2, ya, k3);
#pragma unroll
      for (int c = 0; c < D; ++c) yNew[c] = y[c] + dt * (2.0 / 9.0 * k1[c] + 1.0 / 3.0 * k2[c] + 4.0 / 9.0 * k3[c]);
      ops.rhs(t + dt, yNew, k4);
#pragma unroll
      for (int c = 0; c < D; ++c) {
        const double yLow = y[c] + dt * (7.0 / 24.0 * k1[c] + 1.0 / 4.0 * k2[c] + 1.0 / 3.0 * k3[c] + 1.0 / 8.0 * k4[c]);
        err_y[c] = yNew[c] - yLow;
        fsalNew[c] = k4[c];
      }
    }
    error = ops.norm(yNew, err_y, o);  // scaled RMS norm (:61-65)
    factor = shrink_factor<ORDER>(error);                      // the attempt's one pow (:71 if rejected, :537 if accepted)
  };
  if constexpr (PEEL) {
  attempt();
  if (!(error <= 1.0)) {                                       // :69-70 (NaN goes on to the abort below)
    if (error != error) status |= kStatusNaN;                  // deviation: the reference would spin forever
    else {
      for (;;) {
        dt = dt * factor;                                          // :71
        if (fabs(dt) < o.dtMin) { dt = o.dtMin; limitCounter += 1; }  // :72-74
        else if (o.dtMax < fabs(dt)) { dt = o.dtMax; }             // :75-76
        rejected += 1;
        if (!(limitCounter < 2)) break;                            // :58
        attempt();
        if (error <= 1.0) break;                                   // :69-70
        if (error != error) { status |= kStatusNaN; break; }
      }
    }
  }
  } else {
  while (limitCounter < 2) {  // :58
    attempt();
    if (error <= 1.0) break;                                   // :69-70
    if (error != error) { status |= kStatusNaN; break; }       // deviation: the reference would spin forever
    dt = dt * factor;                                          // :71
    if (fabs(dt) < o.dtMin) { dt = o.dtMin; limitCounter += 1; }  // :72-74
    else if (o.dtMax < fabs(dt)) { dt = o.dtMax; }             // :75-76
    rejected += 1;
  }
  }
#pragma unroll
  for (int c = 0; c < D; ++c) fsalOut[c] = fsalNew[c];  // (may be the array `fsal` itself: every read of it is done)
  return status;
}

// A value that is the same in every lane, moved to scalar registers (loop bounds loaded after a vector store come back in VGPRs: the scalar
// cache is not coherent with them, so the compiler may not use s_load — and would then compare per lane).
NNHIP_DEV int64_t uniform_i64(int64_t v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v & 0xffffffffu));
  const int hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}

// hermiteSpline per component (utils.nim:273-279)
struct HermiteW {
  double h00, h10w, h01, h11w;  // h10*(x2-x1), h11*(x2-x1) pre-multiplied exactly as the expression does
};
NNHIP_HD HermiteW hermite_weights(double x, double x1, double x2) {  // (host too: the fixed-step emission schedule is replayed there)
  const double t = (x - x1) / (x2 - x1);
  const double omt = 1.0 - t;
  HermiteW w;
  w.h00 = (1.0 + 2.0 * t) * (omt * omt);
  const double h10 = t * (omt * omt);
  w.h01 = (t * t) * (3.0 - 2.0 * t);
  const double h11 = (t * t * t) - (t * t);
  w.h10w = h10 * (x2 - x1);
  w.h11w = h11 * (x2 - x1);
  return w;
}
NNHIP_DEV double hermite_apply(const HermiteW& w, double y1, double y2, double dy1, double dy2) {
  return w.h00 * y1 + w.h10w * dy1 + w.h01 * y2 + w.h11w * dy2;  // left-assoc sum (:279)
}

// ------------------------------------------------------------------------------------------------
// Per-IVP driver = one direction of ODESolver's time loop (ode.nim:508-542 forward, :544-584 backward).
// `emit(k, yv)` stores the k-th emitted state of this direction.
// ------------------------------------------------------------------------------------------------
struct DriveIn {
  double tStartEff;  // t0 (forward) or -t0 (backward)
  double tEnd;       // max(tPositive) or -min(tNegative)
  double dtInit;
  const double* tReq;  // tPositive, or tNegative (reversed order as the reference stores it; negated on read when NEG)
  int nReq;
  int useDense;
  int64_t maxSteps;
  StepCtl ctl;
  // Fixed-step, no dense output: (t, dt) do not depend on the state, so the host replays the reference's time loop
  // (dt = min(dt, tEnd - t); t += dt, ode.nim:525,532) once and hands over the resulting step schedule:
  // `uniformFull` steps of dtInit, then `nTail` clipped / ulp-sized closing steps.  uniformFull < 0: not available.
  int64_t uniformFull;
  int nTail;
  double tailDt[4];
  // ... and, with dense output, the emission block of that replay (ode.nim:512-524): requested row k < nEmit is interpolated at the START of
  // step emitStep[k] (non-decreasing) between the states before and after step emitStep[k] - 1, with the Hermite weights emitW[4k .. 4k+3]
  // = (h00, h10*(x2-x1), h01, h11*(x2-x1)) of utils.nim:273-279 — times, step indices and weights are the same for every IVP of the batch.
  const double* emitW;
  const int64_t* emitStep;
  int nEmit;
};
struct DriveOut {
  int emitted;
  int status;
  int64_t steps, rejected;
  double tFinal;  // loop variable t when the direction ended (= tEnd unless truncated by maxSteps / aborted)
  double dtFinal; // the loop's dt at that point (after the controller's update): with tFinal and the state, where a truncated integration resumes
};

// DENSE = false is the lean instantiation for tspan.len == 2 (in.useDense must be 0): the Hermite history lastIter = (t, y, dy)
// (ode.nim:498,526-530) is never read then, and not carrying it frees 2-3 state vectors of registers — the difference between two
// and three waves per SIMD for the 16-component lanes-per-system kernels.  DENSE = true handles both cases.
template <int METHOD, bool NEG, bool DENSE = true, class Ops, class Emit>
NNHIP_DEV void drive(const Ops& ops, const DriveIn& in, const double (&y0)[Ops::D], Emit&& emit, DriveOut& out) {
  constexpr int D = Ops::D;
  constexpr int DH = DENSE ? D : 1;  // size of the history vectors (1 = unused placeholder)
  using MT = MethodTraits<METHOD>;
  double t = in.tStartEff;
  double y[D], fsal[D], yNew[D];
  [[maybe_unused]] double lastDy[DH], dyNow[DH];  // slopes at the two ends of a step whose end is interpolated (methods without FSAL evaluate them there)
#pragma unroll
  for (int c = 0; c < D; ++c) y[c] = y0[c];
  // The reference evaluates f(t0, y, ctx) twice before the forward loop — lastIter.dy (:498) and FSAL (:506) — and g(-t0, y0) once before
  // the backward one (:546).  Same value; the repetition only matters to a right-hand side that mutates its ctx (aux slots), and is
  // removed by the compiler for every other one.
  if constexpr (!NEG) ops.rhs(t, y, fsal);
  ops.rhs(t, y, fsal);  // FSAL = f(t0, y) (:506) / g(-t0, y0) (:546)
  if constexpr (DENSE) {
#pragma unroll
    for (int c = 0; c < D; ++c) lastDy[c] = fsal[c];  // lastIter.dy (:498,:548)
  }
  double dt = in.dtInit;
  [[maybe_unused]] Rk4Dt h4 = rk4_dt(dt);
  double error = 0.0;
  [[maybe_unused]] double factor = 1.0;
  int denseIndex = 0;
  const int high = in.nReq - 1;
  int status = 0;
  int64_t steps = 0, rejected = 0;
  if constexpr (!MT::adaptive) {
    if (in.uniformFull >= 0 && (!DENSE || !in.useDense || !Ops::mutates)) {
      // Pre-computed schedule: the loop carries no FP64 bookkeeping besides t itself.  With dense output the emission block is scheduled
      // too: scalar comparisons of the step counter decide when a row is due; the state before the step is kept only across steps whose
      // end is interpolated, and lastIter.dy / f(t, y) (:521, :530) are evaluated only there — the same calls on the same arguments as the
      // general loop below makes, hence the same bits.  (A right-hand side that mutates its ctx takes the general loop: it must see every
      // evaluation the reference makes, where it makes it.)
      [[maybe_unused]] int ek = 0;
      [[maybe_unused]] int64_t nextEmit = -1;
      [[maybe_unused]] double yPrev[DH], tPrev = t;
      const int64_t total = in.uniformFull + in.nTail;
      if constexpr (DENSE) {
        if (in.useDense && in.nEmit > 0) nextEmit = uniform_i64(in.emitStep[0]);
      }
      // The steps run in segments that end where something other than a plain step is due, so that the step loops themselves are the lean
      // ones (every VALU instruction added to a scalar RK4 step costs 6 % of the solve): first up to the step BEFORE the next scheduled
      // row (`stop` = its index), whose start is kept; then that one step; then the rows due at its end.  One copy of the loops serves all.
      int64_t n = 0, stop = nextEmit >= 0 ? nextEmit - 1 : total;
      [[maybe_unused]] bool startKept = false;
      for (;;) {
        // (trip counts compared with != : the scalar ALU has no ordered 64-bit compare, `n < stop` would be a VALU instruction per step)
        const int64_t stopU = stop < in.uniformFull ? stop : in.uniformFull;
        if (n < stopU) {
          for (int64_t k = stopU - n; k != 0; --k) {
            if constexpr (METHOD == NNHIP_RK4) rk4_step(ops, t, h4, y, yNew);
            else fixed_step<METHOD>(ops, t, dt, y, yNew);
#pragma unroll
            for (int c = 0; c < D; ++c) y[c] = yNew[c];
            t += dt;
          }
          n = stopU;
        }
        if (n < stop) {
          for (int k = (int)(n - in.uniformFull), kEnd = (int)(stop - in.uniformFull); k != kEnd; ++k) {
            const double dk = k == 0 ? in.tailDt[0] : k == 1 ? in.tailDt[1] : k == 2 ? in.tailDt[2] : in.tailDt[3];  // (selects: indexing by k put the whole DriveIn in scratch, 136 B per lane)
            if constexpr (METHOD == NNHIP_RK4) rk4_step(ops, t, rk4_dt(dk), y, yNew);
            else fixed_step<METHOD>(ops, t, dk, y, yNew);
#pragma unroll
            for (int c = 0; c < D; ++c) y[c] = yNew[c];
            t += dk;
          }
          n = stop;
        }
        if (nextEmit < 0) break;  // n == total
        if constexpr (DENSE) {
          if (!startKept) {  // n == nextEmit - 1: the step about to be taken ends where a row is due
            tPrev = t;
#pragma unroll
            for (int c = 0; c < D; ++c) yPrev[c] = y[c];
            startKept = true;
            stop = nextEmit;
            continue;
          }
          ops.rhs(tPrev, yPrev, lastDy);  // lastIter.dy (:530)
          ops.rhs(t, y, dyNow);           // f(t, y, ctx) (:521)
          do {
            HermiteW w;
            w.h00 = in.emitW[4 * ek + 0]; w.h10w = in.emitW[4 * ek + 1]; w.h01 = in.emitW[4 * ek + 2]; w.h11w = in.emitW[4 * ek + 3];
            double yv[D];
#pragma unroll
            for (int c = 0; c < D; ++c) yv[c] = hermite_apply(w, yPrev[c], y[c], lastDy[c], dyNow[c]);
            emit(ek, yv);
            ek += 1;
            nextEmit = ek < in.nEmit ? uniform_i64(in.emitStep[ek]) : -1;
          } while (nextEmit == n);
          startKept = false;
          stop = nextEmit >= 0 ? nextEmit - 1 : total;
        }
      }
      emit(ek, y);  // yPositive.add(y) / yNegative.add(y) (:542,:584), after whatever was emitted
      out.emitted = ek + 1;
      if constexpr (DENSE) out.status = (in.useDense ? (in.maxSteps > 0 && total >= in.maxSteps) : (in.maxSteps > 0 && total >= in.maxSteps && t < in.tEnd)) ? 2 : 0;
      else out.status = (in.maxSteps > 0 && total >= in.maxSteps && t < in.tEnd) ? 2 : 0;
      out.steps = total;
      out.rejected = 0;
      out.tFinal = t;
      out.dtFinal = dt;
      return;
    }
  }
  // next requested time, kept in a register and re-read only after an emission (a per-step global load otherwise)
  [[maybe_unused]] double treq = (DENSE && in.useDense && in.nReq > 0) ? (NEG ? -in.tReq[0] : in.tReq[0]) : 0.0;
  // The emission block at the head of ODESolver's loop (:512-524) interpolates between lastIter = (t, y, dy) at the START of the step just taken and
  // the state at its end.  It is evaluated right AFTER that step (= at the head of the next iteration: nothing happens in between), where the
  // step's own inputs — y and k1 = FSAL — are still in registers and ARE lastIter: the dense instantiation keeps no copy of the history across the
  // step (two state vectors of VGPRs less at the step's peak: the adaptive dense solves ran 10-19 % behind the lean ones, now the same occupancy).
  // Same operations on the same values as at the reference's place, hence the same bits.  Returns true when every requested row has been emitted
  // (`if tPositive.high < denseIndex: break`, :513-514).
  auto emit_due = [&](double tA, const double (&yA)[D], const double (&dyA)[D], const double (&yB)[D], const double (&dyB)[D]) -> bool {
    if constexpr (DENSE) {
      if (!in.useDense) return false;
      if (high < denseIndex) return true;  // :513-514
      if (treq <= t) {
        if constexpr (!MT::fsal && !Ops::mutates) {
          ops.rhs(t, yB, dyNow);    // f(t, y, ctx) per emitted point (:521); same value each time
          ops.rhs(tA, yA, lastDy);  // lastIter.dy = f(t, y, ctx) of the step's start (:530): only ever read here, so evaluated here — the same call
        }
        while (treq <= t) {  // :515
          if constexpr (!MT::fsal && Ops::mutates) ops.rhs(t, yB, dyNow);  // a mutating f: once per emitted point, as the reference calls it (:521)
          const HermiteW w = hermite_weights(treq, tA, t);
          double yv[D];
#pragma unroll
          for (int c = 0; c < D; ++c) yv[c] = hermite_apply(w, yA[c], yB[c], MT::fsal ? dyA[c] : lastDy[c], MT::fsal ? dyB[c] : dyNow[c]);
          emit(denseIndex, yv);
          denseIndex += 1;
          if (high < denseIndex) break;  // :523-524 (the inner loop only: the iteration goes on, the next head check ends the loop)
          treq = NEG ? -in.tReq[denseIndex] : in.tReq[denseIndex];
        }
      }
    }
    return false;
  };
  [[maybe_unused]] double fsalNew[D];
  bool allEmitted = (t < in.tEnd) ? emit_due(t, y, fsal, y, fsal) : false;  // head of the first iteration: lastIter is the initial state (:498)
  while (!allEmitted && t < in.tEnd) {  // :511
    dt = nmin(dt, in.tEnd - t);  // :525
    if constexpr (DENSE && !MT::fsal && Ops::mutates) {
      if (in.useDense) ops.rhs(t, y, lastDy);  // lastIter.dy = f(t, y, ctx) (:530): a mutating f is called once per step, here, as the reference calls it
    }
    const double tOld = t;
    if constexpr (METHOD == NNHIP_RK4) {
      if (dt != h4.dt) h4 = rk4_dt(dt);  // only the clipped last step changes dt
      rk4_step(ops, t, h4, y, yNew);     // :531
      error = 0.0;
    } else if constexpr (!MT::adaptive) {
      fixed_step<METHOD>(ops, t, dt, y, yNew);
      error = 0.0;
    } else {
      status |= embedded_step<METHOD, NNHIP_PEEL_FUSED != 0>(ops, t, dt, y, fsal, fsalNew, yNew, error, in.ctl, rejected, factor);
    }
    t += dt;  // :532
    steps += 1;
    if constexpr (MT::adaptive) {  // :533-541
      if (error == 0.0) dt *= 5.0;
      else dt = dt * factor;  // = shrink_factor<order>(error), evaluated inside embedded_step
      if (dt < in.ctl.dtMin) dt = in.ctl.dtMin;
      else if (in.ctl.dtMax < dt) dt = in.ctl.dtMax;
    }
    const bool cut = in.maxSteps > 0 && steps >= in.maxSteps;
    if constexpr (DENSE) {
      if (!status && !cut && t < in.tEnd) {  // the loop goes on: head of the next iteration
        if constexpr (MT::adaptive) allEmitted = emit_due(tOld, y, fsal, yNew, fsalNew);
        else allEmitted = emit_due(tOld, y, fsal, yNew, fsal);
      }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) y[c] = yNew[c];
    if constexpr (MT::adaptive) {
#pragma unroll
      for (int c = 0; c < D; ++c) fsal[c] = fsalNew[c];
    }
    if (status) break;
    if (cut) { status |= 2; break; }
  }
  // yPositive.add(y) / yNegative.add(y) (:542,:584): appended after whatever was emitted so far
  // (denseIndex stays 0 when tspan.len == 2, so a non-dense solve returns exactly this one row).
  emit(denseIndex, y);
  out.emitted = denseIndex + 1;
  out.status = status;
  out.steps = steps;
  out.rejected = rejected;
  out.tFinal = t;
  out.dtFinal = dt;
}

}  // namespace NNHIP_NS

// ode_kernels.hpp — __global__ kernels built from ode_device.hpp, and their launch records.
//
// Kernel families (all FP64 VALU, no MFMA):
//   solve_tpi_kernel   thread-per-IVP fused solve: a whole ODESolver run (ode.nim:471-586) per thread,
//                      HBM touched only for y0 in and the requested rows out        -> FP64-VALU bound
//   step_tpi_kernel    thread-per-IVP single IntegratorProc call (ode.nim:38,531), state streams
//                      HBM -> VGPR -> HBM once per step                              -> HBM bound
//   solve_lps_kernel / step_lps_kernel  the same two for larger systems: DIM lanes of one wavefront per
//                      system, stage argument vector + error components staged in LDS
//   rk4_stream_vec_kernel  the scalar RK4 instance of the above with 16-byte lane accesses and several
//                      independent loads in flight per lane (the BASELINE.json headline kernel;
//                      algorithmic traffic 16 B per trajectory-step)                 -> HBM bound
#pragma once
#if defined(__HIPCC_RTC__) || defined(NNHIP_CPU_EMU)  // compiled at run time by hiprtc for a user-supplied RHS (ode_rtc.hip): device code only.
#define NNHIP_RTC 1                                    // (NNHIP_CPU_EMU: tests/cpp/hip_cpu_emu.hpp — the kernel BODIES compiled for the host by the
#else                                                  // test suite, lanes as threads; test infrastructure, never part of the library)
#define NNHIP_RTC 0
#include <tuple>
#include <type_traits>
#include <utility>
#endif

#include "ode_device.hpp"

// A/B hooks (scripts/build_variant.sh): peel the first attempt of an adaptive step out of the retry loop (see embedded_step)
#ifndef NNHIP_PEEL_STREAM
#define NNHIP_PEEL_STREAM 1  // step-streaming kernels
#endif
#ifndef NNHIP_PEEL_FUSED
#define NNHIP_PEEL_FUSED 0   // fused solves
#endif

namespace nnhip_abi {

struct SolveArgs {
  const double* y0;
  double* y_out;
  int32_t* ny_out;        // nullable
  int64_t* steps_out;     // nullable
  int64_t* rejected_out;  // nullable
  unsigned long long* agg;  // nullable: kAggSlots x {[0] steps_total [1] rejected_total [2] steps_max [3] ny_min [4] nan_aborts [5] truncated, 2 pad}
  int64_t N;
  int64_t ivpStride, compStride;  // element (i, c) of a state array lives at i*ivpStride + c*compStride
  int64_t rowStride;              // distance between consecutive output rows (= dim*N)
  int n_t;
  const double* tPos;  // device, ascending requested times > t0
  const double* tNeg;  // device, requested times < t0 in DESCENDING order (tNegative as the reference holds it)
  int nPos, nNeg, nZero;
  double t0, tEndPos, tEndNeg, dtInit;
  int useDense;
  int64_t maxSteps;
  StepCtl ctl;
  Params P;
  // per-IVP RHS parameters (parameter sweeps: every IVP of the batch is its own solveODE call with its own ctx.fValues):
  // nullable device array [nPerIvp][N]; parameter k of IVP i = perIvpParams[k*N + i] and overrides P.p[k]
  const double* perIvpParams;
  int nPerIvp;
  int64_t perIvpStride;  // = N of the full batch (a sub-range launch shrinks N but keeps addressing the full table)
  // host-replayed step schedule of the two directions ([0] forward, [1] backward); see DriveIn
  int64_t uniformFull[2];
  int nTail[2];
  double tailDt[2][4];
  // ... and the emission block of that replay when there is dense output (device arrays in the workspace; see DriveIn)
  const double* emitW[2];
  const int64_t* emitStep[2];
  int nEmit[2];
  // Order of integration (divergence binning): nullable device array [N]; work item k integrates IVP perm[k].  Every per-IVP
  // array (y0, y_out, counters, per-IVP parameters) is still addressed by the IVP's own index, so results land in the caller's
  // order without an un-permute pass; neighbouring lanes then hold IVPs with similar step sequences.
  const uint32_t* perm;
  // nullable [N]: how far the IVP got, |t - t0| summed over the directions integrated (= the full span unless max_steps cut it short)
  double* progress_out;
  // Resuming a truncated forward integration (the automatic divergence binning continues from its probe instead of restarting): nullable [N] —
  // the loop's t and dt when the forward branch ended.  With the state (last row) they are everything the loop carries for the methods whose
  // FSAL is f(t, y) of that state (DOPRI54, Tsit54, BS32; RK21 has none): perCall.tStart / dtInit / resume take them back in.  NaN-aborted IVPs
  // report tEnd (nothing to resume).  accumulate: the per-IVP counters are added to what steps_out / rejected_out hold.
  double* tfinal_out;
  double* dtfinal_out;
  int accumulate;
  // Every IVP its own solveODE call (each reference call owns its tspan AND its ODEoptions, ode.nim:589-591, 476-480): device
  // arrays [N], all nullable except tEnd (tEnd == nullptr: feature off).  tspan_i = [tStart_i, tEnd_i]; n_t == 2.  tEnd > tStart
  // integrates forward (rows y0, y(tEnd)), tEnd < tStart backward (rows y(tEnd), y0), tEnd == tStart yields the reference's single
  // row y0.  Option values go through abs() as in newODEoptions (:101-102); an IVP whose options newODEoptions would reject
  // (dtMax < dtMin) or that could never finish (fixed-step dt == 0; dtMin == 0 without max_steps) gets ny = -1 and NaN rows.
  // tGrid != nullptr (instead of tEnd): every IVP its own n_t-point tspan (MODE 3).  Row i of tGrid [N][n_t], prepared on the device by
  // prepare_tspans (ode_sort.hip): the requested times < tStart_i in DESCENDING order first (tNegative as the reference holds it),
  // those > tStart_i in ascending order last (tPositive); tCounts [N][3] = (nNeg, nZero, nPos), nNeg < 0: non-finite tspan (ny = -1).
  struct PerCall {
    const double *tEnd, *tStart, *absTol, *relTol, *dtMax, *dtMin, *dt;
    const double* tGrid;
    const int32_t* tCounts;
    const double* dtInit;  // nullable [N] (MODE 2, adaptive): the first step size of IVP i instead of sqrt(dtMax * dtMin) — a resumed integration's dt
    int resume;            // MODE 2: tStart_i is where a truncated forward integration stopped: tEnd_i <= tStart_i means "already there" (rows y, y; ny = 2), never backward
  } perCall;
};

struct StepArgs {
  int64_t N;
  int64_t ivpStride, compStride;
  const double* t_dev;  // nullable -> t_uniform
  double t_uniform;
  const double* dt_dev;  // nullable -> dt_uniform
  double dt_uniform;
  const double* y_in;
  const double* fsal_in;
  double* y_out;
  double* fsal_out;
  double* dt_used;  // nullable
  double* error;    // nullable
  StepCtl ctl;
  Params P;
  // advance mode (adaptive streaming driver): one iteration of ODESolver's loop body per launch (ode.nim:525-541)
  double tEnd;           // IVPs with t >= tEnd are finished and touch no memory
  double* t_io;          // per-IVP time, read and updated in place
  double* dt_io;         // per-IVP step size, read and updated in place; nullptr: t_io holds (t, dt) of IVP i side by side ([N][2], 16-byte
                         // aligned) — one 16-byte access each way instead of two of 8: the streaming drivers' own workspace is laid out so
  unsigned int* active;  // nullable; kAggSlots flags, set when a workgroup still has IVPs short of tEnd after this launch
  int64_t* steps_io;     // nullable: per-IVP accepted-step counter
  const double* perIvpParams;  // nullable [nPerIvp][N], as in SolveArgs
  int nPerIvp;
  int64_t perIvpStride;
  int stepsPerLaunch;  // advance mode: loop iterations per IVP and launch with the state kept in registers in between (<= 1: one, the
                       // IntegratorProc seam proper; K > 1 moves 8*(4d+5)/K bytes per attempted step — a different traffic model, reported apart)
  int nontemporal;  // advance mode: non-temporal hint on the streamed state arrays (working set beyond the Infinity Cache)
  int recomputeFsal;  // advance mode, DOPRI54 / Tsit54: FSAL is not carried through HBM but re-evaluated as f(t, y) at the start of the launch
                      // (see adv_fsal_in_hbm): 16*d bytes per step less for one more evaluation of f — the same bits
  int noLean;         // advance mode: 1 keeps the general kernels where the lean ones (advance_*_lean_kernel: the streaming driver's own layout as the
                      // kernel's contract) would apply — A/B and parity tests, tuning knob "adv_lean"
  // advance mode WITH dense output (adaptive streaming through the IntegratorProc seam, ode.nim:512-530): tReq == nullptr -> none.
  const double* tReq;   // requested times of this direction as the reference holds them (tPositive ascending / tNegative descending)
  int nReq;
  int useDense;         // tspan.len != 2 (:499-502); 0: no emission block, the only row is the final yPositive.add(y) / yNegative.add(y)
  int negate;           // backward branch: integrate g(t, y) = -f(-t, y); requested times are negated on read
  int emitAfter;        // 1: a launch ends with the emission block of the next iteration (:511-524); 0: the last permitted launch (max_launches)
  int32_t* denseIdx_io; // per-IVP denseIndex
  double* rows;         // result tensor [n_t][dim][N] / [n_t][N][dim]
  int64_t rowStride;
  const int32_t* rowBase;  // forward: emission k of IVP i goes to row rowBase0 + rowBase[i] + k (rowBase == nullptr: no backward branch ran, the
  int rowBase0;            // same base for every IVP); backward (negate): row nReq-1-k
};

constexpr int kBlock = 256;
constexpr int kAggSlots = 64;  // replicated statistics accumulators (see aggregate_stats)

#if !NNHIP_RTC
using SolveLaunchFn = hipError_t (*)(const SolveArgs&, hipStream_t);
using StepLaunchFn = hipError_t (*)(const StepArgs&, int negate, hipStream_t);

struct StreamTune {
  int vec = 4;           // 16-byte accesses in flight per lane and direction: 1, 2, 4, 8
  int mode = 0;          // see rk4_stream_vec_kernel
  int blocksPerCU = 8;   // persistent modes: grid = 256 CUs * blocksPerCU
};
#endif
}  // namespace nnhip_abi

namespace NNHIP_NS {
using namespace nnhip_abi;

#if !NNHIP_RTC

// Launch through hipLaunchKernel so the returned status belongs to THIS launch (hipGetLastError() can hand
// back a stale error left by an unrelated runtime call of the host process).
template <class... KArgs, class... Args>
inline hipError_t launch_kernel(void (*kernel)(KArgs...), dim3 grid, dim3 block, hipStream_t s, Args&&... args) {
  std::tuple<std::decay_t<KArgs>...> pack(std::forward<Args>(args)...);
  void* ptrs[sizeof...(KArgs)];
  int k = 0;
  std::apply([&](auto&... a) { ((ptrs[k++] = (void*)&a), ...); }, pack);
  return hipLaunchKernel((const void*)kernel, grid, block, ptrs, 0, s);
}
#endif  // !NNHIP_RTC

// RHS parameters of IVP i: the batch-wide ones, overridden by this IVP's column of the per-IVP table when there is one
template <class Args>
NNHIP_DEV Params params_of(const Args& a, int64_t i) {
  Params P = a.P;
  if (a.perIvpParams) {
#pragma unroll
    for (int k = 0; k < kMaxParams; ++k)
      if (k < a.nPerIvp) P.p[k] = a.perIvpParams[(int64_t)k * a.perIvpStride + i];
  }
  if (P.ivp) P.ivp += i;  // the context block of a run-time compiled right-hand side: this IVP's column
  if (P.aux) P.aux += i;
  return P;
}

// ------------------------------------------------------------------------------------------------
// fused solve: the body shared by the thread-per-IVP and lanes-per-system kernels.  `out`/`y0p` already
// point at this lane's first owned component of IVP i; D = components owned by the lane.
// Result rows follow ODESolver's assembly: yNegative.reversed ++ yZero ++ yPositive (ode.nim:585-586).
// ------------------------------------------------------------------------------------------------
struct LaneStats {
  unsigned long long steps = 0, rejected = 0;
  int ny = 0x7fffffff, nanAb = 0, trunc = 0;
  double progress = 0.0;
  double tFinal = 0.0, dtFinal = 0.0;  // of the forward branch (SolveArgs::tfinal_out)
};

// MODE: 1 = general (dense output capable), 0 = lean (tspan.len == 2: no Hermite history), 2 = lean + per-IVP 2-point tspan and
// options, 3 = general + per-IVP n_t-point tspan and options
template <int METHOD, int MODE = 1, class OpsF, class OpsB>
NNHIP_DEV void solve_body(const SolveArgs& a0, const OpsF& opsF, const OpsB& opsB, const double* y0p, double* out, LaneStats& ls, int64_t ivp) {
  // the direction bookkeeping of this IVP: the batch-wide one, or its own 2-point tspan
  struct Dir { int nPos, nNeg, nZero, n_t, useDense; double t0, tEndPos, tEndNeg, dtInit; int64_t rowStride, compStride, maxSteps; const double *tPos, *tNeg;
               StepCtl ctl; int64_t uniformFull[2]; int nTail[2]; double tailDt[2][4]; const double* emitW[2]; const int64_t* emitStep[2]; int nEmit[2]; };
  Dir a;
  a.nPos = a0.nPos; a.nNeg = a0.nNeg; a.nZero = a0.nZero; a.n_t = a0.n_t; a.useDense = a0.useDense; a.t0 = a0.t0; a.tEndPos = a0.tEndPos;
  a.tEndNeg = a0.tEndNeg; a.dtInit = a0.dtInit; a.rowStride = a0.rowStride; a.compStride = a0.compStride; a.maxSteps = a0.maxSteps;
  a.tPos = a0.tPos; a.tNeg = a0.tNeg; a.ctl = a0.ctl;
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    a.uniformFull[d] = a0.uniformFull[d]; a.nTail[d] = a0.nTail[d];
    a.emitW[d] = a0.emitW[d]; a.emitStep[d] = a0.emitStep[d]; a.nEmit[d] = a0.nEmit[d];
#pragma unroll
    for (int q = 0; q < 4; ++q) a.tailDt[d][q] = a0.tailDt[d][q];
  }
  constexpr bool DENSE = MODE == 1 || MODE == 3;
  [[maybe_unused]] bool callInvalid = false;
  if constexpr (MODE == 2 || MODE == 3) {  // their own instantiations: per-lane call data costs registers the other kernels do not have to spare
    const SolveArgs::PerCall& pc = a0.perCall;
    if (pc.tStart) a.t0 = pc.tStart[ivp];
    if (pc.absTol) a.ctl.absTol = fabs(pc.absTol[ivp]);
    if (pc.relTol) a.ctl.relTol = fabs(pc.relTol[ivp]);
    if (pc.dtMax) a.ctl.dtMax = fabs(pc.dtMax[ivp]);
    if (pc.dtMin) a.ctl.dtMin = fabs(pc.dtMin[ivp]);
    if constexpr (MethodTraits<METHOD>::adaptive) {
      if (pc.dtMax || pc.dtMin) a.dtInit = sqrt(a.ctl.dtMax * a.ctl.dtMin);             // :491-493
      if constexpr (MODE == 2) { if (pc.dtInit) a.dtInit = pc.dtInit[ivp]; }             // a resumed integration's step size
      callInvalid = a.ctl.dtMax < a.ctl.dtMin || (!(a.ctl.dtMin > 0.0) && a.maxSteps <= 0);   // newODEoptions :95-96 / would never finish
    } else {
      if (pc.dt) a.dtInit = fabs(pc.dt[ivp]);                                             // :495-496
      callInvalid = !(a.dtInit > 0.0);
    }
    if constexpr (MODE == 2) {
      const double te = pc.tEnd[ivp];
      a.nPos = a.t0 < te ? 1 : 0;   // tspan.filterIt(it > t0) (:479)
      a.nNeg = (te < a.t0 && !pc.resume) ? 1 : 0;   // :480 (a resumed forward integration that is already at or past its tEnd is finished)
      a.tEndPos = te;
      a.tEndNeg = -te;
      if (!(te == te) || !(a.t0 == a.t0) || fabs(te) == __longlong_as_double(0x7ff0000000000000LL)) callInvalid = true;  // non-finite spans never end
    } else {
      const int32_t* cnt = pc.tCounts + 3 * ivp;
      const double* row = pc.tGrid + ivp * (int64_t)a.n_t;
      a.nNeg = cnt[0]; a.nZero = cnt[1]; a.nPos = cnt[2];
      if (a.nNeg < 0 || !(a.t0 == a.t0) || fabs(a.t0) == __longlong_as_double(0x7ff0000000000000LL)) { callInvalid = true; a.nNeg = a.nZero = a.nPos = 0; }
      a.tNeg = row;                       // descending: row[0] = max(tNegative) ... row[nNeg-1] = min
      a.tPos = row + (a.n_t - a.nPos);    // ascending, ends with max(tPositive)
      a.tEndPos = a.nPos > 0 ? row[a.n_t - 1] : a.t0;            // tPositive.max (:510)
      a.tEndNeg = a.nNeg > 0 ? -row[a.nNeg - 1] : -a.t0;         // -tNegative.min (:549)
    }
  }
  constexpr int D = OpsF::D;
  if constexpr (MODE == 2 || MODE == 3) {
    if (callInvalid) {
      const double qn = __longlong_as_double(0x7ff8000000000000LL);
      for (int j = 0; j < a.n_t; ++j)
#pragma unroll
        for (int c = 0; c < D; ++c)
          if (opsF.owns(c)) out[(int64_t)j * a.rowStride + c * a.compStride] = qn;
      ls.ny = -1;
      return;
    }
  }
  double y0[D];
#pragma unroll
  for (int c = 0; c < D; ++c) y0[c] = opsF.owns(c) ? y0p[c * a.compStride] : 0.0;
  if (a.nPos == 0) {  // ode.nim:498,506 run whether or not a forward branch follows (visible only to a right-hand side that mutates its ctx)
    double tmp[D];
    opsF.rhs(a.t0, y0, tmp);
    opsF.rhs(a.t0, y0, tmp);
  }
  int rowBase = 0;
  int status = 0;
  DriveIn in;
  in.useDense = a.useDense;
  in.maxSteps = a.maxSteps;
  in.ctl = a.ctl;
  in.dtInit = a.dtInit;
  const int64_t rs = a.rowStride, cs = a.compStride;
  // Forward branch first, as the reference runs them (ode.nim:508-542, then :544-584): the two branches are independent (both restart
  // from y0), so the order only matters to a right-hand side that mutates its ctx — which then sees the reference's call sequence.
  // Result rows are yNegative.reversed ++ yZero ++ yPositive (:585): the forward rows go to base nNeg + nZero, and move up afterwards
  // in the rare case that the backward branch returns fewer rows than requested (reference quirk).
  int mPos = 0;
  const int posBase = a.nNeg + (a.nZero > 0 ? 1 : 0);
  if (a.nPos > 0) {  // ode.nim:508-542
    in.tStartEff = a.t0;
    in.tEnd = a.tEndPos;
    in.tReq = a.tPos;
    in.nReq = a.nPos;
    in.uniformFull = a.uniformFull[0];
    in.nTail = a.nTail[0];
    in.emitW = a.emitW[0]; in.emitStep = a.emitStep[0]; in.nEmit = a.nEmit[0];
#pragma unroll
    for (int q = 0; q < 4; ++q) in.tailDt[q] = a.tailDt[0][q];
    DriveOut o;
    const int nPos = a.nPos;
    const int rb = posBase;
    drive<METHOD, false, DENSE>(opsF, in, y0,
                         [=](int k, const double(&yv)[D]) {
                           if (k < nPos) {
#pragma unroll
                             for (int c = 0; c < D; ++c)
                               if (opsF.owns(c)) out[(int64_t)(rb + k) * rs + c * cs] = yv[c];
                           }
                         },
                         o);
    mPos = o.emitted < nPos ? o.emitted : nPos;
    status |= o.status;
    ls.steps += o.steps;
    ls.rejected += o.rejected;
    ls.progress += o.tFinal - in.tStartEff;
    ls.tFinal = (o.status & kStatusNaN) ? in.tEnd : o.tFinal;
    ls.dtFinal = o.dtFinal;
  }
  // Backward branch (ode.nim:544-584).  Emission k of this branch is element k of yNegative; the result holds
  // yNegative.reversed(), so it lands in row nNeg-1-k.
  if (a.nNeg > 0) {
    in.tStartEff = -a.t0;
    in.tEnd = a.tEndNeg;
    in.tReq = a.tNeg;
    in.nReq = a.nNeg;
    in.uniformFull = a.uniformFull[1];
    in.nTail = a.nTail[1];
    in.emitW = a.emitW[1]; in.emitStep = a.emitStep[1]; in.nEmit = a.nEmit[1];
#pragma unroll
    for (int q = 0; q < 4; ++q) in.tailDt[q] = a.tailDt[1][q];
    DriveOut o;
    const int nNeg = a.nNeg;
    drive<METHOD, true, DENSE>(opsB, in, y0,
                        [=](int k, const double(&yv)[D]) {
                          if (k < nNeg) {
#pragma unroll
                            for (int c = 0; c < D; ++c)
                              if (opsF.owns(c)) out[(int64_t)(nNeg - 1 - k) * rs + c * cs] = yv[c];
                          }
                        },
                        o);
    const int m = o.emitted < nNeg ? o.emitted : nNeg;
    if (m < nNeg) {  // reference quirk: fewer rows than requested -> they close up
      const int shift = nNeg - m;
      for (int j = 0; j < m; ++j)
#pragma unroll
        for (int c = 0; c < D; ++c)
          if (opsF.owns(c)) out[(int64_t)j * rs + c * cs] = out[(int64_t)(j + shift) * rs + c * cs];
    }
    rowBase = m;
    status |= o.status;
    ls.steps += o.steps;
    ls.rejected += o.rejected;
    ls.progress += o.tFinal - in.tStartEff;
  }
  if (a.nZero > 0) {  // `if t0 in tspan` (ode.nim:485-487)
#pragma unroll
    for (int c = 0; c < D; ++c)
      if (opsF.owns(c)) out[(int64_t)rowBase * rs + c * cs] = y0[c];
    rowBase += 1;
  }
  if (mPos > 0 && rowBase != posBase) {  // the backward branch closed up: the forward rows follow it
    for (int j = 0; j < mPos; ++j)
#pragma unroll
      for (int c = 0; c < D; ++c)
        if (opsF.owns(c)) out[(int64_t)(rowBase + j) * rs + c * cs] = out[(int64_t)(posBase + j) * rs + c * cs];
  }
  rowBase += mPos;
  if constexpr (MODE == 2) {
    if (a0.perCall.resume && a.nPos == 0 && a.nNeg == 0 && rowBase == 1 && a.n_t >= 2) {  // resumed where it had already arrived: the final row is the state itself
#pragma unroll
      for (int c = 0; c < D; ++c)
        if (opsF.owns(c)) out[(int64_t)rowBase * rs + c * cs] = y0[c];
      rowBase += 1;
      ls.tFinal = a.t0;
      ls.dtFinal = a.dtInit;
    }
  }
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  for (int j = rowBase; j < a.n_t; ++j)
#pragma unroll
    for (int c = 0; c < D; ++c)
      if (opsF.owns(c)) out[(int64_t)j * rs + c * cs] = qnan;
  ls.ny = rowBase;
  ls.nanAb = (status & kStatusNaN) ? 1 : 0;
  ls.trunc = (status & 2) ? 1 : 0;
}

// Reduction of the per-IVP statistics: wavefront shuffles, then the four waves of the workgroup through LDS, then one atomic per
// workgroup and field; the accumulators are replicated kAggSlots times (slot = workgroup index mod kAggSlots) so 4e4 workgroups do
// not serialise on four addresses.  Must be reached by every thread of the workgroup (it contains a barrier).
NNHIP_DEV void aggregate_stats(unsigned long long* aggBase, LaneStats ls) {
  __shared__ unsigned long long part[kBlock / 64][6];
  unsigned long long steps = ls.steps, rejected = ls.rejected, smax = ls.steps;
  int ny = ls.ny, nanAb = ls.nanAb, trunc = ls.trunc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    steps += __shfl_down(steps, off, 64);
    rejected += __shfl_down(rejected, off, 64);
    const unsigned long long om = __shfl_down(smax, off, 64);
    smax = om > smax ? om : smax;
    const int on = __shfl_down(ny, off, 64);
    ny = on < ny ? on : ny;
    nanAb += __shfl_down(nanAb, off, 64);
    trunc += __shfl_down(trunc, off, 64);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    part[wave][0] = steps; part[wave][1] = rejected; part[wave][2] = smax; part[wave][3] = (unsigned long long)(unsigned)ny;
    part[wave][4] = (unsigned long long)nanAb; part[wave][5] = (unsigned long long)trunc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) {
      part[0][0] += part[w][0]; part[0][1] += part[w][1];
      part[0][2] = part[w][2] > part[0][2] ? part[w][2] : part[0][2];
      part[0][3] = part[w][3] < part[0][3] ? part[w][3] : part[0][3];
      part[0][4] += part[w][4]; part[0][5] += part[w][5];
    }
    unsigned long long* agg = aggBase + (size_t)(blockIdx.x % kAggSlots) * 8;
    atomicAdd(&agg[0], part[0][0]);
    atomicAdd(&agg[1], part[0][1]);
    atomicMax(&agg[2], part[0][2]);
    atomicMin(&agg[3], part[0][3]);
    if (part[0][4]) atomicAdd(&agg[4], part[0][4]);
    if (part[0][5]) atomicAdd(&agg[5], part[0][5]);
  }
}

// Waves per SIMD the thread-per-IVP solve kernels are compiled for: the compiler's own choice everywhere (2, 3 and 5 forced: DOPRI54 / Tsit54 on 1-3 components
// 0-17 % slower, profiles/r04_tpi_occupancy_ab.json) except the lean Vern65 solve of 3-component systems, which it leaves at 2 waves: held to 3, 1.43 -> 1.35 ms.
// -DNNHIP_SOLVE_TPI_WPE=n forces n for all of them (A/B).
template <int METHOD, class RHS, int MODE>
constexpr int solve_tpi_min_waves() { return (METHOD == NNHIP_VERN65 && MODE == 0 && RhsSize<RHS>::value == 3) ? 3 : 1; }
#ifdef NNHIP_SOLVE_TPI_WPE
#define NNHIP_SOLVE_TPI_ATTR __attribute__((amdgpu_waves_per_eu(NNHIP_SOLVE_TPI_WPE, NNHIP_SOLVE_TPI_WPE)))
#else
#define NNHIP_SOLVE_TPI_ATTR __attribute__((amdgpu_waves_per_eu(solve_tpi_min_waves<METHOD, RHS, MODE>())))
#endif
template <int METHOD, class RHS, int MODE = 1>
__global__ __launch_bounds__(kBlock) NNHIP_SOLVE_TPI_ATTR void solve_tpi_kernel(const SolveArgs a) {
  controller_prologue<MethodTraits<METHOD>::adaptive>();
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneStats ls;
  if (k < a.N) {
    const int64_t i = a.perm ? (int64_t)a.perm[k] : k;  // the IVP this work item integrates
    const Params P = params_of(a, i);
    const TpiOps<RHS, false> opsF{P};
    const TpiOps<RHS, true> opsB{P};
    solve_body<METHOD, MODE>(a, opsF, opsB, a.y0 + i * a.ivpStride, a.y_out + i * a.ivpStride, ls, i);
    if (a.ny_out) a.ny_out[i] = ls.ny;
    if (a.steps_out) a.steps_out[i] = (int64_t)ls.steps + (a.accumulate ? a.steps_out[i] : 0);
    if (a.rejected_out) a.rejected_out[i] = (int64_t)ls.rejected + (a.accumulate ? a.rejected_out[i] : 0);
    if (a.progress_out) a.progress_out[i] = ls.progress;
    if (a.tfinal_out) a.tfinal_out[i] = ls.tFinal;
    if (a.dtfinal_out) a.dtfinal_out[i] = ls.dtFinal;
  }
  if (a.agg) aggregate_stats(a.agg, ls);
}

#if !NNHIP_RTC
template <int METHOD, class RHS>
hipError_t launch_solve_tpi(const SolveArgs& a, hipStream_t s) {
  const int64_t grid = (a.N + kBlock - 1) / kBlock;
  if (grid <= 0) return hipSuccess;
  if (a.perCall.tGrid) return launch_kernel(solve_tpi_kernel<METHOD, RHS, 3>, dim3((unsigned)grid), dim3(kBlock), s, a);  // every IVP its own n_t-point tspan
  if (a.perCall.tEnd) return launch_kernel(solve_tpi_kernel<METHOD, RHS, 2>, dim3((unsigned)grid), dim3(kBlock), s, a);  // every IVP its own tEnd
  if (!a.useDense) return launch_kernel(solve_tpi_kernel<METHOD, RHS, 0>, dim3((unsigned)grid), dim3(kBlock), s, a);  // lean: no Hermite history
  return launch_kernel(solve_tpi_kernel<METHOD, RHS, 1>, dim3((unsigned)grid), dim3(kBlock), s, a);
}
#endif

// ---- lanes per system: DIM lanes of one wavefront integrate one DIM-component system -------------
// Lane (s, c) owns component c of system s: its y, k1..kS, yNew are single VGPR doubles; the stage
// argument vector and the squared error components of each system live in LDS (2*DIM doubles per system,
// 4 KiB per 256-thread workgroup).  With the AoS layout a wave's 64 lanes read 512 contiguous bytes.
// Occupancy of the fused lanes-per-system kernels.  With 4 components per lane a 7-stage method holds ~190 VGPRs (2 waves per
// SIMD) even without the dense-output history; asked for 3 waves the allocator fits 168 with 28-56 B of scratch per lane, and the
// VALU-bound solve gains ~9 % (C4 Tsit54 default 7.45 -> 6.75 ms, DOPRI54 6.71 -> 6.17 ms; profiles/r02_c4_fused_ab.txt).  The
// dense instantiations held 215+ VGPRs while they carried the Hermite history across the step; since round 4 (rows interpolated right after the
// step, drive()) they need 203 and are held to 3 waves as well: 16-component Tsit54 / DOPRI54 with one interior row 5.70 -> 5.37 ms, tight 1.65 -> 1.50 ms
// (same box, scripts/ab_dense_wpe.sh).  A/B hooks: -DNNHIP_LPS_WPE=n forces n for all of them, -DNNHIP_LPS_DENSE_WAVES=n for the dense ones.
// LDS slots of the lanes-per-system kernels: per system DIM doubles of stage arguments (ys) and DIM of squared error components (es).
// A/B hook -DNNHIP_LPS_PAD=n: consecutive systems n doubles further apart than DIM (at a stride of exactly 128 B, DIM = 16, every
// system of a wavefront starts in the same LDS bank and the ordered error sum reads es[j] of 8 or 16 systems at once).
#ifndef NNHIP_LPS_PAD
#define NNHIP_LPS_PAD 0  // measured: no effect either way (profiles/r02_pow_tables_ab.txt section 10); hook kept
#endif
template <int DIM>
constexpr int lps_stride() { return DIM + NNHIP_LPS_PAD; }
template <int DIM, int CPL>
constexpr int lps_lds_doubles() { return 2 * (kBlock / (DIM / CPL)) * lps_stride<DIM>(); }

// (not the 9-stage Vern65: held to 3 waves it spills 200-340 B per lane and the 16-component solve takes 14.6 ms instead of 9.9, r03_dim16_variants.json)
#ifndef NNHIP_LPS_DENSE_WAVES
#define NNHIP_LPS_DENSE_WAVES 3  // waves per SIMD asked for the dense-output instantiations (MODE 1 / 3); 1 = leave it to the allocator (A/B)
#endif
template <int METHOD, int CPL, int MODE>
constexpr int lps_solve_waves() { return (METHOD != NNHIP_VERN65 && CPL >= 4) ? (MODE == 0 ? 3 : ((MODE == 1 || MODE == 3) ? NNHIP_LPS_DENSE_WAVES : 1)) : 1; }
#ifndef NNHIP_LPS_WPE
#define NNHIP_LPS_ATTR __attribute__((amdgpu_waves_per_eu(lps_solve_waves<METHOD, CPL, MODE>())))
#else
#define NNHIP_LPS_ATTR __attribute__((amdgpu_waves_per_eu(NNHIP_LPS_WPE, NNHIP_LPS_WPE)))
#endif
template <int METHOD, class RHS, int CPL, bool SHUFFLE_NORM = false, int MODE = 1>
__global__ __launch_bounds__(kBlock) NNHIP_LPS_ATTR void solve_lps_kernel(const SolveArgs a) {
  constexpr int DIM = RHS::dim;
  constexpr int LPSYS = DIM / CPL;  // lanes per system
  static_assert(DIM % CPL == 0 && 64 % LPSYS == 0, "a system must not straddle wavefronts");
  __shared__ double lds[lps_lds_doubles<DIM, CPL>()];
  controller_prologue<MethodTraits<METHOD>::adaptive>();
  const int sysInBlock = threadIdx.x / LPSYS, c = (threadIdx.x % LPSYS) * CPL;
  const int64_t k = (int64_t)blockIdx.x * (kBlock / LPSYS) + sysInBlock;
  LaneStats ls;
  if (k < a.N) {
    const int64_t i = a.perm ? (int64_t)a.perm[k] : k;  // the system this group of lanes integrates
    double* ys = lds + sysInBlock * lps_stride<DIM>();
    double* es = lds + lps_lds_doubles<DIM, CPL>() / 2 + sysInBlock * lps_stride<DIM>();
    const Params P = params_of(a, i);
    const LpsOps<RHS, false, CPL, SHUFFLE_NORM> opsF{P, ys, es, c};
    const LpsOps<RHS, true, CPL, SHUFFLE_NORM> opsB{P, ys, es, c};
    solve_body<METHOD, MODE>(a, opsF, opsB, a.y0 + i * a.ivpStride + c * a.compStride, a.y_out + i * a.ivpStride + c * a.compStride, ls, i);
    if (c == 0) {
      if (a.ny_out) a.ny_out[i] = ls.ny;
      if (a.steps_out) a.steps_out[i] = (int64_t)ls.steps + (a.accumulate ? a.steps_out[i] : 0);
      if (a.rejected_out) a.rejected_out[i] = (int64_t)ls.rejected + (a.accumulate ? a.rejected_out[i] : 0);
      if (a.progress_out) a.progress_out[i] = ls.progress;
      if (a.tfinal_out) a.tfinal_out[i] = ls.tFinal;
      if (a.dtfinal_out) a.dtfinal_out[i] = ls.dtFinal;
    } else {  // count each system once in the aggregate sums
      ls.steps = 0; ls.rejected = 0; ls.nanAb = 0; ls.trunc = 0;
    }
  }
  if (a.agg) aggregate_stats(a.agg, ls);
}

#if !NNHIP_RTC
template <int METHOD, class RHS, int CPL = 1, bool SHUFFLE_NORM = false>
hipError_t launch_solve_lps(const SolveArgs& a, hipStream_t s) {
  constexpr int perBlock = kBlock / (RHS::dim / CPL);
  const int64_t grid = (a.N + perBlock - 1) / perBlock;
  if (grid <= 0) return hipSuccess;
  if (a.perCall.tGrid) return launch_kernel(solve_lps_kernel<METHOD, RHS, CPL, SHUFFLE_NORM, 3>, dim3((unsigned)grid), dim3(kBlock), s, a);
  if (a.perCall.tEnd) return launch_kernel(solve_lps_kernel<METHOD, RHS, CPL, SHUFFLE_NORM, 2>, dim3((unsigned)grid), dim3(kBlock), s, a);
  if (!a.useDense) return launch_kernel(solve_lps_kernel<METHOD, RHS, CPL, SHUFFLE_NORM, 0>, dim3((unsigned)grid), dim3(kBlock), s, a);
  return launch_kernel(solve_lps_kernel<METHOD, RHS, CPL, SHUFFLE_NORM, 1>, dim3((unsigned)grid), dim3(kBlock), s, a);
}
#endif

// Kernel-argument prefetch: the argument struct is ~300 bytes and the compiler fetches it piecemeal, one s_load + s_waitcnt round
// trip per basic block that first touches a field (five serialized scalar round trips before the first vector load of the advance
// kernel).  Pinning the hot fields in SGPRs at the top of the kernel makes them ONE batch of s_loads behind one wait.
#ifdef NNHIP_CPU_EMU  // (register classes of the target: nothing to pin on the host)
#define NNHIP_PIN_SGPR64(x) ((void)(x))
#define NNHIP_PIN_SGPR32(x) ((void)(x))
#define NNHIP_KEEP_VGPR(x) ((void)(x))
#else
#define NNHIP_PIN_SGPR64(x) asm volatile("" ::"s"(__builtin_bit_cast(unsigned long long, (x))))
#define NNHIP_PIN_SGPR32(x) asm volatile("" ::"s"(x))
#define NNHIP_KEEP_VGPR(x) asm volatile("" : "+v"(x))  // the value is "modified" here: whatever produces it (a load) stays above this point
#endif
NNHIP_DEV void pin_step_args(const StepArgs& a) {
  NNHIP_PIN_SGPR64(a.N); NNHIP_PIN_SGPR64(a.ivpStride); NNHIP_PIN_SGPR64(a.compStride);
  NNHIP_PIN_SGPR64(a.y_in); NNHIP_PIN_SGPR64(a.fsal_in); NNHIP_PIN_SGPR64(a.y_out); NNHIP_PIN_SGPR64(a.fsal_out);
  NNHIP_PIN_SGPR64(a.t_io); NNHIP_PIN_SGPR64(a.dt_io); NNHIP_PIN_SGPR64(a.error); NNHIP_PIN_SGPR64(a.tEnd);
  NNHIP_PIN_SGPR64(a.ctl.absTol); NNHIP_PIN_SGPR64(a.ctl.relTol); NNHIP_PIN_SGPR64(a.ctl.dtMax); NNHIP_PIN_SGPR64(a.ctl.dtMin);
  NNHIP_PIN_SGPR64(a.perIvpParams); NNHIP_PIN_SGPR64(a.active); NNHIP_PIN_SGPR64(a.steps_io);
  NNHIP_PIN_SGPR32(a.nPerIvp);
  NNHIP_PIN_SGPR64(a.t_dev); NNHIP_PIN_SGPR64(a.dt_dev); NNHIP_PIN_SGPR64(a.t_uniform); NNHIP_PIN_SGPR64(a.dt_uniform); NNHIP_PIN_SGPR64(a.dt_used);
#pragma unroll
  for (int k = 0; k < 4; ++k) NNHIP_PIN_SGPR64(a.P.p[k]);
}

// ------------------------------------------------------------------------------------------------
// one IntegratorProc call (step-streaming; state in HBM between calls)
// ------------------------------------------------------------------------------------------------
// NT: non-temporal hint on the state arrays (see the advance kernels below: a template parameter, chosen by the host from the
// working-set size; instantiated for the adaptive methods' forward direction only)
template <bool NT>
NNHIP_DEV double ld_state(const double* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
NNHIP_DEV void st_state(double v, double* p) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int METHOD, bool NT = false, class Ops>
NNHIP_DEV void step_body(const StepArgs& a, const Ops& ops, int64_t i, int64_t base, bool writeScalars) {
  constexpr int D = Ops::D;
  double y[D], yNew[D];
#pragma unroll
  for (int c = 0; c < D; ++c) y[c] = ops.owns(c) ? ld_state<NT>(&a.y_in[base + c * a.compStride]) : 0.0;
  const double t = a.t_dev ? a.t_dev[i] : a.t_uniform;
  double dt = a.dt_dev ? a.dt_dev[i] : a.dt_uniform;
  double error = 0.0;
  if constexpr (METHOD == NNHIP_RK4) {
    rk4_step(ops, t, rk4_dt(dt), y, yNew);
    if (a.fsal_out) {  // fixed-step methods return yNew in the FSAL slot (ode.nim:189)
#pragma unroll
      for (int c = 0; c < D; ++c)
        if (ops.owns(c)) a.fsal_out[base + c * a.compStride] = yNew[c];
    }
  } else if constexpr (!MethodTraits<METHOD>::adaptive) {
    fixed_step<METHOD>(ops, t, dt, y, yNew);
    if (a.fsal_out) {
#pragma unroll
      for (int c = 0; c < D; ++c)
        if (ops.owns(c)) a.fsal_out[base + c * a.compStride] = yNew[c];
    }
  } else {
    double fsal[D];
    if constexpr (has_tableau(METHOD)) {  // only the tableau methods read FSAL (k1 = FSAL); RK21 / BS32 ignore it
#pragma unroll
      for (int c = 0; c < D; ++c) fsal[c] = ops.owns(c) ? ld_state<NT>(&a.fsal_in[base + c * a.compStride]) : 0.0;
    }
    int64_t rej = 0;
    double factor;
    embedded_step<METHOD, NNHIP_PEEL_STREAM != 0>(ops, t, dt, y, fsal, fsal, yNew, error, a.ctl, rej, factor);
    if (a.fsal_out) {
#pragma unroll
      for (int c = 0; c < D; ++c)
        if (ops.owns(c)) st_state<NT>(fsal[c], &a.fsal_out[base + c * a.compStride]);
    }
  }
#pragma unroll
  for (int c = 0; c < D; ++c)
    if (ops.owns(c)) st_state<NT>(yNew[c], &a.y_out[base + c * a.compStride]);
  if (writeScalars) {
    if (a.dt_used) a.dt_used[i] = dt;
    if (a.error) a.error[i] = error;
  }
}

template <int METHOD, class RHS, bool NEG, bool NT = false>
__global__ __launch_bounds__(kBlock) void step_tpi_kernel(const StepArgs a) {
  controller_prologue<MethodTraits<METHOD>::adaptive>();
  pin_step_args(a);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.N) return;
  const Params P = params_of(a, i);
  const TpiOps<RHS, NEG> ops{P};
  step_body<METHOD, NT>(a, ops, i, i * a.ivpStride, true);
}

#if !NNHIP_RTC
template <int METHOD, class RHS>
hipError_t launch_step_tpi(const StepArgs& a, int negate, hipStream_t s) {
  const int64_t grid = (a.N + kBlock - 1) / kBlock;
  if (grid <= 0) return hipSuccess;
  if (negate) return launch_kernel(step_tpi_kernel<METHOD, RHS, true>, dim3((unsigned)grid), dim3(kBlock), s, a);
  if constexpr (MethodTraits<METHOD>::adaptive) {
    if (a.nontemporal) return launch_kernel(step_tpi_kernel<METHOD, RHS, false, true>, dim3((unsigned)grid), dim3(kBlock), s, a);
  }
  return launch_kernel(step_tpi_kernel<METHOD, RHS, false>, dim3((unsigned)grid), dim3(kBlock), s, a);
}
#endif

// ------------------------------------------------------------------------------------------------
// advance: ONE iteration of ODESolver's adaptive loop body per IVP and launch (ode.nim:525-541), everything resident
// in HBM between launches:  dt = min(dt, tEnd - t); (y, FSAL, dt, error) = integrator(...); t += dt; controller.
// Algorithmic traffic per attempted step: read y(d)+FSAL(d)+t+dt, write y(d)+FSAL(d)+t+dt+error = 8*(4d+5) B.
// ------------------------------------------------------------------------------------------------
// StepArgs::nontemporal gives the streamed state arrays of the advance kernels the non-temporal hint: once a launch's state
// (y, FSAL, t, dt, error: 8*(2d+3) B per IVP) no longer fits the 256 MiB Infinity Cache, keeping it out of the cache is worth
// +11 % (1e7 Lorenz IVPs: 241 -> 216 us per iteration = 6.3 TB/s); below that it costs 4 % (1e6: 26.1 -> 27.6 us); the host
// picks per call (profiles/r02_pow_tables_ab.txt, section 7).  The hint is a TEMPLATE parameter of the kernel: a run-time branch
// between hinted and plain accesses of the same addresses is merged by the compiler into the plain ones (the hint is dropped).
// A lane's components are contiguous in the AoS layout (compStride == 1): with an even number of them per lane and 16-byte aligned
// arrays the state moves as 16-byte accesses (global_load/store_dwordx4) instead of pairs of 8-byte ones at a 16-byte stride.
template <class Ops>
struct OpsAllOwned { static constexpr bool value = true; };
template <class RHS, bool NEG, int CPL, bool SH>
struct OpsAllOwned<LpsOps<RHS, NEG, CPL, SH>> { static constexpr bool value = RhsSize<RHS>::value == RHS::dim; };
template <class RHS, int CPL>
struct LpsOpsRt;  // the lanes-per-system ops of the dense streaming kernel (direction chosen at run time), defined with it
template <class RHS, int CPL>
struct OpsAllOwned<LpsOpsRt<RHS, CPL>> { static constexpr bool value = RhsSize<RHS>::value == RHS::dim; };
NNHIP_DEV bool adv_vec2(const StepArgs& a) {
#ifdef NNHIP_ADV_NO_VEC2
  return false;
#else
  return a.compStride == 1 && (a.ivpStride & 1) == 0 &&
         ((((uintptr_t)a.y_in | (uintptr_t)a.fsal_in | (uintptr_t)a.y_out | (uintptr_t)a.fsal_out) & 15) == 0);
#endif
}
// Does the FSAL slot of the IntegratorProc signature (ode.nim:38) travel through HBM between launches?
//   BS32, RK21     never: they evaluate k1 = f(t, y) themselves (:203, :224) and nothing reads the slot they return;
//   DOPRI54, Tsit54  unless StepArgs::recomputeFsal: their FSAL is the last stage f(t + dt, yNew) with yNew the stage's own argument
//                  (b = the tableau's last row, c_S = 1), i.e. f of exactly the (t, y) the next launch reads — a pure right-hand side
//                  evaluated there again returns the same bits, for 16*d bytes per step less;
//   Vern65         always: its last stage's argument is not yNew (separate b literals, :426-433).
template <int METHOD>
NNHIP_DEV bool adv_fsal_in_hbm(const StepArgs& a) {
  if constexpr (METHOD == NNHIP_BS32 || METHOD == NNHIP_RK21) return false;
  else if constexpr (METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54) return a.recomputeFsal == 0;
  else return true;
}
template <bool NT, class Ops, int D>
NNHIP_DEV void adv_load_state(const StepArgs& a, const Ops& ops, int64_t base, double (&y)[D], double (&fsal)[D], bool withFsal = true) {
  // (FSAL is zeroed once, up front: zeroing it separately on the two paths below left the state arrays in scratch — 96 B per lane and twice
  // the time for BS32 / RK21 on 16-component systems — because the merged stores addressed them through a run-time offset)
#pragma unroll
  for (int c = 0; c < D; ++c) fsal[c] = 0.0;
  if constexpr (D % 2 == 0 && OpsAllOwned<Ops>::value && !NT) {
    if (adv_vec2(a)) {
#pragma unroll
      for (int c = 0; c < D; c += 2) {
        const double2 v = *reinterpret_cast<const double2*>(&a.y_in[base + c]);
        y[c] = v.x; y[c + 1] = v.y;
      }
      if (withFsal) {
#pragma unroll
        for (int c = 0; c < D; c += 2) {
          const double2 w = *reinterpret_cast<const double2*>(&a.fsal_in[base + c]);
          fsal[c] = w.x; fsal[c + 1] = w.y;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    if (!ops.owns(c)) { y[c] = 0.0; continue; }
    if constexpr (NT) y[c] = __builtin_nontemporal_load(&a.y_in[base + c * a.compStride]);
    else y[c] = a.y_in[base + c * a.compStride];
  }
  if (withFsal) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
      if (!ops.owns(c)) continue;
      if constexpr (NT) fsal[c] = __builtin_nontemporal_load(&a.fsal_in[base + c * a.compStride]);
      else fsal[c] = a.fsal_in[base + c * a.compStride];
    }
  }
}
template <bool NT, class Ops, int D>
NNHIP_DEV void adv_store_state(const StepArgs& a, const Ops& ops, int64_t base, const double (&yNew)[D], const double (&fsal)[D], bool withFsal = true) {
  if constexpr (D % 2 == 0 && OpsAllOwned<Ops>::value && !NT) {
    if (adv_vec2(a)) {
#pragma unroll
      for (int c = 0; c < D; c += 2) *reinterpret_cast<double2*>(&a.y_out[base + c]) = make_double2(yNew[c], yNew[c + 1]);
      if (withFsal) {
#pragma unroll
        for (int c = 0; c < D; c += 2) *reinterpret_cast<double2*>(&a.fsal_out[base + c]) = make_double2(fsal[c], fsal[c + 1]);
      }
      return;
    }
  }
#pragma unroll
  for (int c = 0; c < D; ++c) {
    if (!ops.owns(c)) continue;
    if constexpr (NT) __builtin_nontemporal_store(yNew[c], &a.y_out[base + c * a.compStride]);
    else a.y_out[base + c * a.compStride] = yNew[c];
  }
  if (withFsal) {
#pragma unroll
    for (int c = 0; c < D; ++c) {
      if (!ops.owns(c)) continue;
      if constexpr (NT) __builtin_nontemporal_store(fsal[c], &a.fsal_out[base + c * a.compStride]);
      else a.fsal_out[base + c * a.compStride] = fsal[c];
    }
  }
}
// Shared by the thread-per-IVP and the lanes-per-system form.  `base` addresses this lane's first owned component of IVP i; all
// lanes of a system read the same (t, dt) and compute bit-identical values for them; lane `writeScalars` stores them.
template <int D>
struct AdvState {
  double t, dt, y[D], fsal[D];
  bool live;
};
// (t, dt) of IVP i in either form of StepArgs::t_io / dt_io
NNHIP_DEV void adv_ld_t_dt(const StepArgs& a, int64_t i, double& t, double& dt) {
  if (a.dt_io) { dt = a.dt_io[i]; t = a.t_io[i]; }
  else { const double2 v = reinterpret_cast<const double2*>(a.t_io)[i]; t = v.x; dt = v.y; }
}
NNHIP_DEV void adv_st_t_dt(const StepArgs& a, int64_t i, double t, double dt) {
  if (a.dt_io) { a.t_io[i] = t; a.dt_io[i] = dt; }
  else reinterpret_cast<double2*>(a.t_io)[i] = make_double2(t, dt);
}
// phase 1: t, and — for IVPs still short of tEnd — dt, y, FSAL (finished IVPs touch no other memory)
template <int METHOD, bool NT, class Ops, bool SPECULATE = false>
NNHIP_DEV void adv_fetch(const StepArgs& a, const Ops& ops, int64_t i, int64_t base, AdvState<Ops::D>& s) {
  const bool withFsal = adv_fsal_in_hbm<METHOD>(a);
  if constexpr (SPECULATE) {
    // all loads of the IVP in ONE round trip (finished IVPs read their state for nothing): `t` first would put the state loads
    // behind a dependent branch = two serialized memory latencies per wave.  The empty asm keeps the loads above the branch.
    adv_load_state<NT>(a, ops, base, s.y, s.fsal, withFsal);
    adv_ld_t_dt(a, i, s.t, s.dt);
#pragma unroll
    for (int c = 0; c < Ops::D; ++c) { NNHIP_KEEP_VGPR(s.y[c]); NNHIP_KEEP_VGPR(s.fsal[c]); }
    NNHIP_KEEP_VGPR(s.dt); NNHIP_KEEP_VGPR(s.t);
    s.live = s.t < a.tEnd;  // :511
  } else {
    if (a.dt_io) s.t = a.t_io[i];
    else adv_ld_t_dt(a, i, s.t, s.dt);
    s.live = s.t < a.tEnd;  // :511
    if (s.live) {
      adv_load_state<NT>(a, ops, base, s.y, s.fsal, withFsal);
      if (a.dt_io) s.dt = a.dt_io[i];
    }
  }
}
// phase 2: the loop iteration itself (registers only) ...
template <int D>
struct AdvResult {
  double t, dt, error, y[D], fsal[D];
  bool live;  // the lane advanced an IVP in this launch: the members above are to be written back
  int steps;  // loop iterations it took (1 unless StepArgs::stepsPerLaunch > 1)
};
template <int METHOD, class Ops>
NNHIP_DEV unsigned int adv_compute(const StepArgs& a, const Ops& ops, AdvState<Ops::D>& s, AdvResult<Ops::D>& r) {
  constexpr int D = Ops::D;
  r.live = s.live;
  r.steps = 1;
  if (!s.live) return 0u;
  double t = s.t;
  double dt = nmin(s.dt, a.tEnd - t);  // :525
  double error = 0.0;
  int64_t rej = 0;
  double factor;
  embedded_step<METHOD, NNHIP_PEEL_STREAM != 0>(ops, t, dt, s.y, s.fsal, s.fsal, r.y, error, a.ctl, rej, factor);  // :531
  t += dt;                                                              // :532
  if (error == 0.0) dt *= 5.0;                                          // :534-535
  else dt = dt * factor;                                                // :537 (the factor of the accepted attempt's error)
  if (dt < a.ctl.dtMin) dt = a.ctl.dtMin;                               // :538-539
  else if (a.ctl.dtMax < dt) dt = a.ctl.dtMax;                          // :540-541
  if (error != error) t = a.tEnd;  // NaN abort (same deviation as the fused driver): retire the IVP
#pragma unroll
  for (int c = 0; c < D; ++c) r.fsal[c] = s.fsal[c];
  r.t = t; r.dt = dt; r.error = error;
  return t < a.tEnd ? 1u : 0u;
}
// ... and its write-back
template <int METHOD, bool NT, class Ops>
NNHIP_DEV void adv_commit(const StepArgs& a, const Ops& ops, int64_t i, int64_t base, bool writeScalars, const AdvResult<Ops::D>& r) {
  if (!r.live) return;
  adv_store_state<NT>(a, ops, base, r.y, r.fsal, adv_fsal_in_hbm<METHOD>(a));
  if (writeScalars) {
    adv_st_t_dt(a, i, r.t, r.dt);
    if (a.error) a.error[i] = r.error;
    if (a.steps_io) a.steps_io[i] += r.steps;
  }
}
// MULTI: StepArgs::stepsPerLaunch iterations per launch.  A template parameter, not a run-time trip count of one loop: carrying the
// state around a loop costs the one-iteration kernels a spill and 10-15 % (C3 1e6: 22.9 -> 26.4 us, C4: 123 -> 131 us per launch).
template <int METHOD, bool NT, bool MULTI = false, class Ops>
NNHIP_DEV unsigned int adv_advance(const StepArgs& a, const Ops& ops, int64_t i, int64_t base, bool writeScalars, AdvState<Ops::D>& s) {
  constexpr int D = Ops::D;
  AdvResult<D> r;
  unsigned int more;
  if constexpr (METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54) {
    if (a.recomputeFsal && s.live) ops.rhs(s.t, s.y, s.fsal);  // FSAL = f(t, y): what the previous launch's last stage evaluated (adv_fsal_in_hbm)
  }
  if constexpr (!MULTI) {
    more = adv_compute<METHOD>(a, ops, s, r);  // the IntegratorProc seam proper: one loop iteration per launch
  } else {
    // the IVP keeps iterating ode.nim:525-541 on the registers it has (what K launches would do through HBM; the same operations
    // in the same order, so the same bits) until it reaches tEnd or has taken its K iterations
    int k = 0;
    for (;;) {
      more = adv_compute<METHOD>(a, ops, s, r);
      ++k;
      if (!more || k >= a.stepsPerLaunch) break;
      s.t = r.t; s.dt = r.dt;
#pragma unroll
      for (int c = 0; c < D; ++c) { s.y[c] = r.y[c]; s.fsal[c] = r.fsal[c]; }
    }
    r.steps = k;
  }
  adv_commit<METHOD, NT>(a, ops, i, base, writeScalars, r);
  return more;
}
template <int METHOD, bool NT = false, bool MULTI = false, class Ops>
NNHIP_DEV unsigned int advance_body(const StepArgs& a, const Ops& ops, int64_t i, int64_t base, bool writeScalars) {
  AdvState<Ops::D> s;
  // (A/B -DNNHIP_ADV_TPI_SPECULATE=1: every load of the IVP in ONE round trip instead of `t` first — 2-4 % on C3 (1e6: 21.3 -> 20.6 us, 1e7: 186 -> 183 us),
  // not adopted: IVPs that are finished would keep reading their state, +50 % load traffic in the late launches of a heterogeneous batch)
#ifndef NNHIP_ADV_TPI_SPECULATE
#define NNHIP_ADV_TPI_SPECULATE 0
#endif
  adv_fetch<METHOD, NT, Ops, NNHIP_ADV_TPI_SPECULATE != 0>(a, ops, i, base, s);
  return adv_advance<METHOD, NT, MULTI>(a, ops, i, base, writeScalars, s);
}

// Occupancy of the thread-per-IVP advance kernel.  glibc's pow keeps six 64-bit polynomial constants in VGPRs (an FMA takes
// at most one scalar operand on gfx9), which puts the 7-stage methods on 3-component systems at 131-135 VGPRs = 3 waves per SIMD;
// asked for 4 waves the allocator rematerialises them instead (128 VGPRs, 0-12 B of scratch) and the HBM-bound loop gains 4-12 %
// (1e7 Lorenz IVPs, DOPRI54 298 -> 264 us, Tsit54 286 -> 274 us per iteration; profiles/r02_pow_tables_ab.txt).  Larger systems
// and the 9-stage Vern65 would spill 50-260 B per lane — more traffic than they save — and keep the default allocation.
// -DNNHIP_ADV_TPI_WPE=n overrides (A/B).
#ifdef NNHIP_ADV_TPI_WPE
#define NNHIP_ADV_TPI_ATTR __attribute__((amdgpu_waves_per_eu(NNHIP_ADV_TPI_WPE)))
#else
template <int METHOD, class RHS>
constexpr int adv_tpi_waves() { return (RHS::dim <= 3 && METHOD != NNHIP_VERN65) || RHS::dim <= 2 ? 4 : 1; }
#define NNHIP_ADV_TPI_ATTR __attribute__((amdgpu_waves_per_eu(adv_tpi_waves<METHOD, RHS>())))
#endif
#ifdef NNHIP_ADV_LPS_WPE
#define NNHIP_ADV_LPS_ATTR __attribute__((amdgpu_waves_per_eu(NNHIP_ADV_LPS_WPE)))
#else
#define NNHIP_ADV_LPS_ATTR
#endif
template <int METHOD, class RHS, bool NT = false, bool MULTI = false>
__global__ __launch_bounds__(kBlock) NNHIP_ADV_TPI_ATTR void advance_tpi_kernel(const StepArgs a) {
  static_assert(MethodTraits<METHOD>::adaptive, "fixed-step methods share (t, dt): use the uniform streaming loop");
  controller_prologue();
  pin_step_args(a);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // workgroup size chosen by the launcher (<= kBlock)
  unsigned int stillActive = 0;
  if (i < a.N) {
    const Params P = params_of(a, i);
    const TpiOps<RHS, false> ops{P};
    stillActive = advance_body<METHOD, NT, MULTI>(a, ops, i, i * a.ivpStride, true);
  }
  // "is anyone still integrating?" — a plain flag store per workgroup into one of kAggSlots words (no atomics: 1e5 waves
  // hitting one address cost ~170 us per launch), and only in the launches whose answer the host will read
  if (a.active) {
    if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
  }
}

// The same loop iteration for Vector[float] states of 8 / 16 / 32 ... components (C4's streamed form; ode.nim:525-541 over
// Vector[float]): DIM / CPL lanes of one wavefront per system, one component per lane by default (best coalescing: with the AoS
// layout a wave moves 512 contiguous bytes per array), stage argument vector and error components through LDS (LpsOps).
// Algorithmic traffic 8*(4d+5) B per attempted step: 552 B at d = 16, so 1e6 systems stream 552 MB per launch — far
// beyond the 256 MiB Infinity Cache.
// A block advances SPG tiles of kBlock / LPSYS systems, software-pipelined (NNHIP_ADV_LPS_PIPE): the state of tile g + 1 is loaded
// (unconditionally, one round trip) before tile g is advanced, so a wave's memory latency runs under its own arithmetic — the
// kernel's two floors are close (1e6 x 16: VALU 65 us, HBM 86 us), its waves start together and stay in phase, and with 4 waves per
// SIMD nothing else overlaps them.  Measured (profiles/r02_pow_tables_ab.txt section 11): 125 -> 118 us per iteration with 2 tiles;
// 4 and 8 tiles no better; fetching 2 tiles up front WITHOUT the pipeline (PIPE=0, SPG=2), the speculative single round trip alone
// (NNHIP_ADV_LPS_SPECULATE), 16-byte accesses (adv_vec2) and padded LDS slots each change nothing measurable.
// Tiles per block by components per lane: the prefetched tile costs 2 * CPL + 2 doubles of VGPRs per lane; at 4 components per lane that is the
// difference between 2 waves per SIMD and fewer, and ONE tile per block (no prefetch, twice the blocks) is faster in both FSAL modes
// (1e6 x 16 Tsit54, kernel alone: FSAL re-evaluated 96-98 -> 88.5-91 us, carried 120.6 -> 108.3 us; at 2 per lane 2 tiles stay better: 119 vs 121 us).
#ifdef NNHIP_ADV_LPS_SPG  // A/B hook: forces the number of tiles for every instantiation
constexpr int adv_lps_spg(int) { return NNHIP_ADV_LPS_SPG; }
#else
constexpr int adv_lps_spg(int cpl) { return cpl >= 4 ? 1 : 2; }
#endif
#ifndef NNHIP_ADV_LPS_SPECULATE
#define NNHIP_ADV_LPS_SPECULATE 0
#endif
#ifndef NNHIP_ADV_LPS_PIPE
#define NNHIP_ADV_LPS_PIPE 1
#endif
#ifndef NNHIP_ADV_LPS_DEPTH
#define NNHIP_ADV_LPS_DEPTH 1
#endif
template <int METHOD, class RHS, int CPL = 1, bool MULTI = false>
__global__ __launch_bounds__(kBlock) NNHIP_ADV_LPS_ATTR void advance_lps_kernel(const StepArgs a) {
  static_assert(MethodTraits<METHOD>::adaptive, "fixed-step methods share (t, dt): use the uniform streaming loop");
  constexpr int DIM = RHS::dim;
  constexpr int LPSYS = DIM / CPL;
  constexpr int SPG = adv_lps_spg(CPL);
  static_assert(DIM % CPL == 0 && 64 % LPSYS == 0, "a system must not straddle wavefronts");
  __shared__ double lds[lps_lds_doubles<DIM, CPL>()];
  controller_prologue();
  pin_step_args(a);
  const int sysInBlock = threadIdx.x / LPSYS, c = (threadIdx.x % LPSYS) * CPL;
  constexpr int perBlock = kBlock / LPSYS;
  unsigned int stillActive = 0;
#if NNHIP_ADV_LPS_PIPE
  {
    // software pipeline over the block's SPG tiles: the state of tile g + 1 is loaded (unconditionally, one round trip, index
    // clamped) BEFORE tile g is advanced, so a wave's memory latency runs under its own arithmetic
    double* ys = lds + sysInBlock * lps_stride<DIM>();
    double* es = lds + lps_lds_doubles<DIM, CPL>() / 2 + sysInBlock * lps_stride<DIM>();
    const int64_t i0 = (int64_t)blockIdx.x * SPG * perBlock + sysInBlock;
    const Params P0 = a.P;
    const LpsOps<RHS, false, CPL> ops0{P0, ys, es, c};
    auto prefetch = [&](int64_t i, AdvState<CPL>& s, Params& P) {
      const int64_t ic = i < a.N ? i : a.N - 1;
      adv_load_state<false>(a, ops0, ic * a.ivpStride + c * a.compStride, s.y, s.fsal, adv_fsal_in_hbm<METHOD>(a));
      adv_ld_t_dt(a, ic, s.t, s.dt);
      P = params_of(a, ic);           // per-IVP parameters (sweeps) belong to the tile's loads as well
      if constexpr (SPG > 1) asm volatile("" ::: "memory");  // keeps the tiles' loads in program order: the wait for tile g must not cover tile g + 1's loads
                                                             // (with one tile there is nothing to order, and the clobber would pin the tile's state in scratch
                                                             // for the methods whose FSAL slot is never read: BS32 / RK21 at 4 components per lane, 96 B per lane)
    };
#if NNHIP_ADV_LPS_DEPTH == 2  // A/B: two tiles ahead (state of tiles g + 1 and g + 2 in registers while tile g is advanced)
    AdvState<CPL> st[3];
    Params Ps[3];
    prefetch(i0, st[0], Ps[0]);
    if (SPG > 1) prefetch(i0 + perBlock, st[1], Ps[1]);
#pragma unroll
    for (int g = 0; g < SPG; ++g) {
      const int64_t i = i0 + (int64_t)g * perBlock;
      if (g + 2 < SPG) prefetch(i + 2 * perBlock, st[(g + 2) % 3], Ps[(g + 2) % 3]);
      AdvState<CPL>& cur = st[g % 3];
      cur.live = i < a.N && cur.t < a.tEnd;
      if (i < a.N) {
        const LpsOps<RHS, false, CPL> ops{Ps[g % 3], ys, es, c};
        stillActive |= adv_advance<METHOD, false, MULTI>(a, ops, i, i * a.ivpStride + c * a.compStride, c == 0, cur);
      }
    }
#else
    AdvState<CPL> cur, nxt;
    Params Pcur, Pnxt;
    prefetch(i0, cur, Pcur);
#pragma unroll
    for (int g = 0; g < SPG; ++g) {
      const int64_t i = i0 + (int64_t)g * perBlock;
      if (g + 1 < SPG) prefetch(i + perBlock, nxt, Pnxt);
      cur.live = i < a.N && cur.t < a.tEnd;  // :511
      if (i < a.N) {
        const LpsOps<RHS, false, CPL> ops{Pcur, ys, es, c};
        stillActive |= adv_advance<METHOD, false, MULTI>(a, ops, i, i * a.ivpStride + c * a.compStride, c == 0, cur);
      }
      if (g + 1 < SPG) { cur = nxt; Pcur = Pnxt; }
    }
#endif
    if (a.active) {
      if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
    }
    return;
  }
#endif
  int64_t idx[SPG];
  AdvState<CPL> st[SPG];
#pragma unroll
  for (int g = 0; g < SPG; ++g) {
    idx[g] = ((int64_t)blockIdx.x * SPG + g) * perBlock + sysInBlock;
    st[g].live = false;
    if (idx[g] < a.N) {
      const Params P = params_of(a, idx[g]);
      const LpsOps<RHS, false, CPL> ops{P, lds + sysInBlock * lps_stride<DIM>(), lds + lps_lds_doubles<DIM, CPL>() / 2 + sysInBlock * lps_stride<DIM>(), c};
      adv_fetch<METHOD, false, decltype(ops), NNHIP_ADV_LPS_SPECULATE != 0>(a, ops, idx[g], idx[g] * a.ivpStride + c * a.compStride, st[g]);
    }
  }
#pragma unroll
  for (int g = 0; g < SPG; ++g) {
    if (idx[g] < a.N) {
      const Params P = params_of(a, idx[g]);
      const LpsOps<RHS, false, CPL> ops{P, lds + sysInBlock * lps_stride<DIM>(), lds + lps_lds_doubles<DIM, CPL>() / 2 + sysInBlock * lps_stride<DIM>(), c};
      stillActive |= adv_advance<METHOD, false, MULTI>(a, ops, idx[g], idx[g] * a.ivpStride + c * a.compStride, c == 0, st[g]);
    }
  }
  if (a.active) {
    if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
  }
}

// ---- the same loop iteration for the layout the streaming driver itself sets up (round 5) --------------------------------------------
// nnhip_ode_adaptive_stream_f64_dev's default state: updated in place, (t, dt) side by side, FSAL re-evaluated (DOPRI54 / Tsit54), no error /
// step-count columns, no per-IVP parameters, AoS systems back to back (lanes-per-system form) or SoA planes (thread-per-IVP form).  The general
// kernels above decide all of that per launch from StepArgs — strides, four state pointers, three nullable columns, two (t, dt) forms — and pay
// for it in every wave: 64-bit address arithmetic per array (v_lshl_add_u64, v_mad_u64_u32), pointer selects (v_cndmask), and so many live
// scalars that EXEC masks of the nested accept / reject branches are spilled through v_writelane / v_readlane.  Here the layout is the
// kernel's contract: ONE uniform 64-bit base per block in SGPRs + a 32-bit lane offset (global_load/store ... v_off, s[base]), seven scalar
// arguments.  The arithmetic is the same inlined code (ops.rhs, embedded_step, the controller lines :525-541), hence the same bits.
struct AdvLeanArgs {
  double* y;        // [N][DIM] (lanes-per-system) / [DIM][N] (thread-per-IVP), advanced in place
  double2* td;      // (t, dt) of IVP i
  int64_t N;
  double tEnd;
  StepCtl ctl;
  Params P;         // scalars only (ivp / aux == nullptr)
  unsigned int* active;  // nullable, see StepArgs::active
};
NNHIP_DEV void pin_lean_args(const AdvLeanArgs& a) {
  NNHIP_PIN_SGPR64(a.y); NNHIP_PIN_SGPR64(a.td); NNHIP_PIN_SGPR64(a.N); NNHIP_PIN_SGPR64(a.tEnd);
  NNHIP_PIN_SGPR64(a.ctl.absTol); NNHIP_PIN_SGPR64(a.ctl.relTol); NNHIP_PIN_SGPR64(a.ctl.dtMax); NNHIP_PIN_SGPR64(a.ctl.dtMin);
  NNHIP_PIN_SGPR64(a.active);
}
// ode.nim:525-541 on registers: (t, dt, y) -> (t', dt', yNew); returns t' < tEnd.  FSAL = f(t, y) is evaluated here (adv_fsal_in_hbm).
template <int METHOD, class Ops>
NNHIP_DEV bool adv_lean_iteration(const Ops& ops, const StepCtl& ctl, double tEnd, double& t, double& dt, const double (&y)[Ops::D], double (&yNew)[Ops::D]) {
  static_assert(METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54, "the lean form re-evaluates FSAL: methods whose last stage is f(t + dt, yNew)");
  double fsal[Ops::D];
  ops.rhs(t, y, fsal);
  dt = nmin(dt, tEnd - t);                                               // :525
  double error = 0.0, factor;
  int64_t rej = 0;
  embedded_step<METHOD, NNHIP_PEEL_STREAM != 0>(ops, t, dt, y, fsal, fsal, yNew, error, ctl, rej, factor);  // :531
  t += dt;                                                               // :532
  if (error == 0.0) dt *= 5.0;                                           // :534-535
  else dt = dt * factor;                                                 // :537
  if (dt < ctl.dtMin) dt = ctl.dtMin;                                    // :538-539
  else if (ctl.dtMax < dt) dt = ctl.dtMax;                               // :540-541
  if (error != error) t = tEnd;  // NaN abort (same deviation as the fused driver): retire the IVP
  return t < tEnd;
}
// "anyone short of tEnd?" — asked only by the launch that closes a polling group (a.active != nullptr, uniform): that launch answers like the general kernels,
// ONE store per workgroup after an OR over its waves, into its slot of the host-mapped flag block.  The launches in between (100 of BASELINE's 104) skip the
// reduction's ~20 VALU instructions, its two LDS operations and the barrier that keeps a workgroup's waves from retiring on their own.
#ifndef NNHIP_ADV_LEAN_REPORT  // A/B hook: 0 = every lane still integrating stores the 1 itself (one merged store per wavefront, no barrier; round 5's form)
#define NNHIP_ADV_LEAN_REPORT 1
#endif
NNHIP_DEV void adv_lean_report(const AdvLeanArgs& a, unsigned int stillActive) {
#if NNHIP_ADV_LEAN_REPORT
  if (a.active) {
    if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
  }
#else
  if (a.active && stillActive) a.active[blockIdx.x % kAggSlots] = 1u;
#endif
}
#ifdef NNHIP_ADV_LEAN_WPE
#define NNHIP_ADV_LEAN_ATTR __attribute__((amdgpu_waves_per_eu(NNHIP_ADV_LEAN_WPE)))
#else
#define NNHIP_ADV_LEAN_ATTR
#endif
template <int METHOD, class RHS, int CPL>
__global__ __launch_bounds__(kBlock) NNHIP_ADV_LEAN_ATTR void advance_lps_lean_kernel(const AdvLeanArgs a) {
  constexpr int DIM = RHS::dim;
  constexpr int LPSYS = DIM / CPL;
  constexpr int perBlock = kBlock / LPSYS;
  static_assert(DIM % CPL == 0 && 64 % LPSYS == 0 && CPL % 2 == 0 && RhsSize<RHS>::value == DIM, "whole systems inside a wavefront, 16-byte lane accesses");
  __shared__ double lds[lps_lds_doubles<DIM, CPL>()];  // (untouched — and dropped — for banded right-hand sides with the register-chain norm)
  controller_prologue();
  pin_lean_args(a);
  const unsigned int tid = threadIdx.x;
  const unsigned int sysInBlock = tid / LPSYS;
  const int c = (int)(tid % LPSYS) * CPL;
  const int64_t sys0 = (int64_t)blockIdx.x * perBlock;                   // uniform
  const int64_t left = a.N - sys0;
  const unsigned int nHere = left < (int64_t)perBlock ? (unsigned int)left : (unsigned int)perBlock;  // uniform, >= 1
  double* const yb = a.y + sys0 * DIM;                                    // uniform: the block's systems, back to back
  double2* const tb = a.td + sys0;
  const bool in = sysInBlock < nHere;
  const unsigned int so = in ? sysInBlock : 0u;                           // lanes past the batch read (never write) the block's first system
  const unsigned int yo = in ? tid * (unsigned int)CPL : (unsigned int)c; // = so * DIM + c
  const double2 td = tb[so];
  double y[CPL];
#pragma unroll
  for (int j = 0; j < CPL; j += 2) {
    const double2 v = *reinterpret_cast<const double2*>(&yb[yo + j]);
    y[j] = v.x; y[j + 1] = v.y;
  }
  double t = td.x, dt = td.y;
  // every load of the system in ONE round trip, above the branch on `t` (left alone the compiler sinks the state loads below it: two serialized
  // memory latencies per wave, and the kernel's waves start together and stay in phase).  The general kernel's tile prefetch does the same.
#pragma unroll
  for (int j = 0; j < CPL; ++j) NNHIP_KEEP_VGPR(y[j]);
  NNHIP_KEEP_VGPR(t); NNHIP_KEEP_VGPR(dt);
  unsigned int stillActive = 0;
  if (in && t < a.tEnd) {                                                 // :511
    const LpsOps<RHS, false, CPL> ops{a.P, lds + sysInBlock * lps_stride<DIM>(), lds + lps_lds_doubles<DIM, CPL>() / 2 + sysInBlock * lps_stride<DIM>(), c};
    double yNew[CPL];
    stillActive = adv_lean_iteration<METHOD>(ops, a.ctl, a.tEnd, t, dt, y, yNew) ? 1u : 0u;
#pragma unroll
    for (int j = 0; j < CPL; j += 2) *reinterpret_cast<double2*>(&yb[yo + j]) = make_double2(yNew[j], yNew[j + 1]);
    if (c == 0) tb[so] = make_double2(t, dt);
  }
  adv_lean_report(a, stillActive);
}
// thread-per-IVP form: SoA planes y[c * N + i]; a block's part of plane c starts at y + c * N + blockIdx.x * blockDim.x (uniform)
template <int METHOD, class RHS>
__global__ __launch_bounds__(kBlock) NNHIP_ADV_TPI_ATTR void advance_tpi_lean_kernel(const AdvLeanArgs a) {
  constexpr int D = RHS::dim;
  controller_prologue();
  pin_lean_args(a);
  const unsigned int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * blockDim.x;                    // uniform
  const int64_t left = a.N - i0;
  const bool in = (int64_t)tid < left;
  const unsigned int o = in ? tid : 0u;
  double* const yb = a.y + i0;
  double2* const tb = a.td + i0;
  const double2 td = tb[o];
  double y[D];
#pragma unroll
  for (int c = 0; c < D; ++c) y[c] = (yb + (int64_t)c * a.N)[o];
  double t = td.x, dt = td.y;
  // (t, dt) AND the state in one round trip, above the branch on `t`, as in the lanes-per-system kernel: behind the branch the state loads wait for t to
  // arrive — two dependent memory latencies per wave in a kernel whose waves start together and stay in phase (streamed C3).
  // Finished IVPs then read (never write) their state: in a BASELINE loop that happens in the launches after the last step only.
#pragma unroll
  for (int c = 0; c < D; ++c) NNHIP_KEEP_VGPR(y[c]);
  NNHIP_KEEP_VGPR(t); NNHIP_KEEP_VGPR(dt);
  unsigned int stillActive = 0;
  if (in && t < a.tEnd) {                                                 // :511
    double yNew[D];
    const TpiOps<RHS, false> ops{a.P};
    stillActive = adv_lean_iteration<METHOD>(ops, a.ctl, a.tEnd, t, dt, y, yNew) ? 1u : 0u;
#pragma unroll
    for (int c = 0; c < D; ++c) (yb + (int64_t)c * a.N)[o] = yNew[c];
    tb[o] = make_double2(t, dt);
  }
  adv_lean_report(a, stillActive);
}
#if !NNHIP_RTC
// Does this launch have the layout the lean kernels are written for?  (Everything the streaming driver's default set-up produces.)
#ifndef NNHIP_ADV_LEAN
#define NNHIP_ADV_LEAN 1
#endif
inline bool adv_lean_layout_ok(const StepArgs& a, int dim, bool aos) {  // the launch record describes exactly the state the lean kernels are written for
  if (!a.recomputeFsal || a.stepsPerLaunch > 1 || a.nontemporal) return false;
  if (a.y_in != a.y_out || a.dt_io || !a.t_io || a.error || a.steps_io || a.perIvpParams || a.P.ivp || a.P.aux || a.tReq) return false;
  if ((((uintptr_t)a.y_in | (uintptr_t)a.t_io) & 15) != 0) return false;
  if (aos) return a.compStride == 1 && a.ivpStride == dim;
  return a.ivpStride == 1 && a.compStride == a.N;
}
template <int METHOD>
inline bool adv_lean_applies(const StepArgs& a, int dim, bool aos) {
  if constexpr (!(METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54)) return false;
  else return NNHIP_ADV_LEAN && !a.noLean && adv_lean_layout_ok(a, dim, aos);
}
inline AdvLeanArgs adv_lean_args(const StepArgs& a) {
  AdvLeanArgs l{};
  l.y = a.y_out; l.td = reinterpret_cast<double2*>(a.t_io); l.N = a.N; l.tEnd = a.tEnd; l.ctl = a.ctl; l.P = a.P; l.active = a.active;
  return l;
}
#endif

// ---- adaptive streaming WITH dense output (thread-per-IVP systems) ------------------------------------------------------------
// ODESolver's loop iteration including the emission block :512-524 and the lastIter update :526-530, per IVP and launch, with
// the Hermite history (lastIter.t, .y, .dy) and denseIndex resident in HBM next to (y, FSAL, t, dt).  One kernel serves both
// directions: the backward branch g(t, y) = -f(-t, y) is a run-time flag here (exact sign flips, same bits as the compile-time
// NEG of the fused kernels).
template <class RHS>
struct TpiOpsRt {
  static constexpr int D = RHS::dim;
  static constexpr bool mutates = RhsMutates<RHS>::value;
  const Params& P;
  bool neg;
  NNHIP_DEV static constexpr bool owns(int) { return true; }
  NNHIP_DEV void rhs(double t, const double (&y)[D], double (&dy)[D]) const {
    RHS::eval(neg ? -t : t, y, dy, P);
    if (neg) {
#pragma unroll
      for (int c = 0; c < D; ++c) dy[c] = -dy[c];
    }
  }
  NNHIP_DEV double norm(const double (&yNew)[D], const double (&err_y)[D], const StepCtl& o) const {
    const TpiOps<RHS, false> base{P};
    return base.norm(yNew, err_y, o);
  }
};

// y = y0, FSAL = f(t0, y0) / g(-t0, y0) (:506,:546), t, dt, denseIndex = 0 (lastIter (:498,:548) lives in the advance kernel's registers)
template <class RHS>
__global__ __launch_bounds__(kBlock) void advance_dense_init_kernel(const StepArgs a, const double* __restrict__ y0, double tStartEff, double dtInit) {
  constexpr int D = RHS::dim;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.N) return;
  const Params P = params_of(a, i);
  const TpiOpsRt<RHS> ops{P, a.negate != 0};
  const int64_t base = i * a.ivpStride;
  double y[D], f[D];
#pragma unroll
  for (int c = 0; c < D; ++c) y[c] = y0[base + c * a.compStride];
  ops.rhs(tStartEff, y, f);
#pragma unroll
  for (int c = 0; c < D; ++c) {
    a.y_out[base + c * a.compStride] = y[c];
    a.fsal_out[base + c * a.compStride] = f[c];
  }
  reinterpret_cast<double2*>(a.t_io)[i] = make_double2(tStartEff, dtInit);  // (t, dt) side by side, see advance_dense_body
  a.denseIdx_io[i] = 0;
}

// The emission block of ODESolver's loop (:511-524) for the step from lastIter = (lastT, yOld, lastDy) to (t, yNew, dyNow): every requested
// time up to t is interpolated and stored; returns the new denseIndex, `done` when the last requested time has been emitted.
template <int D>
struct DenseEmitIn {
  double t, lastT, treq, treqNext;
  double yOld[D], yNew[D], lastDy[D], dyNow[D];
  int denseIndex, rowBase;
  int64_t i, base;
  bool neg, lead;
};
template <class OPS>
NNHIP_DEV int dense_emit(const StepArgs& a, const OPS& ops, const DenseEmitIn<OPS::D>& e, bool& done) {
  constexpr int D = OPS::D;
  const int high = a.nReq - 1;
  const int64_t cs = a.compStride;
  int denseIndex = e.denseIndex;
  double treq = e.treq, treqNext = e.treqNext;
  while (treq <= e.t) {  // :515
    const HermiteW w = hermite_weights(treq, e.lastT, e.t);
    const int64_t r = e.neg ? (int64_t)(a.nReq - 1 - denseIndex) : (int64_t)(e.rowBase + a.rowBase0 + denseIndex);
#pragma unroll
    for (int c = 0; c < D; ++c)
      if (ops.owns(c))  // written once, never read back by the loop: kept out of the caches the state lives in
        st_state<true>(hermite_apply(w, e.yOld[c], e.yNew[c], e.lastDy[c], e.dyNow[c]), &a.rows[r * a.rowStride + e.base + c * cs]);
    denseIndex += 1;
    if (high < denseIndex) { done = true; break; }  // :513-514 / :523-524: the loop ends without another step
    treq = treqNext;
    if (treq <= e.t && denseIndex < high) treqNext = e.neg ? -a.tReq[denseIndex + 1] : a.tReq[denseIndex + 1];
  }
  if (e.lead) a.denseIdx_io[e.i] = denseIndex;
  return denseIndex;
}
// One loop iteration of one IVP; `ops` owns D of its components (all of them: thread-per-IVP; CPL: lanes-per-system — the lanes
// of a system read the same per-IVP scalars, so they take every branch together; the `lead` lane stores the scalars).
// A launch runs the reference's loop body from `dt = min(dt, tEnd - t)` (:525) through the controller (:541) AND the top of the NEXT
// iteration (:511-524: the requested times the step just taken has passed) — the same statements in the same order per IVP, cut one
// third of an iteration later than the reference's `while`.  At that cut both ends of the step are in registers, so lastIter =
// (t, y, dy) (:526-530) never goes to HBM: a launch moves the 8*(4d+4) bytes of the state plus denseIndex, and 8*d per emitted row.
// `emitAfter == 0` (the last launch a caller's max_launches permits) leaves the emission to the iteration that never comes, as the
// fused solve does when max_steps ends its loop.
// NT: non-temporal hint on the streamed state arrays, chosen by the host from the working-set size like the loop without dense
// output (a template parameter: see StepArgs::nontemporal)
template <int METHOD, bool NT = false, bool SPEC = false, class OPS>
NNHIP_DEV unsigned int advance_dense_body(const StepArgs& a, const OPS& ops, int64_t i, int64_t base, bool lead, bool neg) {
  constexpr int D = OPS::D;
  using MT = MethodTraits<METHOD>;
  // the loop variables (t, dt) of an IVP sit side by side in the driver's workspace (t_io = [N][2]): one 16-byte access each way
  double2* const td_io = reinterpret_cast<double2*>(a.t_io) + i;
  // denseIndex and the requested times it points at are only needed after the step; fetched first (with t, in one round trip), the
  // dependent read of the times passes under the state loads and the stage arithmetic instead of at the tail of every wave
  int denseIndex = a.useDense ? a.denseIdx_io[i] : 0;  // <= nReq - 1: an IVP whose last requested time has been emitted is retired below
  const double2 td = *td_io;
  double t = td.x;
  // DOPRI54 / Tsit54 re-evaluate FSAL instead (StepArgs::recomputeFsal); RK21 never reads the slot; BS32's stepper does not either, but the
  // slot it returns (k4 = f(t + dt, yNew)) is lastIter.dy of the dense output (MethodTraits::fsal), so here it travels
  const bool withFsal = METHOD == NNHIP_BS32 ? true : adv_fsal_in_hbm<METHOD>(a);
  double y[D], yNew[D], fsal[D];
  // the state through the loop's own accessors (16-byte accesses where a lane's components are contiguous).  SPEC (lanes-per-system kernel): issued
  // with t, in one round trip, as advance_lps_kernel's tile fetch does — a finished system then reads its state for nothing
  if constexpr (SPEC) adv_load_state<NT>(a, ops, base, y, fsal, withFsal);
  if (!(t < a.tEnd)) return 0u;  // :511
  if constexpr (!SPEC) adv_load_state<NT>(a, ops, base, y, fsal, withFsal);
  if constexpr (METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54) {
    if (a.recomputeFsal) ops.rhs(t, y, fsal);
  }
  double dt = td.y;
  const int high = a.nReq - 1;
  // A step rarely passes more than one requested time: the next two are fetched.  Neighbouring IVPs are mostly at the same denseIndex:
  // a wave that agrees on it reads the two times through the scalar cache into SGPRs (no vector-memory instruction — what this
  // HBM-bound kernel is short of — and no VGPRs held across the stages); a wave that does not gathers them per lane after the step.
  bool uniformIdx = false;
  double sTreq = 0.0, sTreqNext = 0.0;
  int rowBase = 0;
  if (a.useDense) {
    const int idx0 = __builtin_amdgcn_readfirstlane(denseIndex);
    uniformIdx = __all(denseIndex == idx0) != 0;
    sTreq = a.tReq[idx0];
    sTreqNext = a.tReq[idx0 < high ? idx0 + 1 : high];
    if (a.rowBase) rowBase = a.rowBase[i];
  }
  dt = nmin(dt, a.tEnd - t);  // :525
  // lastIter = (t, y, dy) (:526-530), in registers: dy = FSAL for FSAL methods; f(t, y) otherwise, evaluated only if a requested
  // time falls into this step (the same call on the same operands, hence the same bits)
  const double lastT = t;
  DenseEmitIn<D> e;
  if constexpr (MT::fsal) {
#pragma unroll
    for (int c = 0; c < D; ++c) e.lastDy[c] = fsal[c];
  } else if constexpr (OPS::mutates) {
    if (a.useDense) ops.rhs(t, y, e.lastDy);  // a mutating f: lastIter.dy = f(t, y, ctx) once per step, as the reference calls it (:530)
  }
  double error = 0.0, factor;
  int64_t rej = 0;
  embedded_step<METHOD, NNHIP_PEEL_STREAM != 0>(ops, t, dt, y, fsal, fsal, yNew, error, a.ctl, rej, factor);  // :531
  t += dt;                                                                      // :532
  if (error == 0.0) dt *= 5.0;
  else dt = dt * factor;
  if (dt < a.ctl.dtMin) dt = a.ctl.dtMin;
  else if (a.ctl.dtMax < dt) dt = a.ctl.dtMax;
  if (error != error) t = a.tEnd;  // NaN abort, as in the fused driver
  bool done = false;
#ifndef NNHIP_DENSE_NO_FENCE
  // everything the emission block derives from denseIndex / rowBase (row addresses, the index of the next requested time) is computed
  // AFTER the stages: left to the compiler it is hoisted to the top of the kernel and held in VGPRs across them
  NNHIP_KEEP_VGPR(denseIndex); NNHIP_KEEP_VGPR(rowBase);
#endif
  if (a.useDense && a.emitAfter && t < a.tEnd) {  // :511-524 of the next iteration
    double treq = sTreq, treqNext = sTreqNext;
    if (!uniformIdx) {
      treq = a.tReq[denseIndex];
      treqNext = a.tReq[denseIndex < high ? denseIndex + 1 : high];
    }
    if (neg) { treq = -treq; treqNext = -treqNext; }
    if (treq <= t) {
      if constexpr (!MT::fsal) {
        ops.rhs(t, yNew, e.dyNow);                                // f(t, y, ctx) (:521)
        if constexpr (!OPS::mutates) ops.rhs(lastT, y, e.lastDy);  // lastIter.dy (:530), lazily
      } else {
#pragma unroll
        for (int c = 0; c < D; ++c) e.dyNow[c] = fsal[c];
      }
#pragma unroll
      for (int c = 0; c < D; ++c) { e.yOld[c] = y[c]; e.yNew[c] = yNew[c]; }
      e.t = t; e.lastT = lastT; e.treq = treq; e.treqNext = treqNext;
      e.denseIndex = denseIndex; e.rowBase = rowBase; e.i = i; e.base = base; e.neg = neg; e.lead = lead;
      [[maybe_unused]] const int firstEmitted = denseIndex;
      denseIndex = dense_emit(a, ops, e, done);
      if constexpr (!MT::fsal && OPS::mutates) {  // a mutating f: f(t, y, ctx) once per emitted point (:521) — the first was made above
        for (int q = firstEmitted + 1; q < denseIndex; ++q) { double again[D]; ops.rhs(t, yNew, again); }
      }
    }
  }
  adv_store_state<NT>(a, ops, base, yNew, fsal, withFsal);
  // every requested time has been emitted: the reference leaves the loop here and its final yPositive.add(y) falls outside the
  // requested rows.  Retire the IVP (denseIndex == nReq tells the finalize kernel that nothing is left to add).
  if (done) t = a.tEnd;
  if (lead) {
    *td_io = make_double2(t, dt);
    if (a.error) a.error[i] = error;
    if (a.steps_io) a.steps_io[i] += 1;
  }
  return t < a.tEnd ? 1u : 0u;
}

// Workgroup size: one wave, as launch_advance_tpi's default (one-wave workgroups retire and refill sooner)
template <bool NT>
constexpr int adv_dense_block() { return 64; }
// Held to 4 waves per SIMD like advance_tpi_kernel (NNHIP_ADV_TPI_ATTR).  Unconstrained the kernel sits at 145-147 VGPRs for d = 3 (3 waves);
// what kept it from fitting 128 was not what the emission keeps alive (parking lastIter in LDS, re-reading it in the emission branch,
// compiling the block as a function of its own, fencing its address arithmetic behind the stages: all 50-600 B of scratch at 4 waves) but
// SGPR pressure: with the emission block's extra uniform arguments, pin_step_args' pinned kernel arguments leave the allocator 52 B of
// scratch per lane at 4 waves; without the pins 12 B (2 VGPRs), like the loop without dense output.  1e7 Lorenz IVPs: 205-223 -> 196-200 us
// per launch (n_t = 2 / 11), 270 -> 242 us with a row emitted per launch.
template <int METHOD, class RHS, bool NT = false>
__global__ __launch_bounds__(kBlock) NNHIP_ADV_TPI_ATTR void advance_dense_tpi_kernel(const StepArgs a) {
  static_assert(MethodTraits<METHOD>::adaptive, "adaptive methods only");
  controller_prologue();
  // (no pin_step_args here: with the extra uniform arguments of the emission block the pinned SGPRs push the allocator from 12 to 52 B of scratch at 4 waves per SIMD)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // workgroup size chosen by the launcher (<= kBlock)
  unsigned int stillActive = 0;
  if (i < a.N) {
    const Params P = params_of(a, i);
    const bool neg = a.negate != 0;
    const TpiOpsRt<RHS> ops{P, neg};
    stillActive = advance_dense_body<METHOD, NT>(a, ops, i, i * a.ivpStride, true, neg);
  }
  if (a.active) {
    if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
  }
}

// Lanes-per-system form (Vector[float] states of 8 / 16 / 32 ... components, and run-time compiled per-component systems): the
// direction is a kernel-wide flag, so the two compile-time variants of LpsOps are selected by one wave-uniform branch.
template <class RHS, int CPL>
struct LpsOpsRt {
  static constexpr int D = CPL;
  static constexpr bool mutates = RhsMutates<RHS>::value;
  LpsOps<RHS, false, CPL> f;
  LpsOps<RHS, true, CPL> b;
  bool neg;
  NNHIP_DEV bool owns(int j) const { return f.owns(j); }
  NNHIP_DEV void rhs(double t, const double (&y)[CPL], double (&dy)[CPL]) const {
    if (neg) b.rhs(t, y, dy);
    else f.rhs(t, y, dy);
  }
  NNHIP_DEV double norm(const double (&yNew)[CPL], const double (&err_y)[CPL], const StepCtl& o) const { return f.norm(yNew, err_y, o); }
};

#ifndef NNHIP_ADV_DENSE_LPS_SPECULATE
#define NNHIP_ADV_DENSE_LPS_SPECULATE 1
#endif
template <int METHOD, class RHS, int CPL = 1>
__global__ __launch_bounds__(kBlock) void advance_dense_lps_kernel(const StepArgs a) {
  static_assert(MethodTraits<METHOD>::adaptive, "adaptive methods only");
  constexpr int DIM = RHS::dim;
  constexpr int LPSYS = DIM / CPL;
  static_assert(DIM % CPL == 0 && 64 % LPSYS == 0, "a system must not straddle wavefronts");
  __shared__ double lds[lps_lds_doubles<DIM, CPL>()];
  controller_prologue();
  const int sysInBlock = threadIdx.x / LPSYS, c = (threadIdx.x % LPSYS) * CPL;
  constexpr int perBlock = kBlock / LPSYS;
  const int64_t i = (int64_t)blockIdx.x * perBlock + sysInBlock;
  unsigned int stillActive = 0;
  if (i < a.N) {
    const Params P = params_of(a, i);
    const bool neg = a.negate != 0;
    double* ys = lds + sysInBlock * lps_stride<DIM>();
    double* es = lds + lps_lds_doubles<DIM, CPL>() / 2 + sysInBlock * lps_stride<DIM>();
    const LpsOpsRt<RHS, CPL> ops{{P, ys, es, c}, {P, ys, es, c}, neg};
    stillActive = advance_dense_body<METHOD, false, NNHIP_ADV_DENSE_LPS_SPECULATE != 0>(a, ops, i, i * a.ivpStride + c * a.compStride, c == 0, neg);
  }
  if (a.active) {
    if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
  }
}

// After a direction's loop: yPositive.add(y) / yNegative.add(y) (:542,:584) at row denseIndex if it is still a requested row, the
// backward rows closed up when fewer than requested were produced (reference quirk), and the per-IVP row count.
//   mode 0: backward direction finished   (ny[i] = rows produced)
//   mode 1: `t0 in tspan` row             (rows[ny[i]] = y0; ny[i] += 1)
//   mode 2: forward direction finished    (ny[i] += rows produced)
//   mode 3: NaN fill of rows ny[i] .. n_t-1
// When both directions are asked for, the forward one runs FIRST (as in the reference, ode.nim:508-542 then :544-584 — observable only through
// a mutable ctx), its rows placed where they belong if the backward branch returns all of its rows:
//   mode 5: forward direction finished first (rows at rowBase0 + k; fwd[i] = rows produced)
//   mode 4: after the backward rows and the t0 row: move the forward rows up if the backward branch returned fewer rows; ny[i] += fwd[i]
template <int DIM_UNUSED = 0>
__global__ __launch_bounds__(kBlock) void advance_dense_finalize_kernel(const StepArgs a, int mode, int dim, const double* __restrict__ y0, int32_t* __restrict__ ny,
                                                                        int n_t, int32_t* __restrict__ fwd = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.N) return;
  const int64_t base = i * a.ivpStride, cs = a.compStride, rs = a.rowStride;
  if (mode == 5) {
    const int k = a.denseIdx_io[i];
    if (k < a.nReq)
      for (int c = 0; c < dim; ++c) a.rows[(int64_t)(a.rowBase0 + k) * rs + base + c * cs] = a.y_in[base + c * cs];
    fwd[i] = k + 1 < a.nReq ? k + 1 : a.nReq;
    return;
  }
  if (mode == 4) {
    const int rb = ny[i], m = fwd[i];
    if (rb != a.rowBase0)
      for (int j = 0; j < m; ++j)
        for (int c = 0; c < dim; ++c) a.rows[(int64_t)(rb + j) * rs + base + c * cs] = a.rows[(int64_t)(a.rowBase0 + j) * rs + base + c * cs];
    ny[i] = rb + m;
    return;
  }
  if (mode == 0 || mode == 2) {
    const int k = a.denseIdx_io[i];
    const int produced = k + 1 < a.nReq ? k + 1 : a.nReq;
    if (mode == 0) {
      if (k < a.nReq)
        for (int c = 0; c < dim; ++c) a.rows[(int64_t)(a.nReq - 1 - k) * rs + base + c * cs] = a.y_in[base + c * cs];
      if (produced < a.nReq) {
        const int shift = a.nReq - produced;
        for (int j = 0; j < produced; ++j)
          for (int c = 0; c < dim; ++c) a.rows[(int64_t)j * rs + base + c * cs] = a.rows[(int64_t)(j + shift) * rs + base + c * cs];
      }
      ny[i] = produced;
    } else {
      const int rb = ny[i];
      if (k < a.nReq)
        for (int c = 0; c < dim; ++c) a.rows[(int64_t)(rb + k) * rs + base + c * cs] = a.y_in[base + c * cs];
      ny[i] = rb + produced;
    }
  } else if (mode == 1) {
    const int rb = ny[i];
    for (int c = 0; c < dim; ++c) a.rows[(int64_t)rb * rs + base + c * cs] = y0[base + c * cs];
    ny[i] = rb + 1;
  } else {
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    for (int j = ny[i]; j < n_t; ++j)
      for (int c = 0; c < dim; ++c) a.rows[(int64_t)j * rs + base + c * cs] = qnan;
  }
}

// (t, dt)[i] = (t0, dt0), the dense driver's interleaved form
template <int UNUSED = 0>
__global__ __launch_bounds__(kBlock) void fill_td_kernel(double2* __restrict__ td, int64_t n, double t0, double dt0) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) td[i] = make_double2(t0, dt0);
}

// t[i] = t0, dt[i] = dt0: the per-IVP loop variables before the first iteration (ode.nim:477, 491-493)
template <int UNUSED = 0>  // a template only so that the header can be included by every translation unit
__global__ __launch_bounds__(kBlock) void fill_t_dt_kernel(double* __restrict__ t, double* __restrict__ dt, int64_t n, double t0, double dt0) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i < n) { t[i] = t0; dt[i] = dt0; }
}

#if !NNHIP_RTC
template <int METHOD, class RHS>
hipError_t launch_advance_tpi(const StepArgs& a, int block, hipStream_t s) {
  if constexpr (MethodTraits<METHOD>::adaptive) {
    const int bs = (block == 64 || block == 128) ? block : kBlock;  // tuning knob "adv_block"
    const int64_t grid = (a.N + bs - 1) / bs;
    if (grid <= 0) return hipSuccess;
    if constexpr (METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54) {
      if (adv_lean_applies<METHOD>(a, RHS::dim, false)) return launch_kernel(advance_tpi_lean_kernel<METHOD, RHS>, dim3((unsigned)grid), dim3(bs), s, adv_lean_args(a));
    }
    if (a.stepsPerLaunch > 1) return launch_kernel(advance_tpi_kernel<METHOD, RHS, false, true>, dim3((unsigned)grid), dim3(bs), s, a);
    if (a.nontemporal) return launch_kernel(advance_tpi_kernel<METHOD, RHS, true>, dim3((unsigned)grid), dim3(bs), s, a);
    return launch_kernel(advance_tpi_kernel<METHOD, RHS, false>, dim3((unsigned)grid), dim3(bs), s, a);
  } else {
    return hipErrorInvalidValue;
  }
}
#endif

#if !NNHIP_RTC
// components per lane of the lanes-per-system advance kernel, from the fused kernels' value `ca`
// (A/B hook: -DNNHIP_ADV_CPL_MAX=1|2|4 caps it)
#ifndef NNHIP_ADV_CPL_MAX  // measured, 1e6 16-component systems: round 2 (two tiles per block, error column, FSAL carried) 1 -> 200 us, 2 -> 152 us, 4 -> 170 us
#define NNHIP_ADV_CPL_MAX 4  // per iteration: with one component per lane the controller (norm, division, sqrt, pow: ~200 VALU) runs 16x per system.  Round 3
#endif                       // (one tile per block at 4 per lane, see adv_lps_spg): FSAL carried 2 -> 118-120 us, 4 -> 108 us; re-evaluated 2 -> 106 us, 4 -> 88.5-91 us
#define NNHIP_ADV_CPL(ca) ((ca) < NNHIP_ADV_CPL_MAX ? (ca) : NNHIP_ADV_CPL_MAX)
// CPLR: components per lane when FSAL is re-evaluated instead of carried (StepArgs::recomputeFsal).  The launch then moves 8*(2d+4) bytes
// per step instead of 8*(4d+4) and is no longer bound by HBM but by the VALU, where halving the replicated controller arithmetic once more
// pays: 1e6 x 16, Tsit54: 2 per lane 106-108 us, 4 per lane 96.6-97.3 us, 8 per lane 129 us (tools/microbench/mb_adv c4quick).
template <int METHOD, class RHS, int CPL, int CPLR = CPL>
hipError_t launch_advance_lps(const StepArgs& a, int block, hipStream_t s) {
  if constexpr (CPLR != CPL && (METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54)) {
    if (a.recomputeFsal) return launch_advance_lps<METHOD, RHS, CPLR, CPLR>(a, block, s);
  }
  if constexpr (MethodTraits<METHOD>::adaptive) {
    constexpr int perBlock = kBlock / (RHS::dim / CPL) * adv_lps_spg(CPL);
    const int64_t grid = (a.N + perBlock - 1) / perBlock;
    if (grid <= 0) return hipSuccess;
    if constexpr ((METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54) && CPL % 2 == 0 && adv_lps_spg(CPL) == 1 && RhsSize<RHS>::value == RHS::dim) {
      if (adv_lean_applies<METHOD>(a, RHS::dim, true)) return launch_kernel(advance_lps_lean_kernel<METHOD, RHS, CPL>, dim3((unsigned)grid), dim3(kBlock), s, adv_lean_args(a));
    }
    if (a.stepsPerLaunch > 1) return launch_kernel(advance_lps_kernel<METHOD, RHS, CPL, true>, dim3((unsigned)grid), dim3(kBlock), s, a);
    return launch_kernel(advance_lps_kernel<METHOD, RHS, CPL>, dim3((unsigned)grid), dim3(kBlock), s, a);
  } else {
    return hipErrorInvalidValue;
  }
}
#endif

#if !NNHIP_RTC
// The lean kernels on their own — what the opt-in FMA-contracted build (namespace nnhip_fast, tuning knob "fp_contract") instantiates of the streaming path:
// the two kernels BASELINE's streamed C3 / C4 run on, over the compiled-in right-hand sides.  The caller has checked adv_lean_layout_ok; a launch record of
// any other shape is refused, not reinterpreted.
template <int METHOD, class RHS>
hipError_t launch_advance_tpi_lean_only(const StepArgs& a, int block, hipStream_t s) {
  if (!adv_lean_layout_ok(a, RHS::dim, false)) return hipErrorInvalidValue;
  const int bs = (block == 64 || block == 128) ? block : kBlock;
  const int64_t grid = (a.N + bs - 1) / bs;
  if (grid <= 0) return hipSuccess;
  return launch_kernel(advance_tpi_lean_kernel<METHOD, RHS>, dim3((unsigned)grid), dim3(bs), s, adv_lean_args(a));
}
template <int METHOD, class RHS, int CPL>
hipError_t launch_advance_lps_lean_only(const StepArgs& a, int, hipStream_t s) {
  if (!adv_lean_layout_ok(a, RHS::dim, true)) return hipErrorInvalidValue;
  constexpr int perBlock = kBlock / (RHS::dim / CPL);
  const int64_t grid = (a.N + perBlock - 1) / perBlock;
  if (grid <= 0) return hipSuccess;
  return launch_kernel(advance_lps_lean_kernel<METHOD, RHS, CPL>, dim3((unsigned)grid), dim3(kBlock), s, adv_lean_args(a));
}
#endif

template <int METHOD, class RHS, bool NEG, int CPL = 1>
__global__ __launch_bounds__(kBlock) void step_lps_kernel(const StepArgs a) {
  constexpr int DIM = RHS::dim;
  constexpr int LPSYS = DIM / CPL;  // lanes per system (CPL > 1 only for systems wider than a wavefront)
  static_assert(DIM % CPL == 0 && 64 % LPSYS == 0, "a system must not straddle wavefronts");
  __shared__ double lds[lps_lds_doubles<DIM, CPL>()];
  controller_prologue<MethodTraits<METHOD>::adaptive>();
  pin_step_args(a);
  const int sysInBlock = threadIdx.x / LPSYS, c = (threadIdx.x % LPSYS) * CPL;
  const int64_t i = (int64_t)blockIdx.x * (kBlock / LPSYS) + sysInBlock;
  if (i >= a.N) return;
  const Params P = params_of(a, i);
  const LpsOps<RHS, NEG, CPL> ops{P, lds + sysInBlock * lps_stride<DIM>(), lds + lps_lds_doubles<DIM, CPL>() / 2 + sysInBlock * lps_stride<DIM>(), c};
  step_body<METHOD>(a, ops, i, i * a.ivpStride + c * a.compStride, c == 0);
}

#if !NNHIP_RTC
template <int METHOD, class RHS>
hipError_t launch_step_lps(const StepArgs& a, int negate, hipStream_t s) {
  constexpr int perBlock = kBlock / RHS::dim;
  const int64_t grid = (a.N + perBlock - 1) / perBlock;
  if (grid <= 0) return hipSuccess;
  if (negate) return launch_kernel(step_lps_kernel<METHOD, RHS, true>, dim3((unsigned)grid), dim3(kBlock), s, a);
  return launch_kernel(step_lps_kernel<METHOD, RHS, false>, dim3((unsigned)grid), dim3(kBlock), s, a);
}
#endif

// ------------------------------------------------------------------------------------------------
// Headline kernel: scalar RK4 step over a flat array of n independent float64 states, uniform (t, dt).
// Each lane moves VEC*16 bytes per direction with VEC independent 16-byte loads in flight; one
// workgroup covers a contiguous kBlock*2*VEC-element tile so every wave-level access is a full
// 1 KiB coalesced segment.  Algorithmic HBM traffic: 8 B read + 8 B written per trajectory-step.
// ------------------------------------------------------------------------------------------------
// MODE 0: plain loads/stores, one tile per workgroup.   MODE 1: non-temporal loads+stores (streaming hint).
// MODE 2: persistent grid-stride over tiles (grid = a few workgroups per CU).  MODE 3: MODE 2 + non-temporal.
// workgroup size of the headline kernel (A/B hook -DNNHIP_RK4_STREAM_BLOCK=64|128|256)
#ifndef NNHIP_RK4_STREAM_BLOCK
#define NNHIP_RK4_STREAM_BLOCK 256
#endif
constexpr int kRk4Block = NNHIP_RK4_STREAM_BLOCK;
template <class RHS1, bool NEG, int VEC, int MODE>
// The scalar arguments come first and as scalars (not structs): ode_tu_rk4_stream.hip is compiled with kernarg preloading, so
// yin, yout, n, t, dt, dt/2, dt/6 arrive in SGPRs with the wave and the first global_load needs no scalar-load round trip.
__global__ __launch_bounds__(kRk4Block) void rk4_stream_vec_kernel(const double* yin, double* yout,  // may alias (in-place stepping): no __restrict__
                                                                int64_t n, double t, double h_dt, double h_hdt, double h_dt6, const Params P) {
  static_assert(RHS1::dim == 1, "scalar RHS only");
  const Rk4Dt h{h_dt, h_hdt, h_dt6};
  constexpr bool NT = (MODE & 1) != 0;
  constexpr bool PERSIST = (MODE & 2) != 0;
  constexpr int64_t TILE = (int64_t)kRk4Block * 2 * VEC;
  const TpiOps<RHS1, NEG> ops{P};
  const int64_t nTiles = (n + TILE - 1) / TILE;
  for (int64_t tileIdx = blockIdx.x; tileIdx < nTiles; tileIdx += PERSIST ? (int64_t)gridDim.x : nTiles) {
    const int64_t tile = tileIdx * TILE;
    if (tile + TILE <= n) {  // full tile: unguarded 16-byte lane accesses, VEC independent loads in flight
      double2 v[VEC];
      const double2* src = reinterpret_cast<const double2*>(yin + tile) + threadIdx.x;
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        if constexpr (NT) {
          v[u].x = __builtin_nontemporal_load(&src[u * kRk4Block].x);
          v[u].y = __builtin_nontemporal_load(&src[u * kRk4Block].y);
        } else {
          v[u] = src[u * kRk4Block];
        }
      }
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        double a0[1] = {v[u].x}, a1[1] = {v[u].y}, r0[1], r1[1];
        rk4_step(ops, t, h, a0, r0);
        rk4_step(ops, t, h, a1, r1);
        v[u].x = r0[0];
        v[u].y = r1[0];
      }
      double2* dst = reinterpret_cast<double2*>(yout + tile) + threadIdx.x;
#pragma unroll
      for (int u = 0; u < VEC; ++u) {
        if constexpr (NT) {
          __builtin_nontemporal_store(v[u].x, &dst[u * kRk4Block].x);
          __builtin_nontemporal_store(v[u].y, &dst[u * kRk4Block].y);
        } else {
          dst[u * kRk4Block] = v[u];
        }
      }
    } else {  // ragged tail tile: scalar, bounds-checked
      for (int64_t j = tile + threadIdx.x; j < n; j += kRk4Block) {
        double a0[1] = {yin[j]}, r0[1];
        rk4_step(ops, t, h, a0, r0);
        yout[j] = r0[0];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// The same treatment for ANY fixed-step method and any thread-per-IVP system (RK4_step[Vector] etc., ode.nim:107-189) with
// uniform (t, dt): a lane advances IPL IVPs, moving them with 16-byte accesses (SoA: two neighbouring IVPs per component
// plane; AoS: the 2*dim contiguous doubles of two neighbouring IVPs), all loads issued before the arithmetic, one workgroup
// per contiguous tile of kBlock*IPL IVPs.  Half (a quarter) as many waves as step_tpi_kernel and twice as wide accesses:
// RK4 Lorenz at 1e7 IVPs went from 4.06 TB/s to the figure in DESIGN.md §6.  Algorithmic traffic 16*dim B per IVP-step.
// `yin` may alias `yout` (in-place stepping): a lane reads its own IVPs before it writes them; no __restrict__.
// ------------------------------------------------------------------------------------------------
struct FixedVecArgs {
  const double* yin;
  double* yout;
  double* fsalOut;      // nullable: fixed-step IntegratorProcs hand yNew back in the FSAL slot too (ode.nim:189)
  const double* tDev;   // nullable -> t   (per-IVP times: every IVP of the reference is its own solveODE call)
  const double* dtDev;  // nullable -> dt
  int64_t N;
  int aos;
  double t, dt;
  Params P;
};

// NT: non-temporal hint on the state accesses, chosen by the host when the arrays no longer fit the Infinity Cache (a template
// parameter: see the advance kernels).  IPL = 2 (4 was measured: no gain, profiles/r02_fixed_vec_ab.json).
template <int METHOD, class RHS, bool NEG, int IPL, bool NT = false>
__global__ __launch_bounds__(kBlock) void fixed_stream_vec_kernel(const FixedVecArgs a) {
  static_assert(!MethodTraits<METHOD>::adaptive && IPL % 2 == 0, "fixed-step methods, an even number of IVPs per lane");
  auto ld2 = [](const double* p) -> double2 {
    if constexpr (NT) { double2 r; r.x = __builtin_nontemporal_load(p); r.y = __builtin_nontemporal_load(p + 1); return r; }
    else return *reinterpret_cast<const double2*>(p);
  };
  auto st2 = [](double2 v, double* p) {
    if constexpr (NT) { __builtin_nontemporal_store(v.x, p); __builtin_nontemporal_store(v.y, p + 1); }
    else *reinterpret_cast<double2*>(p) = v;
  };
  constexpr int D = RHS::dim;
  constexpr int PAIRS = IPL / 2;
  NNHIP_PIN_SGPR64(a.yin); NNHIP_PIN_SGPR64(a.yout); NNHIP_PIN_SGPR64(a.fsalOut); NNHIP_PIN_SGPR64(a.tDev); NNHIP_PIN_SGPR64(a.dtDev);
  NNHIP_PIN_SGPR64(a.N); NNHIP_PIN_SGPR32(a.aos); NNHIP_PIN_SGPR64(a.t); NNHIP_PIN_SGPR64(a.dt);
#pragma unroll
  for (int k = 0; k < 4; ++k) NNHIP_PIN_SGPR64(a.P.p[k]);
  const TpiOps<RHS, NEG> ops{a.P};
  const int64_t N = a.N;
  const int aos = a.aos;
  const double* yin = a.yin;
  double* yout = a.yout;
  const int64_t tile = (int64_t)blockIdx.x * kBlock * IPL;
  const bool uniform = !a.tDev && !a.dtDev;
  [[maybe_unused]] const Rk4Dt h4u = rk4_dt(a.dt);
  auto one = [&](double t, double dt, const double (&y)[D], double (&r)[D]) {
    if constexpr (METHOD == NNHIP_RK4) rk4_step(ops, t, uniform ? h4u : rk4_dt(dt), y, r);
    else fixed_step<METHOD>(ops, t, dt, y, r);
  };
  if (tile + (int64_t)kBlock * IPL <= N) {
    double2 v[PAIRS][D];
    double2 tt[PAIRS], dd[PAIRS];
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
      const int64_t i = tile + (int64_t)u * 2 * kBlock + 2 * threadIdx.x;
      if (aos) {
#pragma unroll
        for (int c = 0; c < D; ++c) v[u][c] = ld2(yin + i * D + 2 * c);
      } else {
#pragma unroll
        for (int c = 0; c < D; ++c) v[u][c] = ld2(yin + (int64_t)c * N + i);
      }
      tt[u] = a.tDev ? *reinterpret_cast<const double2*>(a.tDev + i) : double2{a.t, a.t};
      dd[u] = a.dtDev ? *reinterpret_cast<const double2*>(a.dtDev + i) : double2{a.dt, a.dt};
    }
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
      double ya[D], yb[D], ra[D], rb[D];
      if (aos) {  // v[u] holds [a_0 .. a_{D-1}, b_0 .. b_{D-1}] as D double2
#pragma unroll
        for (int c = 0; c < D; ++c) {
          ya[c] = (c % 2 == 0) ? v[u][c / 2].x : v[u][c / 2].y;
          yb[c] = ((c + D) % 2 == 0) ? v[u][(c + D) / 2].x : v[u][(c + D) / 2].y;
        }
      } else {
#pragma unroll
        for (int c = 0; c < D; ++c) { ya[c] = v[u][c].x; yb[c] = v[u][c].y; }
      }
      one(tt[u].x, dd[u].x, ya, ra);
      one(tt[u].y, dd[u].y, yb, rb);
      if (aos) {
#pragma unroll
        for (int c = 0; c < D; ++c) {
          if (c % 2 == 0) v[u][c / 2].x = ra[c]; else v[u][c / 2].y = ra[c];
          if ((c + D) % 2 == 0) v[u][(c + D) / 2].x = rb[c]; else v[u][(c + D) / 2].y = rb[c];
        }
      } else {
#pragma unroll
        for (int c = 0; c < D; ++c) { v[u][c].x = ra[c]; v[u][c].y = rb[c]; }
      }
    }
#pragma unroll
    for (int u = 0; u < PAIRS; ++u) {
      const int64_t i = tile + (int64_t)u * 2 * kBlock + 2 * threadIdx.x;
      if (aos) {
#pragma unroll
        for (int c = 0; c < D; ++c) st2(v[u][c], yout + i * D + 2 * c);
        if (a.fsalOut) {
#pragma unroll
          for (int c = 0; c < D; ++c) st2(v[u][c], a.fsalOut + i * D + 2 * c);
        }
      } else {
#pragma unroll
        for (int c = 0; c < D; ++c) st2(v[u][c], yout + (int64_t)c * N + i);
        if (a.fsalOut) {
#pragma unroll
          for (int c = 0; c < D; ++c) st2(v[u][c], a.fsalOut + (int64_t)c * N + i);
        }
      }
    }
  } else {  // ragged last tile: one IVP at a time, bounds-checked
    const int64_t is = aos ? D : 1, cs = aos ? 1 : N;
    for (int64_t i = tile + threadIdx.x; i < N; i += kBlock) {
      double y1[D], r[D];
#pragma unroll
      for (int c = 0; c < D; ++c) y1[c] = yin[i * is + c * cs];
      one(a.tDev ? a.tDev[i] : a.t, a.dtDev ? a.dtDev[i] : a.dt, y1, r);
#pragma unroll
      for (int c = 0; c < D; ++c) {
        yout[i * is + c * cs] = r[c];
        if (a.fsalOut) a.fsalOut[i * is + c * cs] = r[c];
      }
    }
  }
}

// dy = f(t, y) alone over a batch (pins the RHS library; also used for user-compiled RHS)
template <class RHS>
__global__ __launch_bounds__(kBlock) void rhs_batch_kernel(int64_t N, int64_t ivpStride, int64_t compStride, double t,
                                                           const double* __restrict__ y, double* __restrict__ dy, const Params P) {
  constexpr int D = RHS::dim, SIZE = RhsSize<RHS>::value;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  double yv[D], d[D];
#pragma unroll
  for (int c = 0; c < D; ++c) yv[c] = c < SIZE ? y[i * ivpStride + c * compStride] : 0.0;
  Params Pi = P;
  if (Pi.ivp) Pi.ivp += i;
  if (Pi.aux) Pi.aux += i;
  RHS::eval(t, yv, d, Pi);
#pragma unroll
  for (int c = 0; c < SIZE; ++c) dy[i * ivpStride + c * compStride] = d[c];
}

// The rows due at the head of an iteration of ODESolver's loop (ode.nim:512-524), for the host-driven fixed-step dense streaming driver, in ONE
// pass over the two ends of the step: lastIter.dy = f(lastT, lastY) (:530), f(t, y) (:521) and hermiteSpline (utils.nim:273-279) per requested time
// — 2 state reads + 1 write per row instead of the 9 array passes of three separate launches.  neg: the backward branch's g(t, y) = -f(-t, y).
struct DenseRows {
  int n;            // rows of this launch (<= 8)
  double treq[8];   // requested times (already negated for the backward branch)
  double* out[8];   // row tensors, laid out like the state
};
template <class RHS>
__global__ __launch_bounds__(kBlock) void dense_rows_kernel(int64_t N, int64_t ivpStride, int64_t compStride, double tA, double tB, int neg,
                                                            const double* __restrict__ yA, const double* __restrict__ yB, const DenseRows r, const Params P) {
  constexpr int D = RHS::dim, SIZE = RhsSize<RHS>::value;
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  double a[D], b[D], da[D], db[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    a[c] = c < SIZE ? yA[i * ivpStride + c * compStride] : 0.0;
    b[c] = c < SIZE ? yB[i * ivpStride + c * compStride] : 0.0;
  }
  Params Pi = P;
  if (Pi.ivp) Pi.ivp += i;
  if (Pi.aux) Pi.aux += i;
  RHS::eval(neg ? -tA : tA, a, da, Pi);
  RHS::eval(neg ? -tB : tB, b, db, Pi);
  if (neg) {
#pragma unroll
    for (int c = 0; c < D; ++c) { da[c] = -da[c]; db[c] = -db[c]; }
  }
  for (int k = 0; k < r.n; ++k) {
    const HermiteW w = hermite_weights(r.treq[k], tA, tB);
    double* o = r.out[k];
#pragma unroll
    for (int c = 0; c < SIZE; ++c) o[i * ivpStride + c * compStride] = hermite_apply(w, a[c], b[c], da[c], db[c]);
  }
}

#if !NNHIP_RTC
template <class RHS>
hipError_t launch_dense_rows(int64_t N, int64_t is, int64_t cs, double tA, double tB, int neg, const double* yA, const double* yB, const DenseRows& r, const Params& P,
                             hipStream_t s) {
  const int64_t grid = (N + kBlock - 1) / kBlock;
  if (grid <= 0 || r.n <= 0) return hipSuccess;
  return launch_kernel(dense_rows_kernel<RHS>, dim3((unsigned)grid), dim3(kBlock), s, N, is, cs, tA, tB, neg, yA, yB, r, P);
}
#endif

#if !NNHIP_RTC

template <class RHS1, int VEC, int MODE>
hipError_t launch_rk4_stream_vec(const double* yin, double* yout, int64_t n, double t, double dt, const Params& P, int negate,
                                 const StreamTune& tune, hipStream_t s) {
  const int64_t per = (int64_t)kRk4Block * 2 * VEC;
  int64_t grid = (n + per - 1) / per;
  if (grid <= 0) return hipSuccess;
  if (MODE & 2) { const int64_t cap = 256LL * tune.blocksPerCU; if (grid > cap) grid = cap; }
  const Rk4Dt h{dt, 0.5 * dt, dt / 6.0};  // host IEEE double ops == the device's (this TU is built -ffp-contract=off)
  if (negate) return launch_kernel(rk4_stream_vec_kernel<RHS1, true, VEC, MODE>, dim3((unsigned)grid), dim3(kRk4Block), s, yin, yout, n, t, h.dt, h.hdt, h.dt6, P);
  return launch_kernel(rk4_stream_vec_kernel<RHS1, false, VEC, MODE>, dim3((unsigned)grid), dim3(kRk4Block), s, yin, yout, n, t, h.dt, h.hdt, h.dt6, P);
}

using FixedVecLaunchFn = hipError_t (*)(const FixedVecArgs& a, int negate, int nontemporal, hipStream_t s);
template <int METHOD, class RHS>
hipError_t launch_fixed_stream_vec(const FixedVecArgs& a, int negate, int nontemporal, hipStream_t s) {
  if constexpr (!MethodTraits<METHOD>::adaptive) {
    if (a.N <= 0) return hipSuccess;
    const int64_t per = (int64_t)kBlock * 2;
    const dim3 grid((unsigned)((a.N + per - 1) / per)), block(kBlock);
    if (negate) return launch_kernel(fixed_stream_vec_kernel<METHOD, RHS, true, 2, false>, grid, block, s, a);
    if (nontemporal) return launch_kernel(fixed_stream_vec_kernel<METHOD, RHS, false, 2, true>, grid, block, s, a);
    return launch_kernel(fixed_stream_vec_kernel<METHOD, RHS, false, 2, false>, grid, block, s, a);
  } else {
    return hipErrorInvalidValue;
  }
}

// ------------------------------------------------------------------------------------------------
// Dispatch tables: (rhs_kind, dim) -> launcher, one table per integrator (one translation unit each,
// so the method x RHS instantiations compile in parallel).
// ------------------------------------------------------------------------------------------------
#define NNHIP_FOR_EACH_TPI_RHS(X)                                                                                  \
  X(NNHIP_RHS_NEG_Y, 1, RhsNegY<1>) X(NNHIP_RHS_NEG_Y, 2, RhsNegY<2>) X(NNHIP_RHS_NEG_Y, 3, RhsNegY<3>)            \
  X(NNHIP_RHS_NEG_Y, 4, RhsNegY<4>) X(NNHIP_RHS_LINEAR, 1, RhsLinear<1>) X(NNHIP_RHS_LINEAR, 2, RhsLinear<2>)      \
  X(NNHIP_RHS_LINEAR, 3, RhsLinear<3>) X(NNHIP_RHS_LINEAR, 4, RhsLinear<4>) X(NNHIP_RHS_AFFINE_T, 1, RhsAffineT<1>) \
  X(NNHIP_RHS_AFFINE_T, 2, RhsAffineT<2>) X(NNHIP_RHS_AFFINE_T, 3, RhsAffineT<3>)                                  \
  X(NNHIP_RHS_AFFINE_T, 4, RhsAffineT<4>) X(NNHIP_RHS_LORENZ, 3, RhsLorenz) X(NNHIP_RHS_VANDERPOL, 2, RhsVanDerPol) \
  X(NNHIP_RHS_RING, 4, RhsRing<4>)

// Systems integrated by several lanes of one wavefront.  X(kind, dim, RHS type, CPL adaptive, CPL fixed-step):
// components per lane of the FUSED kernels, from the A/B on MI355X (profiles/r01_dim16_variants.txt; 1e6 16-dim
// systems, Tsit54: 1 comp/lane 10.9 ms, 2: 7.9, 4: 7.6, 8: 13.1, all-in-one-lane 7.4; Vern65: 4/lane 9.8 vs 13.9).
// Fewer lanes per system = less redundant controller work (norm, sqrt, root run once per LANE); more
// components per lane = more VGPRs (85 / 115 / 193 / 256) and less latency hiding.  The step-streaming kernels
// are HBM-bound and keep 1 component per lane (best coalescing, fewest registers).
#define NNHIP_FOR_EACH_LPS_RHS(X)                                                                  \
  X(NNHIP_RHS_RING, 8, RhsRing<8>, 2, 2) X(NNHIP_RHS_RING, 16, RhsRing<16>, 4, 2) X(NNHIP_RHS_RING, 32, RhsRing<32>, 4, 2) \
  X(NNHIP_RHS_NEG_Y, 16, RhsNegY<16>, 4, 2) X(NNHIP_RHS_LINEAR, 16, RhsLinear<16>, 4, 2) X(NNHIP_RHS_AFFINE_T, 16, RhsAffineT<16>, 4, 2)

// dim-16 systems also have a register-resident thread-per-IVP fused kernel (all 512 VGPR+AGPR of a lane, one
// wave per SIMD) and other mappings kept for A/B runs; tuning knob "dim16_variant": 0 default (components per lane
// from the table above), 1 whole system in one lane, 2 lanes-per-system 16x1, 3 / 4 = 4x4 / 16x1 with the
// wavefront-shuffle error norm (adaptive methods only).
#define NNHIP_FOR_EACH_WIDE_TPI_RHS(X) X(NNHIP_RHS_RING, 16, RhsRing<16>)

template <int METHOD>
SolveLaunchFn find_solve_tpi(int rhs_kind, int dim, int dim16_variant) {
#define X(kind, d, T) \
  if (rhs_kind == kind && dim == d) return &launch_solve_tpi<METHOD, T>;
  NNHIP_FOR_EACH_TPI_RHS(X)
  if (dim16_variant == 1) { NNHIP_FOR_EACH_WIDE_TPI_RHS(X) }
#undef X
  if (dim16_variant == 2 && rhs_kind == NNHIP_RHS_RING && dim == 16) return &launch_solve_lps<METHOD, RhsRing<16>, 1>;
  if constexpr (MethodTraits<METHOD>::adaptive) {  // A/B: wavefront-shuffle error norm instead of the ordered LDS sum
    if (dim16_variant == 3 && rhs_kind == NNHIP_RHS_RING && dim == 16) return &launch_solve_lps<METHOD, RhsRing<16>, 4, true>;
    if (dim16_variant == 4 && rhs_kind == NNHIP_RHS_RING && dim == 16) return &launch_solve_lps<METHOD, RhsRing<16>, 1, true>;
  }
#define X(kind, d, T, CA, CF) \
  if (rhs_kind == kind && dim == d) return &launch_solve_lps<METHOD, T, (MethodTraits<METHOD>::adaptive ? CA : CF)>;
  NNHIP_FOR_EACH_LPS_RHS(X)
#undef X
  return nullptr;
}
template <int METHOD>
StepLaunchFn find_step_tpi(int rhs_kind, int dim) {
#define X(kind, d, T) \
  if (rhs_kind == kind && dim == d) return &launch_step_tpi<METHOD, T>;
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
#define X(kind, d, T, CA, CF) \
  if (rhs_kind == kind && dim == d) return &launch_step_lps<METHOD, T>;
  NNHIP_FOR_EACH_LPS_RHS(X)
#undef X
  return nullptr;
}

// dense adaptive streaming: (init, advance) launchers of one (method, RHS)
struct DenseAdvLaunch {
  hipError_t (*init)(const StepArgs&, const double* y0, double tStartEff, double dtInit, hipStream_t);
  hipError_t (*advance)(const StepArgs&, hipStream_t);
};
template <class RHS>
hipError_t launch_advance_dense_init(const StepArgs& a, const double* y0, double tStartEff, double dtInit, hipStream_t s) {
  const int64_t grid = (a.N + kBlock - 1) / kBlock;
  if (grid <= 0) return hipSuccess;
  return launch_kernel(advance_dense_init_kernel<RHS>, dim3((unsigned)grid), dim3(kBlock), s, a, y0, tStartEff, dtInit);
}
template <int METHOD, class RHS>
hipError_t launch_advance_dense(const StepArgs& a, hipStream_t s) {
  if constexpr (MethodTraits<METHOD>::adaptive) {
    const int bs = a.nontemporal ? adv_dense_block<true>() : adv_dense_block<false>();
    const int64_t grid = (a.N + bs - 1) / bs;
    if (grid <= 0) return hipSuccess;
    if (a.nontemporal) return launch_kernel(advance_dense_tpi_kernel<METHOD, RHS, true>, dim3((unsigned)grid), dim3(bs), s, a);
    return launch_kernel(advance_dense_tpi_kernel<METHOD, RHS>, dim3((unsigned)grid), dim3(bs), s, a);
  } else {
    return hipErrorInvalidValue;
  }
}
template <int METHOD, class RHS, int CPL>
hipError_t launch_advance_dense_lps(const StepArgs& a, hipStream_t s) {
  if constexpr (MethodTraits<METHOD>::adaptive) {
    constexpr int perBlock = kBlock / (RHS::dim / CPL);
    const int64_t grid = (a.N + perBlock - 1) / perBlock;
    if (grid <= 0) return hipSuccess;
    return launch_kernel(advance_dense_lps_kernel<METHOD, RHS, CPL>, dim3((unsigned)grid), dim3(kBlock), s, a);
  } else {
    return hipErrorInvalidValue;
  }
}
// init == nullptr: the driver initialises the state with the generic pieces (copies + the RHS batch kernel)
#ifndef NNHIP_ADV_DENSE_CPL_MAX  // components per lane of the lanes-per-system DENSE advance kernel: 1e6 x 16 Tsit54, 11 requested times: 2 per lane 121 us, 4 per lane 117 us per iteration
#define NNHIP_ADV_DENSE_CPL_MAX 4
#endif
template <int METHOD>
DenseAdvLaunch find_advance_dense_tpi(int rhs_kind, int dim) {
#define X(kind, d, T) \
  if (rhs_kind == kind && dim == d) return DenseAdvLaunch{&launch_advance_dense_init<T>, &launch_advance_dense<METHOD, T>};
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
#define X(kind, d, T, CA, CF) \
  if (rhs_kind == kind && dim == d) return DenseAdvLaunch{nullptr, &launch_advance_dense_lps<METHOD, T, ((CA) < NNHIP_ADV_DENSE_CPL_MAX ? (CA) : NNHIP_ADV_DENSE_CPL_MAX)>};
  NNHIP_FOR_EACH_LPS_RHS(X)
#undef X
  return DenseAdvLaunch{nullptr, nullptr};
}

template <int METHOD>
FixedVecLaunchFn find_fixed_vec_tpi(int rhs_kind, int dim) {
#define X(kind, d, T) \
  if (rhs_kind == kind && dim == d) return &launch_fixed_stream_vec<METHOD, T>;
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
  return nullptr;
}

template <int METHOD>
StepLaunchFn find_advance_tpi(int rhs_kind, int dim) {
#define X(kind, d, T) \
  if (rhs_kind == kind && dim == d) return &launch_advance_tpi<METHOD, T>;
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
#define X(kind, d, T, CA, CF) \
  if (rhs_kind == kind && dim == d) return &launch_advance_lps<METHOD, T, NNHIP_ADV_CPL(CA), ((CA) < 4 ? (CA) : 4)>;
  NNHIP_FOR_EACH_LPS_RHS(X)
#undef X
  return nullptr;
}

// (rhs_kind, dim) -> the lean advance kernel alone (nullptr: the right-hand side has none); thread-per-IVP planes or lanes-per-system rows as find_advance_tpi picks them
template <int METHOD>
StepLaunchFn find_advance_lean(int rhs_kind, int dim) {
  static_assert(METHOD == NNHIP_DOPRI54 || METHOD == NNHIP_TSIT54, "the lean kernels re-evaluate FSAL");
#define X(kind, d, T) \
  if (rhs_kind == kind && dim == d) return &launch_advance_tpi_lean_only<METHOD, T>;
  NNHIP_FOR_EACH_TPI_RHS(X)
#undef X
#define X(kind, d, T, CA, CF)                                                                       \
  if (rhs_kind == kind && dim == d) {                                                               \
    constexpr int CPL = (CA) < 4 ? (CA) : 4; /* = launch_advance_lps's CPLR: FSAL re-evaluated */   \
    if constexpr (CPL % 2 == 0 && adv_lps_spg(CPL) == 1 && RhsSize<T>::value == T::dim) return &launch_advance_lps_lean_only<METHOD, T, CPL>; \
    else return nullptr;                                                                            \
  }
  NNHIP_FOR_EACH_LPS_RHS(X)
#undef X
  return nullptr;
}

// one translation unit per integrator (ode_tu_method.hip compiled with -DNNHIP_TU_METHOD=<id>)
#define NNHIP_FOR_EACH_METHOD(X)                                                                                       \
  X(NNHIP_RK4, rk4) X(NNHIP_DOPRI54, dopri54) X(NNHIP_TSIT54, tsit54) X(NNHIP_VERN65, vern65) X(NNHIP_BS32, bs32)        \
  X(NNHIP_RK21, rk21) X(NNHIP_HEUN2, heun2) X(NNHIP_RALSTON2, ralston2) X(NNHIP_KUTTA3, kutta3) X(NNHIP_HEUN3, heun3)    \
  X(NNHIP_RALSTON3, ralston3) X(NNHIP_SSPRK3, ssprk3) X(NNHIP_RALSTON4, ralston4) X(NNHIP_KUTTA4, kutta4)
#define X(id, name)                                                      \
  SolveLaunchFn find_solve_##name(int rhs_kind, int dim, int dim16_variant);  \
  StepLaunchFn find_step_##name(int rhs_kind, int dim);                  \
  StepLaunchFn find_advance_##name(int rhs_kind, int dim);               \
  FixedVecLaunchFn find_fixed_vec_##name(int rhs_kind, int dim);         \
  DenseAdvLaunch find_advance_dense_##name(int rhs_kind, int dim);
NNHIP_FOR_EACH_METHOD(X)
#undef X
// scalar RK4 streaming (vectorised); rhs_kind must be an elementwise kind. Defined in ode_tu_rk4_stream.hip
hipError_t launch_rk4_stream(int rhs_kind, const double* yin, double* yout, int64_t n, double t, double dt, const Params& P,
                             int negate, const StreamTune& tune, hipStream_t s);
bool rk4_stream_supported(int rhs_kind);

#endif  // !NNHIP_RTC

}  // namespace NNHIP_NS

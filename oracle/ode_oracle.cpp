// =============================================================================
// ode_oracle.cpp — CPU ORACLE for the numericalnim ODE path.  TEST INFRASTRUCTURE ONLY.
//
// This file is a CPU restatement, operation for operation, of the reference's ODE solver
//   /root/reference/src/numericalnim/ode.nim      (steppers, controller, driver, dispatch)
//   /root/reference/src/numericalnim/utils.nim    (Vector[T] arithmetic, hermiteSpline, hermiteInterpolate, linspace)
// and of the consumers on either side of it
//   /root/reference/src/numericalnim/interpolate.nim  (newHermiteSpline eval / derivEval + extrapolation)
//   /root/reference/src/numericalnim/integrate.nim    (cumtrapz / cumsimpson for discrete points and for functions)
// It exists only so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
// check / time the HIP product path against it.  NOTHING in the product path (numericalnim_amd/,
// include/) may include, link or call anything in oracle/.
//
// Parity pin status — by the letter of the rule, PARITY UNPINNED at the bit level: the reference is Nim and no
// Nim compiler exists in the build image, so the reference itself cannot be run here, and it stores no bit-level
// golden vectors.  What exists instead: the oracle is
// pinned against every known-answer test the reference holds for this path
// (tests/test_ode.nim:24-257: all 14 integrators vs exp(-0.1 t) on linspace(-10,10,100) incl.
// `t == tspan`, at the reference's own tolerances; tests/test_vector.nim operator semantics;
// tests/test_utils.nim:15-23 linspace; tests/test_integrate.nim:67-95 cumtrapz / cumsimpson, discrete and
// function forms) — see tests/test_oracle_reference_kats.py — and against the
// survey's independent scratch known-answer values, and — bit for bit — against a second
// restatement written independently from ode.nim / utils.nim in plain Python floats (oracle/py_restatement.py;
// tests/test_oracle_two_restatements.py: 14 integrators, scalar and Vector states, both directions, dense rows).
// Since round 4 it is also compared, bit for bit, with an EXECUTION OF THE REFERENCE'S OWN TEXT by a Nim-subset interpreter written
// here (oracle/nim_subset.py, nim_subset_quad.py; vectors under tests/golden/reference_text_*.json): every ODE row, and since rounds
// 5 / 6 the f4 consumers incl. sortAndTrimDataset.  An interpreter of our own is a stand-in for the compiler the image lacks: the
// strongest pin obtainable here, formally not a run of the reference.
// Bit-level identity with a Nim build is by construction (same IEEE-754 double operations in the same order,
// compiled -ffp-contract=off, no fast-math, the same libm pow), not by execution.
//
// Build: see oracle/Makefile  (g++ -O3 -ffp-contract=off -fno-fast-math; same bits as -O2)
// =============================================================================
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------------------------
// Vector[T] — utils.nim:15-17.  Every binary operator allocates a fresh seq (utils.nim:61,115,...)
// exactly like the reference; that allocation cost is part of the CPU baseline for vector states.
// ---------------------------------------------------------------------------------------------
struct Vec {
  std::vector<double> components;  // utils.nim:16
};

static inline void checkVectorSizes(const Vec& a, const Vec& b) {  // utils.nim:22-26
  if (a.components.size() == b.components.size()) return;
  throw std::invalid_argument("Vectors must have the same size.");
}
static inline Vec operator+(const Vec& v1, const Vec& v2) {  // utils.nim:59-64
  checkVectorSizes(v1, v2);
  Vec r; r.components.resize(v1.components.size());
  for (size_t i = 0; i < v1.components.size(); ++i) r.components[i] = v1.components[i] + v2.components[i];
  return r;
}
static inline Vec operator-(const Vec& v1, const Vec& v2) {  // utils.nim:113-118
  checkVectorSizes(v1, v2);
  Vec r; r.components.resize(v1.components.size());
  for (size_t i = 0; i < v1.components.size(); ++i) r.components[i] = v1.components[i] - v2.components[i];
  return r;
}
static inline Vec operator*(double d, const Vec& v1) {  // utils.nim:176-180  (v1[i] * d)
  Vec r; r.components.resize(v1.components.size());
  for (size_t i = 0; i < v1.components.size(); ++i) r.components[i] = v1.components[i] * d;
  return r;
}
static inline Vec operator-(const Vec& v1) {  // utils.nim:214-218
  Vec r; r.components.resize(v1.components.size());
  for (size_t i = 0; i < v1.components.size(); ++i) r.components[i] = -v1.components[i];
  return r;
}
static inline Vec nabs(const Vec& v1) {  // utils.nim:219-223
  Vec r; r.components.resize(v1.components.size());
  for (size_t i = 0; i < v1.components.size(); ++i) r.components[i] = std::fabs(v1.components[i]);
  return r;
}
static inline Vec dotAdd(double d, const Vec& v1) {  // `+.` utils.nim:78-79 -> `+`(d, v1) :72-76 (v1[i] + d)
  Vec r; r.components.resize(v1.components.size());
  for (size_t i = 0; i < v1.components.size(); ++i) r.components[i] = v1.components[i] + d;
  return r;
}
static inline Vec dotMul(const Vec& v1, const Vec& v2) {  // `*.` utils.nim:186-191
  checkVectorSizes(v1, v2);
  Vec r; r.components.resize(v1.components.size());
  for (size_t i = 0; i < v1.components.size(); ++i) r.components[i] = v1.components[i] * v2.components[i];
  return r;
}
static inline Vec dotDiv(const Vec& v1, const Vec& v2) {  // `/.` utils.nim:192-197
  checkVectorSizes(v1, v2);
  Vec r; r.components.resize(v1.components.size());
  for (size_t i = 0; i < v1.components.size(); ++i) r.components[i] = v1.components[i] / v2.components[i];
  return r;
}
static inline int nsize(const Vec& v) { return (int)v.components.size(); }  // utils.nim:57
static inline double nsum(const Vec& v) {  // utils.nim:243-250 -> norm(v,1) :233-235 -> std/math sum: left-to-right from 0.0
  double result = 0.0;
  for (double x : v.components) result = result + x;
  return result;
}

// scalar shims — ode.nim:45-55
static inline double nabs(double d) { return std::fabs(d); }
static inline double dotAdd(double d1, double d2) { return d1 + d2; }  // ode.nim:45-46
static inline double dotDiv(double d1, double d2) { return d1 / d2; }  // ode.nim:48-49
static inline double dotMul(double d1, double d2) { return d1 * d2; }  // ode.nim:51-52
static inline int nsize(double) { return 1; }                          // ode.nim:54
static inline double nsum(double d) { return d; }                      // ode.nim:55

// Nim system min/max for floats: `if x <= y: x else: y` / `if y <= x: x else: y`
static inline double nmin(double x, double y) { return (x <= y) ? x : y; }
static inline double nmax(double x, double y) { return (y <= x) ? x : y; }

// std/math `^` with Natural exponent (cases 2 and 3 are plain products)
static inline double sq(double x) { return x * x; }
static inline double cube(double x) { return x * x * x; }

// hermiteSpline — utils.nim:273-279
template <class T>
static inline T hermiteSpline(double x, double x1, double x2, const T& y1, const T& y2, const T& dy1, const T& dy2) {
  const double t = (x - x1) / (x2 - x1);
  const double h00 = (1.0 + 2.0 * t) * sq(1.0 - t);
  const double h10 = t * sq(1.0 - t);
  const double h01 = sq(t) * (3.0 - 2.0 * t);
  const double h11 = cube(t) - sq(t);
  return h00 * y1 + h10 * (x2 - x1) * dy1 + h01 * y2 + h11 * (x2 - x1) * dy2;
}

// ---------------------------------------------------------------------------------------------
// ODEoptions — ode.nim:26-34, newODEoptions ode.nim:78-102
// ---------------------------------------------------------------------------------------------
struct ODEoptions {
  double dt, dtMax, dtMin, tStart, absTol, relTol, scaleMax, scaleMin;
};

static ODEoptions newODEoptions(double dt = 1e-4, double absTol = 1e-4, double relTol = 1e-4, double dtMax = 1e-2,
                                double dtMin = 1e-4, double scaleMax = 4.0, double scaleMin = 0.1, double tStart = 0.0) {
  if (std::fabs(dtMax) < std::fabs(dtMin)) throw std::invalid_argument("dtMin must be less than dtMax");   // :95-96
  if (std::fabs(scaleMax) < 1) throw std::invalid_argument("scaleMax must be bigger than 1");             // :97-98
  if (1 < std::fabs(scaleMin)) throw std::invalid_argument("scaleMin must be smaller than 1");            // :99-100
  ODEoptions o;                                                                                             // :101-102
  o.dt = std::fabs(dt); o.absTol = std::fabs(absTol); o.relTol = std::fabs(relTol); o.dtMax = std::fabs(dtMax);
  o.dtMin = std::fabs(dtMin); o.scaleMax = std::fabs(scaleMax); o.scaleMin = std::fabs(scaleMin); o.tStart = tStart;
  return o;
}

// ODEProc[T] — ode.nim:36.  A Nim closure is (fn pointer, environment); model it the same way so the
// CPU baseline pays the same indirect call.  `ctx` (NumContext) is the env: a flat double array here.
struct Counters {
  int64_t rhsEvals = 0, steps = 0, rejected = 0;
  int nanAbort = 0;
};
template <class T>
struct ODEProc {
  T (*fn)(double t, const T& y, const void* env);
  const void* env;
  Counters* counters;
  bool countEvals = true;  // false for the wrapper g(t,y) = -f(-t,y): the inner f call is the one counted
  inline T operator()(double t, const T& y) const {
    if (counters && countEvals) counters->rhsEvals++;
    return fn(t, y, env);
  }
};

template <class T>
struct StepResult {  // (T, T, float, float) — ode.nim:38
  T yNew, fsal;
  double dt, error;
};

// commonAdaptiveMethodCode tail — ode.nim:61-76.  Returns true when the retry loop must `break`.
// DEVIATION (documented in DESIGN.md): with error = NaN the reference loops forever
// (`error <= 1` is false, dt becomes NaN, neither clamp fires, limitCounter never moves).
// The oracle and the HIP path both abort the trajectory instead and flag it.
template <class T>
static inline bool adaptiveTail(const T& yNew, const T& error_y, int order, double absTol, double relTol, double dtMin,
                                double dtMax, double& dt, double& error, int& limitCounter, Counters* cnt) {
  const T totalTol = dotAdd(absTol, relTol * nabs(yNew));              // :61
  const T err1 = dotDiv(error_y, totalTol);                            // :62
  const T err1_square = dotMul(err1, err1);                            // :63
  const double size = (double)nsize(err1);                             // :64
  error = std::sqrt(1 / size * nsum(err1_square));                     // :65
  if (error <= 1) return true;                                         // :69-70
  if (error != error) { if (cnt) cnt->nanAbort = 1; return true; }     // deviation: see above
  dt = dt * nmin(4, nmax(0.125, 0.9 * std::pow(1 / error, 1.0 / order)));  // :71
  if (std::fabs(dt) < dtMin) { dt = dtMin; limitCounter += 1; }        // :72-74
  else if (dtMax < std::fabs(dt)) { dt = dtMax; }                      // :75-76
  if (cnt) cnt->rejected++;
  return false;
}

// ---- fixed-step steppers ---------------------------------------------------------------------
template <class T> static StepResult<T> HEUN2_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // ode.nim:107-113
  const T k1 = f(t, y);
  const T k2 = f(t + dt, y + dt * k1);
  const T yNew = y + 0.5 * dt * (k1 + k2);
  return {yNew, yNew, dt, 0.0};
}
template <class T> static StepResult<T> RALSTON2_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // :115-121
  const T k1 = f(t, y);
  const T k2 = f(t + 2.0 / 3.0 * dt, y + 2.0 / 3.0 * dt * k1);
  const T yNew = y + dt * (0.25 * k1 + 0.75 * k2);
  return {yNew, yNew, dt, 0.0};
}
template <class T> static StepResult<T> KUTTA3_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // :123-130
  const T k1 = f(t, y);
  const T k2 = f(t + 0.5 * dt, y + 0.5 * dt * k1);
  const T k3 = f(t + dt, y - dt * k1 + 2.0 * dt * k2);
  const T yNew = y + dt * (1.0 / 6.0 * k1 + 2.0 / 3.0 * k2 + 1.0 / 6.0 * k3);
  return {yNew, yNew, dt, 0.0};
}
template <class T> static StepResult<T> HEUN3_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // :132-139
  const T k1 = f(t, y);
  const T k2 = f(t + 1.0 / 3.0 * dt, y + 1.0 / 3.0 * dt * k1);
  const T k3 = f(t + 2.0 / 3.0 * dt, y + 2.0 / 3.0 * dt * k2);
  const T yNew = y + dt * (0.25 * k1 + 0.75 * k3);
  return {yNew, yNew, dt, 0.0};
}
template <class T> static StepResult<T> RALSTON3_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // :141-148
  const T k1 = f(t, y);
  const T k2 = f(t + 1.0 / 2.0 * dt, y + 1.0 / 2.0 * dt * k1);
  const T k3 = f(t + 3.0 / 4.0 * dt, y + 3.0 / 4.0 * dt * k2);
  const T yNew = y + dt * (2.0 / 9.0 * k1 + 1.0 / 3.0 * k2 + 4.0 / 9.0 * k3);
  return {yNew, yNew, dt, 0.0};
}
template <class T> static StepResult<T> SSPRK3_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // :150-157
  const T k1 = f(t, y);
  const T k2 = f(t + dt, y + dt * k1);
  const T k3 = f(t + 0.5 * dt, y + 0.25 * dt * (k1 + k2));
  const T yNew = y + dt * (1.0 / 6.0 * k1 + 1.0 / 6.0 * k2 + 2.0 / 3.0 * k3);
  return {yNew, yNew, dt, 0.0};
}
template <class T> static StepResult<T> RALSTON4_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // :160-168
  const T k1 = f(t, y);
  const T k2 = f(t + 0.4 * dt, y + 0.4 * dt * k1);
  const T k3 = f(t + 0.45573725 * dt, y + dt * (0.29697761 * k1 + 0.15875964 * k2));
  const T k4 = f(t + dt, y + dt * (0.21810040 * k1 - 3.05096516 * k2 + 3.83286476 * k3));
  const T yNew = y + dt * (0.17476028 * k1 - 0.55148066 * k2 + 1.20553560 * k3 + 0.17118478 * k4);
  return {yNew, yNew, dt, 0.0};
}
template <class T> static StepResult<T> KUTTA4_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // :170-178
  const T k1 = f(t, y);
  const T k2 = f(t + 1.0 / 3.0 * dt, y + 1.0 / 3.0 * dt * k1);
  const T k3 = f(t + 2.0 / 3.0 * dt, y + dt * (-1.0 / 3.0 * k1 + k2));
  const T k4 = f(t + dt, y + dt * (k1 - k2 + k3));
  const T yNew = y + dt * (1.0 / 8.0 * k1 + 3.0 / 8.0 * k2 + 3.0 / 8.0 * k3 + 1.0 / 8.0 * k4);
  return {yNew, yNew, dt, 0.0};
}
template <class T> static StepResult<T> RK4_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt, const ODEoptions&) {  // :180-189
  const T k1 = f(t, y);
  const T k2 = f(t + 0.5 * dt, y + 0.5 * dt * k1);
  const T k3 = f(t + 0.5 * dt, y + 0.5 * dt * k2);
  const T k4 = f(t + dt, y + dt * k3);
  const T yNew = y + dt / 6.0 * (k1 + 2.0 * (k2 + k3) + k4);
  return {yNew, yNew, dt, 0.0};
}

// ---- adaptive steppers -----------------------------------------------------------------------
template <class T> static StepResult<T> RK21_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt_in, const ODEoptions& options) {  // :191-210
  const double absTol = options.absTol, relTol = options.relTol, dtMax = options.dtMax, dtMin = options.dtMin;
  T k1, k2, yNew, yLow;
  double error = 0.0; int limitCounter = 0; double dt = dt_in;
  while (limitCounter < 2) {
    k1 = f(t, y);
    k2 = f(t + dt, y + dt * k1);
    yNew = y + dt * 0.5 * (k1 + k2);
    yLow = y + dt * k1;
    const T error_y = yNew - yLow;
    if (adaptiveTail(yNew, error_y, 2, absTol, relTol, dtMin, dtMax, dt, error, limitCounter, f.counters)) break;
  }
  return {yNew, yNew, dt, error};
}
template <class T> static StepResult<T> BS32_step(const ODEProc<T>& f, double t, const T& y, const T&, double dt_in, const ODEoptions& options) {  // :212-234
  const double absTol = options.absTol, relTol = options.relTol, dtMax = options.dtMax, dtMin = options.dtMin;
  T k1, k2, k3, k4, yNew, yLow;
  double error = 0.0; int limitCounter = 0; double dt = dt_in;
  while (limitCounter < 2) {
    k1 = f(t, y);
    k2 = f(t + 0.5 * dt, y + 0.5 * dt * k1);
    k3 = f(t + 0.75 * dt, y + 0.75 * dt * k2);
    yNew = y + dt * (2.0 / 9.0 * k1 + 1.0 / 3.0 * k2 + 4.0 / 9.0 * k3);
    k4 = f(t + dt, yNew);
    yLow = y + dt * (7.0 / 24.0 * k1 + 1.0 / 4.0 * k2 + 1.0 / 3.0 * k3 + 1.0 / 8.0 * k4);
    const T error_y = yNew - yLow;
    if (adaptiveTail(yNew, error_y, 3, absTol, relTol, dtMin, dtMax, dt, error, limitCounter, f.counters)) break;
  }
  return {yNew, k4, dt, error};
}

// The Butcher tableaux of the three tableau methods, as name/value lists: the step procs below declare them as their constants
// (in the reference's declaration order, ode.nim:240-282, 310-352, 380-443) and oracle_tableau() exports the very same values,
// so that tests/test_reference_text_pin.py can compare them bit for bit with the `const` sections of the reference's text.
#define DOPRI54_CONSTS(X) \
  X(c2, 1.0 / 5.0) X(c3, 3.0 / 10.0) X(c4, 4.0 / 5.0) X(c5, 8.0 / 9.0) X(c6, 1.0) X(c7, 1.0) X(a21, 1.0 / 5.0) X(a31, 3.0 / 40.0) \
  X(a32, 9.0 / 40.0) X(a41, 44.0 / 45.0) X(a42, -56.0 / 15.0) X(a43, 32.0 / 9.0) X(a51, 19372.0 / 6561.0) \
  X(a52, -25360.0 / 2187.0) X(a53, 64448.0 / 6561.0) X(a54, -212.0 / 729.0) X(a61, 9017.0 / 3168.0) X(a62, -355.0 / 33.0) \
  X(a63, 46732.0 / 5247.0) X(a64, 49.0 / 176.0) X(a65, -5103.0 / 18656.0) X(a71, 35.0 / 384.0) X(a72, 0.0) X(a73, 500.0 / 1113.0) \
  X(a74, 125.0 / 192.0) X(a75, -2187.0 / 6784.0) X(a76, 11.0 / 84.0) X(b1, a71) X(b2, a72) X(b3, a73) X(b4, a74) X(b5, a75) \
  X(b6, a76) X(bHat1, 5179.0 / 57600.0) X(bHat2, 0.0) X(bHat3, 7571.0 / 16695.0) X(bHat4, 393.0 / 640.0) \
  X(bHat5, -92097.0 / 339200.0) X(bHat6, 187.0 / 2100.0) X(bHat7, 1.0 / 40.0)
#define TSIT54_CONSTS(X) \
  X(c2, 0.161) X(c3, 0.327) X(c4, 0.9) X(c5, 0.9800255409045097) X(c6, 1.0) X(c7, 1.0) X(a21, 0.161) \
  X(a31, -0.008480655492356989) X(a32, 0.335480655492357) X(a41, 2.8971530571054935) X(a42, -6.359448489975075) \
  X(a43, 4.3622954328695815) X(a51, 5.325864828439257) X(a52, -11.748883564062828) X(a53, 7.4955393428898365) \
  X(a54, -0.09249506636175525) X(a61, 5.86145544294642) X(a62, -12.92096931784711) X(a63, 8.159367898576159) \
  X(a64, -0.071584973281401) X(a65, -0.028269050394068383) X(a71, 0.09646076681806523) X(a72, 0.01) X(a73, 0.4798896504144996) \
  X(a74, 1.379008574103742) X(a75, -3.290069515436081) X(a76, 2.324710524099774) X(b1, a71) X(b2, a72) X(b3, a73) X(b4, a74) \
  X(b5, a75) X(b6, a76) X(bHat1, -0.001780011052226) X(bHat2, -0.000816434459657) X(bHat3, 0.007880878010262) \
  X(bHat4, -0.144711007173263) X(bHat5, 0.582357165452555) X(bHat6, -0.458082105929187) X(bHat7, 1.0 / 66.0)
#define VERN65_CONSTS(X) \
  X(c2, 0.06) X(c3, 0.09593333333333333) X(c4, 0.1439) X(c5, 0.4973) X(c6, 0.9725) X(c7, 0.9995) X(c8, 1.0) X(c9, 1.0) \
  X(a21, 0.06) X(a31, 0.019239962962962962) X(a32, 0.07669337037037037) X(a41, 0.035975) X(a42, 0.0) X(a43, 0.107925) \
  X(a51, 1.3186834152331484) X(a52, 0.0) X(a53, -5.042058063628562) X(a54, 4.220674648395414) X(a61, -41.87259166432751) \
  X(a62, 0.0) X(a63, 159.43256216313748) X(a64, -122.11921356501004) X(a65, 5.531743066200053) X(a71, -54.430156935316504) \
  X(a72, 0.0) X(a73, 207.06725136501848) X(a74, -158.61081378459) X(a75, 6.991816585950242) X(a76, -0.01859723106220323) \
  X(a81, -54.66374178728198) X(a82, 0.0) X(a83, 207.95280625538936) X(a84, -159.2889574744995) X(a85, 7.018743740796944) \
  X(a86, -0.018338785905045722) X(a87, -0.0005119484997882099) X(a91, 0.03438957868357036) X(a92, 0.0) X(a93, 0.0) \
  X(a94, 0.25826245556335037) X(a95, 0.4209371189673537) X(a96, 4.405396469669310) X(a97, -176.48311902429865) \
  X(a98, 172.36413340141507) X(b1, 0.03438957868357036) X(b2, 0.0) X(b3, 0.0) X(b4, 0.25826245556335034) \
  X(b5, 0.42093711896735372) X(b6, 4.4053964696693102) X(b7, -176.48311902429866) X(b8, 172.36413340141507) \
  X(bHat1, 0.04909967648382) X(bHat2, 0.0) X(bHat3, 0.0) X(bHat4, 0.22511122295165) X(bHat5, 0.46946822530296) \
  X(bHat6, 0.80657922499889) X(bHat7, 0.0) X(bHat8, -0.60711948917780) X(bHat9, 0.05686113944048)

template <class T> static StepResult<T> DOPRI54_step(const ODEProc<T>& f, double t, const T& y, const T& FSAL, double dt_in, const ODEoptions& options) {  // :237-305
#define X(n, v) constexpr double n = v;
  DOPRI54_CONSTS(X)
#undef X
  const double absTol = options.absTol, relTol = options.relTol, dtMax = options.dtMax, dtMin = options.dtMin;
  T k1, k2, k3, k4, k5, k6, k7, yNew, yLow;
  double error = 0.0; int limitCounter = 0; double dt = dt_in;
  while (limitCounter < 2) {
    k1 = FSAL;
    k2 = f(t + dt * c2, y + dt * (a21 * k1));
    k3 = f(t + dt * c3, y + dt * (a31 * k1 + a32 * k2));
    k4 = f(t + dt * c4, y + dt * (a41 * k1 + a42 * k2 + a43 * k3));
    k5 = f(t + dt * c5, y + dt * (a51 * k1 + a52 * k2 + a53 * k3 + a54 * k4));
    k6 = f(t + dt * c6, y + dt * (a61 * k1 + a62 * k2 + a63 * k3 + a64 * k4 + a65 * k5));
    k7 = f(t + dt * c7, y + dt * (a71 * k1 + a72 * k2 + a73 * k3 + a74 * k4 + a75 * k5 + a76 * k6));
    yNew = y + dt * (b1 * k1 + b2 * k2 + b3 * k3 + b4 * k4 + b5 * k5 + b6 * k6);
    yLow = y + dt * (bHat1 * k1 + bHat2 * k2 + bHat3 * k3 + bHat4 * k4 + bHat5 * k5 + bHat6 * k6 + bHat7 * k7);
    const T error_y = yNew - yLow;
    if (adaptiveTail(yNew, error_y, 5, absTol, relTol, dtMin, dtMax, dt, error, limitCounter, f.counters)) break;
  }
  return {yNew, k7, dt, error};
}

template <class T> static StepResult<T> TSIT54_step(const ODEProc<T>& f, double t, const T& y, const T& FSAL, double dt_in, const ODEoptions& options) {  // :307-374
#define X(n, v) constexpr double n = v;
  TSIT54_CONSTS(X)
#undef X
  const double absTol = options.absTol, relTol = options.relTol, dtMax = options.dtMax, dtMin = options.dtMin;
  T k1, k2, k3, k4, k5, k6, k7, yNew;
  double error = 0.0; int limitCounter = 0; double dt = dt_in;
  while (limitCounter < 2) {
    k1 = FSAL;
    k2 = f(t + dt * c2, y + dt * (a21 * k1));
    k3 = f(t + dt * c3, y + dt * (a31 * k1 + a32 * k2));
    k4 = f(t + dt * c4, y + dt * (a41 * k1 + a42 * k2 + a43 * k3));
    k5 = f(t + dt * c5, y + dt * (a51 * k1 + a52 * k2 + a53 * k3 + a54 * k4));
    k6 = f(t + dt * c6, y + dt * (a61 * k1 + a62 * k2 + a63 * k3 + a64 * k4 + a65 * k5));
    k7 = f(t + dt * c7, y + dt * (a71 * k1 + a72 * k2 + a73 * k3 + a74 * k4 + a75 * k5 + a76 * k6));
    yNew = y + dt * (b1 * k1 + b2 * k2 + b3 * k3 + b4 * k4 + b5 * k5 + b6 * k6);
    const T error_y = dt * (bHat1 * k1 + bHat2 * k2 + bHat3 * k3 + bHat4 * k4 + bHat5 * k5 + bHat6 * k6 + bHat7 * k7);
    if (adaptiveTail(yNew, error_y, 5, absTol, relTol, dtMin, dtMax, dt, error, limitCounter, f.counters)) break;
  }
  return {yNew, k7, dt, error};
}

template <class T> static StepResult<T> VERN65_step(const ODEProc<T>& f, double t, const T& y, const T& FSAL, double dt_in, const ODEoptions& options) {  // :377-468
#define X(n, v) constexpr double n = v;
  VERN65_CONSTS(X)
#undef X
  const double absTol = options.absTol, relTol = options.relTol, dtMax = options.dtMax, dtMin = options.dtMin;
  T k1, k2, k3, k4, k5, k6, k7, k8, k9, yNew, yLow;
  double error = 0.0; int limitCounter = 0; double dt = dt_in;
  while (limitCounter < 2) {
    k1 = FSAL;
    k2 = f(t + dt * c2, y + dt * (a21 * k1));
    k3 = f(t + dt * c3, y + dt * (a31 * k1 + a32 * k2));
    k4 = f(t + dt * c4, y + dt * (a41 * k1 + a42 * k2 + a43 * k3));
    k5 = f(t + dt * c5, y + dt * (a51 * k1 + a52 * k2 + a53 * k3 + a54 * k4));
    k6 = f(t + dt * c6, y + dt * (a61 * k1 + a62 * k2 + a63 * k3 + a64 * k4 + a65 * k5));
    k7 = f(t + dt * c7, y + dt * (a71 * k1 + a72 * k2 + a73 * k3 + a74 * k4 + a75 * k5 + a76 * k6));
    k8 = f(t + dt * c8, y + dt * (a81 * k1 + a82 * k2 + a83 * k3 + a84 * k4 + a85 * k5 + a86 * k6 + a87 * k7));
    k9 = f(t + dt * c9, y + dt * (a91 * k1 + a92 * k2 + a93 * k3 + a94 * k4 + a95 * k5 + a96 * k6 + a97 * k7 + a98 * k8));
    yNew = y + dt * (b1 * k1 + b2 * k2 + b3 * k3 + b4 * k4 + b5 * k5 + b6 * k6 + b7 * k7 + b8 * k8);
    yLow = y + dt * (bHat1 * k1 + bHat2 * k2 + bHat3 * k3 + bHat4 * k4 + bHat5 * k5 + bHat6 * k6 + bHat7 * k7 + bHat8 * k8 + bHat9 * k9);
    const T error_y = yNew - yLow;
    if (adaptiveTail(yNew, error_y, 6, absTol, relTol, dtMin, dtMax, dt, error, limitCounter, f.counters)) break;
  }
  return {yNew, k9, dt, error};
}

template <class T>
using IntegratorProc = StepResult<T> (*)(const ODEProc<T>&, double, const T&, const T&, double, const ODEoptions&);  // ode.nim:38

struct Method { int id; const char* name; bool useFSAL; double order; bool adaptive; };
// solveODE dispatch table — ode.nim:607-649
static const Method kMethods[] = {
    {0, "rk4", false, 4.0, false},      {1, "dopri54", true, 5.0, true},   {2, "tsit54", true, 5.0, true},
    {3, "vern65", true, 6.0, true},     {4, "bs32", true, 3.0, true},      {5, "rk21", false, 2.0, true},
    {6, "heun2", false, 2.0, false},    {7, "ralston2", false, 2.0, false}, {8, "kutta3", false, 3.0, false},
    {9, "heun3", false, 3.0, false},    {10, "ralston3", false, 3.0, false}, {11, "ssprk3", false, 3.0, false},
    {12, "ralston4", false, 4.0, false}, {13, "kutta4", false, 4.0, false},
};
template <class T> static IntegratorProc<T> stepperFor(int id) {
  switch (id) {
    case 0: return RK4_step<T>;      case 1: return DOPRI54_step<T>;  case 2: return TSIT54_step<T>;
    case 3: return VERN65_step<T>;   case 4: return BS32_step<T>;     case 5: return RK21_step<T>;
    case 6: return HEUN2_step<T>;    case 7: return RALSTON2_step<T>; case 8: return KUTTA3_step<T>;
    case 9: return HEUN3_step<T>;    case 10: return RALSTON3_step<T>; case 11: return SSPRK3_step<T>;
    case 12: return RALSTON4_step<T>; case 13: return KUTTA4_step<T>;
  }
  return nullptr;
}
static const Method* methodByName(const char* name) {  // integrator.toLower() — ode.nim:607
  std::string s(name);
  for (auto& ch : s) ch = (char)std::tolower((unsigned char)ch);
  for (const Method& m : kMethods) if (s == m.name) return &m;
  return nullptr;  // ode.nim:651 ValueError
}

// ---------------------------------------------------------------------------------------------
// ODESolver — ode.nim:471-586.  Returns (t, y) with y possibly SHORTER than t (reference quirk:
// requested times strictly inside the last step are never emitted).
// ---------------------------------------------------------------------------------------------
template <class T>
static void ODESolver(const ODEProc<T>& f, const T& y0, const std::vector<double>& tspan /*sorted*/, const ODEoptions& options,
                      IntegratorProc<T> integrator, bool useFSAL, double order, bool adaptive,
                      std::vector<double>& tOut, std::vector<T>& yOut, int64_t maxSteps = -1) {
  const double t0 = options.tStart;                                                    // :476
  double t = t0;
  std::vector<double> tPositive, tNegative;
  for (double x : tspan) if (x > t0) tPositive.push_back(x);                           // :479
  for (double x : tspan) if (x < t0) tNegative.push_back(x);                           // :480
  std::reverse(tNegative.begin(), tNegative.end());
  std::vector<T> yPositive, yNegative;
  T y = y0;                                                                            // :482 clone
  std::vector<T> yZero; std::vector<double> tZero;
  if (std::find(tspan.begin(), tspan.end(), t0) != tspan.end()) { yZero.push_back(y); tZero.push_back(t0); }  // :485-487
  const double dtMax = options.dtMax, dtMin = options.dtMin;
  double dt, dtInit;
  if (adaptive) { dtInit = std::sqrt(dtMax * dtMin); dt = dtInit; }                    // :491-493
  else { dtInit = options.dt; dt = dtInit; }                                           // :494-496
  struct Iter { double t; T y; T dy; };
  Iter lastIter{t0, y, f(t0, y)};                                                      // :498
  const bool useDense = (tspan.size() != 2);                                           // :499-502
  long denseIndex = 0;
  double error = 0.0;
  T FSAL = f(t0, y);                                                                   // :506
  double tEnd;
  Counters* cnt = f.counters;
  if (0 < tPositive.size()) {                                                          // :508
    dt = dtInit;
    tEnd = tPositive[0]; for (size_t i = 1; i < tPositive.size(); ++i) tEnd = nmax(tEnd, tPositive[i]);  // :510
    const long high = (long)tPositive.size() - 1;
    while (t < tEnd) {                                                                 // :511
      if (useDense) {
        if (high < denseIndex) break;                                                  // :513-514
        while (tPositive[denseIndex] <= t) {                                           // :515
          if (useFSAL) yPositive.push_back(hermiteSpline(tPositive[denseIndex], lastIter.t, t, lastIter.y, y, lastIter.dy, FSAL));
          else         yPositive.push_back(hermiteSpline(tPositive[denseIndex], lastIter.t, t, lastIter.y, y, lastIter.dy, f(t, y)));
          denseIndex += 1;
          if (high < denseIndex) break;                                                // :523-524
        }
      }
      dt = nmin(dt, tEnd - t);                                                         // :525
      if (useDense) {
        if (useFSAL) lastIter = Iter{t, y, FSAL};                                      // :528
        else         lastIter = Iter{t, y, f(t, y)};                                   // :530
      }
      StepResult<T> r = integrator(f, t, y, FSAL, dt, options);                        // :531
      y = r.yNew; FSAL = r.fsal; dt = r.dt; error = r.error;
      t += dt;                                                                         // :532
      if (cnt) cnt->steps++;
      if (adaptive) {
        if (error == 0.0) dt *= 5;                                                     // :534-535
        else dt = dt * nmin(4, nmax(0.125, 0.9 * std::pow(1 / error, 1 / order)));     // :537
        if (dt < dtMin) dt = dtMin;                                                    // :538-539
        else if (dtMax < dt) dt = dtMax;                                               // :540-541
      }
      if (cnt && cnt->nanAbort) break;                        // deviation: NaN abort (see adaptiveTail)
      if (maxSteps >= 0 && cnt && cnt->steps >= maxSteps) break;
    }
    yPositive.push_back(y);                                                            // :542
  }
  if (0 < tNegative.size()) {                                                          // :544
    // g(t, y) = -f(-t, y) — a closure over f, exactly as the reference builds it              // :545
    struct GEnv { ODEProc<T> f; } genv{f};
    struct Shim { static T call(double tt, const T& yy, const void* env) { const GEnv* e = (const GEnv*)env; return -(e->f)(-tt, yy); } };
    const ODEProc<T> g{Shim::call, &genv, cnt, false};
    FSAL = g(-t0, y0);                                                                 // :546
    dt = dtInit;                                                                       // :547
    lastIter = Iter{-t0, y0, FSAL};                                                    // :548
    double mn = tNegative[0]; for (size_t i = 1; i < tNegative.size(); ++i) mn = nmin(mn, tNegative[i]);
    tEnd = -mn;                                                                        // :549
    t = -t0;                                                                           // :550
    y = y0;                                                                            // :551
    denseIndex = 0;                                                                    // :552
    const long high = (long)tNegative.size() - 1;
    while (t < tEnd) {                                                                 // :553
      if (useDense) {
        if (high < denseIndex) break;                                                  // :555-556
        while (-tNegative[denseIndex] <= t) {                                          // :557
          if (useFSAL) yNegative.push_back(hermiteSpline(-tNegative[denseIndex], lastIter.t, t, lastIter.y, y, lastIter.dy, FSAL));
          else         yNegative.push_back(hermiteSpline(-tNegative[denseIndex], lastIter.t, t, lastIter.y, y, lastIter.dy, g(t, y)));
          denseIndex += 1;
          if (high < denseIndex) break;                                                // :565-566
        }
      }
      dt = nmin(dt, tEnd - t);                                                         // :567
      if (useDense) {
        if (useFSAL) lastIter = Iter{t, y, FSAL};                                      // :570
        else         lastIter = Iter{t, y, g(t, y)};                                   // :572
      }
      StepResult<T> r = integrator(g, t, y, FSAL, dt, options);                        // :573
      y = r.yNew; FSAL = r.fsal; dt = r.dt; error = r.error;
      t += dt;                                                                         // :574
      if (cnt) cnt->steps++;
      if (adaptive) {
        if (error == 0.0) dt *= 5;                                                     // :576-577
        else dt = dt * nmin(4, nmax(0.125, 0.9 * std::pow(1 / error, 1 / order)));     // :579
        if (dt < dtMin) dt = dtMin;                                                    // :580-581
        else if (dtMax < dt) dt = dtMax;                                               // :582-583
      }
      if (cnt && cnt->nanAbort) break;
      if (maxSteps >= 0 && cnt && cnt->steps >= maxSteps) break;
    }
    yNegative.push_back(y);                                                            // :584
  }
  // :585-586
  tOut.clear(); yOut.clear();
  for (auto it = tNegative.rbegin(); it != tNegative.rend(); ++it) tOut.push_back(*it);
  for (double x : tZero) tOut.push_back(x);
  for (double x : tPositive) tOut.push_back(x);
  for (auto it = yNegative.rbegin(); it != yNegative.rend(); ++it) yOut.push_back(*it);
  for (auto& v : yZero) yOut.push_back(v);
  for (auto& v : yPositive) yOut.push_back(v);
}

// solveODE — ode.nim:589-651
template <class T>
static int solveODE(const ODEProc<T>& f, const T& y0, const double* tspan, int nT, const ODEoptions& options,
                    const char* integrator, std::vector<double>& tOut, std::vector<T>& yOut) {
  const Method* m = methodByName(integrator);
  if (!m) return -2;  // ValueError "... is not a valid integrator" :651
  std::vector<double> ts(tspan, tspan + nT);
  std::sort(ts.begin(), ts.end());  // tspan.sorted()
  ODESolver<T>(f, y0, ts, options, stepperFor<T>(m->id), m->useFSAL, m->order, m->adaptive, tOut, yOut);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// RHS library restated on the CPU (definitions: include/nnhip_ode.h, enum nnhip_rhs_kind).
// The reference takes an arbitrary user closure; these are the closures the tests/bench supply.
// ---------------------------------------------------------------------------------------------
enum RhsKind { RHS_NEG_Y = 0, RHS_LINEAR = 1, RHS_LORENZ = 2, RHS_RING = 3, RHS_AFFINE_T = 4, RHS_VANDERPOL = 5,
               RHS_DUFFING = 6 /* oracle-only: checks run-time compiled user RHS */,
               RHS_COS_T = 7 /* oracle-only integrand a*cos(t): tests/test_integrate.nim:5-6 */,
               RHS_POLY_T = 8 /* oracle-only integrand ((a t + b) t)(1 + c) + d for component c */,
               RHS_HEAT = 9 /* oracle-only: method-of-lines heat equation, checks wide run-time compiled systems */,
               RHS_MATVEC = 10 /* oracle-only: dy = s * (A y) + g with the closure's own d x d matrix A (ctx.tValues, commonTypes.nim:4-27) */,
               RHS_LORENZ_ZCROSS = 11 /* oracle-only: Lorenz whose closure MUTATES its ctx (ode.nim:599): counts crossings of z = 25 between calls */ };

static double rhsScalar(double t, const double& y, const void* env) {
  const double* p = (const double*)env;  // p[0] = kind, p[1..] = params
  switch ((int)p[0]) {
    case RHS_NEG_Y: return -y;                      // ode.nim:16-17
    case RHS_LINEAR: return p[1] * y;               // tests/test_ode.nim:5   (-0.1 * y)
    case RHS_AFFINE_T: return p[1] * y + p[2] * t;  // time-dependent probe
    case RHS_COS_T: return p[1] * std::cos(t);      // tests/test_integrate.nim:5
    case RHS_POLY_T: return ((p[1] * t + p[2]) * t) * 1.0 + p[3];
  }
  return NAN;
}
static Vec rhsVector(double t, const Vec& y, const void* env) {
  const double* p = (const double*)env;
  const size_t d = y.components.size();
  Vec r; r.components.resize(d);
  switch ((int)p[0]) {
    case RHS_NEG_Y: for (size_t i = 0; i < d; ++i) r.components[i] = -y.components[i]; break;
    case RHS_LINEAR: for (size_t i = 0; i < d; ++i) r.components[i] = y.components[i] * p[1]; break;  // tests/test_ode.nim:6 (-0.1 * y -> y[i]*d)
    case RHS_AFFINE_T: for (size_t i = 0; i < d; ++i) r.components[i] = p[1] * y.components[i] + p[2] * t; break;
    case RHS_LORENZ: {  // sigma, rho, beta = p[1..3]
      const double x = y.components[0], yy = y.components[1], z = y.components[2];
      r.components[0] = p[1] * (yy - x);
      r.components[1] = x * (p[2] - z) - yy;
      r.components[2] = x * yy - p[3] * z;
      break;
    }
    case RHS_RING: {  // y_i' = -((i+1)/d) * y_i + c * y_{(i+1) mod d}, c = p[1]
      for (size_t i = 0; i < d; ++i)
        r.components[i] = -((double)(i + 1) / (double)d) * y.components[i] + p[1] * y.components[(i + 1) % d];
      break;
    }
    case RHS_VANDERPOL: {  // mu = p[1]
      const double x = y.components[0], v = y.components[1];
      r.components[0] = v;
      r.components[1] = p[1] * ((1.0 - x * x) * v) - x;
      break;
    }
    case RHS_DUFFING: {  // delta, alpha, beta, gamma = p[1..4]: x' = v; v' = -delta v - alpha x - beta x^3 + gamma t
      const double x = y.components[0], v = y.components[1];
      r.components[0] = v;
      r.components[1] = ((-p[1] * v - p[2] * x) - p[3] * (x * x * x)) + p[4] * t;
      break;
    }
    case RHS_COS_T: for (size_t i = 0; i < d; ++i) r.components[i] = p[1] * std::cos(t); break;  // tests/test_integrate.nim:6 (cos(x) * ctx["a"])
    case RHS_POLY_T: for (size_t i = 0; i < d; ++i) r.components[i] = ((p[1] * t + p[2]) * t) * (1.0 + (double)i) + p[3]; break;
    case RHS_HEAT: {  // y_i' = kappa * ((y_{i-1} - 2 y_i) + y_{i+1}), zero boundary values; kappa = p[1]
      for (size_t i = 0; i < d; ++i) {
        const double left = i > 0 ? y.components[i - 1] : 0.0, right = i + 1 < d ? y.components[i + 1] : 0.0;
        r.components[i] = p[1] * ((left - 2.0 * y.components[i]) + right);
      }
      break;
    }
    case RHS_MATVEC: {  // env: [kind, s, g[0..d), A[0..d*d) row-major]: dy_r = s * (((A_r0 y_0) + A_r1 y_1) + ...) + g_r
      const double* g = p + 2;
      const double* A = g + d;
      for (size_t i = 0; i < d; ++i) {
        double acc = A[i * d] * y.components[0];
        for (size_t c = 1; c < d; ++c) acc = acc + A[i * d + c] * y.components[c];
        r.components[i] = p[1] * acc + g[i];
      }
      break;
    }
    case RHS_LORENZ_ZCROSS: {  // env: [kind, sigma, rho, beta, count, last z, calls] — the last three are the closure's mutable ctx
      double* m = const_cast<double*>(p) + 4;
      const double x = y.components[0], yy = y.components[1], z = y.components[2];
      if (m[2] > 0.0 && (m[1] - 25.0) * (z - 25.0) < 0.0) m[0] = m[0] + 1.0;
      m[1] = z;
      m[2] = m[2] + 1.0;
      r.components[0] = p[1] * (yy - x);
      r.components[1] = x * (p[2] - z) - yy;
      r.components[2] = x * yy - p[3] * z;
      break;
    }
    default: for (size_t i = 0; i < d; ++i) r.components[i] = NAN;
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// Cumulative quadrature of a function of x ("march + Hermite resample"):
//   hermiteInterpolate     utils.nim:282-312
//   cumsimpson(Y, X)       integrate.nim:329-375     (generic T restatement of oracle_cumsimpson below)
//   cumtrapz(f, X, ctx, dx)    integrate.nim:138-175
//   cumsimpson(f, X, ctx, dx)  integrate.nim:377-400
// ---------------------------------------------------------------------------------------------
template <class T>
static std::vector<T> hermiteInterpolate(const std::vector<double>& x, const std::vector<double>& t, const std::vector<T>& y,
                                         const std::vector<T>& dy) {
  std::vector<T> result;
  const long xHigh = (long)x.size() - 1, tHigh = (long)t.size() - 1;
  long xIndex = 0;
  if (std::is_sorted(x.begin(), x.end())) {  // isSorted(x): non-decreasing (:290)
    bool done = false;
    for (long i = 0; i <= tHigh - 1 && !done; ++i) {  // :291
      while (t[i] <= x[xIndex] && x[xIndex] < t[i + 1]) {  // :292
        result.push_back(hermiteSpline<T>(x[xIndex], t[i], t[i + 1], y[i], y[i + 1], dy[i], dy[i + 1]));
        xIndex += 1;
        if (xHigh < xIndex) { done = true; break; }  // :295-298
      }
    }
    if (x[xHigh] == t[tHigh]) result.push_back(y[tHigh]);  // :299-300
  } else {  // :302-311
    for (double a : x) {
      bool found = false;
      for (long i = 0; i <= tHigh - 1; ++i)
        if (t[i] <= a && a < t[i + 1]) {
          result.push_back(hermiteSpline<T>(a, t[i], t[i + 1], y[i], y[i + 1], dy[i], dy[i + 1]));
          found = true;
          break;  // break forblock
        }
      if (found) continue;
      if (a == t[tHigh]) result.push_back(y[tHigh]);
      else throw std::invalid_argument("x not in interval");  // ValueError :311
    }
  }
  return result;
}

// sortAndTrimDataset (utils.nim:404-407) = sortDataset (:384-402) followed by removeDuplicates (:360-381), for scalar series y[k] over one x.
struct SortedTrimmed {
  std::vector<double> x;
  std::vector<std::vector<double>> y;
};
static SortedTrimmed sortAndTrimDataset(const std::vector<double>& x, const std::vector<std::vector<double>>& y) {
  const size_t n = x.size();
  for (double v : x)
    if (v != v) throw std::domain_error("NaN in x: the comparison sort of sortDataset has no defined order for it");  // (not a reference error: no defined result to restate)
  // sortDataset: var zipped_x = zip(x, toSeq(0 .. xLen-1)); zipped_x.sort(sortOrder) (:392-393) — tuples compare field by field: by x (cmp: <, ==), then by index
  std::vector<std::pair<double, long>> zipped(n);
  for (size_t i = 0; i < n; ++i) zipped[i] = {x[i], (long)i};
  std::sort(zipped.begin(), zipped.end(), [](const std::pair<double, long>& a, const std::pair<double, long>& b) {
    if (a.first < b.first) return true;
    if (a.first == b.first) return a.second < b.second;
    return false;
  });
  SortedTrimmed s;
  s.x.resize(n);
  s.y.assign(y.size(), std::vector<double>(n));
  for (size_t i = 0; i < n; ++i) {  // result.y[resultIdx][i] = y[resultIdx][idxNew] (:399-402)
    s.x[i] = zipped[i].first;
    for (size_t k = 0; k < y.size(); ++k) s.y[k][i] = y[k][(size_t)zipped[i].second];
  }
  // removeDuplicates: findDuplicates(x) = the index lists of getIndexTable(x) longer than 1 (:347-357): a Table keyed by float — equal keys (-0.0 == 0.0,
  // and hash(x) hashes x + 0.0) share an entry.  x is sorted here, so the members of an entry are adjacent: dups = runs of equal values, ascending indices.
  std::vector<size_t> idxDelete;
  for (size_t i = 0; i < n;) {
    size_t j = i + 1;
    while (j < n && s.x[j] == s.x[i]) ++j;
    for (size_t d = i; d < j && j - i > 1; ++d)        // `for i in dups` — dups[0] included, which compares equal to itself unless it is NaN (:369-372)
      for (size_t k = 0; k < y.size(); ++k)
        if (s.y[k][d] != s.y[k][i]) throw std::invalid_argument("impure y-duplicates was found");  // ValueError :372
    for (size_t d = i + 1; d < j; ++d) idxDelete.push_back(d);  // idxDelete.add dups[1 .. ^1] (:376)
    i = j;
  }
  for (size_t q = idxDelete.size(); q-- > 0;) {  // delete (:338-344): indices in descending order
    s.x.erase(s.x.begin() + (long)idxDelete[q]);
    for (auto& yk : s.y) yk.erase(yk.begin() + (long)idxDelete[q]);
  }
  return s;
}

template <class T>
static std::vector<T> cumsimpsonSorted(const std::vector<T>& Y, const std::vector<double>& X, const std::vector<double>& callerX);
template <class T>
static std::vector<T> cumsimpsonDiscrete(const std::vector<T>& Y, const std::vector<double>& X) {  // X sorted, duplicate-free
  return cumsimpsonSorted<T>(Y, X, X);
}
// the body of cumsimpson(Y, X) after its first line (:341-375): X, Y = (xSorted, ySorted); callerX = the argument X, which the last line interpolates back to
template <class T>
static std::vector<T> cumsimpsonSorted(const std::vector<T>& Y, const std::vector<double>& X, const std::vector<double>& callerX) {
  int N = (int)X.size();
  const int n = N;
  bool evenN = false;
  if (N < 3) throw std::invalid_argument("X and Y must have at least 3 elements to perform Simpson, use cumtrapz instead");
  if (N % 2 == 0) { evenN = true; N -= 1; }
  std::vector<T> y, dy;
  std::vector<double> xs;
  T integral = Y[0] - Y[0];
  y.push_back(integral); dy.push_back(Y[0]); xs.push_back(X[0]);
  for (int i = 0; i < (N - 1) / 2; ++i) {
    const double h1 = X[2 * i + 1] - X[2 * i];
    const double h2 = X[2 * i + 2] - X[2 * i + 1];
    const double alpha = (2.0 * cube(h2) - cube(h1) + 3.0 * h1 * sq(h2)) / (6.0 * h2 * (h2 + h1));
    const double beta = (cube(h2) + cube(h1) + 3.0 * h1 * h2 * (h2 + h1)) / (6.0 * h2 * h1);
    const double eta = (2.0 * cube(h1) - cube(h2) + 3.0 * h2 * sq(h1)) / (6.0 * h1 * (h2 + h1));
    integral = integral + (alpha * Y[2 * i + 2] + beta * Y[2 * i + 1] + eta * Y[2 * i]);  // `+=` :359
    y.push_back(integral); dy.push_back(Y[2 * i + 2]); xs.push_back(X[2 * i + 2]);
  }
  if (evenN) {
    const int last = n - 1;
    const double h1 = X[last - 1] - X[last - 2];
    const double h2 = X[last] - X[last - 1];
    const double alpha = (2.0 * sq(h2) + 3.0 * h1 * h2) / (6.0 * (h1 + h2));
    const double beta = (sq(h2) + 3.0 * h1 * h2) / (6.0 * h1);
    const double eta = -(cube(h2)) / (6.0 * h1 * (h1 + h2));
    integral = integral + (eta * Y[last - 2] + beta * Y[last - 1] + alpha * Y[last]);
    y.push_back(integral); dy.push_back(Y[last]); xs.push_back(X[last]);
  }
  return hermiteInterpolate<T>(callerX, xs, y, dy);  // :375
}

template <class T, class F>
static std::vector<T> cumtrapzFn(F f, const std::vector<double>& X, double dx) {  // integrate.nim:138-175
  std::vector<double> times;
  std::vector<T> dy, y;
  double t = *std::min_element(X.begin(), X.end());
  const double tEnd = *std::max_element(X.begin(), X.end()) + 1.0;  // "make sure to get the endpoint as well"
  T dyTemp = f(t), dyPrev = dyTemp;
  T integral = dyTemp - dyTemp;
  times.push_back(t); dy.push_back(dyTemp); y.push_back(integral);
  t += dx;
  while (t <= tEnd) {
    dyPrev = dyTemp;
    dyTemp = f(t);
    integral = integral + (0.5 * dx) * (dyPrev + dyTemp);  // integral += 0.5 * dx * (dyPrev + dyTemp)
    times.push_back(t); dy.push_back(dyTemp); y.push_back(integral);
    t += dx;
  }
  return hermiteInterpolate<T>(X, times, y, dy);
}

static long nimToInt(double v) { return (long)std::round(v); }  // system.toInt: rounds half away from zero

template <class T, class F>
static std::vector<T> cumsimpsonFn(F f, const std::vector<double>& X, double dx) {  // integrate.nim:377-400
  const double lo = *std::min_element(X.begin(), X.end()), hi = *std::max_element(X.begin(), X.end());
  const long N = nimToInt((hi - lo) / dx) + 2;
  std::vector<double> t;  // linspace(lo, hi, N), utils.nim:498-507
  const double step = (hi - lo) / (double)(N - 1);
  t.push_back(lo);
  for (long i = 1; i <= N - 2; ++i) t.push_back(lo + step * (double)i);
  t.push_back(hi);
  std::vector<T> dy;
  for (double x : t) dy.push_back(f(x));
  // cumsimpson(dy, t) sorts and trims (sortAndTrimDataset); a linspace with step > 0 is already sorted and duplicate-free,
  // anything else is refused by the entry point before we get here.
  const std::vector<T> ys = cumsimpsonDiscrete<T>(dy, t);
  return hermiteInterpolate<T>(X, t, ys, dy);
}

}  // namespace oracle

// =============================================================================================
// C entry points for ctypes (tests / smoke / bench cpu_baseline only)
// =============================================================================================
extern "C" {

struct oracle_options { double dt, dtMax, dtMin, tStart, absTol, relTol, scaleMax, scaleMin; };
struct oracle_stats { int64_t rhs_evals, steps, rejected; int32_t n_t, n_y, nan_abort, _pad; };

// newODEoptions (ode.nim:78-102). Returns 0, or -1 (ValueError) with nothing written.
int oracle_new_options(oracle_options* out, double dt, double absTol, double relTol, double dtMax, double dtMin,
                       double scaleMax, double scaleMin, double tStart) {
  try {
    oracle::ODEoptions o = oracle::newODEoptions(dt, absTol, relTol, dtMax, dtMin, scaleMax, scaleMin, tStart);
    std::memcpy(out, &o, sizeof(o));
    return 0;
  } catch (const std::invalid_argument&) { return -1; }
}

// Is `name` a valid integrator (case-insensitive)? returns method id or -2.
int oracle_integrator_id(const char* name) {
  const oracle::Method* m = oracle::methodByName(name);
  return m ? m->id : -2;
}

// The constants the tableau step procs above are compiled with, in declaration order: `names` receives them '\n'-separated, `values`
// their values.  Returns the count (or -1: not a tableau method, -2: buffers too small).  For tests/test_reference_text_pin.py.
int oracle_tableau(const char* integrator, char* names, int names_cap, double* values, int values_cap) {
  struct Entry { const char* name; double value; };
  std::vector<Entry> e;
  const oracle::Method* m = oracle::methodByName(integrator);
  if (!m) return -1;
  const std::string nm = m->name;
#define X(n, v) constexpr double n = v;
#define Y(n, v) e.push_back({#n, n});
  if (nm == "dopri54") { DOPRI54_CONSTS(X) DOPRI54_CONSTS(Y) }
  else if (nm == "tsit54") { TSIT54_CONSTS(X) TSIT54_CONSTS(Y) }
  else if (nm == "vern65") { VERN65_CONSTS(X) VERN65_CONSTS(Y) }
  else return -1;
#undef X
#undef Y
  std::string joined;
  for (size_t i = 0; i < e.size(); ++i) { if (i) joined += '\n'; joined += e[i].name; }
  if ((int)joined.size() + 1 > names_cap || (int)e.size() > values_cap) return -2;
  std::memcpy(names, joined.c_str(), joined.size() + 1);
  for (size_t i = 0; i < e.size(); ++i) values[i] = e[i].value;
  return (int)e.size();
}

// One IVP. dim == 0 → scalar `float` state path (T = float); dim >= 1 → Vector[float] path of that length.
// y_out is [n_t][max(dim,1)] row-major; only the first stats->n_y rows are written (reference quirk A.8).
int oracle_solve_ode(int rhs_kind, const double* rhs_params, int n_params, int dim, const double* y0, const double* tspan,
                     int n_t, const oracle_options* opt, const char* integrator, double* t_out, double* y_out,
                     oracle_stats* stats) {
  using namespace oracle;
  std::vector<double> env(1 + (size_t)n_params);
  env[0] = rhs_kind;
  for (int i = 0; i < n_params; ++i) env[1 + i] = rhs_params[i];
  ODEoptions o; std::memcpy(&o, opt, sizeof(o));
  Counters cnt;
  std::vector<double> tO;
  int rc;
  int ny = 0;
  try {
    if (dim == 0) {
      ODEProc<double> f{rhsScalar, env.data(), &cnt};
      std::vector<double> yO;
      rc = solveODE<double>(f, y0[0], tspan, n_t, o, integrator, tO, yO);
      if (rc) return rc;
      ny = (int)yO.size();
      for (int j = 0; j < ny; ++j) y_out[j] = yO[j];
    } else {
      ODEProc<Vec> f{rhsVector, env.data(), &cnt};
      Vec v0; v0.components.assign(y0, y0 + dim);
      std::vector<Vec> yO;
      rc = solveODE<Vec>(f, v0, tspan, n_t, o, integrator, tO, yO);
      if (rc) return rc;
      ny = (int)yO.size();
      for (int j = 0; j < ny; ++j)
        for (int c = 0; c < dim; ++c) y_out[(size_t)j * dim + c] = yO[j].components[c];
    }
  } catch (const std::invalid_argument&) { return -1; }
  for (size_t j = 0; j < tO.size(); ++j) t_out[j] = tO[j];
  if (stats) {
    stats->rhs_evals = cnt.rhsEvals; stats->steps = cnt.steps; stats->rejected = cnt.rejected;
    stats->n_t = (int)tO.size(); stats->n_y = ny; stats->nan_abort = cnt.nanAbort; stats->_pad = 0;
  }
  return 0;
}

// Batch of independent IVPs = the reference called once per IVP (ode.nim:589). Layout of y0 / y_out:
//   layout 0 (SoA): y0[c*N + i], y_out[(j*dimv + c)*N + i];  layout 1 (AoS): y0[i*dimv + c], y_out[(j*N + i)*dimv + c]
// with dimv = max(dim,1). ny_out[i] (nullable) = number of y rows the reference returns for IVP i;
// rows >= ny_out[i] are filled with NaN. steps_out / rejected_out nullable per-IVP counters.
// n_threads > 1 uses OpenMP over IVPs (all-cores CPU baseline); 1 = the single-threaded reference.
int oracle_solve_ode_batch(int rhs_kind, const double* rhs_params, int n_params, int dim, int layout, const double* y0,
                           int64_t N, const double* tspan, int n_t, const oracle_options* opt, const char* integrator,
                           double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out, int64_t* rejected_out,
                           int n_threads) {
  if (oracle_integrator_id(integrator) < 0) return -2;
  const int dimv = dim > 0 ? dim : 1;
  int rc_all = 0;
#pragma omp parallel for schedule(static) num_threads(n_threads) if (n_threads > 1)
  for (int64_t i = 0; i < N; ++i) {
    std::vector<double> y0i(dimv), tO(n_t), yO((size_t)n_t * dimv);
    for (int c = 0; c < dimv; ++c) y0i[c] = layout == 0 ? y0[(size_t)c * N + i] : y0[(size_t)i * dimv + c];
    oracle_stats st{};
    int rc = oracle_solve_ode(rhs_kind, rhs_params, n_params, dim, y0i.data(), tspan, n_t, opt, integrator, tO.data(),
                              yO.data(), &st);
    if (rc) { rc_all = rc; continue; }
    if (i == 0 && t_out) for (int j = 0; j < st.n_t; ++j) t_out[j] = tO[j];
    for (int j = 0; j < n_t; ++j)
      for (int c = 0; c < dimv; ++c) {
        const double v = j < st.n_y ? yO[(size_t)j * dimv + c] : NAN;
        if (layout == 0) y_out[((size_t)j * dimv + c) * N + i] = v;
        else y_out[((size_t)j * N + i) * dimv + c] = v;
      }
    if (ny_out) ny_out[i] = st.n_y;
    if (steps_out) steps_out[i] = st.steps;
    if (rejected_out) rejected_out[i] = st.rejected;
  }
  return rc_all;
}

// A batch whose members each have their own ctx (N reference calls, each closure capturing its own NumContext): the closure's
// environment of IVP i = [kind, shared[0..n_shared), per_ivp[r*N + i] for r < rows, aux_io[k*N + i] for k < n_aux]; the aux part is
// what the closure mutates (ode.nim:599) and is copied back after the call.  Layout of y0 / y_out as oracle_solve_ode_batch.
int oracle_solve_ode_batch_ctx(int rhs_kind, const double* shared, int n_shared, const double* per_ivp, int rows, double* aux_io, int n_aux,
                               int dim, int layout, const double* y0, int64_t N, const double* tspan, int n_t, const oracle_options* opt,
                               const char* integrator, double* t_out, double* y_out, int32_t* ny_out, int64_t* steps_out,
                               int64_t* rejected_out, int n_threads) {
  if (oracle_integrator_id(integrator) < 0) return -2;
  const int dimv = dim > 0 ? dim : 1;
  int rc_all = 0;
#pragma omp parallel for schedule(static) num_threads(n_threads) if (n_threads > 1)
  for (int64_t i = 0; i < N; ++i) {
    std::vector<double> y0i(dimv), tO(n_t), yO((size_t)n_t * dimv), par((size_t)n_shared + rows + n_aux);
    for (int k = 0; k < n_shared; ++k) par[k] = shared[k];
    for (int r = 0; r < rows; ++r) par[(size_t)n_shared + r] = per_ivp[(size_t)r * N + i];
    for (int k = 0; k < n_aux; ++k) par[(size_t)n_shared + rows + k] = aux_io[(size_t)k * N + i];
    for (int c = 0; c < dimv; ++c) y0i[c] = layout == 0 ? y0[(size_t)c * N + i] : y0[(size_t)i * dimv + c];
    oracle_stats st{};
    // oracle_solve_ode copies the parameters into the closure's env; the mutable tail is needed back, so the env is built here
    using namespace oracle;
    std::vector<double> env(1 + par.size());
    env[0] = rhs_kind;
    for (size_t k = 0; k < par.size(); ++k) env[1 + k] = par[k];
    ODEoptions o; std::memcpy(&o, opt, sizeof(o));
    Counters cnt;
    std::vector<double> tV;
    std::vector<Vec> yV;
    int rc;
    try {
      ODEProc<Vec> f{rhsVector, env.data(), &cnt};
      Vec v0; v0.components.assign(y0i.begin(), y0i.end());
      rc = solveODE<Vec>(f, v0, tspan, n_t, o, integrator, tV, yV);
    } catch (const std::invalid_argument&) { rc = -1; }
    if (rc) { rc_all = rc; continue; }
    st.n_t = (int)tV.size(); st.n_y = (int)yV.size(); st.steps = cnt.steps; st.rejected = cnt.rejected;
    if (i == 0 && t_out) for (int j = 0; j < st.n_t; ++j) t_out[j] = tV[j];
    for (int j = 0; j < n_t; ++j)
      for (int c = 0; c < dimv; ++c) {
        const double v = j < st.n_y ? yV[j].components[c] : NAN;
        if (layout == 0) y_out[((size_t)j * dimv + c) * N + i] = v;
        else y_out[((size_t)j * N + i) * dimv + c] = v;
      }
    for (int k = 0; k < n_aux; ++k) aux_io[(size_t)k * N + i] = env[1 + (size_t)n_shared + rows + k];
    if (ny_out) ny_out[i] = st.n_y;
    if (steps_out) steps_out[i] = st.steps;
    if (rejected_out) rejected_out[i] = st.rejected;
  }
  return rc_all;
}

// One IntegratorProc call (ode.nim:38) on one IVP — for checking the step-streaming kernels.
// in: t, y[dimv], fsal[dimv], dt;  out: y_new[dimv], fsal_new[dimv], *dt_used, *error
int oracle_step(int rhs_kind, const double* rhs_params, int n_params, int dim, const char* integrator, const oracle_options* opt,
                double t, const double* y, const double* fsal, double dt, double* y_new, double* fsal_new, double* dt_used,
                double* error) {
  using namespace oracle;
  const Method* m = methodByName(integrator);
  if (!m) return -2;
  std::vector<double> env(1 + (size_t)n_params);
  env[0] = rhs_kind;
  for (int i = 0; i < n_params; ++i) env[1 + i] = rhs_params[i];
  ODEoptions o; std::memcpy(&o, opt, sizeof(o));
  if (dim == 0) {
    ODEProc<double> f{rhsScalar, env.data(), nullptr};
    StepResult<double> r = stepperFor<double>(m->id)(f, t, y[0], fsal[0], dt, o);
    y_new[0] = r.yNew; fsal_new[0] = r.fsal; *dt_used = r.dt; *error = r.error;
  } else {
    ODEProc<Vec> f{rhsVector, env.data(), nullptr};
    Vec vy, vf; vy.components.assign(y, y + dim); vf.components.assign(fsal, fsal + dim);
    StepResult<Vec> r = stepperFor<Vec>(m->id)(f, t, vy, vf, dt, o);
    for (int c = 0; c < dim; ++c) { y_new[c] = r.yNew.components[c]; fsal_new[c] = r.fsal.components[c]; }
    *dt_used = r.dt; *error = r.error;
  }
  return 0;
}

// RHS evaluation alone (to pin the RHS library definitions).
int oracle_rhs(int rhs_kind, const double* rhs_params, int n_params, int dim, double t, const double* y, double* dy) {
  using namespace oracle;
  std::vector<double> env(1 + (size_t)n_params);
  env[0] = rhs_kind;
  for (int i = 0; i < n_params; ++i) env[1 + i] = rhs_params[i];
  if (dim == 0) { dy[0] = rhsScalar(t, y[0], env.data()); return 0; }
  Vec vy; vy.components.assign(y, y + dim);
  Vec r = rhsVector(t, vy, env.data());
  for (int c = 0; c < dim; ++c) dy[c] = r.components[c];
  return 0;
}

// hermiteSpline (utils.nim:273-279) on scalars, and linspace (utils.nim:498-507)
double oracle_hermite_spline(double x, double x1, double x2, double y1, double y2, double dy1, double dy2) {
  return oracle::hermiteSpline<double>(x, x1, x2, y1, y2, dy1, dy2);
}
// The step-size controller's factor alone (ode.nim:71 / :537): min(4, max(0.125, 0.9 * pow(1/error, 1/order))) with the C
// library's pow, as Nim's std/math pow resolves to.  Lets the tests compare the device factor with the reference's libm directly.
void oracle_controller_factor(const double* error, int64_t n, int order_i, double* out) {
  const double order = (double)order_i;
  for (int64_t i = 0; i < n; ++i) out[i] = oracle::nmin(4, oracle::nmax(0.125, 0.9 * std::pow(1 / error[i], 1 / order)));
}
int oracle_linspace(double x1, double x2, int N, double* out) {
  if (N <= 0) return -1;  // ValueError utils.nim:500-501
  const double dx = (x2 - x1) / (double)(N - 1);
  int k = 0;
  out[k++] = x1;
  for (int i = 1; i <= N - 2; ++i) out[k++] = x1 + dx * (double)i;
  out[k++] = x2;  // N == 1 yields two points in the reference as well (x1, x2); caller must size N+1
  return k;
}

// newHermiteSpline(X, Y, dY) + eval / derivEval (src/numericalnim/interpolate.nim:114-115 findInterval, :186-217 handlers,
// :299-345 / :346-390 extrapolation) for one scalar series.  X must already be sorted and duplicate-free (what
// sortAndTrimDataset, :231, produces).  extrap: 0 Constant 1 Edge 2 Linear 3 Native 4 Error.  Returns -1 for Error out of range.
int oracle_hermite_interp(const double* X, int n, const double* Y, const double* dY, const double* xq, int nq, int deriv, int extrap,
                          double extrapValue, double* out) {
  auto handler = [&](double x, bool d) -> double {  // eval_hermitespline :186-201 / derivEval_hermitespline :203-217
    // findInterval (:114-115): clamp(lowerbound(X, x) - 1, 0, high - 1)
    int k = (int)(std::lower_bound(X, X + n, x) - X) - 1;
    if (k < 0) k = 0;
    if (k > n - 2) k = n - 2;
    const double xDiff = X[k + 1] - X[k];
    const double t = (x - X[k]) / xDiff;
    const double t2 = t * t;
    const double p1 = Y[k], p2 = Y[k + 1], m1 = dY[k], m2 = dY[k + 1];
    if (!d) {
      const double t3 = t2 * t;
      const double h00 = 2 * t3 - 3 * t2 + 1;
      const double h10 = t3 - 2 * t2 + t;
      const double h01 = -2 * t3 + 3 * t2;
      const double h11 = t3 - t2;
      return h00 * p1 + h10 * xDiff * m1 + h01 * p2 + h11 * xDiff * m2;
    }
    const double h00 = 6 * t2 - 6 * t;
    const double h10 = 3 * t2 - 4 * t + 1;
    const double h01 = -6 * t2 + 6 * t;
    const double h11 = 3 * t2 - 2 * t;
    return (h00 * p1 + h10 * xDiff * m1 + h01 * p2 + h11 * xDiff * m2) / xDiff;
  };
  for (int q = 0; q < nq; ++q) {
    const double x = xq[q];
    const bool xLeft = x < X[0], xRight = x > X[n - 1];
    if (xLeft || xRight) {  // :317-341 / :364-388
      if (extrap == 0) { out[q] = extrapValue; continue; }
      if (extrap == 1) {
        if (!deriv) out[q] = xLeft ? Y[0] : Y[n - 1];
        else out[q] = xLeft ? handler(X[0], true) : handler(X[n - 1], true);
        continue;
      }
      if (extrap == 2) {
        const double x0 = xLeft ? X[0] : X[n - 2], x1 = xLeft ? X[1] : X[n - 1];
        const double y0 = !deriv ? (xLeft ? Y[0] : Y[n - 2]) : handler(x0, true);
        const double y1 = !deriv ? (xLeft ? Y[1] : Y[n - 1]) : handler(x1, true);
        const double k = (x - x0) / (x1 - x0);
        out[q] = y0 + k * (y1 - y0);
        continue;
      }
      if (extrap == 4) return -1;
    }
    out[q] = handler(x, deriv != 0);
  }
  return 0;
}

// newHermiteSpline(X, Y) without derivatives (src/numericalnim/interpolate.nim:241-253): three-point difference slopes for one
// scalar series; X sorted and duplicate-free, n >= 2.
int oracle_hermite_slopes(const double* X, int n, const double* Y, double* dY) {
  if (n < 2) return -1;
  const int highest = n - 1;
  dY[0] = (Y[1] - Y[0]) / (X[1] - X[0]);                                               // :247
  dY[highest] = (Y[highest] - Y[highest - 1]) / (X[highest] - X[highest - 1]);         // :248-249
  for (int i = 1; i <= highest - 1; ++i)                                               // :250-252
    dY[i] = 0.5 * ((Y[i + 1] - Y[i]) / (X[i + 1] - X[i]) + (Y[i] - Y[i - 1]) / (X[i] - X[i - 1]));
  return 0;
}

// cumtrapz(Y, X) for discrete points (src/numericalnim/integrate.nim:120-135) on one scalar series; X sorted and
// duplicate-free (sortAndTrimDataset's postcondition).  trapz(Y, X) (:104-117) equals the last entry for finite data.
// X in any order: `let (xSorted, ySorted) = sortAndTrimDataset(@X, @Y)` (:130) first.  Returns the number of rows (distinct abscissae), -1 for the
// reference's ValueError (impure duplicates), -2 for NaN in X (no defined order in the reference's sort: nothing to restate).
int oracle_cumtrapz(const double* Xc, int nc, const double* Yc, double* out) {
  oracle::SortedTrimmed st;
  try { st = oracle::sortAndTrimDataset(std::vector<double>(Xc, Xc + nc), {std::vector<double>(Yc, Yc + nc)}); }
  catch (const std::invalid_argument&) { return -1; }
  catch (const std::domain_error&) { return -2; }
  const std::vector<double>&X = st.x, &Y = st.y[0];
  const int n = (int)X.size();
  out[0] = Y[0] - Y[0];            // "get the right kind of zero" (:131)
  double integral = Y[0] - Y[0];   // :132
  for (int i = 0; i <= n - 2; ++i) {
    integral += 0.5 * (X[i + 1] - X[i]) * (Y[i + 1] + Y[i]);  // :134
    out[i + 1] = integral;
  }
  return n;
}

// sortAndTrimDataset(x, @[y_0 .. y_{nY-1}]) (utils.nim:404-407) on scalar series.  Returns the rows of the result, -1 for ValueError (impure duplicates),
// -2 for NaN in X.
int oracle_sort_and_trim(const double* X, int n, const double* const* Y, int nY, double* Xout, double* const* Yout) {
  std::vector<std::vector<double>> ys;
  for (int k = 0; k < nY; ++k) ys.emplace_back(Y[k], Y[k] + n);
  oracle::SortedTrimmed st;
  try { st = oracle::sortAndTrimDataset(std::vector<double>(X, X + n), ys); }
  catch (const std::invalid_argument&) { return -1; }
  catch (const std::domain_error&) { return -2; }
  std::copy(st.x.begin(), st.x.end(), Xout);
  for (int k = 0; k < nY; ++k) std::copy(st.y[(size_t)k].begin(), st.y[(size_t)k].end(), Yout[k]);
  return (int)st.x.size();
}

// cumsimpson(Y, X) for discrete points (src/numericalnim/integrate.nim:329-375) on one scalar series: composite Simpson on
// pairs of intervals (non-uniform weights :354-359, odd-tail correction :364-373) gives the integral at every second
// point; hermiteInterpolate (utils.nim:282-312, sorted branch) with dy = Y fills in all points of X.
int oracle_cumsimpson(const double* X, int n, const double* Y, double* out) {
  // X in any order: `var (xSorted, ySorted) = sortAndTrimDataset(@X, @Y)` (:340); the rule on the sorted data (:341-373, cumsimpsonSorted above, the restatement
  // shared with the function form); the result at the CALLER's abscissae: hermiteInterpolate(X, xs, y, dy) (:375), sorted or unsorted branch as X is.
  // Returns the number of rows, -1 for the reference's ValueError (impure duplicates, fewer than 3 distinct abscissae), -2 for NaN in X.
  const std::vector<double> callerX(X, X + n);
  try {
    const oracle::SortedTrimmed st = oracle::sortAndTrimDataset(callerX, {std::vector<double>(Y, Y + n)});
    const std::vector<double> r = oracle::cumsimpsonSorted<double>(st.y[0], st.x, callerX);
    std::copy(r.begin(), r.end(), out);
    return (int)r.size();
  } catch (const std::invalid_argument&) { return -1; }
  catch (const std::domain_error&) { return -2; }
}

// cumtrapz(f, X, ctx, dx) (rule 0) / cumsimpson(f, X, ctx, dx) (rule 1) with f(x) := rhs(x, y = 0, params) — the integrand is a
// function of x alone (NumContextProc, integrate.nim:9).  dim == 0: T = float; dim >= 1: T = Vector[float].
// out is [n_x][max(dim,1)]; returns the number of rows produced (the reference can return fewer than n_x), -1 for a ValueError.
int oracle_cumquad_fn(int rule, int rhs_kind, const double* rhs_params, int n_params, int dim, const double* X, int n_x, double dx,
                      double* out) {
  using namespace oracle;
  std::vector<double> env(1 + (size_t)n_params);
  env[0] = rhs_kind;
  for (int i = 0; i < n_params; ++i) env[1 + i] = rhs_params[i];
  const std::vector<double> x(X, X + n_x);
  try {
    if (dim == 0) {
      auto f = [&](double t) { return rhsScalar(t, 0.0, env.data()); };
      const std::vector<double> r = rule == 0 ? cumtrapzFn<double>(f, x, dx) : cumsimpsonFn<double>(f, x, dx);
      for (size_t j = 0; j < r.size(); ++j) out[j] = r[j];
      return (int)r.size();
    }
    Vec zero; zero.components.assign((size_t)dim, 0.0);
    auto f = [&](double t) { return rhsVector(t, zero, env.data()); };
    const std::vector<Vec> r = rule == 0 ? cumtrapzFn<Vec>(f, x, dx) : cumsimpsonFn<Vec>(f, x, dx);
    for (size_t j = 0; j < r.size(); ++j)
      for (int c = 0; c < dim; ++c) out[j * (size_t)dim + c] = r[j].components[c];
    return (int)r.size();
  } catch (const std::invalid_argument&) { return -1; }
}

// Vector operator probes (tests/test_vector.nim semantics). op: 0 '+', 1 '-', 2 scalar*V, 3 abs, 4 *. 5 /. 6 d +. V
int oracle_vector_op(int op, const double* a, int na, const double* b, int nb, double d, double* out) {
  using namespace oracle;
  Vec va, vb; va.components.assign(a, a + na); if (b) vb.components.assign(b, b + nb);
  try {
    Vec r;
    switch (op) {
      case 0: r = va + vb; break;
      case 1: r = va - vb; break;
      case 2: r = d * va; break;
      case 3: r = nabs(va); break;
      case 4: r = dotMul(va, vb); break;
      case 5: r = dotDiv(va, vb); break;
      case 6: r = dotAdd(d, va); break;
      case 7: out[0] = nsum(va); return 1;
      default: return -2;
    }
    for (size_t i = 0; i < r.components.size(); ++i) out[i] = r.components[i];
    return (int)r.components.size();
  } catch (const std::invalid_argument&) { return -1; }  // ValueError utils.nim:26
}

}  // extern "C"

// C++ host-side parity harness: reads like /root/reference/tests/test_ode.nim (same RHS, y0, tspan, option sets,
// tolerances and checks), driving the HIP backend through include/numericalnim_hip.hpp.  Built and run by
// tests/test_gpu_cpp_host.py on the GPU box (g++ + libnnhip_ode.so).
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "numericalnim_hip.hpp"

using namespace numericalnim;

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { ++failures; std::printf("  CHECK failed: %s (line %d)\n", #cond, __LINE__); } \
  } while (0)

static std::vector<double> linspace(double x1, double x2, int N) {  // utils.nim:498-507
  std::vector<double> r;
  const double dx = (x2 - x1) / (double)(N - 1);
  r.push_back(x1);
  for (int i = 1; i <= N - 2; ++i) r.push_back(x1 + dx * (double)i);
  r.push_back(x2);
  return r;
}
static bool isClose(double a, double b, double tol) { return std::fabs(a - b) <= tol; }  // utils.nim:270,474-479

int main() {
  // test_ode.nim:5-16
  NumContext<double> ctx;
  ctx.setF("a", -0.1);                       // proc f(x, y, ctx) = -0.1 * y
  const RhsSpec f = rhsLinear();
  const ODEoptions oo = newODEoptions(/*dt=*/1e-6, /*absTol=*/1e-4, /*relTol=*/1e-8);   // newODEoptions(relTol=1e-8, dt=1e-6)
  const ODEoptions ooVector = newODEoptions(/*dt=*/1e-2, 1e-4, /*relTol=*/1e-8);
  const std::vector<double> tspan = linspace(-10.0, 10.0, 100);
  OdeBatch y0 = OdeBatch::zeros(1, 1);
  y0.at(0, 0) = 1.0;
  OdeBatch y0Vector = OdeBatch::zeros(1, 3, NNHIP_LAYOUT_AOS);
  for (int c = 0; c < 3; ++c) y0Vector.at(0, c) = 1.0;

  struct Case { const char* name; const char* integ; const ODEoptions* opt; double tol; };
  const Case scalar[] = {{"DOPRI54, default", "dopri54", nullptr, 1e-4}, {"DOPRI54, tol = 1e-8", "dopri54", &oo, 1e-8},
                         {"RK4, default", "rk4", nullptr, 1e-4},         {"RK4, dt = 1e-6", "rk4", &oo, 1e-8},
                         {"Tsit54, default", "tsit54", nullptr, 1e-4},   {"Tsit54, tol = 1e-8", "tsit54", &oo, 1e-8}};
  for (const Case& c : scalar) {
    std::printf("test \"%s\"\n", c.name);
    OdeSolution s = solveODE(f, y0, tspan, c.opt ? *c.opt : DEFAULT_ODEoptions(), &ctx, c.integ);
    CHECK(s.t == tspan);                                      // check t == tspan
    for (size_t i = 0; i < s.y.size(); ++i) CHECK(isClose(s.y[i].at(0, 0), std::exp(-0.1 * tspan[i]), c.tol));
  }
  const Case vec[] = {{"DOPRI54 Vector, default", "dopri54", nullptr, 1e-4}, {"DOPRI54 Vector, tol = 1e-8", "dopri54", &ooVector, 1e-8},
                      {"RK4 Vector, default", "rk4", nullptr, 1e-4},         {"RK4 Vector, dt = 1e-2", "rk4", &ooVector, 1e-8},
                      {"Tsit54 Vector, default", "tsit54", nullptr, 1e-4},   {"Tsit54 Vector, dt = 1e-2", "tsit54", &ooVector, 1e-8}};
  for (const Case& c : vec) {
    std::printf("test \"%s\"\n", c.name);
    OdeSolution s = solveODE(f, y0Vector, tspan, c.opt ? *c.opt : DEFAULT_ODEoptions(), &ctx, c.integ);
    CHECK(s.t == tspan);
    for (size_t i = 0; i < s.y.size(); ++i) {  // isClose on Vector: norm2(a-b)/len <= tol (utils.nim:252)
      double n2 = 0.0;
      for (int k = 0; k < 3; ++k) { const double d = s.y[i].at(0, k) - std::exp(-0.1 * tspan[i]); n2 += d * d; }
      CHECK(std::sqrt(n2) / 3.0 <= c.tol);
    }
  }
  // an arbitrary RHS handed over as source must reproduce the compiled-in one bit for bit
  {
    std::printf("test \"user RHS from source\"\n");
    const RhsSpec fsrc = rhsFromSource(1, "dy[0] = y[0] * p[0];", {"a"});
    OdeSolution a = solveODE(fsrc, y0, tspan, DEFAULT_ODEoptions(), &ctx, "tsit54");
    OdeSolution b = solveODE(f, y0, tspan, DEFAULT_ODEoptions(), &ctx, "tsit54");
    for (size_t i = 0; i < a.y.size(); ++i) CHECK(a.y[i].at(0, 0) == b.y[i].at(0, 0));
    bool threwSrc = false;
    try { rhsFromSource(1, "dy[0] = undefined_symbol;"); } catch (const std::invalid_argument&) { threwSrc = true; }
    CHECK(threwSrc);
  }
  // a parameter sweep: every IVP with its own `a` must equal separate solves with that `a`
  {
    std::printf("test \"parameter sweep\"\n");
    OdeBatch yb = OdeBatch::zeros(3, 1);
    for (int i = 0; i < 3; ++i) yb.at(i, 0) = 1.0;
    const std::vector<double> as = {-0.1, -0.5, 0.25};
    OdeSolution all = solveODE(f, yb, tspan, DEFAULT_ODEoptions(), &ctx, "dopri54", 0, 1, {as});
    for (int i = 0; i < 3; ++i) {
      NumContext<double> ci;
      ci.setF("a", as[i]);
      OdeBatch y1 = OdeBatch::zeros(1, 1);
      y1.at(0, 0) = 1.0;
      OdeSolution one = solveODE(f, y1, tspan, DEFAULT_ODEoptions(), &ci, "dopri54");
      for (size_t j = 0; j < one.y.size(); ++j) CHECK(all.y[j].at(i, 0) == one.y[j].at(0, 0));
    }
    // divergence binning below the C ABI: caller's key and automatic probe give the bits of the unsorted solve, rows in caller order
    OdeBatch yw = OdeBatch::zeros(257, 1);
    std::vector<double> aw(257);
    for (int i = 0; i < 257; ++i) { yw.at(i, 0) = 1.0 + 0.001 * i; aw[i] = -0.05 - 0.037 * ((i * 73) % 257); }
    const OdeSolution plain = solveODE(f, yw, tspan, DEFAULT_ODEoptions(), &ctx, "tsit54", 0, 1, {aw});
    const OdeSolution byKey = solveODE(f, yw, tspan, DEFAULT_ODEoptions(), &ctx, "tsit54", 0, 1, {aw}, aw);
    const OdeSolution byProbe = solveODE(f, yw, tspan, DEFAULT_ODEoptions(), &ctx, "tsit54", 0, 1, {aw}, {}, true);
    CHECK(byKey.t == plain.t && byProbe.t == plain.t && byKey.ny == plain.ny && byProbe.ny == plain.ny);
    for (size_t j = 0; j < plain.y.size(); ++j) CHECK(byKey.y[j].data == plain.y[j].data && byProbe.y[j].data == plain.y[j].data);
  }
  // N separate calls in one launch: call i owns its tspan end and its option object, and must equal the 1-IVP solve with them
  {
    std::printf("test \"per-call tspan and options\"\n");
    const int n = 9;
    OdeBatch yb = OdeBatch::zeros(n, 1);
    std::vector<double> tEnd(n);
    std::vector<ODEoptions> opts(n);
    for (int i = 0; i < n; ++i) {
      yb.at(i, 0) = 1.0 + 0.125 * i;
      tEnd[i] = (i == 4) ? 0.25 : (i % 2 ? -1.0 - 0.5 * i : 0.5 + 0.75 * i);   // both directions; IVP 4: tEnd == tStart below
      opts[i] = newODEoptions(1e-3 * (1 + i), i % 3 ? 1e-6 : 1e-9, i % 3 ? 1e-6 : 1e-9, 0.05 + 0.01 * i, 1e-5, 4.0, 0.1, i == 4 ? 0.25 : 0.0);
    }
    opts[7].dtMax = 1e-6;  // dtMax < dtMin: newODEoptions raises for this call alone (ode.nim:95-96)
    for (const char* integ : {"tsit54", "rk4"}) {
      const OdeSolution all = solveODECalls(f, yb, tEnd, opts, &ctx, integ);
      CHECK(all.y.size() == 2 && all.ny.size() == (size_t)n);
      for (int i = 0; i < n; ++i) {
        if (i == 7 && std::string(integ) == "tsit54") { CHECK(all.ny[i] == -1 && std::isnan(all.y[0].at(i, 0)) && std::isnan(all.y[1].at(i, 0))); continue; }
        OdeBatch y1 = OdeBatch::zeros(1, 1);
        y1.at(0, 0) = yb.at(i, 0);
        const OdeSolution one = solveODE(f, y1, {opts[i].tStart, tEnd[i]}, opts[i], &ctx, integ);
        CHECK(all.ny[i] == one.ny[0]);
        for (int j = 0; j < one.ny[0]; ++j) CHECK(all.y[j].at(i, 0) == one.y[j].at(0, 0));
      }
      const OdeSolution same = solveODECalls(f, yb, tEnd, {opts[2]}, &ctx, integ);   // one option object for every call
      OdeBatch y1 = OdeBatch::zeros(1, 1);
      y1.at(0, 0) = yb.at(5, 0);
      const OdeSolution one = solveODE(f, y1, {opts[2].tStart, tEnd[5]}, opts[2], &ctx, integ);
      for (int j = 0; j < 2; ++j) CHECK(same.y[j].at(5, 0) == one.y[j].at(0, 0));
    }
    bool threwLen = false;
    try { solveODECalls(f, yb, {1.0, 2.0}, opts, &ctx); } catch (const std::invalid_argument&) { threwLen = true; }
    CHECK(threwLen);
    // every call its own 4-point tspan (unsorted, both sides of its tStart): rows and times of the 1-IVP solves
    std::vector<std::vector<double>> tspans(n), tOut;
    for (int i = 0; i < n; ++i) tspans[i] = {opts[i].tStart + 0.4 + 0.05 * i, opts[i].tStart - 0.3, opts[i].tStart, opts[i].tStart + 0.1};
    opts[7] = opts[6];
    const OdeSolution grid = solveODECalls(f, yb, tspans, tOut, opts, &ctx, "dopri54");
    CHECK(grid.y.size() == 4 && tOut.size() == (size_t)n);
    for (int i = 0; i < n; ++i) {
      OdeBatch y1 = OdeBatch::zeros(1, 1);
      y1.at(0, 0) = yb.at(i, 0);
      const OdeSolution one = solveODE(f, y1, tspans[i], opts[i], &ctx, "dopri54");
      CHECK(tOut[i] == one.t && grid.ny[i] == one.ny[0]);
      for (int j = 0; j < one.ny[0]; ++j) CHECK(grid.y[j].at(i, 0) == one.y[j].at(0, 0));
    }
  }
  // the consumers, as tests/test_integrate.nim:19-21, 67-95 and tests/test_interpolate.nim:5-18, 104-145 use them
  {
    std::printf("test \"cumtrapz / cumsimpson / HermiteSpline\"\n");
    const double pi = 3.14159265358979323846;
    const std::vector<double> X = linspace(0.0, 1.5 * pi, 17);
    std::vector<OdeBatch> Y(X.size(), OdeBatch::zeros(2, 1));  // two series: 2 cos x and 4 cos x
    for (size_t j = 0; j < X.size(); ++j) { Y[j].at(0, 0) = 2.0 * std::cos(X[j]); Y[j].at(1, 0) = 4.0 * std::cos(X[j]); }
    const auto ct = cumtrapz(Y, X), cs = cumsimpson(Y, X);
    for (size_t j = 0; j < X.size(); ++j) {
      CHECK(isClose(ct[j].at(0, 0), 2.0 * std::sin(X[j]), 1e-1) && isClose(cs[j].at(0, 0), 2.0 * std::sin(X[j]), 1e-3));  // :67-70, :82-85
      CHECK(isClose(cs[j].at(1, 0), 4.0 * std::sin(X[j]), 2e-3));
    }
    const RhsSpec acos = rhsFromSource(1, "dy[0] = p[0] * cos(t);", {"a"}, {{"a", 2.0}}, "acos_cpp");
    for (double dx : {1e-5, 0.1}) {  // "cumtrapz func ..." / "cumsimpson func ..." (:72-95)
      const auto ft = cumtrapz<double>(acos, X, nullptr, dx), fs = cumsimpson<double>(acos, X, nullptr, dx);
      CHECK(ft.size() == X.size() && fs.size() == X.size());
      for (size_t j = 0; j < X.size(); ++j) CHECK(isClose(ft[j].at(0, 0), 2.0 * std::sin(X[j]), 1e-1) && isClose(fs[j].at(0, 0), 2.0 * std::sin(X[j]), 1e-3));
    }
    const auto sw = cumsimpson<double>(acos, X, nullptr, 1e-2, {{1.0, 2.0, 3.0}});  // a parameter sweep: three amplitudes at once
    for (size_t j = 0; j < X.size(); ++j) for (int i = 0; i < 3; ++i) CHECK(isClose(sw[j].at(i, 0), (i + 1.0) * std::sin(X[j]), 1e-3));
    const std::vector<double> t = linspace(0.0, 10.0, 100);
    std::vector<OdeBatch> ys(t.size(), OdeBatch::zeros(1, 1)), dys(t.size(), OdeBatch::zeros(1, 1));
    for (size_t j = 0; j < t.size(); ++j) { ys[j].at(0, 0) = std::sin(t[j]); dys[j].at(0, 0) = std::cos(t[j]); }
    std::vector<double> tTest;  // arange(0.0, 10.0, 0.2345)
    for (int i = 0; i <= (int)std::floor(10.0 / 0.2345); ++i) tTest.push_back(0.0 + (double)i * 0.2345);
    const HermiteSpline h2 = newHermiteSpline(t, ys, dys), h1 = newHermiteSpline(t, ys);
    const auto e2 = h2.eval(t), e1 = h1.eval(t), b2 = h2.eval(tTest), d2 = h2.derivEval(tTest), d1 = h1.derivEval(tTest);
    for (size_t j = 0; j < t.size(); ++j) CHECK(isClose(e2[j].at(0, 0), ys[j].at(0, 0), 1e-15) && isClose(e1[j].at(0, 0), ys[j].at(0, 0), 1e-15));
    for (size_t j = 0; j < tTest.size(); ++j) {
      CHECK(isClose(b2[j].at(0, 0), std::sin(tTest[j]), 1e-4));
      CHECK(std::fabs(d2[j].at(0, 0) - std::cos(tTest[j])) < 1e-5 && std::fabs(d1[j].at(0, 0) - std::cos(tTest[j])) < 2e-3);
    }
    bool threwX = false;
    try { h2.eval({11.0}, ExtrapolateKind::Error); } catch (const std::invalid_argument&) { threwX = true; }
    CHECK(threwX);
  }
  {  // NumContext in full: every system its own 3 x 3 matrix (per-IVP tValues), a shared forcing vector, and a mutable call counter
    const int n = 64, d = 3;
    const RhsSpec fm = rhsFromSourceCtx(d, "for (int r = 0; r < dim; ++r) { double acc = A(r*dim)*y[0]; for (int k = 1; k < dim; ++k) acc = acc + A(r*dim+k)*y[k];"
                                           " dy[r] = p[0]*acc + g[r]; } aux(0) = aux(0) + 1.0;", {"s"}, {{"g", d, false}, {"A", d * d, true}}, /*nAux=*/1, {}, "mirror_matvec");
    NumContext<double> c2;
    c2.setF("s", 0.5);
    std::vector<double> g = {0.1, -0.2, 0.05}, A((size_t)d * d * n), aux0((size_t)n, 0.0);
    for (int i = 0; i < n; ++i)
      for (int r = 0; r < d; ++r)
        for (int k = 0; k < d; ++k) A[(size_t)(r * d + k) * n + i] = (r == k ? -1.0 - 0.01 * i : 0.1 * (r - k));   // row r*d+k of IVP i
    bindCtx(fm, g, A, aux0, 1, n);
    OdeBatch yb = OdeBatch::zeros(n, d);
    for (int i = 0; i < n; ++i) for (int k = 0; k < d; ++k) yb.at(i, k) = 1.0 + 0.1 * k;
    const OdeSolution all = solveODE(fm, yb, {0.0, 1.0}, DEFAULT_ODEoptions(), &c2, "tsit54");
    const std::vector<double> calls = readAux(fm, 1, n);
    // IVP 5 alone, bound as a batch of one with its own matrix: the same bits (N reference calls, each closure its own ctx)
    std::vector<double> A5((size_t)d * d), one0(1, 0.0);
    for (int q = 0; q < d * d; ++q) A5[q] = A[(size_t)q * n + 5];
    bindCtx(fm, g, A5, one0, 1, 1);
    OdeBatch y5 = OdeBatch::zeros(1, d);
    for (int k = 0; k < d; ++k) y5.at(0, k) = yb.at(5, k);
    const OdeSolution one = solveODE(fm, y5, {0.0, 1.0}, DEFAULT_ODEoptions(), &c2, "tsit54");
    for (int k = 0; k < d; ++k) CHECK(all.y[1].at(5, k) == one.y[1].at(0, k));
    CHECK(calls[5] == readAux(fm, 1, 1)[0] && calls[5] > 100.0);   // f(t0) twice, then six stages per attempt (ode.nim:498,506,362-374)
    CHECK(std::fabs(all.y[1].at(5, 0)) < 1.0);
  }
  // error behaviour: ValueError analogues
  bool threw = false;
  try { solveODE(f, y0, tspan, DEFAULT_ODEoptions(), &ctx, "rk5"); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { newODEoptions(1e-4, 1e-4, 1e-4, /*dtMax=*/1e-5, /*dtMin=*/1e-4); } catch (const std::invalid_argument&) { threw = true; }
  CHECK(threw);
  std::printf(failures ? "FAILED (%d)\n" : "OK\n", failures);
  return failures ? 1 : 0;
}

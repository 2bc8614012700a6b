// adv_poll_schedule.hpp — how many launches the adaptive streaming driver enqueues before it asks the device "is anyone still integrating?".
// Plain C++ (no HIP): used by ode_capi_stream.hip and, as it stands, by tests/cpp/test_poll_schedule.cpp, which replays the driver's loop against a
// simulated batch.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>

namespace nnhip {
struct AdvPollSchedule {
  // uniform: groups of `checkEvery` launches (a caller's explicit check_every; hipGraph replay).  Otherwise the library's own schedule: every step
  // is at most dtMax long — the first is sqrt(dtMax * dtMin), the controller clamps the others (ode.nim:538-541, :72-76), :525 only shortens — so no
  // IVP can reach tEnd in fewer than n0 = ceil((tEnd - t0) / dtMax) iterations: those launches go out unpolled (in slices of at most 4096: a batch that
  // NaN-aborts retires early, and nothing enqueued can be taken back), then groups of 2, 2, 4, 8, 8 ... .  A homogeneous batch at loose tolerances needs
  // n0 + 2 iterations (the two short first steps) and ends after n0 + 4 launches: BASELINE's C3 / C4 took 112 launches for their 102 iterations with
  // uniform groups of 8, 104 now.
  bool uniform = true;
  int checkEvery = 8;
  int64_t n0 = 0, maxLaunches = 0, issued = 0;
  int tail = 0;
  static AdvPollSchedule make(bool uniformGroups, int checkEvery, double t0, double tEnd, double dtMax, int stepsPerLaunch, int64_t maxLaunches) {
    AdvPollSchedule s;
    s.uniform = uniformGroups; s.checkEvery = checkEvery > 0 ? checkEvery : 8; s.maxLaunches = maxLaunches;
    if (!uniformGroups) {
      const double span = (tEnd - t0) / dtMax * (1.0 - 1e-9);  // (the margin: t accumulates rounding errors of a few ulp per step)
      s.n0 = span < 1e15 ? (int64_t)std::ceil(span) : (int64_t)1 << 50;  // (NaN, e.g. dtMax = 0 with t0 = tEnd: the comparison is false -> "unbounded", capped by the slices)
      if (s.n0 < 0) s.n0 = 0;
      if (stepsPerLaunch > 1) s.n0 = (s.n0 + stepsPerLaunch - 1) / stepsPerLaunch;
    }
    return s;
  }
  // size of the next group; the caller enqueues that many launches and then calls issued_group(n)
  int next() {
    if (uniform) return checkEvery;
    int64_t n;
    if (n0 - issued >= 2) n = std::min<int64_t>(n0 - issued, 4096);
    else { n = tail < 2 ? 2 : tail == 2 ? 4 : 8; ++tail; }
    if (maxLaunches > 0 && issued + n > maxLaunches) n = std::max<int64_t>(1, maxLaunches - issued);
    return (int)n;
  }
  void issued_group(int n) { issued += n; }
};
// The driver's loop around the schedule: the host always has the NEXT group enqueued before it waits for the answer of the current one, so the device
// never idles on the host.  issue(n, half) enqueues n launches whose last one reports "work left" into flag block `half` and returns 0 or an error code (< 0);
// wait(half) blocks until that group has run and returns 1 (someone is still integrating), 0 (nobody) or an error code (< 0).
template <class Issue, class Wait>
int adv_poll_loop(AdvPollSchedule& s, Issue&& issue, Wait&& wait, int64_t* launches_out) {
  int64_t g = 0;
  int n = s.next();
  int rc = issue(n, 0);
  if (rc) return rc;
  s.issued_group(n);
  for (;;) {
    const bool more = !(s.maxLaunches > 0 && s.issued >= s.maxLaunches);
    if (more) {  // keep the device busy while the host waits for group g's answer
      n = s.next();
      rc = issue(n, (int)((g + 1) & 1));
      if (rc) return rc;
      s.issued_group(n);
    }
    const int any = wait((int)(g & 1));
    if (any < 0) return any;
    if (!any || !more) {
      if (more) {  // the speculative group (it found nothing left to do)
        const int w = wait((int)((g + 1) & 1));
        if (w < 0) return w;
      }
      break;
    }
    ++g;
  }
  if (launches_out) *launches_out = s.issued;
  return 0;
}
}  // namespace nnhip

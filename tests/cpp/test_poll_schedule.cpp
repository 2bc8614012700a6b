// test_poll_schedule.cpp — the adaptive streaming driver's polling loop (numericalnim_amd/csrc/adv_poll_schedule.hpp, the code the library runs)
// against a simulated batch whose slowest IVP needs `need` loop iterations: prints one line per case, "launches <n>".
//   test_poll_schedule <uniform 0|1> <check_every> <t0> <tEnd> <dtMax> <steps_per_launch> <max_launches> <need>
#include <cstdio>
#include <cstdlib>

#include "../../numericalnim_amd/csrc/adv_poll_schedule.hpp"

int main(int argc, char** argv) {
  if (argc < 9) return 64;
  const bool uniform = std::atoi(argv[1]) != 0;
  const int checkEvery = std::atoi(argv[2]);
  const double t0 = std::atof(argv[3]), tEnd = std::atof(argv[4]), dtMax = std::atof(argv[5]);
  const int K = std::atoi(argv[6]);
  const long long maxLaunches = std::atoll(argv[7]), need = std::atoll(argv[8]);   // launches after which nobody is left (iterations / K, rounded up by the caller)
  nnhip::AdvPollSchedule s = nnhip::AdvPollSchedule::make(uniform, checkEvery, t0, tEnd, dtMax, K, maxLaunches);
  long long done = 0, last[2] = {0, 0}, groups = 0, unpolledBeforeNeed = 0;
  auto issue = [&](int n, int half) -> int {
    if (n < 1) return -1;
    done += n; last[half] = done; ++groups;
    return 0;
  };
  auto wait = [&](int half) -> int { return last[half] < need ? 1 : 0; };   // the flag of a group = work left after its LAST launch
  int64_t launches = 0;
  const int rc = nnhip::adv_poll_loop(s, issue, wait, &launches);
  (void)unpolledBeforeNeed;
  std::printf("rc %d launches %lld groups %lld n0 %lld\n", rc, (long long)launches, groups, (long long)s.n0);
  return rc ? 1 : 0;
}

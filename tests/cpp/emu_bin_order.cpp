// emu_bin_order.cpp — TEST INFRASTRUCTURE: the two kernels of the order of integration (numericalnim_amd/csrc/sort_kernels.hpp: bin_count_kernel,
// bin_place_kernel — 1024 threads per workgroup, LDS histograms, a wavefront-shuffle scan) run on the host (tests/cpp/hip_cpu_emu.hpp) over keys read from
// stdin (one hex float per line; "nan" / "inf" accepted), with the key range computed here the way key_range_kernel leaves it (images of the smallest and
// largest FINITE key).  Prints the order, one index per line.
//   g++ -std=c++20 -O1 -DNNHIP_CPU_EMU -I tests/cpp -I numericalnim_amd/csrc -pthread tests/cpp/emu_bin_order.cpp
#include "sort_kernels.hpp"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace nnhip::sortk;

int main() {
  std::vector<double> keys;
  char buf[128];
  while (std::fgets(buf, sizeof buf, stdin)) keys.push_back(std::strtod(buf, nullptr));
  const int64_t n = (int64_t)keys.size();
  unsigned long long range[2] = {~0ULL, 0ULL};
  for (double v : keys)
    if (v == v && std::fabs(v) != INFINITY) {
      const unsigned long long o = ordered_img(v);
      if (o < range[0]) range[0] = o;
      if (o > range[1]) range[1] = o;
    }
  std::vector<uint16_t> bins(n);
  std::vector<uint32_t> hist(kBins, 0), cursor(kBins, 0), perm(n, 0xffffffffu);
  const unsigned blocks = (unsigned)((n + kBinThreads * kBinItems - 1) / (kBinThreads * kBinItems));
  hipemu::launch(bin_count_kernel, dim3(blocks), dim3(kBinThreads), (const double*)keys.data(), (const unsigned long long*)range, bins.data(), hist.data(), n);
  hipemu::launch(bin_place_kernel, dim3(blocks), dim3(kBinThreads), (const uint16_t*)bins.data(), (const uint32_t*)hist.data(), cursor.data(), perm.data(), n,
                 (const unsigned long long*)range, 0.0);
  for (uint32_t p : perm) std::printf("%u\n", p);
  return 0;
}

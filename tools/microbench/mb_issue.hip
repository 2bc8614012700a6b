// mb_issue.hip — how fast can ONE wave issue 1 KiB vector-memory instructions on gfx950?  One issuing wave per workgroup, G workgroups;
// each issues NI instructions back to back (addresses stream through a big buffer), s_memtime around the issue and around the drain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ unsigned long long g_t[4];
template <int MODE, int NI>
__global__ __launch_bounds__(256) void issue_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t strideElems, int reps) {
  __shared__ __attribute__((aligned(16))) char lds[NI * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave != 0) return;
  const unsigned ldsBase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
  unsigned long long tIssue = 0, tAll = 0;
  for (int r = 0; r < reps; ++r) {
    const int64_t base = ((int64_t)blockIdx.x * reps + r) * strideElems;  // NI KiB per (block, rep)
    v2d v[NI];
    if (MODE == 2) { for (int k = 0; k < NI; ++k) v[k] = v2d{(double)k, (double)lane}; }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < NI; ++k) {
      const double* p = src + base + k * 128 + lane * 2;
      double* q = dst + base + k * 128 + lane * 2;
      if (MODE == 0) {  // LDS-DMA
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(ldsBase + k * 1024));
      } else if (MODE == 1) {  // register load
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v[k]) : "v"(p) : "memory");
      } else {  // store
        asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(q), "v"(v[k]));
      }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_readcyclecounter();
    if (MODE == 1) { double acc = 0; for (int k = 0; k < NI; ++k) acc += v[k].x; if (acc == 12345.678) dst[0] = acc; }
    tIssue += t1 - t0; tAll += t2 - t0;
  }
  if (lane == 0) { atomicAdd(&g_t[0], tIssue); atomicAdd(&g_t[1], tAll); atomicAdd(&g_t[2], 1ull); }
}
template <int MODE, int NI>
static void run(const char* name, int grid, const double* src, double* dst, int reps) {
  unsigned long long z[4] = {0, 0, 0, 0}, h[4];
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_t), z, sizeof(z)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  issue_kernel<MODE, NI><<<grid, 256>>>(src, dst, (int64_t)NI * 128, reps);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_t), sizeof(h)));
  const double n = (double)h[2] * reps * NI;
  printf("%-28s grid %5d x %2d instr x %3d reps: issue %7.1f cycles/instr, issue+drain %7.1f cycles/instr per wave; kernel %.1f us -> %.0f GB/s\n", name, grid, NI, reps,
         h[0] / n, h[1] / n, ms * 1e3, (double)grid * reps * NI * 1024 / (ms * 1e-3) * 1e-9);
}
int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  const size_t bytes = (size_t)3 << 30;
  double *src, *dst; CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes));
  CK(hipMemset(src, 0, bytes)); CK(hipMemset(dst, 0, bytes));
  for (int grid : {256, 768, 1024, 2048}) {
    const int reps = 32;
    if (only < 0 || only == 0) run<0, 16>("LDS-DMA load 16x1KiB", grid, src, dst, reps);
    if (only < 0 || only == 1) run<1, 16>("register load 16x1KiB", grid, src, dst, reps);
    if (only < 0 || only == 2) run<2, 16>("store 16x1KiB", grid, src, dst, reps);
    if (only < 0 || only == 0) run<0, 32>("LDS-DMA load 32x1KiB", grid, src, dst, reps);
    if (only < 0 || only == 2) run<2, 32>("store 32x1KiB", grid, src, dst, reps);
  }
  return 0;
}

// mb_module.cpp — does a kernel loaded at run time (hiprtc + hipModuleLoadData) run as fast as the same kernel linked into the binary?
// Two kernels: a tight FP64 loop (fits any instruction cache) and a long unrolled one (~24 KB of straight-line FP64 code per iteration).
//   build: hipcc --offload-arch=gfx950 -O3 -o mb_module mb_module.cpp -lhiprtc      run: ./mb_module
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <cstdio>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

#define BODY_TIGHT "double a = x[i], b = 1.0000001, c = 0.5; for (int k = 0; k < iters * 256; ++k) { a = a * b + c; c = c * b + a; } x[i] = a + c;"
#define REP4(s) s s s s
#define REP16(s) REP4(REP4(s))
#define REP256(s) REP16(REP16(s))
#define STEP "a = a * b + c; c = c * b + a; b = b * 0.99999 + 1e-9;"
#define BODY_LONG "double a = x[i], b = 1.0000001, c = 0.5; for (int k = 0; k < iters; ++k) { " REP256(STEP) " } x[i] = a + c;"

extern "C" __global__ void tight_static(double* x, int iters) { const int i = blockIdx.x * blockDim.x + threadIdx.x; double a = x[i], b = 1.0000001, c = 0.5; for (int k = 0; k < iters * 256; ++k) { a = a * b + c; c = c * b + a; } x[i] = a + c; }
extern "C" __global__ void long_static(double* x, int iters) { const int i = blockIdx.x * blockDim.x + threadIdx.x; double a = x[i], b = 1.0000001, c = 0.5; for (int k = 0; k < iters; ++k) {
#define S a = a * b + c; c = c * b + a; b = b * 0.99999 + 1e-9;
#define S4 S S S S
#define S16 S4 S4 S4 S4
#define S256 S16 S16 S16 S16 S16 S16 S16 S16 S16 S16 S16 S16 S16 S16 S16 S16
  S256 } x[i] = a + c; }

#define BODY_LDS "__shared__ double sh[2112]; const int tix = threadIdx.x; double a = x[i]; sh[tix] = a; for (int k = 0; k < iters * 64; ++k) { sh[tix + 256 * (k & 7)] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, \"wavefront\"); __builtin_amdgcn_wave_barrier(); a = a * 1.0000001 + sh[(tix ^ 1) + 256 * (k & 7)]; __builtin_amdgcn_wave_barrier(); } x[i] = a;"
#define BODY_TAB "double a = x[i]; for (int k = 0; k < iters * 64; ++k) { a = a * 1.0000001 + tab[(k + (int)a) & 255]; } x[i] = a;"
__device__ const double tab_static[256] = {1.0, 2.0, 3.0};
extern "C" __global__ void lds_static(double* x, int iters) { const int i = blockIdx.x * blockDim.x + threadIdx.x; __shared__ double sh[2112]; const int tix = threadIdx.x; double a = x[i]; sh[tix] = a; for (int k = 0; k < iters * 64; ++k) { sh[tix + 256 * (k & 7)] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); a = a * 1.0000001 + sh[(tix ^ 1) + 256 * (k & 7)]; __builtin_amdgcn_wave_barrier(); } x[i] = a; }
extern "C" __global__ void tab_static_k(double* x, int iters) { const int i = blockIdx.x * blockDim.x + threadIdx.x; const double* tab = tab_static; double a = x[i]; for (int k = 0; k < iters * 64; ++k) { a = a * 1.0000001 + tab[(k + (int)a) & 255]; } x[i] = a; }

int main() {
  const int n = 1 << 22, iters = 40;
  double* x; CK(hipMalloc(&x, sizeof(double) * n)); CK(hipMemset(x, 0, sizeof(double) * n));
  const std::string src = std::string("extern \"C\" __global__ void tight_rtc(double* x, int iters) { const int i = blockIdx.x * blockDim.x + threadIdx.x; ") + BODY_TIGHT + " }\n" +
                          "extern \"C\" __global__ void long_rtc(double* x, int iters) { const int i = blockIdx.x * blockDim.x + threadIdx.x; " + BODY_LONG + " }\n" +
                          "extern \"C\" __global__ void lds_rtc(double* x, int iters) { const int i = blockIdx.x * blockDim.x + threadIdx.x; " + BODY_LDS + " }\n" +
                          "__device__ const double tab_rtc[256] = {1.0, 2.0, 3.0};\n"
                          "extern \"C\" __global__ void tab_rtc_k(double* x, int iters) { const int i = blockIdx.x * blockDim.x + threadIdx.x; const double* tab = tab_rtc; " + BODY_TAB + " }\n";
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "m.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return 2;
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off"};
  if (hiprtcCompileProgram(prog, 3, opts) != HIPRTC_SUCCESS) { size_t l; hiprtcGetProgramLogSize(prog, &l); std::string log(l, 0); hiprtcGetProgramLog(prog, &log[0]); printf("%s\n", log.c_str()); return 3; }
  size_t sz; hiprtcGetCodeSize(prog, &sz); std::vector<char> code(sz); hiprtcGetCode(prog, code.data());
  hipModule_t mod; CK(hipModuleLoadData(&mod, code.data()));
  hipFunction_t ft, fl, fs, fb; CK(hipModuleGetFunction(&ft, mod, "tight_rtc")); CK(hipModuleGetFunction(&fl, mod, "long_rtc"));
  CK(hipModuleGetFunction(&fs, mod, "lds_rtc")); CK(hipModuleGetFunction(&fb, mod, "tab_rtc_k"));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, auto&& launch) {
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) { hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("%-14s %8.3f ms\n", name, best);
  };
  int it = iters; void* args[] = {&x, &it};
  timeit("tight static", [&] { hipLaunchKernelGGL(tight_static, dim3(n / 256), dim3(256), 0, 0, x, iters); });
  timeit("tight module", [&] { hipModuleLaunchKernel(ft, n / 256, 1, 1, 256, 1, 1, 0, 0, args, nullptr); });
  timeit("long static", [&] { hipLaunchKernelGGL(long_static, dim3(n / 256), dim3(256), 0, 0, x, iters); });
  timeit("long module", [&] { hipModuleLaunchKernel(fl, n / 256, 1, 1, 256, 1, 1, 0, 0, args, nullptr); });
  timeit("lds static", [&] { hipLaunchKernelGGL(lds_static, dim3(n / 256), dim3(256), 0, 0, x, iters); });
  timeit("lds module", [&] { hipModuleLaunchKernel(fs, n / 256, 1, 1, 256, 1, 1, 0, 0, args, nullptr); });
  timeit("table static", [&] { hipLaunchKernelGGL(tab_static_k, dim3(n / 256), dim3(256), 0, 0, x, iters); });
  timeit("table module", [&] { hipModuleLaunchKernel(fb, n / 256, 1, 1, 256, 1, 1, 0, 0, args, nullptr); });
  hipDeviceptr_t dptr; size_t bytes;  // where does the module's code live?  (kernel object addresses are printed by AMD_LOG_LEVEL=4)
  (void)dptr; (void)bytes;
  return 0;
}

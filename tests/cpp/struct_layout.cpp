// struct_layout.cpp — TEST INFRASTRUCTURE: the byte layout of the kernel argument structs (field offsets as the compiler lays them out — the same on host and device: plain
// aggregates of 8-byte scalars and pointers), printed as JSON for tools/gfx950_isa_interp.py's callers, which build kernarg buffers for the interpreted code object.
#include "ode_kernels.hpp"

#include <cstddef>
#include <cstdio>
using namespace nnhip;
#define F(S, m) std::printf("%s\"%s\": %zu", first ? "" : ", ", #m, offsetof(S, m)), first = false
int main() {
  bool first = true;
  std::printf("{\"AdvLeanArgs\": {");
  F(AdvLeanArgs, y); F(AdvLeanArgs, td); F(AdvLeanArgs, N); F(AdvLeanArgs, tEnd); F(AdvLeanArgs, ctl); F(AdvLeanArgs, P); F(AdvLeanArgs, active);
  std::printf(", \"sizeof\": %zu}, ", sizeof(AdvLeanArgs));
  first = true;
  std::printf("\"StepArgs\": {");
  F(StepArgs, N); F(StepArgs, ivpStride); F(StepArgs, compStride); F(StepArgs, t_dev); F(StepArgs, t_uniform); F(StepArgs, dt_dev); F(StepArgs, dt_uniform);
  F(StepArgs, y_in); F(StepArgs, fsal_in); F(StepArgs, y_out); F(StepArgs, fsal_out); F(StepArgs, dt_used); F(StepArgs, error); F(StepArgs, ctl); F(StepArgs, P);
  F(StepArgs, tEnd); F(StepArgs, t_io); F(StepArgs, dt_io); F(StepArgs, active); F(StepArgs, steps_io); F(StepArgs, perIvpParams); F(StepArgs, nPerIvp);
  F(StepArgs, perIvpStride); F(StepArgs, stepsPerLaunch); F(StepArgs, nontemporal); F(StepArgs, recomputeFsal); F(StepArgs, noLean); F(StepArgs, tReq);
  F(StepArgs, nReq); F(StepArgs, useDense); F(StepArgs, negate); F(StepArgs, emitAfter); F(StepArgs, denseIdx_io); F(StepArgs, rows); F(StepArgs, rowStride);
  F(StepArgs, rowBase); F(StepArgs, rowBase0);
  std::printf(", \"sizeof\": %zu}, ", sizeof(StepArgs));
  first = true;
  std::printf("\"Params\": {");
  F(Params, p); F(Params, shared); F(Params, ivp); F(Params, aux); F(Params, stride);
  std::printf(", \"sizeof\": %zu}, ", sizeof(Params));
  first = true;
  std::printf("\"StepCtl\": {");
  F(StepCtl, absTol); F(StepCtl, relTol); F(StepCtl, dtMax); F(StepCtl, dtMin);
  std::printf(", \"sizeof\": %zu}, \"kBlock\": %d, \"kAggSlots\": %d, \"kMaxParams\": %d}\n", sizeof(StepCtl), kBlock, kAggSlots, kMaxParams);
  return 0;
}

// adv_experiments.hpp — round-3 EXPERIMENTAL variants of the streamed adaptive kernels (advance_lps_kernel / advance_tpi_kernel of
// numericalnim_amd/csrc/ode_kernels.hpp).  NOT part of the product library: instantiated only by tools/microbench/mb_adv.hip, which
// times them against the production kernels and demands bitwise equality.  What each one tests, what was measured and why none of them
// replaced the production kernel is written up in profiles/r03_c4_stream_experiments.md.  All of them reuse the production arithmetic
// (adv_compute / adv_commit / embedded_step): same expressions, same bits.
#pragma once
#include "../../numericalnim_amd/csrc/ode_kernels.hpp"

namespace NNHIP_NS {

// ---- round 3: persistent, store-deferred pipeline ---------------------------------------------------------------------------------
// What held the two-tile pipeline above back (read off its ISA): gfx9 has ONE in-order counter (vmcnt) for vector loads AND stores,
// and the write-back of tile g sits behind exec-dependent branches, so the compiler cannot count the stores in flight and waits for
// "everything" (s_waitcnt vmcnt(0)) where tile g + 1's prefetched state is first used — i.e. right after tile g's stores were issued:
// every tile after the first paid one full store round trip, exposed (which is also why 4 and 8 tiles per block were no better than 2).
// Here a workgroup walks tiles b, b + G, b + 2G, ... (G = resident workgroups) with the order of one iteration chosen so that the
// ONE conservative wait only ever covers old requests:
//     wait for tile g's state  ->  issue tile g-1's stores  ->  issue tile g+1's loads  ->  advance tile g (registers only)
// When the wait executes, the requests in flight are tile g's loads and tile g-2's stores, both issued a whole compute phase
// (>= 1300 VALU cycles) earlier; the new stores and loads then travel under tile g's arithmetic.  Same expressions, same bits.
// -DNNHIP_ADV_TIMING (tools/microbench only): per-wave s_memtime stamps around the phases of one iteration, summed into g_adv_timing
// [0] wait  [1] issue of stores + loads  [2] arithmetic  [3] tiles  [4] whole kernel per wave  [5] waves
#ifdef NNHIP_ADV_TIMING
__device__ unsigned long long g_adv_timing[16];
#define NNHIP_ADV_TIMING_DECL unsigned long long tm_[4] = {0, 0, 0, 0}, tacc_[3] = {0, 0, 0}, ttiles_ = 0; const unsigned long long tstart_ = __builtin_readcyclecounter();
#define NNHIP_ADV_TIMING_MARK(k) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tm_[k] = __builtin_readcyclecounter(); asm volatile("" ::: "memory"); if (k == 3) { tacc_[0] += tm_[1] - tm_[0]; tacc_[1] += tm_[2] - tm_[1]; tacc_[2] += tm_[3] - tm_[2]; ttiles_ += 1; } }
#define NNHIP_ADV_TIMING_FLUSH if ((threadIdx.x & 63) == 0) { atomicAdd(&g_adv_timing[0], tacc_[0]); atomicAdd(&g_adv_timing[1], tacc_[1]); atomicAdd(&g_adv_timing[2], tacc_[2]); atomicAdd(&g_adv_timing[3], ttiles_); atomicAdd(&g_adv_timing[4], (unsigned long long)__builtin_readcyclecounter() - tstart_); atomicAdd(&g_adv_timing[5], 1ull); }
#define NNHIP_MC_T0 unsigned long long mt_[4] = {0, 0, 0, 0}, mprev_ = __builtin_readcyclecounter(); const unsigned long long mc0_ = mprev_, mr0_ = __builtin_amdgcn_s_memrealtime();
#define NNHIP_MC_T(k) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_readcyclecounter(); mt_[k] += now_ - mprev_; mprev_ = now_; }
#define NNHIP_MC_TFLUSH(base, n) if ((threadIdx.x & 63) == 0) { for (int q_ = 0; q_ < 4; ++q_) atomicAdd(&g_adv_timing[base + q_], mt_[q_]); atomicAdd(&g_adv_timing[n], 1ull); \
    if (n == 8) { atomicAdd(&g_adv_timing[10], (unsigned long long)__builtin_readcyclecounter() - mc0_); atomicAdd(&g_adv_timing[11], (unsigned long long)__builtin_amdgcn_s_memrealtime() - mr0_); } }
#else
#define NNHIP_ADV_TIMING_DECL
#define NNHIP_ADV_TIMING_MARK(k)
#define NNHIP_ADV_TIMING_FLUSH
#define NNHIP_MC_T0
#define NNHIP_MC_T(k)
#define NNHIP_MC_TFLUSH(base, n)
#endif
template <int D>
NNHIP_DEV void adv_touch(AdvState<D>& s) {  // first use of a prefetched state: this is where the compiler places the wait
#pragma unroll
  for (int c = 0; c < D; ++c) asm volatile("" : "+v"(s.y[c]), "+v"(s.fsal[c]));
  asm volatile("" : "+v"(s.t), "+v"(s.dt));
}
// PINGPONG: the two halves of a workgroup alternate between the memory part and the arithmetic part of an iteration, in opposite
// phase (the second half starts one barrier late).  Why: a wave's 15 vector-memory instructions are issued in order and the issue
// stalls on the CU's memory pipeline (s_memtime stamps: ~6000 cycles to issue them when all 16 waves of a CU do so at once — the
// CU's share of HBM bandwidth, ~10 B per cycle, IS that back-pressure), the arbiter serves the waves of a SIMD round-robin, so all
// of them leave the issue phase together, wait together and then share the VALU together: memory and arithmetic take turns instead
// of overlapping (wait 4556 + issue 6147 + arithmetic 3408 cycles per tile).  With the halves in opposite phase one half's
// arithmetic always runs beside the other half's memory phase.
// PROBE (tools/microbench only; results are NOT valid): 1 = arithmetic only (the first tile's state is advanced once per tile, nothing else
// is loaded or stored), 2 = memory only (every tile is loaded and written back unchanged).
template <int METHOD, class RHS, int CPL = 1, bool PINGPONG = false, int PROBE = 0>
__global__ __launch_bounds__(kBlock) NNHIP_ADV_LPS_ATTR void advance_lps_persist_kernel(const StepArgs a) {
  static_assert(MethodTraits<METHOD>::adaptive, "fixed-step methods share (t, dt): use the uniform streaming loop");
  constexpr int DIM = RHS::dim;
  constexpr int LPSYS = DIM / CPL;
  static_assert(DIM % CPL == 0 && 64 % LPSYS == 0, "a system must not straddle wavefronts");
  constexpr int perTile = kBlock / LPSYS;
  __shared__ double lds[lps_lds_doubles<DIM, CPL>()];
  controller_prologue();
  pin_step_args(a);
  const int sysInBlock = threadIdx.x / LPSYS, c = (threadIdx.x % LPSYS) * CPL;
  double* ys = lds + sysInBlock * lps_stride<DIM>();
  double* es = lds + lps_lds_doubles<DIM, CPL>() / 2 + sysInBlock * lps_stride<DIM>();
  const Params P = a.P;  // batch-wide parameters only (the launcher sends per-IVP tables to advance_lps_kernel)
  const LpsOps<RHS, false, CPL> ops{P, ys, es, c};
  const int64_t nTiles = (a.N + perTile - 1) / perTile;
  auto prefetch = [&](int64_t tile, AdvState<CPL>& s) {
    const int64_t i = tile * perTile + sysInBlock;
    const int64_t ic = i < a.N ? i : a.N - 1;  // ragged last tile: a valid address, the lane is switched off below
    adv_load_state<false>(a, ops, ic * a.ivpStride + c * a.compStride, s.y, s.fsal);
    s.dt = a.dt_io[ic];
    s.t = a.t_io[ic];
  };
  unsigned int stillActive = 0;
  AdvState<CPL> cur, nxt;
  AdvResult<CPL> res;
  res.live = false;
  int64_t iRes = 0;
  int64_t tile = blockIdx.x;
  const bool lateHalf = PINGPONG && threadIdx.x >= kBlock / 2;  // wave-uniform
  if (tile < nTiles) prefetch(tile, cur);
  if (lateHalf) __builtin_amdgcn_s_barrier();
  NNHIP_ADV_TIMING_DECL
  for (; tile < nTiles; tile += gridDim.x) {
    NNHIP_ADV_TIMING_MARK(0)
    adv_touch(cur);                                                                          // the wait
    NNHIP_ADV_TIMING_MARK(1)
    if constexpr (PROBE != 1) adv_commit<METHOD, false>(a, ops, iRes, iRes * a.ivpStride + c * a.compStride, c == 0, res);     // tile g-1 goes out
    const int64_t nt = tile + gridDim.x;
    if constexpr (PROBE == 1) nxt = cur;
    else if (nt < nTiles) prefetch(nt, nxt);                                                 // tile g+1 comes in
    if constexpr (PINGPONG) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    NNHIP_ADV_TIMING_MARK(2)
    const int64_t i = tile * perTile + sysInBlock;
    cur.live = i < a.N && cur.t < a.tEnd;  // :511
    if constexpr (PROBE == 2) {
      res.live = cur.live; res.t = cur.t; res.dt = cur.dt; res.error = 0.0;
#pragma unroll
      for (int j = 0; j < CPL; ++j) { res.y[j] = cur.y[j]; res.fsal[j] = cur.fsal[j]; }
    } else {
      stillActive |= adv_compute<METHOD>(a, ops, cur, res);
    }
    iRes = i;
    cur = nxt;
    if constexpr (PROBE == 1) {  // keep integrating the first tile's systems (a fused solve in registers)
      cur.t = res.t; cur.dt = res.dt;
#pragma unroll
      for (int j = 0; j < CPL; ++j) { cur.y[j] = res.y[j]; cur.fsal[j] = res.fsal[j]; }
    }
    if constexpr (PINGPONG) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    NNHIP_ADV_TIMING_MARK(3)
  }
  if (PINGPONG && !lateHalf) __builtin_amdgcn_s_barrier();
  adv_commit<METHOD, false>(a, ops, iRes, iRes * a.ivpStride + c * a.compStride, c == 0, res);
  NNHIP_ADV_TIMING_FLUSH
  if (a.active) {
    if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
  }
}

// ---- round 3: mover / consumer wave specialisation ---------------------------------------------------------------------------------
// What the counters and s_memtime stamps of the symmetric kernels above say (tools/microbench, profiles/r03_c4_*): a CU's vector-memory
// pipeline accepts requests only as fast as the CU's share of HBM bandwidth (~10 B per cycle), so at saturation a wave's in-order
// vector-memory instructions BLOCK AT ISSUE for about as long as a whole round trip — and a wave blocked at issue does no arithmetic.
// Whatever the order of one iteration (loads early, stores deferred, two halves in opposite phase), every wave serialises
// "queue for the memory pipe" and "arithmetic", and the sum, not the maximum, of the two floors is paid (memory alone 82-88 us,
// arithmetic alone ~60-70 us, together 123-127 us at 1e6 x 16 whether 2 or 4 components per lane, 2 or 4 waves per SIMD).
// The way out is to give the queueing to a wave that has nothing else to do:
//   wave 0 of a workgroup (the MOVER) owns every global access: it streams the state of upcoming tiles into an LDS ring with LDS-DMA
//   (global_load_lds_dwordx4: no registers, no VALU), and writes finished tiles back (ds_read_b128 -> global_store_dwordx4);
//   waves 1..3 (CONSUMERS) run the loop iteration (adv_compute: the same expressions as everywhere else) from LDS to LDS and never
//   issue a vector-memory instruction (pow's table gathers aside), so they never queue.
// Per round r (= the tile the consumers work on) the mover issues  stores(r-1), t-loads(r+4), state-loads(r+2)  and then waits
// until state-loads(r+1) have landed (s_waitcnt vmcnt(N) with N = the instructions issued since: every vector-memory instruction of
// the mover is issued unconditionally, finished IVPs are switched off through EXEC, so N is a compile-time constant) and publishes
// tile r+1.  Three LDS slots per workgroup (r: in work, r+1: landing, r+2 = r-1: written back, then refilled).  The times `t` run two
// tiles ahead of the state in their own ring: state-loads are masked with `t < tEnd` so that finished IVPs move no state (a launch
// over a finished batch reads 8 B per IVP, as with the symmetric kernels).
// Layout requirements (the launcher checks them, anything else takes advance_lps_kernel): AoS (compStride 1, ivpStride DIM), every
// component slot real, batch-wide parameters, an error array, no step counters.
struct AdvMcLds {  // byte offsets inside the workgroup's LDS block; T = systems per tile, D = DIM
  template <int T, int D> static constexpr int slot_bytes() { return 2 * T * D * 8 + 2 * T * 8 + T * 4; }  // y, fsal, dt, err, adv
  template <int T, int D> static constexpr int y(int s) { return s * slot_bytes<T, D>(); }
  template <int T, int D> static constexpr int fsal(int s) { return y<T, D>(s) + T * D * 8; }
  template <int T, int D> static constexpr int dt(int s) { return fsal<T, D>(s) + T * D * 8; }
  template <int T, int D> static constexpr int err(int s) { return dt<T, D>(s) + T * 8; }
  template <int T, int D> static constexpr int adv(int s) { return err<T, D>(s) + T * 8; }
  static constexpr int kSlots = 3, kTRing = 8;
  template <int T, int D> static constexpr int tring(int k) { return kSlots * slot_bytes<T, D>() + k * T * 8; }
  template <int T, int D> static constexpr int flags() { return tring<T, D>(kTRing); }  // pub[3], done[3 consumers][3 slots] (uint32)
  template <int T, int D> static constexpr int total() { return flags<T, D>() + 64; }
};
typedef double nnhip_v2d __attribute__((ext_vector_type(2)));
// The mover's vector-memory instructions.  Address = wave-uniform base (SGPR pair) + per-lane byte offset (one VGPR that never
// changes): no per-instruction address arithmetic — the mover's scalar and vector ALU instructions compete for issue slots with the
// consumers' arithmetic, and every one it does not need is latency it does not pay.  Lanes are switched off through EXEC (mask = a
// compare result, i.e. already a scalar pair), so switched-off lanes need no valid address and every instruction is issued.
// one exec-masked 16-byte-per-lane LDS-DMA: lane l's 16 bytes land at ldsBase + 16*l (ldsBase wave-uniform)
// (the SGPR-base + 32-bit-offset form of these instructions faulted on the MI355X boxes used here; one 64-bit add per instruction instead)
NNHIP_DEV void mc_dma16(const void* sbase, unsigned voff, unsigned ldsBase, unsigned long long mask) {
  unsigned keep;
  unsigned long long ex;
  const char* p = reinterpret_cast<const char*>(sbase) + voff;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %4\n\ts_and_b64 exec, exec, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %2, off\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
      : "=&s"(keep), "=&s"(ex) : "v"(p), "s"(mask), "s"(ldsBase));
}
NNHIP_DEV void mc_st16(double* sbase, unsigned voff, nnhip_v2d v, unsigned long long mask) {
  unsigned long long ex;
  char* p = reinterpret_cast<char*>(sbase) + voff;
  asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_store_dwordx4 %1, %2, off\n\ts_nop 1\n\ts_mov_b64 exec, %0"
               : "=&s"(ex) : "v"(p), "v"(v), "s"(mask));
}
NNHIP_DEV void mc_st8(double* sbase, unsigned voff, double v, unsigned long long mask) {
  unsigned long long ex;
  char* p = reinterpret_cast<char*>(sbase) + voff;
  asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_store_dwordx2 %1, %2, off\n\ts_nop 0\n\ts_mov_b64 exec, %0"
               : "=&s"(ex) : "v"(p), "v"(v), "s"(mask));
}
NNHIP_DEV void mc_fence() { asm volatile("" ::: "memory"); }  // compiler-level: LDS accesses and the asm vector-memory helpers stay on their side
NNHIP_DEV unsigned mc_lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; }
NNHIP_DEV void mc_spin(volatile unsigned* flag, unsigned want) {
  while (*flag != want) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

#ifndef NNHIP_ADV_MC_WPE
#define NNHIP_ADV_MC_WPE 3
#endif
// PROBE (tools/microbench only, results invalid): 2 = the consumers hand every tile straight back (memory path alone); PRIO: s_setprio of the mover
// SUB: mover/consumer groups per workgroup.  A workgroup's waves are placed on the CU's four SIMDs cyclically, so with 4-wave groups
// the movers of all groups (waves 0, 4, 8) share one SIMD and no consumer competes with a mover for issue slots: measured, a mover
// beside arithmetic-heavy waves streams ~25 % slower (a consumer that sleeps costs nothing, one that issues VALU does).
template <int METHOD, class RHS, int CPL, int PROBE = 0, int PRIO = 0, int SUB = 1>
__global__ __launch_bounds__(kBlock * SUB) __attribute__((amdgpu_waves_per_eu(NNHIP_ADV_MC_WPE, NNHIP_ADV_MC_WPE))) void advance_lps_mc_kernel(const StepArgs a) {
  static_assert(MethodTraits<METHOD>::adaptive, "fixed-step methods share (t, dt): use the uniform streaming loop");
  constexpr int DIM = RHS::dim, LPSYS = DIM / CPL, SW = 64 / LPSYS, NCW = 3, T = NCW * SW;
  static_assert(DIM % CPL == 0 && 64 % LPSYS == 0 && RhsSize<RHS>::value == DIM && DIM % 2 == 0, "layout");
  using L = AdvMcLds;
  constexpr int YB = T * DIM * 8;                       // bytes of a tile's y (and FSAL) block
  constexpr int KY = (YB + 1023) / 1024;                // 1 KiB DMA / store instructions per block
  constexpr int KL = 2 * KY + 1, KS = 2 * KY + 3;       // vector-memory instructions of state-loads(r) / stores(r)
  static_assert(KS + 1 + KL <= 60, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(16))) char ldsAll[SUB * L::total<T, DIM>()];
  controller_prologue();
  const int waveAll = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
  const int sub = waveAll >> 2, wave = waveAll & 3;
  char* lds = ldsAll + sub * L::total<T, DIM>();
  const int64_t vblock = (int64_t)blockIdx.x * SUB + sub, vgrid = (int64_t)gridDim.x * SUB;  // the group's index among all groups
  const int64_t nTiles = (a.N + T - 1) / T;
  const int64_t nj = vblock < nTiles ? (nTiles - vblock + vgrid - 1) / vgrid : 0;  // this group's tiles: vblock + j * vgrid
  volatile unsigned* flags = reinterpret_cast<volatile unsigned*>(lds + L::flags<T, DIM>());
  if ((threadIdx.x & 255) < 16) flags[threadIdx.x & 255] = 0u;
  __syncthreads();
  unsigned int stillActive = 0;
  const unsigned ldsBase = __builtin_amdgcn_readfirstlane(mc_lds_addr(lds));
  if (wave == 0) {
    // ------------------------------------------------ mover ------------------------------------------------
    if constexpr (PRIO > 0) __builtin_amdgcn_s_setprio(PRIO);
    auto tileSys0 = [&](int64_t j) { return (vblock + j * vgrid) * T; };
    // per-lane constants: byte offsets inside a tile's blocks, and which system a lane's chunk of instruction k belongs to
    const unsigned off16 = (unsigned)lane * 16u, off8 = (unsigned)lane * 8u;
    const int sysOfLane8 = lane / (DIM / 2);  // system of chunk k*64 + lane  =  k * (128 / DIM) + lane / (DIM / 2)
    static_assert(64 % (DIM / 2) == 0, "a 1 KiB instruction covers whole systems");
    constexpr int SPI = 64 / (DIM / 2);       // systems per 1 KiB instruction
    // All LDS reads of a phase come first, then its vector-memory instructions back to back.
    auto nValid = [&](int64_t j) -> int {  // systems of tile j that exist (0 for a tile this workgroup does not have)
      if (j < 0 || j >= nj) return 0;
      const int64_t left = a.N - tileSys0(j);
      return left >= T ? T : (int)left;
    };
    auto issueT = [&](int64_t j) {  // t of tile j -> ring entry j & 7 (16 B per lane: two systems; N is even)
      const int nv = nValid(j);
      const int64_t s0 = nv ? tileSys0(j) : 0;
      mc_dma16(a.t_io + s0, off16, ldsBase + L::tring<T, DIM>((int)(j & 7)), __builtin_amdgcn_ballot_w64(2 * lane < nv));
    };
    auto issueL = [&](int64_t j) {  // y, FSAL (masked per system with the consumers' predicate: exists and t < tEnd), dt of tile j -> slot j % 3
      const int slot = (int)(((j % 3) + 3) % 3);
      const int nv = nValid(j);
      const int64_t s0 = nv ? tileSys0(j) : 0;
      const char* ring = lds + L::tring<T, DIM>((int)(j & 7));
      double tv[KY];
#pragma unroll
      for (int k = 0; k < KY; ++k) {
        const int sys = k * SPI + sysOfLane8;
        tv[k] = *reinterpret_cast<const double*>(ring + (sys < T ? sys : 0) * 8);
      }
      const nnhip_v2d t2 = *reinterpret_cast<const nnhip_v2d*>(ring + (2 * lane < T ? lane : 0) * 16);
      mc_fence();
      const double* yb = a.y_in + s0 * DIM;
      const double* fb = a.fsal_in + s0 * DIM;
#pragma unroll
      for (int k = 0; k < KY; ++k) {
        const unsigned long long live = __builtin_amdgcn_ballot_w64(k * SPI + sysOfLane8 < nv && tv[k] < a.tEnd);
        mc_dma16(yb + k * 128, off16, ldsBase + L::y<T, DIM>(slot) + k * 1024, live);
        mc_dma16(fb + k * 128, off16, ldsBase + L::fsal<T, DIM>(slot) + k * 1024, live);
      }
      mc_dma16(a.dt_io + s0, off16, ldsBase + L::dt<T, DIM>(slot),
               __builtin_amdgcn_ballot_w64((2 * lane < nv && t2.x < a.tEnd) || (2 * lane + 1 < nv && t2.y < a.tEnd)));
      mc_fence();
    };
    auto issueS = [&](int64_t j) {  // results of tile j (slot j % 3, t from the ring) -> global, masked by the consumers' adv flags
      const int slot = (int)(((j % 3) + 3) % 3);
      const int nv = nValid(j);
      const int64_t s0 = nv ? tileSys0(j) : 0;
      nnhip_v2d vy[KY], vf[KY];
      unsigned av[KY];
#pragma unroll
      for (int k = 0; k < KY; ++k) {
        const int sys = k * SPI + sysOfLane8;
        const int chunk = k * 64 + lane;
        av[k] = *reinterpret_cast<const unsigned*>(lds + L::adv<T, DIM>(slot) + (sys < T ? sys : 0) * 4);
        vy[k] = *reinterpret_cast<const nnhip_v2d*>(lds + L::y<T, DIM>(slot) + (chunk < T * DIM / 2 ? chunk : 0) * 16);
        vf[k] = *reinterpret_cast<const nnhip_v2d*>(lds + L::fsal<T, DIM>(slot) + (chunk < T * DIM / 2 ? chunk : 0) * 16);
      }
      const int sysl = lane < T ? lane : 0;
      const unsigned al = *reinterpret_cast<const unsigned*>(lds + L::adv<T, DIM>(slot) + sysl * 4);
      const double tv = *reinterpret_cast<const double*>(lds + L::tring<T, DIM>((int)(j & 7)) + sysl * 8);
      const double dv = *reinterpret_cast<const double*>(lds + L::dt<T, DIM>(slot) + sysl * 8);
      const double ev = *reinterpret_cast<const double*>(lds + L::err<T, DIM>(slot) + sysl * 8);
      mc_fence();
      double* yb = a.y_out + s0 * DIM;
      double* fb = a.fsal_out + s0 * DIM;
#pragma unroll
      for (int k = 0; k < KY; ++k) {
        const unsigned long long adv = __builtin_amdgcn_ballot_w64(k * SPI + sysOfLane8 < nv && av[k] != 0u);
        mc_st16(yb + k * 128, off16, vy[k], adv);
        mc_st16(fb + k * 128, off16, vf[k], adv);
      }
      const unsigned long long adv = __builtin_amdgcn_ballot_w64(lane < nv && al != 0u);
      mc_st8(a.t_io + s0, off8, tv, adv);
      mc_st8(a.dt_io + s0, off8, dv, adv);
      mc_st8(a.error + s0, off8, ev, adv);
      mc_fence();
    };
    static_assert(T <= 64, "one lane per system for the scalar write-back");
    // prologue: t of tiles 0..3, then the state of tiles 0 and 1; tile 0 is published once it has landed
    issueT(0); issueT(1); issueT(2); issueT(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issueL(0); issueL(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KL) : "memory");
    flags[0] = 1u;  // pub[slot 0] = tile 0 (+1)
    NNHIP_MC_T0
    for (int64_t r = 0; r <= nj; ++r) {
      if (r > 0 && r - 1 < nj) {  // consumers done with tile r-1?
        const int slotp = (int)((r - 1) % 3);
        for (;;) {  // all consumers' flags in one round trip
          unsigned f[NCW];
#pragma unroll
          for (int w = 0; w < NCW; ++w) f[w] = flags[3 + w * 3 + slotp];
          bool all = true;
#pragma unroll
          for (int w = 0; w < NCW; ++w) all = all && f[w] == (unsigned)r;
          if (all) break;
          __builtin_amdgcn_s_sleep(1);
        }
        mc_fence();
      }
      NNHIP_MC_T(0)
      issueS(r - 1);
      NNHIP_MC_T(1)
      issueT(r + 4);
      issueL(r + 2);
      NNHIP_MC_T(2)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KS + 1 + KL) : "memory");  // state-loads(r+1) and everything older have landed
      if (r + 1 < nj) flags[(int)((r + 1) % 3)] = (unsigned)(r + 2);
      mc_fence();
      NNHIP_MC_T(3)
    }
    NNHIP_MC_TFLUSH(0, 8)
  } else {
    // ----------------------------------------------- consumers -----------------------------------------------
    const int cw = wave - 1;
    const int sysInWave = lane / LPSYS, c = (lane % LPSYS) * CPL;
    const int sysInTile = cw * SW + sysInWave;
    const Params P = a.P;
    NNHIP_MC_T0
    for (int64_t r = 0; r < nj; ++r) {
      const int slot = (int)(r % 3);
      NNHIP_MC_T(2)
      mc_spin(&flags[slot], (unsigned)(r + 1));
      NNHIP_MC_T(0)
      double* ys = reinterpret_cast<double*>(lds + L::y<T, DIM>(slot)) + sysInTile * DIM;     // this system's y block: state in, stage arguments, y out
      double* es = reinterpret_cast<double*>(lds + L::fsal<T, DIM>(slot)) + sysInTile * DIM;  // FSAL in, error components, FSAL out
      double* tp = reinterpret_cast<double*>(lds + L::tring<T, DIM>((int)(r & 7))) + sysInTile;
      double* dp = reinterpret_cast<double*>(lds + L::dt<T, DIM>(slot)) + sysInTile;
      const int64_t sidx = (vblock + r * vgrid) * T + sysInTile;
      AdvState<CPL> cur;
      AdvResult<CPL> res;
      // one LDS round trip for everything (a finished system's slot holds stale state: read, never used)
      cur.t = *tp;
      cur.dt = *dp;
#pragma unroll
      for (int j2 = 0; j2 < CPL; j2 += 2) {
        const nnhip_v2d v = *reinterpret_cast<const nnhip_v2d*>(ys + c + j2), w = *reinterpret_cast<const nnhip_v2d*>(es + c + j2);
        cur.y[j2] = v.x; cur.y[j2 + 1] = v.y; cur.fsal[j2] = w.x; cur.fsal[j2 + 1] = w.y;
      }
      cur.live = sidx < a.N && cur.t < a.tEnd;  // :511 (the mover's predicate)
      const LpsOps<RHS, false, CPL> ops{P, ys, es, c};
      if constexpr (PROBE >= 2) {
        res.live = cur.live; res.t = cur.t; res.dt = cur.dt; res.error = 0.0;
#pragma unroll
        for (int j2 = 0; j2 < CPL; ++j2) { res.y[j2] = cur.y[j2]; res.fsal[j2] = cur.fsal[j2]; }
        if constexpr (PROBE == 3) {  // synthetic arithmetic: ~500 independent-ish FP64 VALU instructions, no LDS
          double q0 = cur.t, q1 = cur.dt, q2 = 1.0, q3 = 2.0;
          for (int it = 0; it < 125; ++it) asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %2, %2, %3, %0\n\tv_fma_f64 %3, %3, %0, %1" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
          if (q0 == 1.2345e300) res.error = q0 + q1 + q2 + q3;
        }
        if constexpr (PROBE >= 10 && PROBE < 20) {  // idle delay: (PROBE - 9) * 1024 cycles asleep, no VALU
          for (int it = 0; it < (PROBE - 9) * 16; ++it) __builtin_amdgcn_s_sleep(1);
        }
        if constexpr (PROBE >= 20) {  // (PROBE - 19) * 250 FP64 fma
          double q0 = cur.t, q1 = cur.dt, q2 = 1.0, q3 = 2.0;
          for (int it = 0; it < (PROBE - 19) * 62; ++it) asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %1, %1, %2, %3\n\tv_fma_f64 %2, %2, %3, %0\n\tv_fma_f64 %3, %3, %0, %1" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
          if (q0 == 1.2345e300) res.error = q0 + q1 + q2 + q3;
        }
        if constexpr (PROBE == 4) {  // synthetic LDS traffic: 250 reads of this system's block
          double acc = 0.0;
          for (int it = 0; it < 250; ++it) { acc += *reinterpret_cast<volatile double*>(es + ((c + it) & (DIM - 1))); }
          if (acc == 1.2345e300) res.error = acc;
        }
        if constexpr (PROBE == 5) {  // synthetic arithmetic, FP32 VALU
          float q0 = (float)cur.t, q1 = (float)cur.dt, q2 = 1.0f, q3 = 2.0f;
          for (int it = 0; it < 250; ++it) asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %2, %2, %3, %0\n\tv_fma_f32 %3, %3, %0, %1" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
          if (q0 == 1.2345e30f) res.error = q0 + q1 + q2 + q3;
        }
      } else {
        stillActive |= adv_compute<METHOD>(a, ops, cur, res);
      }
      if (res.live) {
#pragma unroll
        for (int j2 = 0; j2 < CPL; j2 += 2) {
          *reinterpret_cast<nnhip_v2d*>(ys + c + j2) = nnhip_v2d{res.y[j2], res.y[j2 + 1]};
          *reinterpret_cast<nnhip_v2d*>(es + c + j2) = nnhip_v2d{res.fsal[j2], res.fsal[j2 + 1]};
        }
      }
      if (c == 0) {
        if (res.live) { *tp = res.t; *dp = res.dt; *(reinterpret_cast<double*>(lds + L::err<T, DIM>(slot)) + sysInTile) = res.error; }
        *(reinterpret_cast<unsigned*>(lds + L::adv<T, DIM>(slot)) + sysInTile) = res.live ? 1u : 0u;
      }
      mc_fence();  // DS operations of one wave execute in issue order: the flag lands after the results
      flags[3 + cw * 3 + slot] = (unsigned)(r + 1);
      NNHIP_MC_T(1)
    }
    NNHIP_MC_TFLUSH(4, 9)
  }
  if (a.active) {
    if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
  }
}

// The same pipeline for thread-per-IVP systems (C3's streamed form).  One tile = blockDim.x IVPs; the prefetched state of the next
// tile waits in registers (8 doubles for a 3-component system).
template <int METHOD, class RHS, bool NT = false>
__global__ __launch_bounds__(kBlock) NNHIP_ADV_TPI_ATTR void advance_tpi_persist_kernel(const StepArgs a) {
  static_assert(MethodTraits<METHOD>::adaptive, "fixed-step methods share (t, dt): use the uniform streaming loop");
  constexpr int D = RHS::dim;
  controller_prologue();
  pin_step_args(a);
  const Params P = a.P;
  const TpiOps<RHS, false> ops{P};
  const int64_t nTiles = (a.N + blockDim.x - 1) / blockDim.x;
  auto prefetch = [&](int64_t tile, AdvState<D>& s) {
    const int64_t i = tile * blockDim.x + threadIdx.x;
    const int64_t ic = i < a.N ? i : a.N - 1;
    adv_load_state<NT>(a, ops, ic * a.ivpStride, s.y, s.fsal);
    s.dt = a.dt_io[ic];
    s.t = a.t_io[ic];
  };
  unsigned int stillActive = 0;
  AdvState<D> cur, nxt;
  AdvResult<D> res;
  res.live = false;
  int64_t iRes = 0;
  int64_t tile = blockIdx.x;
  if (tile < nTiles) prefetch(tile, cur);
  for (; tile < nTiles; tile += gridDim.x) {
    adv_touch(cur);
    adv_commit<METHOD, NT>(a, ops, iRes, iRes * a.ivpStride, true, res);
    const int64_t nt = tile + gridDim.x;
    if (nt < nTiles) prefetch(nt, nxt);
    const int64_t i = tile * blockDim.x + threadIdx.x;
    cur.live = i < a.N && cur.t < a.tEnd;  // :511
    stillActive |= adv_compute<METHOD>(a, ops, cur, res);
    iRes = i;
    cur = nxt;
  }
  adv_commit<METHOD, NT>(a, ops, iRes, iRes * a.ivpStride, true, res);
  if (a.active) {
    if (__syncthreads_or((int)stillActive) && threadIdx.x == 0) a.active[blockIdx.x % kAggSlots] = 1u;
  }
}


// Two IVPs per lane (experiment, round 3): the SoA state of IVPs 2p and 2p + 1 moves as 16-byte accesses (y[c][2p .. 2p+1], two (t, dt)
// pairs = 32 contiguous bytes), the two steps are computed one after the other on the lane.  Requires the SoA layout, an even N, (t, dt)
// side by side and FSAL re-evaluated (mode 3 of the harness): 2.5 vector-memory instructions per IVP and direction instead of 4.
template <int METHOD, class RHS, bool NT = false>
__global__ __launch_bounds__(kBlock) void advance_tpi_pair_kernel(const StepArgs a) {
  constexpr int D = RHS::dim;
  controller_prologue();
  pin_step_args(a);
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i0 = 2 * p;
  if (i0 >= a.N) return;
  const Params P = a.P;
  const TpiOps<RHS, false> ops{P};
  double2* const td = reinterpret_cast<double2*>(a.t_io) + i0;
  const double2 tdA = td[0], tdB = td[1];
  double2 yv[D];
#pragma unroll
  for (int c = 0; c < D; ++c) {
    typedef double nt_d2 __attribute__((ext_vector_type(2)));
    if constexpr (NT) { const nt_d2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_d2*>(&a.y_in[c * a.compStride + i0])); yv[c] = make_double2(v.x, v.y); }
    else yv[c] = *reinterpret_cast<const double2*>(&a.y_in[c * a.compStride + i0]);
  }
  AdvState<D> s;
  AdvResult<D> r, rA;
  double2 tdOut[2] = {tdA, tdB};
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    s.t = k ? tdB.x : tdA.x;
    s.dt = k ? tdB.y : tdA.y;
#pragma unroll
    for (int c = 0; c < D; ++c) { s.y[c] = k ? yv[c].y : yv[c].x; s.fsal[c] = 0.0; }
    s.live = s.t < a.tEnd;
    if (s.live) ops.rhs(s.t, s.y, s.fsal);
    adv_compute<METHOD>(a, ops, s, r);
    if (!r.live) { r.t = s.t; r.dt = s.dt;
#pragma unroll
      for (int c = 0; c < D; ++c) r.y[c] = s.y[c]; }
    if (k == 0) rA = r;
  }
  // write both IVPs back (a finished IVP rewrites the values it read)
#pragma unroll
  for (int c = 0; c < D; ++c) {
    typedef double nt_d2 __attribute__((ext_vector_type(2)));
    if constexpr (NT) { nt_d2 v; v.x = rA.y[c]; v.y = r.y[c]; __builtin_nontemporal_store(v, reinterpret_cast<nt_d2*>(&a.y_out[c * a.compStride + i0])); }
    else *reinterpret_cast<double2*>(&a.y_out[c * a.compStride + i0]) = make_double2(rA.y[c], r.y[c]);
  }
  td[0] = make_double2(rA.t, rA.dt);
  td[1] = make_double2(r.t, r.dt);
  (void)tdOut;
}

}  // namespace NNHIP_NS

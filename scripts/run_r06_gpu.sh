#!/bin/bash
# Round 6: the measurements DESIGN.md section 6 quotes.  Stages, each one gpurun call (box time is budgeted):
#   gpurun --timeout 3000 -- 'bash scripts/run_r06_gpu.sh tests'      the whole -m gpu suite
#   gpurun --timeout 1500 -- 'bash scripts/run_r06_gpu.sh bench'      default bench line + rocprofv3 stats / PMC of the headline kernel (both cache regimes)
#   gpurun --timeout 2400 -- 'bash scripts/run_r06_gpu.sh adaptive'   streamed C3 / C4: general vs lean vs FMA-contracted lean kernels, uniform vs automatic polling (same box, same process),
#                                                                     kernel-trace stats and PMC sets of both C4 kernels and of C3 at 1e6
#   gpurun --timeout 1500 -- 'bash scripts/run_r06_gpu.sh rest'       fused configs, extras, divergence binning, torch-free C5 harness
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STAGE="${1:-all}"
if [ "$STAGE" = tests ] || [ "$STAGE" = all ]; then
  # first contact: smoke as the driver runs it, then the suite WITHOUT -x (one first-contact failure must not hide the rest), per-test timeout so a hung kernel cannot eat the box
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r06_smoke.log; tail -3 gpurun_out/r06_smoke.log
  timeout 2400 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/r06_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r06_pytest_gpu.log
  tail -40 gpurun_out/r06_pytest_gpu.log
fi
if [ "$STAGE" = bench ] || [ "$STAGE" = all ]; then
  timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.log 2>&1; grep '^{' gpurun_out/bench_default.log | tail -1 > gpurun_out/r06_bench_default.json
  bash scripts/profile_gpu.sh > gpurun_out/profile.log 2>&1
  tail -1 gpurun_out/r06_bench_default.json | cut -c1-900
fi
if [ "$STAGE" = adaptive ] || [ "$STAGE" = all ]; then
  # default = the recorded configuration (general kernels, groups of 8); the opt-in settings one by one, same process, same box
  ADV_BENCH_MODES=default,lean,lean_auto_poll,general_auto_poll,lean_auto_poll_fp_contract timeout 900 python scripts/bench_adaptive_stream.py > gpurun_out/r06_bench_adaptive_stream.json 2> gpurun_out/bench_adaptive_stream.err
  # the same command once more for C4: a second run shows the box's own spread
  ADV_BENCH_MODES=default,lean,lean_auto_poll_fp_contract ADV_BENCH_INTEGRATORS=tsit54 ADV_BENCH_ONLY=C4 timeout 300 python scripts/bench_adaptive_stream.py > gpurun_out/r06_bench_adaptive_stream_c4_repeat.json 2>> gpurun_out/bench_adaptive_stream.err
  for cfg in C4:c4 C3_lorenz_N1e+06:c3; do
    only=${cfg%%:*}; tag=${cfg##*:}
    ADV_BENCH_ONLY=$only ADV_BENCH_MODES=default,lean,lean_auto_poll_fp_contract timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_adv_stats_$tag -o adv -- \
      python scripts/bench_adaptive_stream.py > gpurun_out/prof_adv_stats_$tag.log 2>&1
  done
  # PMC: the general, the lean and the contracted lean kernel of streamed C4 (Tsit54), and streamed C3 at 1e6 (DOPRI54), counters only, one group per pass
  ADV_BENCH_INTEGRATORS=tsit54 ADV_BENCH_MODES=lean ADV_BENCH_ONLY=C4 KFILTER="nnhip::advance_lps_lean_kernel<2," TAG=c4lean PMC_GROUPS=3 bash scripts/profile_c4_stream.sh > gpurun_out/prof_c4lean.log 2>&1
  ADV_BENCH_INTEGRATORS=tsit54 ADV_BENCH_MODES=default ADV_BENCH_ONLY=C4 KFILTER="advance_lps_kernel<2," TAG=c4general PMC_GROUPS=3 bash scripts/profile_c4_stream.sh > gpurun_out/prof_c4general.log 2>&1
  ADV_BENCH_INTEGRATORS=tsit54 ADV_BENCH_MODES=lean_auto_poll_fp_contract ADV_BENCH_ONLY=C4 KFILTER="nnhip_fast::advance_lps_lean_kernel<2," TAG=c4contracted PMC_GROUPS=3 bash scripts/profile_c4_stream.sh > gpurun_out/prof_c4contracted.log 2>&1
  ADV_BENCH_INTEGRATORS=dopri54 ADV_BENCH_MODES=lean ADV_BENCH_ONLY=C3_lorenz_N1e+06 KFILTER="nnhip::advance_tpi_lean_kernel<1," TAG=c3lean PMC_GROUPS=5 bash scripts/profile_c4_stream.sh > gpurun_out/prof_c3lean.log 2>&1
  ADV_BENCH_INTEGRATORS=dopri54 ADV_BENCH_MODES=default ADV_BENCH_ONLY=C3_lorenz_N1e+06 KFILTER="advance_tpi_kernel<1," TAG=c3general PMC_GROUPS=2 bash scripts/profile_c4_stream.sh > gpurun_out/prof_c3general.log 2>&1
  # the kernels alone, torch-free (tools/microbench/mb_adv.hip): general vs lean, bit-exact and FMA-contracted builds, 60 timed iterations each, interleaved and repeated
  make -C tools/microbench mb_adv mb_adv_contract > gpurun_out/mb_build.log 2>&1
  timeout 300 tools/microbench/mb_adv lean > gpurun_out/r06_mb_adv_lean.txt 2>&1
  timeout 300 tools/microbench/mb_adv_contract lean > gpurun_out/r06_mb_adv_lean_contracted.txt 2>&1
  python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_adaptive_stream.json"))
for k, v in d.items():
    print(k, "us/iter %.2f launches %d iterations %d equal %s" % (v["us_per_iteration"], v["launches"], v["iterations"], v["equal_to_fused"]))
PY
fi
if [ "$STAGE" = rest ] || [ "$STAGE" = all ]; then
  timeout 300 python scripts/bench_configs.py > gpurun_out/r06_bench_configs.json 2> gpurun_out/bench_configs.err
  timeout 300 python scripts/bench_extra.py > gpurun_out/r06_bench_extra.json 2> gpurun_out/bench_extra.err
  timeout 300 python scripts/bench_divergence.py > gpurun_out/r06_bench_divergence.json 2> gpurun_out/bench_divergence.err
  g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include tests/cpp/bench_c5.cpp -L numericalnim_amd/csrc -lnnhip_ode -L /opt/rocm/lib -lamdhip64 \
      -Wl,-rpath,$PWD/numericalnim_amd/csrc -Wl,-rpath,/opt/rocm/lib -o /tmp/bench_c5 && /tmp/bench_c5 --gpus 1 --steps 10 --warmup 2 --verify 2>/dev/null | grep "^{" > gpurun_out/r06_bench_c5_cpp.json
  tail -c 400 gpurun_out/r06_bench_divergence.json
  # does this box's HIP runtime load zstd-compressed offload bundles (format version 2) of this library?  `make COMPRESS=1`: 44 MB -> 14 MB, opt-in until this says yes
  rm -rf /tmp/nnc && mkdir -p /tmp/nnc/numericalnim_amd && cp -r numericalnim_amd/csrc /tmp/nnc/numericalnim_amd/csrc && cp -r include /tmp/nnc/include
  (cd /tmp/nnc/numericalnim_amd/csrc && rm -f *.o libnnhip_ode.so && make COMPRESS=1 -j"$(nproc)" > /tmp/nnc/build.log 2>&1; ls -la libnnhip_ode.so) > gpurun_out/r06_compressed_build.txt 2>&1
  NNHIP_LIB=/tmp/nnc/numericalnim_amd/csrc/libnnhip_ode.so timeout 300 python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r06_compressed_build.txt 2>&1; echo "smoke on the compressed build rc=$?" >> gpurun_out/r06_compressed_build.txt
  tail -3 gpurun_out/r06_compressed_build.txt
fi

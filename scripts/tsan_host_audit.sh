#!/bin/bash
# One-off audit (no GPU needed): the HOST code of libnnhip_ode.so under ThreadSanitizer.  The host-logic translation units are rebuilt with
# -Xarch_host -fsanitize=thread (device code unchanged; the kernel objects of the normal build are linked as they are), tests/cpp/fake_hip.cpp stands in
# for the HIP runtime (kernels do nothing), and three multi-threaded drivers run on it: tests/cpp/bench_multithread_launch.cpp, tests/fake_hip_scenarios.py
# (1 and 3 fake devices) and tests/fake_hip_thread_stress.py (8 threads x every family of entry, eager and graph-replayed, recorded and opt-in kernels).
# SAN=address runs the same drivers under AddressSanitizer + UBSan instead (heap misuse in the host code: use after free, overruns, bad frees).
# Output: one line per run with the number of ThreadSanitizer reports; the reports themselves under $OUT (default /tmp/nnhip_tsan).
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SAN="${SAN:-thread}"
OUT="${OUT:-/tmp/nnhip_${SAN}san}"
if [ "$SAN" = thread ]; then FS="-fsanitize=thread"; RTL=libclang_rt.tsan-x86_64.so; PAT="WARNING: ThreadSanitizer"; else FS="-fsanitize=address,undefined -fno-sanitize-recover=undefined"; RTL=libclang_rt.asan-x86_64.so; PAT="ERROR: AddressSanitizer\|runtime error:"; fi
CL=/opt/rocm/lib/llvm/bin/clang++
RT="$(dirname "$($CL -print-file-name=$RTL 2>/dev/null || echo /opt/rocm/lib/llvm/lib/clang/22/lib/linux/x)")"
[ -f "$RT/$RTL" ] || RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
mkdir -p "$OUT" && cd "$ROOT/numericalnim_amd/csrc" || exit 1
make -s embedded_headers.inc >/dev/null
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fPIC -Wno-unused-function $(for o in $FS; do printf -- "-Xarch_host %s " "$o"; done)"
for f in ode_capi ode_capi_calls ode_capi_stream ode_capi_aux ode_capi_quad ode_multigpu ode_sort; do hipcc $FLAGS -c $f.hip -o "$OUT/$f.o" & done
hipcc $FLAGS -DNNHIP_BUILD_ROCM_PATH='"/opt/rocm"' -c ode_rtc.hip -o "$OUT/ode_rtc.o" & wait
hipcc --offload-arch=gfx950 -shared -fPIC $FS -shared-libsan -o "$OUT/libnnhip_ode.so" "$OUT"/*.o ode_tu_*.o -lhiprtc -ldl 2>/dev/null || { echo "link failed (build the library first: make -C numericalnim_amd/csrc)"; exit 1; }
$CL -std=c++17 -O1 -g -shared -fPIC $FS -shared-libsan -w -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include "$ROOT/tests/cpp/fake_hip.cpp" -o "$OUT/libfakehip.so" || exit 1
ln -sf libfakehip.so "$OUT/librccl.so.1"
$CL -O1 -g -std=c++17 $FS -shared-libsan -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I "$ROOT/include" "$ROOT/tests/cpp/bench_multithread_launch.cpp" -L "$OUT" -lnnhip_ode \
    -L /opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,"$OUT" -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,"$RT" -o "$OUT/mt_harness" || exit 1
run() {  # tag, devices, command...
  local tag=$1 dev=$2; shift 2
  NNHIP_LIB="$OUT/libnnhip_ode.so" LD_PRELOAD="$RT/$RTL $OUT/libfakehip.so" FAKE_HIP_LIB="$OUT/libfakehip.so" FAKE_HIP_DEVICES=$dev LD_LIBRARY_PATH="$OUT:$RT" \
    ASAN_OPTIONS="detect_leaks=0 halt_on_error=0 exitcode=0 allocator_may_return_null=1" UBSAN_OPTIONS="print_stacktrace=1" TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 exitcode=0 second_deadlock_stack=1" timeout 1800 "$@" > "$OUT/$tag.out" 2> "$OUT/$tag.err"
  echo "$tag: rc=$? last line: $(tail -1 "$OUT/$tag.out" | cut -c1-80) | sanitizer ($SAN) reports: $(grep -c "$PAT" "$OUT/$tag.err")"
  grep SUMMARY "$OUT/$tag.err" | sort | uniq -c | sort -rn | head -12
}
cd "$ROOT"
run mt_harness 1 "$OUT/mt_harness" --rk4-steps 50 --reps 2
run scenarios_1dev 1 python tests/fake_hip_scenarios.py
run scenarios_3dev 3 python tests/fake_hip_scenarios.py
STRESS_KNOBS="stream_graph=0" run stress_eager 1 python tests/fake_hip_thread_stress.py
STRESS_KNOBS="stream_graph=1" run stress_graph 1 python tests/fake_hip_thread_stress.py
STRESS_KNOBS="stream_graph=1" STRESS_TOGGLE_KNOBS=1 run stress_graph_knobs_toggled 1 python tests/fake_hip_thread_stress.py
STRESS_KNOBS="stream_graph=1" STRESS_RELEASE=1 run stress_graph_release_between_rounds 1 python tests/fake_hip_thread_stress.py
STRESS_KNOBS="stream_graph=0,adv_lean=1,adv_auto_poll=1,fp_contract=1" run stress_opt_in 1 python tests/fake_hip_thread_stress.py
if [ "$SAN" != thread ]; then
  # the boundary's error behaviour: every entry of include/nnhip_ode.h with hostile arguments (tests/fake_hip_abi_arg_fuzz.py), one process per entry
  mkdir -p "$OUT/fuzz"; bad=0
  for n in $(python -c "from numericalnim_amd import _lib; print(' '.join(_lib.SIGNATURES))"); do
    NNHIP_LIB="$OUT/libnnhip_ode.so" LD_PRELOAD="$RT/$RTL $OUT/libfakehip.so" FAKE_HIP_LIB="$OUT/libfakehip.so" FAKE_HIP_DEVICES=2 LD_LIBRARY_PATH="$OUT:$RT" \
      ASAN_OPTIONS="detect_leaks=0 exitcode=77 allocator_may_return_null=1" UBSAN_OPTIONS="print_stacktrace=1" timeout 120 python tests/fake_hip_abi_arg_fuzz.py $n ${FUZZ_TRIALS:-200} > "$OUT/fuzz/$n.out" 2> "$OUT/fuzz/$n.err"
    rc=$?
    if [ $rc -ne 0 ] || ! grep -q "^DONE" "$OUT/fuzz/$n.out"; then
      bad=$((bad + 1)); echo "arg fuzz $n: rc=$rc last call: $(grep -v '^DONE' "$OUT/fuzz/$n.out" | tail -1 | cut -c1-260)"
      grep -m3 "ERROR: AddressSanitizer\|runtime error\|SUMMARY\|fake_hip\|AssertionError" "$OUT/fuzz/$n.err" | cut -c1-240
    fi
  done
  echo "arg fuzz: $(ls "$OUT/fuzz"/*.out | wc -l) entries x ${FUZZ_TRIALS:-200} hostile calls, $bad entries with a finding"
fi

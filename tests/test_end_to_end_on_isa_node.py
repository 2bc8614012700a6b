"""The library END TO END without a GPU: its real host logic (C ABI, Python mirror) on a fake node (tests/cpp/fake_hip.cpp: N devices, tracked memory, RCCL with real
data movement) whose every kernel launch is executed by the gfx950 interpreter from the library's OWN compiled code objects or from what hiprtc just compiled
(tests/isa_backed_node.py, tools/gfx950_isa_interp.py) — and the RESULTS compared with the reference's text and the oracle.  tests/isa_node_scenarios.py holds the
scenarios; each runs in a subprocess (LD_PRELOAD):
  * golden fixtures through nn.solveODE: rows, row counts, output times == the reference's text, accepted / rejected counts == the oracle's (29 solves by default;
    NNHIP_ISA_NODE_FULL=1: all 160, recorded in profiles/r05_isa_node_full.txt);
  * what needs several GPUs or had never run: context blocks cut along 3 and 8 shards == the oracle's closures; mutable slots back from their shards; two threads with
    their own contexts; the adaptive streaming loop over the lean kernels with the automatic polling (102 iterations, 104 launches, the fused solve's bits) and the dense
    driver; config C5's shape on 3 devices with the RCCL reassembly; the consumers' entries == the reference's text; the order of integration over
    four key shapes; the general advance kernels through the same driver (knob adv_lean = 0): the lean kernels' bits and launches.
TEST INFRASTRUCTURE, five to six orders of magnitude slower than a GPU; the product has no CPU path and this is not one (nothing in the package or the library refers to
it; it needs LD_PRELOAD, the ROCm LLVM tools and the build tree's object files)."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.slow  # the whole file runs on the ISA-backed fake node (minutes)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "numericalnim_amd", "csrc")


@pytest.fixture(scope="module")
def fake(tmp_path_factory, nn):
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/rccl/rccl.h") or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs g++, the ROCm headers and the ROCm LLVM tools")
    if not os.path.exists(os.path.join(CSRC, "ode_tu_m_tsit54.o")):
        pytest.skip("needs the build tree's object files (python -c 'import __graft_entry__ as g; g.build()')")
    d = tmp_path_factory.mktemp("isa_node")
    lib = str(d / "libfakehip.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "fake_hip.cpp"), "-o", lib, "-ldl"])
    os.symlink(lib, str(d / "librccl.so.1"))
    return str(d), lib


def _run(fake, devices, scenario, timeout=900):
    d, lib = fake
    env = dict(os.environ, LD_PRELOAD=lib, FAKE_HIP_LIB=lib, FAKE_HIP_DEVICES=str(devices), LD_LIBRARY_PATH=d + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "isa_node_scenarios.py"), scenario], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "ALL OK" and "live device allocations: 0" in lines[-2], lines[-4:]
    return r.stdout


def test_golden_fixtures_end_to_end(fake):
    out = _run(fake, 1, "golden_fixtures")
    assert "golden solves == the reference's text" in out


@pytest.mark.parametrize("scenario,devices,needle", [
    ("sharded_context_results", 3, "== the oracle's closures at 3 and 8 shards"),
    ("mutable_slots_results", 3, "== the oracle's environments"),
    ("two_threads_results", 2, "each its own result"),
    ("streaming_results", 1, "C3 104 launches, C4 104 launches"),
    ("c5_shape_with_results", 3, "every device holds the oracle's"),
    ("c5_shape_with_results", 8, "C5 shape on 8 devices"),
    ("device_resident_shards_results", 4, "fused + gathered + streamed == the oracle's closures on 4 devices"),
    ("consumers_results", 1, "== the reference's text"),
    ("bin_order_results", 1, "permutations ascending from slice to slice"),
])
def test_first_executions_with_results(fake, scenario, devices, needle):
    out = _run(fake, devices, scenario)
    assert needle in out, out[-600:]

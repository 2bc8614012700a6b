"""The Python that only ever runs on a GPU box — __graft_entry__.smoke(), bench.py, the package's device-tensor paths — executed here, without a GPU, on the
ISA-backed fake node (tests/isa_backed_node.py) through tests/fake_torch: a stand-in for the handful of PyTorch names that Python uses, whose "cuda" tensors live in
the fake node's tracked device memory.  The kernels that run are the library's own gfx950 code objects (tools/gfx950_isa_interp.py), the host logic is the library's,
the Python is the file the driver will run.  Test infrastructure: what it shows is that these files execute and produce checked results; it says nothing about
PyTorch, timing or a real device.  The wider run (the -m gpu test files themselves under the same stand-ins) is scripts/run_gpu_suite_on_isa_node.py; its record for
this round is profiles/r05_gpu_suite_on_isa_node.txt."""
import json
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.slow  # the whole file runs on the ISA-backed fake node (minutes)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def node_env(tmp_path_factory, nn):
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/rccl/rccl.h") or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("needs g++, the ROCm headers and the ROCm LLVM tools")
    d = tmp_path_factory.mktemp("fake_node")
    lib = str(d / "libfakehip.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "fake_hip.cpp"), "-o", lib])
    os.symlink(lib, str(d / "librccl.so.1"))
    env = dict(os.environ, LD_PRELOAD=lib, FAKE_HIP_LIB=lib, FAKE_HIP_DEVICES="1", LD_LIBRARY_PATH=str(d) + ":" + os.environ.get("LD_LIBRARY_PATH", ""),
               PYTHONPATH=os.path.join(ROOT, "tests", "fake_torch") + ":" + ROOT)
    env.pop("WORLD_SIZE", None)
    return env


def test_smoke_runs_on_the_isa_node(node_env):
    """__graft_entry__.smoke() as the driver calls it: fused + streamed RK4, DOPRI54 on Lorenz, the dense Tsit54 streaming solve in both directions — bit-exact vs the oracle."""
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, env=node_env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2500:])
    assert "smoke ok" in r.stdout and "bit-exact vs the oracle" in r.stdout, r.stdout[-500:]


def test_bench_line_on_the_isa_node(node_env):
    """bench.py's required path + the fused leg + the CPU baseline at a size the interpreter finishes in seconds: one JSON line, the contract's keys, the timed
    kernel's result equal to the oracle's, the launch count of the step-streaming solve exact."""
    r = subprocess.run([sys.executable, "bench.py", "--n-ivp", "4096", "--rk4-steps", "8", "--steps", "2", "--warmup", "1", "--adaptive-n", "0", "--beyond-cache-n", "4096",
                        "--cpu-sample", "1000", "--cpu-ivps-per-thread", "100"], cwd=ROOT, env=node_env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["parity_max_abs_err_vs_oracle"] == 0.0 and d["parity_checked_ivps"] == 1000
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"]) and d["roofline"]["bound"] == "hbm"
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["all_cores"]["cores"] >= 1
    assert d["fused_solve"]["bitwise_equal_to_stream"] is True
    assert d["beyond_infinity_cache"]["launches"] == 100
    assert "informational_errors" not in d, d["informational_errors"]     # (--adaptive-n 0 skips the 1e6-sized legs; the others ran)


@pytest.mark.skipif(not os.environ.get("NNHIP_ISA_NODE_FULL"), reason="~7 min: every informational leg of bench.py at 192 IVPs / systems (NNHIP_ISA_NODE_FULL=1)")
def test_every_leg_of_bench_on_the_isa_node(node_env):
    r = subprocess.run([sys.executable, "bench.py", "--n-ivp", "4096", "--rk4-steps", "8", "--steps", "2", "--warmup", "1", "--adaptive-n", "192", "--beyond-cache-n", "8192",
                        "--cpu-sample", "1000", "--cpu-ivps-per-thread", "100", "--cpu-adaptive-sample", "64"], cwd=ROOT, env=dict(node_env, NNHIP_BENCH_CHILD_TIMEOUT="2400"),
                       capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2500:])
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "informational_errors" not in d, d["informational_errors"]
    for name in ("C3_dopri54_lorenz_1e6", "C4_tsit54_ring16_1e6"):
        c = d["adaptive_configs"][name]
        assert c["streamed_bitwise_equal_to_fused"] and c["loop_iterations"] <= c["streamed_launches"] <= c["loop_iterations"] + 16 and c["cpu_baseline"]["max_abs_dev_gpu_vs_cpu"] == 0.0
        # the opt-in settings come back from bench.py's child process (bench.streamed_opt_in_parent)
        assert set(c["streamed_opt_in"]) == {"lean", "lean_auto_poll", "lean_auto_poll_fp_contract"} and all(o["within_north_star_tolerance"] for o in c["streamed_opt_in"].values())
        assert c["streamed_opt_in"]["lean"]["bitwise_equal_to_fused"] and c["streamed_opt_in"]["lean_auto_poll"]["bitwise_equal_to_fused"]
    assert all(v["within_tolerance"] for v in d["fused_solve_fp_contract"].values())
    assert d["heterogeneous_batches"]["sweep_bitwise_equal"] and d["heterogeneous_batches"]["calls_bitwise_equal"]


_TWO_THREAD_BIND = r'''
import ctypes as C, os, threading
import numpy as np, torch
import numericalnim_amd as nn
F = C.CDLL(os.environ["FAKE_HIP_LIB"]); F.fake_hip_owns.restype = C.c_int
dev = torch.device("cuda:0")
f = nn.Rhs.custom(2, "dy[0] = g[0] * y[1]; dy[1] = -g[1] * y[0];", tvalues={"g": 2}, name="osc")     # one Rhs object, shared by the threads
step = threading.Barrier(2)
ok = [None, None]
def worker(k):
    ctx = nn.newNumContext(tValues={"g": np.array([1.0 + k, 2.0 + k])})
    if k == 1:
        step.wait()                      # thread 0 has bound
    f.bind(ctx, dev)
    ptr = f._bound_tls.bound[0].data_ptr()     # (the address only: holding the tensor would itself keep the block alive)
    if k == 0:
        step.wait()
    step.wait()                          # both have bound: thread 1's bind must not have freed what thread 0's binding points at
    ok[k] = bool(F.fake_hip_owns(C.c_void_p(ptr), C.c_size_t(16))) and list((C.c_double * 2).from_address(ptr)) == [1.0 + k, 2.0 + k]
th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
[t.start() for t in th]; [t.join() for t in th]
assert ok == [True, True], ok
print("BOTH BINDINGS ALIVE")
'''


def test_a_bind_of_one_thread_does_not_free_what_another_threads_binding_points_at(node_env):
    """Found by tests/test_ctx_block.py's two-thread test on the fake node (the allocation registry refused a kernel's read of a freed block): the device tensors
    a bind uploads were kept alive in ONE slot of the Rhs object, so thread A's bind freed the block thread B's (thread-local) library binding still pointed at —
    on a GPU the caching allocator would have handed that block to A's next upload and B would have integrated with A's context.  The slot is per thread now."""
    r = subprocess.run([sys.executable, "-c", _TWO_THREAD_BIND], cwd=ROOT, env=node_env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "BOTH BINDINGS ALIVE" in r.stdout, (r.stdout[-500:], r.stderr[-2500:])


def _line(r):
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-2500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]          # only rank 0 prints
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_on_the_isa_node(node_env):
    """`python bench.py --gpus 2` with WORLD_SIZE unset (the shape of the driver's N = 1 line with another N): bench.py starts its own two ranks through
    `python -m torch.distributed.run` — here the stand-in launcher and a file-exchange process group (tests/fake_torch/torch/distributed) — each rank integrates its
    contiguous shard on the node, every overlapped all-gather is compared with the solve it belongs to, rank 0 prints the one line.  The multi-rank path of bench.py
    (buffer rotation, max-over-ranks timing through an all-reduce, placement check) had never executed since it was changed in round 5."""
    env = {k: v for k, v in node_env.items() if k not in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--verify-gathers", "--steps", "3", "--warmup", "1", "--n-ivp", "4096", "--rk4-steps", "8"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _line(r)
    assert d["n_gpus"] == 2 and d["world_size_reported_by_backend"] == 2 and d["config"]["backend"] == "gloo" and d["scaling"] == "weak"
    assert d["gathers_verified"] == 4 and d["allgather_ms_per_solve"] > 0 and "cpu_baseline" not in d
    pr = d["ms_per_step_per_rank"]
    assert 0 < pr["min"] <= pr["max"] == d["ms_per_step"]
    assert abs(d["value"] - 2 * 4096 * 8 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-9     # whole-job aggregate


def test_the_drivers_eight_rank_line_on_an_eight_device_fake_node(node_env):
    """The SCALE run's launch line as the task states it — `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P
    bench.py --gpus 8 --steps K --warmup W`, backend "nccl", one rank per device — on an 8-device fake node: config C5's shape end to end through bench.py."""
    env = dict(node_env, FAKE_HIP_DEVICES="8")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29611",
                        "bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--verify-gathers", "--n-ivp", "1000", "--rk4-steps", "6"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    d = _line(r)
    assert d["n_gpus"] == 8 and d["world_size_reported_by_backend"] == 8 and "backend" not in d["config"]        # (nccl is the default and is not spelled out)
    assert d["gathers_verified"] == 4 and d["config"]["final_state_allgather"] and d["config"]["allgather_overlapped_with_next_solve"]
    assert abs(d["value"] - 8 * 1000 * 6 * 3 / (d["ms_per_step"] * 3e-3)) / d["value"] < 1e-9
    # more ranks than devices over the backend that wants one device per rank: refused with a message, not a hang (2 devices, 8 ranks)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29612",
                        "bench.py", "--gpus", "8", "--n-ivp", "1000", "--rk4-steps", "6"], cwd=ROOT, env=dict(node_env, FAKE_HIP_DEVICES="2"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "need 8 devices" in r.stderr

"""The HOST logic of the multi-device entries on a fake node: tests/cpp/fake_hip.cpp stands in for the HIP runtime (LD_PRELOAD) and for RCCL ("librccl.so.1") in a
subprocess — FAKE_HIP_DEVICES devices, device memory as tracked host allocations (a copy outside the allocation it addresses, a double free, a width beyond a pitch
abort the process), RCCL's collectives with their real data movement, kernel launches that do nothing.  tests/fake_hip_scenarios.py then drives the library through
its C ABI: context blocks cut along the shards (host entry, both device-resident entries, more shards than devices), mutable slots out and back, four threads binding
their own contexts to one source, the reassembly of sharded tensors (SoA plane by plane, ragged, an empty shard) — code that needs >= 2 GPUs or was written in a
round without one, and had never executed.  Nothing is solved here (no kernel runs); the product has no CPU path and this is not one."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fake(tmp_path_factory, nn):
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("needs g++ and the ROCm headers")
    d = tmp_path_factory.mktemp("fake_hip")
    lib = str(d / "libfakehip.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-Wno-unused-result", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "fake_hip.cpp"), "-o", lib])
    os.symlink(lib, str(d / "librccl.so.1"))
    return str(d), lib


def _env(fake, devices):
    d, lib = fake
    return dict(os.environ, LD_PRELOAD=lib, FAKE_HIP_LIB=lib, FAKE_HIP_DEVICES=str(devices), LD_LIBRARY_PATH=d + ":" + os.environ.get("LD_LIBRARY_PATH", ""))


@pytest.mark.parametrize("devices", [1, 3, 8])
def test_multi_device_host_logic(fake, devices):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_hip_scenarios.py")], env=_env(fake, devices), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = r.stdout.strip().splitlines()
    assert lines[-1] == "ALL OK", lines[-5:]
    assert "live device allocations after nnhip_release(): 0" in r.stdout     # nothing leaked: bindings, shard copies, staging, workspaces
    if devices >= 3:
        assert "RCCL reassembly on %d devices: 6 tensors placed" % devices in r.stdout and "gathered on %d devices" % devices in r.stdout


def test_the_stand_in_catches_an_out_of_bounds_copy(fake):
    """(the harness itself: a device-side range that runs over the end of its allocation aborts the process)"""
    code = ("import ctypes as C, os\nF = C.CDLL(os.environ['FAKE_HIP_LIB'])\np = C.c_void_p()\nassert F.hipMalloc(C.byref(p), C.c_size_t(64)) == 0\n"
            "buf = (C.c_char * 128)()\nassert F.hipMemcpy(p, buf, C.c_size_t(64), 1) == 0\nprint('in bounds ok', flush=True)\nF.hipMemcpy(p, buf, C.c_size_t(72), 1)\nprint('NOT REACHED')\n")
    r = subprocess.run([sys.executable, "-c", code], env=_env(fake, 1), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "in bounds ok" in r.stdout and "NOT REACHED" not in r.stdout and "not inside one live device allocation" in r.stderr


def _build_harness(tmp_path, src):
    libdir = os.path.join(ROOT, "numericalnim_amd", "csrc")
    exe = str(tmp_path / src.replace(".cpp", ""))
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", src), "-L", libdir, "-lnnhip_ode", "-L", "/opt/rocm/lib", "-lamdhip64", "-lpthread",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


@pytest.mark.parametrize("gpus", [1, 2, 8])
def test_c5_harness_on_a_fake_eight_gpu_node(fake, tmp_path, gpus):
    """tests/cpp/bench_c5.cpp — config C5 through ONE C call from a process without PyTorch: shards resident per device, the graph-replayed step-streaming solve on
    every device's own stream and worker thread, RCCL reassembly on a second stream — at 1, 2 and 8 of 8 fake devices: it runs to its JSON line (no --verify: nothing
    is computed here).  On hardware this harness has only ever run with one GPU."""
    import json
    exe = _build_harness(tmp_path, "bench_c5.cpp")
    r = subprocess.run([exe, "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--n-per-gpu", "100000", "--rk4-steps", "20"], env=_env(fake, 8),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == gpus and out["scaling"] == "weak" and out["value"] > 0


def test_multithread_launch_harness_on_the_fake_node(fake, tmp_path):
    """tests/cpp/bench_multithread_launch.cpp: 1, 2, 4, 8 host threads driving their own streams through the library at once, eager and graph-replayed — the
    process-wide graph cache hit and released from several threads: no crash, no bad free, no copy out of bounds."""
    import json
    exe = _build_harness(tmp_path, "bench_multithread_launch.cpp")
    r = subprocess.run([exe, "--rk4-steps", "50", "--reps", "2"], env=_env(fake, 1), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-800:], r.stderr[-1500:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] is True and out["c5_eager_G8"]["launches_per_s_aggregate"] > 0

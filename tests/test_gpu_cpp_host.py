"""Builds and runs the C++ host-side mirror harness (tests/cpp/test_ode_mirror.cpp over include/numericalnim_hip.hpp),
which restates the scalar and Vector RK4/DOPRI54/Tsit54 cases of the reference's tests/test_ode.nim."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(nn, tmp_path):
    exe = str(tmp_path / "test_ode_mirror")
    libdir = os.path.join(ROOT, "numericalnim_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_ode_mirror.cpp"),
                           "-L", libdir, "-lnnhip_ode", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")


def test_cpp_device_entries_standalone(nn, tmp_path):
    """Device-pointer entries, hipGraph replay, adaptive streaming, RCCL reassembly and hiprtc from a process WITHOUT PyTorch
    (system ROCm runtime / hiprtc / librccl only) — what a compiled or Nim host sees."""
    exe = str(tmp_path / "test_device_entries")
    libdir = os.path.join(ROOT, "numericalnim_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_device_entries.cpp"), "-L", libdir, "-lnnhip_ode", "-L", "/opt/rocm/lib",
                           "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")


def test_c5_cpp_harness(nn, tmp_path):
    """tests/cpp/bench_c5.cpp: config C5 through ONE C call from a process without PyTorch — shards resident per device, step-streaming
    solve, RCCL reassembly on a second stream — at every device count this box has (1 on the one-GPU boxes), verified, bench.py's JSON keys."""
    import json
    import torch
    exe = str(tmp_path / "bench_c5")
    libdir = os.path.join(ROOT, "numericalnim_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "bench_c5.cpp"), "-L", libdir, "-lnnhip_ode", "-L", "/opt/rocm/lib",
                           "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    for g in sorted({1, torch.cuda.device_count()}):
        r = subprocess.run([exe, "--gpus", str(g), "--steps", "3", "--warmup", "1", "--n-per-gpu", "1000000", "--rk4-steps", "200", "--verify"],
                           capture_output=True, text=True, timeout=600)
        print(r.stdout[-2000:], r.stderr[-2000:])
        assert r.returncode == 0
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["n_gpus"] == g and out["verified"] is True and out["value"] > 1e9 and out["unit"] == "trajectory-steps/s"


def test_multithread_launch_harness(nn, tmp_path):
    """tests/cpp/bench_multithread_launch.cpp: G = 1, 2, 4, 8 host threads, each driving its own stream and shard on device 0 through the worker
    of the one-call multi-GPU entry (nnhip_ode_fixed_stream_f64_dev), eager and graph-replayed — the process-wide graph cache is hit from
    several threads at once, entries are released while other threads launch.  Here: it runs and every solve takes its steps; the numbers
    are recorded by scripts/run_r04_gpu.sh into profiles/r04_multithread_launch.json."""
    import json
    exe = str(tmp_path / "bench_multithread_launch")
    libdir = os.path.join(ROOT, "numericalnim_amd", "csrc")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "bench_multithread_launch.cpp"), "-L", libdir, "-lnnhip_ode", "-L", "/opt/rocm/lib",
                           "-lamdhip64", "-lpthread", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    r = subprocess.run([exe, "--rk4-steps", "100", "--reps", "2"], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] is True and out["c5_eager_G8"]["launches_per_s_aggregate"] > 0

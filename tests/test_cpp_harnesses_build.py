"""The torch-free C++ harnesses the -m gpu suite compiles on the GPU box (tests/test_gpu_cpp_host.py: what a compiled or Nim host sees of the library) must at
least COMPILE AND LINK against the built library here: a signature that drifted from include/nnhip_ode.h would otherwise first show as a compiler error on the GPU
box.  Nothing is executed (no device)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("src", ["test_ode_mirror.cpp", "test_device_entries.cpp", "bench_c5.cpp", "bench_multithread_launch.cpp"])
def test_harness_compiles_and_links(nn, tmp_path, src):
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("needs g++ and the ROCm headers")
    libdir = os.path.join(ROOT, "numericalnim_amd", "csrc")
    exe = str(tmp_path / src.replace(".cpp", ""))
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", src), "-L", libdir, "-lnnhip_ode", "-L", "/opt/rocm/lib", "-lamdhip64", "-lpthread",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    assert os.path.exists(exe)

"""Run-time compiled right-hand sides under BOTH compilers, and the fall-back paths between them.

By default user source is compiled by the libhiprtc of the ROCm this library was built with, loaded into a link namespace of its own when the
host process bundles another one (PyTorch wheels do; numericalnim_amd/csrc/ode_rtc.hip, rtc_state).  The parity suites of the run-time
compiled kernels must hold with the process's own compiler too (NNHIP_HIPRTC=process), and a private compiler that turns out unusable — a
compilation that fails where the process's own succeeds, a code object the process's runtime refuses — must hand over to the process's
own without the caller noticing (fault injection: NNHIP_HIPRTC_INJECT).  The compiler is chosen once per process, so every mode is its own
pytest process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, args, timeout=1500):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    print(r.stdout[-3000:], r.stderr[-1500:])
    return r


def test_parity_suites_with_the_process_own_hiprtc(nn, dev):
    """tests/test_user_rhs.py, tests/test_ctx_block.py and the run-time instantiated sizes of tests/test_gpu_any_dim.py, bit for bit against the
    oracle / the ahead-of-time kernels, compiled by the libhiprtc the process itself resolves."""
    r = _run({"NNHIP_HIPRTC": "process", "NNHIP_EXPECT_RTC": "process"},
             ["tests/test_user_rhs.py", "tests/test_ctx_block.py", "tests/test_gpu_any_dim.py", "tests/test_gpu_hiprtc_modes.py::test_which_compiler",
              "--deselect", "tests/test_gpu_any_dim.py::test_wide_user_system_method_of_lines"])  # (24 wide-system compiles: they take the main run's compiler)
    assert r.returncode == 0, r.stdout[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout


@pytest.mark.parametrize("inject", ["compile", "load"])
def test_private_compiler_failure_hands_over_to_the_process_own(nn, dev, inject):
    r = _run({"NNHIP_HIPRTC_INJECT": inject, "NNHIP_EXPECT_RTC": "handover:" + inject},
             ["tests/test_user_rhs.py::test_user_lorenz_equals_builtin_bitwise", "tests/test_user_rhs.py::test_user_per_component_rhs_equals_builtin_ring",
              "tests/test_gpu_hiprtc_modes.py::test_which_compiler"])
    assert r.returncode == 0, r.stdout[-2000:]


def test_which_compiler(nn, dev):
    """(Helper, meaningful inside the sub-processes above: NNHIP_EXPECT_RTC says what nnhip_rtc_compiler() must report AFTER the other tests of the
    process have compiled something.)  In the main test process: the default choice is reported and names a libhiprtc."""
    import torch
    text = nn._lib.lib().nnhip_rtc_compiler().decode()
    expect = os.environ.get("NNHIP_EXPECT_RTC", "")
    assert "libhiprtc" in text
    if expect == "process":
        assert "NNHIP_HIPRTC=process" in text and "link namespace" not in text.split(";")[0]
    elif expect.startswith("handover:"):
        # a private compiler existed only if the process bundles a libhiprtc other than the build's; then the injected failure must have retired it
        if "the process already uses the build's libhiprtc" not in text and "does not exist" not in text:
            assert "link namespace" not in text.split(";")[0], text
            assert ("succeeded with the process's own" in text) if expect.endswith("compile") else ("could not load a code object" in text), text

"""Two pieces of round 5 that can be checked without a GPU, against the code the library runs:
* the adaptive streaming driver's polling loop (numericalnim_amd/csrc/adv_poll_schedule.hpp — plain C++, included by ode_capi_stream.hip) replayed against
  a simulated batch: BASELINE's C3 / C4 shape takes 104 launches for its 102 iterations (uniform groups of 8: 112);
* the order of integration's two kernels (numericalnim_amd/csrc/sort_kernels.hpp) with their BODIES executed on the host (tests/cpp/hip_cpu_emu.hpp: test
  infrastructure, lanes as threads): slices linear in value when the keys touch or straddle zero — the round-4 advice's "speed-up vanished silently" case."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# NNHIP_EMU_SANITIZE=1: the kernel bodies run under AddressSanitizer + UBSan (out-of-bounds lane accesses of partly filled workgroups, misaligned vector
# accesses, signed overflow in index arithmetic abort the run) — several times slower, so opt-in; profiles/LAB_NOTES_r05.md records a full pass
SANITIZE = ["-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-DNNHIP_EMU_THREADS"] if os.environ.get("NNHIP_EMU_SANITIZE") else []  # (lanes as OS threads there: ASan does not follow the default engine's swapcontext)


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("needs g++")
    d = tmp_path_factory.mktemp("cpu_exes")
    poll, order = str(d / "poll"), str(d / "order")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "test_poll_schedule.cpp"), "-o", poll])
    subprocess.check_call(["g++", "-std=c++20", "-O1", "-DNNHIP_CPU_EMU", *SANITIZE, "-Wno-attributes", "-I", os.path.join(ROOT, "tests", "cpp"), "-I",
                           os.path.join(ROOT, "numericalnim_amd", "csrc"), "-pthread", os.path.join(ROOT, "tests", "cpp", "emu_bin_order.cpp"), "-o", order])
    return poll, order


def _poll(exe, uniform, check_every, t0, t_end, dt_max, k, max_launches, need):
    r = subprocess.run([exe] + [str(x) for x in (int(uniform), check_every, t0, t_end, dt_max, k, max_launches, need)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    f = r.stdout.split()
    return dict(launches=int(f[3]), groups=int(f[5]), n0=int(f[7]))


def test_polling_schedule(exes):
    poll, _ = exes
    # BASELINE C3 / C4: tspan [0, 1], dtMax 1e-2, 102 iterations (the two short first steps + 100)
    assert _poll(poll, False, 0, 0.0, 1.0, 0.01, 1, 0, 102) == dict(launches=104, groups=3, n0=100)
    assert _poll(poll, True, 8, 0.0, 1.0, 0.01, 1, 0, 102)["launches"] == 112             # round 4: uniform groups of 8
    assert _poll(poll, True, 5, 0.0, 1.5, 1.0, 1, 0, 61)["launches"] == 70                # a caller's check_every is taken as given
    # never fewer launches than iterations, never more than two trailing groups beyond them; long tails poll every 8
    rng = np.random.default_rng(0)
    for _ in range(300):
        t_end = float(rng.uniform(0.01, 50.0))
        dt_max = float(10 ** rng.uniform(-3, 1))
        k = int(rng.choice([1, 1, 1, 2, 5]))
        n0 = int(np.ceil(t_end / dt_max * (1 - 1e-9)))
        need_iters = n0 + int(rng.integers(0, 400))
        need = -(-need_iters // k)
        got = _poll(poll, False, 0, 0.0, t_end, dt_max, k, 0, need)
        assert got["n0"] == -(-n0 // k)
        assert need <= got["launches"] <= need + 16, (t_end, dt_max, k, need, got)
        assert got["groups"] <= 3 + (need - got["n0"]) // 8 + got["n0"] // 4096 + 3
    # max_launches cuts the unpolled part too; a huge lower bound goes out in slices the host can still watch
    assert _poll(poll, False, 0, 0.0, 1.0, 0.01, 1, 50, 102)["launches"] == 50
    big = _poll(poll, False, 0, 0.0, 1e4, 0.01, 1, 0, 1_000_003)
    assert big["launches"] <= 1_000_003 + 16 and big["groups"] >= 1_000_000 // 4096


def _order(exe, keys):
    inp = "\n".join(("nan" if v != v else "inf" if v == np.inf else "-inf" if v == -np.inf else float(v).hex()) for v in keys) + "\n"
    r = subprocess.run([exe], input=inp, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-300:]
    o = np.array([int(x) for x in r.stdout.split()], dtype=np.int64)
    assert np.array_equal(np.sort(o), np.arange(len(keys)))   # a permutation: every IVP once
    return o


def test_bin_order_kernels_on_the_host(exes):
    _, order = exes
    rng = np.random.default_rng(3)
    n = 9500   # three workgroups of 1024 threads x 4 keys, the last one partly filled
    cases = {
        "uniform_with_a_zero": np.concatenate([[0.0], rng.uniform(0.0, 10.0, n - 1)]),
        "steps_left_with_finished_ivps": -np.concatenate([np.zeros(100), rng.uniform(1.0, 500.0, n - 100)]),
        "both_signs": rng.uniform(-5.0, 10.0, n),
        "one_sign_narrow": rng.uniform(100.0, 101.0, n),
        "negative_zero_and_positives": np.concatenate([[-0.0], rng.uniform(0.5, 3.0, n - 1)]),
    }
    for name, keys in cases.items():
        rng.shuffle(keys)
        k = keys[_order(order, keys)]
        viol = float((np.maximum.accumulate(k) - k).max())       # a key may precede a smaller one only inside its own slice
        # (the logarithmic image is cut by a power-of-two shift: its slices are up to twice the ideal width)
        assert viol <= (2.02 if name == "one_sign_narrow" else 1.01) * (keys.max() - keys.min()) / 4094, (name, viol)
    # same-signed keys over six decades keep the logarithmic image: a slice is a fixed RATIO wide (the heavy, small-progress IVPs are not one bin)
    keys = -10.0 ** rng.uniform(-6.0, 0.0, n)
    k = -keys[_order(order, keys)]
    ratio = float((k / np.minimum.accumulate(k)).max())            # descending in |key|: how far above the running minimum a later key may be
    assert ratio < 1.01, ratio
    # non-finite keys go last
    keys = rng.uniform(-1.0, 1.0, n)
    keys[::97] = np.nan
    keys[5::1013] = np.inf
    o = _order(order, keys)
    bad = ~np.isfinite(keys[o])
    assert bad.sum() == (~np.isfinite(keys)).sum() and bad[-bad.sum():].all()

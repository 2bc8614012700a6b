"""bench.py's CPU-baseline legs (cpu_baseline_c2, cpu_baseline_adaptive) are plain functions of numpy arrays: run here on reduced samples, with the
"GPU result" stood in for by the oracle itself, so that a typo in them cannot cost the round its bench line."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_c2_leg(oracle):
    b = _bench()
    from numericalnim_amd import distributed as nd
    O = oracle
    dt, nsteps = 2.0 ** -10, 64
    ref = O.solve_ode_batch(O.RHS_NEG_Y, [], nd.c2_y0_numpy(0, 3000), 3000, 0, [0.0, nsteps * dt], O.new_options(dt=dt), "rk4")["y"][-1, 0]
    cb, check, k = b.cpu_baseline_c2(O, nd.c2_y0_numpy, 100000, nsteps, dt, nsteps * dt, 3000, 500, 4, gpu_first=ref)
    assert check == 0.0 and k == 3000
    assert cb["cores"] == 1 and cb["kind"] == "port" and cb["value"] > 0 and cb["unit"] == "trajectory-steps/s"
    ac = cb["all_cores"]
    assert ac["cores"] == 4 and ac["value"] > 0 and abs(ac["speedup_over_1_core"] - ac["value"] / cb["value"]) < 1e-9 and "3000 IVPs" in ac["sample"]
    # a single-core host: the all-cores leg is the one-core sample again
    cb1, _, _ = b.cpu_baseline_c2(O, nd.c2_y0_numpy, 100000, nsteps, dt, nsteps * dt, 2000, 500, 1)
    assert "first 2000 IVPs" in cb1["all_cores"]["sample"]


def test_adaptive_legs(oracle):
    b = _bench()
    O = oracle
    n = 400
    y3 = np.stack([1.0 + (np.arange(n) % 1024) * 2.0 ** -20, np.ones(n), np.ones(n)])
    y16 = 1.0 + np.arange(16)[None, :] / 16 + ((np.arange(n) % 1024) * 2.0 ** -20)[:, None]
    for name, yh, layout, integ, d, kind, par in (("C3", y3, 0, "dopri54", 3, O.RHS_LORENZ, [10.0, 28.0, 8.0 / 3.0]), ("C4", y16, 1, "tsit54", 16, O.RHS_RING, [0.1])):
        n1 = 64
        first = np.ascontiguousarray(yh[:, :n1]) if layout == 0 else np.ascontiguousarray(yh[:n1])
        gpu = O.solve_ode_batch(kind, par, first, n1, d, [0.0, 1.0], O.new_options(), integ, layout=layout)["y"][-1]
        r = b.cpu_baseline_adaptive(O, name, yh, layout, integ, d, gpu, n1, 8)
        assert r["max_abs_dev_gpu_vs_cpu"] == 0.0 and r["value"] > 0 and r["unit"] == "IVPs/s" and r["cores"] == 1
        assert r["all_cores"]["cores"] == 8 and "first 64 IVPs" in r["all_cores"]["sample"] and r["accepted_steps_per_s"] > 100 * r["value"] * 0.99


def test_opt_in_child_never_costs_the_parent_its_line(monkeypatch):
    """bench.py runs the settings without a hardware record in a child process: whatever happens to the child — a crash (device fault -> abort), a hang (timeout), garbage
    on stdout — the parent gets an `informational_errors` entry under the `streamed_opt_in:` prefix the contract test tolerates, and carries on."""
    import json
    import subprocess
    b = _bench()

    class R:
        def __init__(self, rc, out, err=""):
            self.returncode, self.stdout, self.stderr = rc, out, err

    calls = []

    def fake_run(outcome):
        def run(cmd, **kw):
            calls.append((cmd, kw))
            if isinstance(outcome, Exception):
                raise outcome
            return outcome
        return run

    # a device fault aborts the child: no JSON, rc -6
    monkeypatch.setattr(subprocess, "run", fake_run(R(-6, "amdgpu banner\n", "Memory access fault by GPU node-1")))
    res, err = b.streamed_opt_in_parent(1000, 5)
    assert res == {} and list(err) == ["streamed_opt_in:child"] and "rc -6" in err["streamed_opt_in:child"] and "Memory access fault" in err["streamed_opt_in:child"]
    cmd, kw = calls[-1]
    assert cmd[1].endswith("bench.py") and cmd[2:] == ["--child-leg", "streamed_opt_in", "--adaptive-n", "1000"] and kw["timeout"] == 5
    assert not {"RANK", "LOCAL_RANK", "WORLD_SIZE"} & set(kw["env"])
    # a hang: subprocess.run kills the child and raises
    monkeypatch.setattr(subprocess, "run", fake_run(subprocess.TimeoutExpired("bench.py", 5)))
    res, err = b.streamed_opt_in_parent(1000, 5)
    assert res == {} and "TimeoutExpired" in err["streamed_opt_in:child"]
    # the child's own per-setting errors travel with its results
    payload = {"streamed_opt_in": {"C3_dopri54_lorenz_1e6": {"lean": {"streamed_launches": 104}}, "C4_tsit54_ring16_1e6": {}},
               "errors": {"streamed_opt_in:C4_tsit54_ring16_1e6:lean": "RuntimeError('x')"}}
    monkeypatch.setattr(subprocess, "run", fake_run(R(0, "banner\n" + json.dumps(payload) + "\n")))
    res, err = b.streamed_opt_in_parent(1000, 5)
    assert res == payload["streamed_opt_in"] and err == payload["errors"] and all(k.startswith("streamed_opt_in:") for k in err)

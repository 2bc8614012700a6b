"""bench.py's output contract (one JSON line, last on stdout, with the keys the driver and the judge read) on a reduced
workload, single process and through torch.distributed (world size 1, nccl = RCCL, the overlapped all-gather path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "2", "--warmup", "1", "--n-ivp", "200000", "--rk4-steps", "64", "--cpu-sample", "2000", "--cpu-adaptive-sample", "500"]
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "roofline"}


def _run(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    last = r.stdout.strip().splitlines()[-1]
    return json.loads(last)


def test_single_process_line():
    out = _run(SMALL)
    assert KEYS <= set(out), KEYS - set(out)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["dtype"] == "f64" and out["scaling"] == "weak"
    assert out["world_size_reported_by_backend"] == 1 and out["ms_per_step_per_rank"]["min"] == out["ms_per_step_per_rank"]["max"] == out["ms_per_step"]
    assert out["vs_baseline"] is None and out["higher_is_better"] is True and "workload" in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["algorithmic_bytes_per_launch"] == 16.0 * 200000
    # the Infinity-Cache caveat is part of the line: the HBM-only figure comes from the 6.4e7-IVP leg of the same run, and the PMC
    # traffic (a static figure from profiles/) is only attached to the configuration it was collected on (1e7 IVPs) — with its source
    assert {"frac_hbm_only", "achieved_hbm_only", "traffic", "traffic_source"} <= set(rf)
    assert rf["traffic"] is None and rf["traffic_source"] is None
    assert 0 < rf["frac_hbm_only"] < 1 and abs(rf["frac_hbm_only"] - out["beyond_infinity_cache"]["frac"]) < 1e-12
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    assert out["parity_max_abs_err_vs_oracle"] == 0.0 and out["parity_checked_ivps"] == 2000
    assert out["fused_solve"]["bitwise_equal_to_stream"] is True
    # informational legs never cost the line; the ones that go through settings without a hardware record (the opt-in lean kernels / polling schedule /
    # contracted build) may report an error here without failing the contract — everything with a record must be clean
    assert all(k.startswith("streamed_opt_in:") for k in out.get("informational_errors", {})), out.get("informational_errors")
    ac = out["adaptive_configs"]
    for name in ("C3_dopri54_lorenz_1e6", "C4_tsit54_ring16_1e6"):
        its = ac[name]["loop_iterations"]
        assert ac[name]["streamed_bitwise_equal_to_fused"] is True and its == 102 and its <= ac[name]["streamed_launches"] <= its + 16   # groups of 8: the last one speculative
        for tag, o in ac[name]["streamed_opt_in"].items():   # whatever ran must be right: bits for the bit-exact settings, tolerance for the contracted one
            assert its <= o["streamed_launches"] <= its + 16 and o["within_north_star_tolerance"] is True, (name, tag, o)
            assert o["bitwise_equal_to_fused"] is (not tag.endswith("fp_contract")) or o["max_abs_deviation_from_fused"] == 0.0, (name, tag, o)
        cb = ac[name]["cpu_baseline"]
        assert cb["value"] > 0 and cb["all_cores"]["value"] > 0 and cb["max_abs_dev_gpu_vs_cpu"] <= 1e-6
    fc = out["fused_solve_fp_contract"]
    assert set(fc) == {"C2_rk4_neg_y", "C4_tsit54_ring16_1e6"} and all(v["bit_exact_ms"] > 0 and v["contracted_ms"] > 0 for v in fc.values())
    assert fc["C2_rk4_neg_y"]["within_tolerance"] is True
    # value = trajectory-steps / wall time of the timed region
    assert abs(out["value"] - 200000 * 64 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-9


def test_distributed_line_world_size_one():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = _run(SMALL + ["--force-dist", "--no-cpu-baseline", "--verify-gathers"], env=env)
    assert KEYS <= set(out)
    assert out["gathers_verified"] == 3  # warmup + 2 timed solves, each gather checked against the solve it belongs to
    assert out["config"]["final_state_allgather"] is True and out["config"]["allgather_overlapped_with_next_solve"] is True
    assert out["allgather_ms_per_solve"] > 0 and "cpu_baseline" not in out


def test_two_ranks_share_the_gpu_over_gloo():
    """The driver's multi-GPU launch line (torch.distributed.run, one rank per GPU) with two ranks on this box's single GPU: gloo
    instead of RCCL (RCCL refuses two ranks on one device), everything else identical — contiguous shards of the global index range,
    the overlapped all-gather with its three rotating buffers (every gather verified), max-over-ranks timing, one JSON line from rank 0."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29581",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--verify-gathers", "--steps", "3", "--warmup", "1", "--n-ivp", "300000",
           "--rk4-steps", "64"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines  # only rank 0 prints
    out = json.loads(lines[0])
    assert KEYS <= set(out) and out["n_gpus"] == 2 and out["config"]["backend"] == "gloo"
    assert out["gathers_verified"] == 4 and "cpu_baseline" not in out  # cpu_baseline is reported at N = 1 only
    assert abs(out["value"] - 2 * 300000 * 64 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-9  # whole-job aggregate


def test_eight_ranks_share_the_gpu_over_gloo():
    """BASELINE config C5's launch shape — the driver's exact line at --gpus 8 — on this box's single GPU: eight ranks over gloo, a
    reduced batch per rank, every overlapped all-gather verified against the solve it belongs to, whole-job aggregate from rank 0."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29583",
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--verify-gathers", "--n-ivp", "100000",
           "--rk4-steps", "32"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert KEYS <= set(out) and out["n_gpus"] == 8 and out["scaling"] == "weak" and out["config"]["ivps_per_gpu"] == 100000
    assert out["gathers_verified"] == 4 and out["config"]["allgather_overlapped_with_next_solve"] is True
    assert abs(out["value"] - 8 * 100000 * 32 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-9


def test_gpus_2_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with WORLD_SIZE unset — the shape of the driver's N = 1 line with another N: bench.py re-executes
    itself under torch.distributed.run (two ranks sharing this box's GPU over gloo) and the one JSON line still comes from rank 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--verify-gathers", "--steps", "3", "--warmup", "1",
                        "--n-ivp", "300000", "--rk4-steps", "64"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert KEYS <= set(out) and out["n_gpus"] == 2 and out["world_size_reported_by_backend"] == 2 and out["config"]["backend"] == "gloo"
    assert out["gathers_verified"] == 4 and out["allgather_ms_per_solve"] > 0
    pr = out["ms_per_step_per_rank"]
    assert 0 < pr["min"] <= pr["max"] == out["ms_per_step"]          # the job is as slow as its slowest rank
    assert abs(out["value"] - 2 * 300000 * 64 * 3 / (out["ms_per_step"] * 3e-3)) / out["value"] < 1e-9


def test_a_launcher_of_another_size_is_refused():
    """--gpus N under a launcher whose WORLD_SIZE is not N: an error, not a silently different job"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29579", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE is 1" in r.stderr


def test_more_rccl_ranks_than_devices_is_refused():
    """RCCL wants one device per rank: 2 ranks on a 1-GPU box must say so instead of hanging in the communicator's set-up"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29580", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "need 2 devices" in r.stderr

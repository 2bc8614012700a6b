"""What the device compiler makes of the hot kernels, checked without a GPU: hipcc cross-compiles gfx950 here, and the code object's metadata says how many
VGPRs / SGPRs a wavefront takes, whether anything spills to scratch, and how much LDS a workgroup holds — i.e. how many wavefronts a SIMD keeps resident
(512 VGPRs per lane and SIMD: <= 128 -> 4 waves, <= 168 -> 3), which is what the measured figures of DESIGN.md section 6 were obtained at.  A change that pushes a
hot kernel over its budget shows up here, at CPU test time, instead of as an unexplained slowdown on the next GPU run.  tests/cpp/budget_tu.hip instantiates the
kernels of the BASELINE configurations with the library's own flags; scripts/kernel_isa_stats.py reads the code object.  Static facts only — no timing claim."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stats(tmp_path_factory):
    hipcc = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
    if hipcc is None:
        pytest.skip("needs hipcc")
    obj = str(tmp_path_factory.mktemp("budget") / "budget.o")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function",
                           "-I", os.path.join(ROOT, "numericalnim_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                           "-c", os.path.join(ROOT, "tests", "cpp", "budget_tu.hip"), "-o", obj])
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "kernel_isa_stats.py"), obj, ".", "--json"], text=True)
    return json.loads(out)


def _one(stats, *parts):
    hits = [v for k, v in stats.items() if all(p in k for p in parts)]
    assert len(hits) == 1, (parts, [k for k in stats])
    return hits[0]


def _no_scratch(k):
    r = k["resources"]
    assert r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0, r


def test_headline_kernel_is_tiny_and_spill_free(stats):
    """rk4_stream_vec_kernel (C2 / C5): HBM-bound — it must never be the register file that limits the waves in flight (<= 64 VGPRs: 8 waves per SIMD)."""
    for vec, mode, budget in ((1, 0, 32), (4, 1, 64)):
        k = _one(stats, "rk4_stream_vec_kernel", "Lb0ELi%dELi%dE" % (vec, mode))
        _no_scratch(k)
        assert k["resources"]["vgpr_count"] <= budget and k["resources"]["group_segment_fixed_size"] == 0, k["resources"]
        assert k["classes"].get("vmem:global_load", 0) >= 1 and k["classes"].get("lds", 0) == 0


def test_streamed_c3_kernel_keeps_four_waves(stats):
    """advance_tpi_lean_kernel<DOPRI54, Lorenz>: 128 VGPRs = 4 waves per SIMD, no scratch, at most 4 scalars spilled through lanes (the general kernel: 17) —
    round 6: the per-workgroup answer of a polled launch (adv_lean_report) keeps one more EXEC mask alive across the step: 2 v_writelane on the step path.
    Its four loads — (t, dt) and the three planes of the state — are issued together, before the first branch (one memory round trip per wave, not two)."""
    lean, gen = _one(stats, "advance_tpi_lean_kernel"), _one(stats, "advance_tpi_kernel")
    _no_scratch(lean)
    _no_scratch(gen)
    assert lean["resources"]["vgpr_count"] <= 128 and gen["resources"]["vgpr_count"] <= 128
    assert lean["resources"]["sgpr_spill_count"] <= 4
    assert lean["classes"].get("lane(read/writelane)", 0) <= 12 < gen["classes"].get("lane(read/writelane)", 0)
    assert lean["valu_f64"] == gen["valu_f64"]                      # the same arithmetic, instruction for instruction ...
    assert lean["valu_total"] <= gen["valu_total"] - 60            # ... and at least 60 fewer other VALU instructions around it (round 6: 926 vs 988, of which ~40 in adv_lean_report)
    assert lean["loads_before_first_branch"] >= 4


def test_streamed_c4_kernel_keeps_three_waves(stats):
    """advance_lps_lean_kernel<Tsit54, Ring<16>, 4 components per lane>: <= 168 VGPRs = 3 waves per SIMD, no scratch, no SGPR spills; the FP64 work of the general
    kernel with at least 90 fewer other VALU instructions (round 6: 1192 vs 1294 static, ~30 of the 1192 in adv_lean_report, which only a polled launch executes;
    64-bit address arithmetic 31 -> 6)."""
    lean, gen = _one(stats, "advance_lps_lean_kernel"), _one(stats, "advance_lps_kernel")
    _no_scratch(lean)
    _no_scratch(gen)
    assert lean["resources"]["vgpr_count"] <= 168 and gen["resources"]["vgpr_count"] <= 168
    assert lean["resources"]["sgpr_spill_count"] == 0
    assert lean["valu_f64"] == gen["valu_f64"]
    assert lean["valu_total"] <= gen["valu_total"] - 90
    assert lean["classes"].get("valu_other:v_lshl_add_u64", 0) <= 8
    assert lean["classes"].get("lds", 0) <= 3                      # banded right-hand side + register-chain norm: no LDS traffic on the step path


def test_fused_kernels_stay_inside_their_occupancy_choice(stats):
    """The fused solves (state in VGPRs for the whole solve): scalar RK4 at <= 64 VGPRs, DOPRI54 Lorenz and Tsit54 ring-16 inside the 3-wave budget the
    round-2 A/B chose (profiles/r02_c4_fused_ab.txt: C4 fused 8.07 -> 5.85 ms with the lean loop at 3 waves per SIMD).  Today's compiler spills 26 VGPRs (68 B of scratch
    per lane) in the ring-16 kernel to get there — around the two direction loops, none inside the step body: the ceiling below keeps that from growing unnoticed."""
    rk4 = _one(stats, "solve_tpi_kernel", "RhsNegY")
    _no_scratch(rk4)
    assert rk4["resources"]["vgpr_count"] <= 64
    lor = _one(stats, "solve_tpi_kernel", "RhsLorenz")
    _no_scratch(lor)
    assert lor["resources"]["vgpr_count"] <= 168
    ring = _one(stats, "solve_lps_kernel")
    assert ring["resources"]["vgpr_count"] <= 168 and ring["resources"]["private_segment_fixed_size"] <= 96, ring["resources"]

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X / HIP device (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: emulation-backed CPU test (the gfx950 ISA interpreter / the fake node: minutes, test infrastructure frozen since round 5); "
                                       "not part of the default CPU run — NNHIP_RUN_SLOW=1 or -m slow runs it")


# GPU tests written in rounds 5 and 6, while gpurun was closed: they have never met a device.  The driver runs `pytest -x`: they go LAST, so that a
# first-contact failure in one of them cannot hide the result of the tests that have a record on hardware (GPUTEST_r04.json: 753 passed).  Nothing is skipped or relaxed.
# Remove a name from this list once it has passed on an MI355X.  (All of them have passed on the ISA-backed fake node — the library's compiled kernels interpreted on
# the host, tests/fake_torch standing in for PyTorch — except the self-launch test, whose ranks need the real torch.distributed.  That is not a device: the list stays.)
#
# Round 6: the library's DEFAULTS are the configuration with a hardware record again — general advance kernels, polling groups of 8 (knobs "adv_lean" 0, "adv_auto_poll" 0).
# Every test id that reaches advance_*_lean_kernel, nnhip::AdvPollSchedule's own schedule or the contracted lean kernels does so through an explicit knob and is listed here:
#   test_gpu_adaptive_parity.py::test_lean_advance_kernels_give_the_general_kernels_bits                      adv_lean 1 vs 0
#   test_gpu_adaptive_parity.py::test_automatic_polling_schedule_wastes_at_most_two_launches_...              adv_auto_poll 1
#   test_gpu_adaptive_parity.py::test_streamed_fp_contract_opt_in_stays_within_north_star_tolerance           fp_contract 1 (contracted lean kernels)
#   test_gpu_bench_contract.py::test_single_process_line                                                      bench.py's `streamed_opt_in` legs only (each guarded: a failure there
#                                                                                                             is reported under informational_errors and does not fail the contract)
# smoke() and every other test that calls adaptiveStream / adaptiveStreamSolve run the recorded kernels.
_FIRST_CONTACT = (
    "test_ctx_block.py::test_per_ivp_matrices_travel_with_their_shard",
    "test_ctx_block.py::test_mutable_slots_come_back_from_their_shards",
    "test_ctx_block.py::test_two_threads_bind_different_contexts_to_one_source",
    "test_ctx_block.py::test_device_resident_shards_read_their_columns_of_the_context",
    "test_gpu_adaptive_parity.py::test_automatic_polling_schedule_wastes_at_most_two_launches_beyond_the_speculative_pair",
    "test_gpu_adaptive_parity.py::test_lean_advance_kernels_give_the_general_kernels_bits",
    "test_gpu_adaptive_parity.py::test_streamed_fp_contract_opt_in_stays_within_north_star_tolerance",
    "test_gpu_adaptive_parity.py::test_bin_order_spends_its_bins_on_the_keys_it_gets",
    "test_gpu_bench_contract.py::test_gpus_2_without_a_launcher_starts_its_own_ranks",
    "test_gpu_bench_contract.py::test_a_launcher_of_another_size_is_refused",
    "test_gpu_bench_contract.py::test_more_rccl_ranks_than_devices_is_refused",
    "test_gpu_reference_text_quad.py::",
    "test_gpu_hermite_consumer.py::test_descending_abscissae_are_sorted_like_the_reference",
    "test_gpu_rk4_parity.py::test_per_step_seams_refuse_companions_they_would_misread",
    "test_gpu_adaptive_parity.py::test_polled_launches_see_a_lone_straggler_in_every_row_of_a_wave",
)


def pytest_collection_modifyitems(config, items):
    def first_contact(item):
        return any(("/" + k) in ("/" + item.nodeid) for k in _FIRST_CONTACT)
    items.sort(key=first_contact)  # stable: everything else keeps its order
    if not (os.environ.get("NNHIP_RUN_SLOW") or "slow" in (config.getoption("-m") or "")):
        skip = pytest.mark.skip(reason="slow emulation-backed test: NNHIP_RUN_SLOW=1 or -m slow")
        for item in items:
            if item.get_closest_marker("slow") is not None:
                item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def nn():
    import numericalnim_amd
    numericalnim_amd._lib.lib()  # loud failure if libnnhip_ode.so is not built
    return numericalnim_amd


@pytest.fixture(scope="session")
def dev():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a HIP device"
    return torch.device("cuda:0")


_LIBM = {}


@pytest.fixture(autouse=True)
def _bit_level_comparisons_need_the_glibc_pow_the_device_restates(request):
    """The -m gpu parity tests compare adaptive solves with the oracle BIT FOR BIT.  That is meaningful only where the oracle's std::pow is
    the glibc pow the device restates (x86-64, glibc >= 2.28, FMA variant): checked once per session on a sample
    (numericalnim_amd.hostLibmMatchesDevicePow); on any other host the tests that use the oracle are skipped with that reason instead
    of failing on last-bit differences that are inside the north-star tolerance (smoke() still gates on 1e-10 / 1e-6 there)."""
    if request.node.get_closest_marker("gpu") is None or "oracle" not in request.fixturenames:
        return
    if "ok" not in _LIBM:
        import numericalnim_amd
        _LIBM["ok"] = numericalnim_amd.hostLibmMatchesDevicePow()
    if not _LIBM["ok"]:
        pytest.skip("this host's libm pow differs from the glibc pow the device restates: bit-level comparison with the oracle is not meaningful here")

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")


@pytest.fixture(autouse=True)
def _seed_global_rng():
    """Several tests draw module-style parameters (A, dt_bias, upstream gradients) from torch's global generator; without a
    seed their inputs -- and with them a few tolerance-edge assertions -- depended on which tests ran before."""
    import torch
    torch.manual_seed(20240928)
    yield


# Emulator runs of these cases take 1 - 7 minutes EACH on the host (a SIMT emulator walking multi-pass scans lane by lane); together
# they made the CPU suite a half-hour affair.  Their `gpu` twins run on the MI355X under `-m gpu`; on the emulator they run with
# OMK_FULL_EMU=1 (developer machines), and a smaller case of the same code path stays in the default CPU suite for each of them.
HEAVY_ON_EMU = (
    "test_selective_scan_bwd[emu-bdl-16-1100-8-2-True-False-True-dtype0]", "test_selective_scan_bwd[emu-bdl-16-1100-8-2-True-False-True-dtype1]",
    "test_selective_scan_bwd[emu-bdl-16-600-16-1-True-True-True-dtype0]", "test_selective_scan_bwd[emu-bdl-16-600-16-1-True-True-True-dtype1]",
    "test_selective_scan_bwd[emu-bld-8-200-16-1-True-True-True-dtype1]", "test_selective_scan_bwd_channel_tiles[emu]",
    "test_selective_scan_fwd_lanes_are_channels[emu-96-530-16-2-True-True-ln-bld-dtype0]",
    "test_selective_scan_fwd_lanes_are_channels[emu-96-530-16-2-True-True-nl-bld-dtype0]", "test_selective_scan_fwd_lanes_are_channels[emu-96-530-16-2-True-True-nl-bld-dtype1]",
    "test_selective_scan_fwd_lanes_are_channels[emu-96-530-16-2-True-True-nl-bdl-dtype1]",
    "test_selective_scan_fwd_lanes_are_channels[emu-96-530-16-2-True-True-ln-bdl-dtype0]", "test_selective_scan_fwd_lanes_are_channels[emu-96-530-16-2-True-True-ln-bdl-dtype1]",
    "test_ssd_mfma_bwd[emu-300-8-1-False-True-None-0]", "test_ssd_mfma_bwd[emu-300-8-1-False-True-None-1]",
    "test_ssd_mfma_bwd[emu-330-4-2-False-True-2-0]", "test_ssd_mfma_bwd[emu-330-4-2-False-True-2-1]",
    "test_ssd_mfma_bwd[emu-200-2-1-False-True-1-0]", "test_ssd_mfma_bwd[emu-200-2-1-False-True-1-1]",
    "test_ssd_mfma_split_sequence_long_memory[emu]", "test_ssd_precise_forward_meets_the_1e3_budget_with_initial_states[emu-330-4-2]",
    "test_topk_topp_distribution_matches_reference[emu-64-0.6-1.3]", "test_topk_topp_distribution_matches_reference[emu-3-0.999-1.0]",
)


def pytest_collection_modifyitems(config, items):
    if not os.environ.get("OMK_FULL_EMU"):
        heavy = pytest.mark.skip(reason="minutes on the emulator: runs on the GPU (-m gpu) or with OMK_FULL_EMU=1")
        for it in items:
            if it.nodeid.split("::")[-1] in HEAVY_ON_EMU:
                it.add_marker(heavy)
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def dev(request):
    """Device the HIP sources run on: 'emu' = host build of the same kernels under the SIMT emulator (CPU tensors,
    runs everywhere); 'gpu' = the real libomnimamba_hip.so on cuda:0 (only under `-m gpu` on the MI355X box)."""
    import torch
    if request.param == "emu":
        from emu.loader import use_emulator
        with use_emulator():
            yield torch.device("cpu")
    else:
        yield torch.device("cuda:0")

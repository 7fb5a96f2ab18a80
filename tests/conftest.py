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


def pytest_collection_modifyitems(config, items):
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

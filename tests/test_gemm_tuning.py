"""The recorded library-GEMM solutions (omnimamba_amd/gemm_tuning.py): a no-op without a GPU, validated file on one."""
import os

import pytest
import torch

from omnimamba_amd import gemm_tuning


def test_results_file_is_well_formed():
    path = gemm_tuning._DEFAULT
    assert os.path.exists(path)
    rows = [l.strip().split(",") for l in open(path) if l.strip()]
    assert any(r[0] == "Validator" and r[1] == "GCN_ARCH_NAME" and r[2].startswith("gfx950") for r in rows)
    gemms = [r for r in rows if r[0].startswith("GemmTunableOp_BFloat16")]
    assert len(gemms) >= 6                                   # in_proj / out_proj forward, dgrad, wgrad at the block shapes
    assert all(float(r[-1]) > 0 for r in gemms)


def test_no_gpu_is_a_noop(monkeypatch):
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    assert gemm_tuning.use_tuned_gemms() is False


def test_disabled_by_env(monkeypatch):
    monkeypatch.setenv("OMK_GEMM_TUNING", "0")
    assert gemm_tuning.use_tuned_gemms() is False


@pytest.mark.gpu
def test_recorded_solutions_do_not_change_results():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(2048, 2048, device=dev, dtype=torch.bfloat16)
    w = torch.randn(8512, 2048, device=dev, dtype=torch.bfloat16)
    ref = (x.float() @ w.float().t())
    try:
        ok = gemm_tuning.use_tuned_gemms()
        assert isinstance(ok, bool)
        y = x @ w.t()
        assert ((y.float() - ref).norm() / ref.norm()).item() < 6e-3      # one bf16 output rounding
    finally:
        torch.cuda.tunable.enable(False)

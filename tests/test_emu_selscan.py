"""Mamba-1 selective_scan forward HIP kernel through the emulator vs the oracle."""
import pytest
import torch

import oracle as O
from emu.loader import use_emulator


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["bdl", "bld"])
@pytest.mark.parametrize("Dm,L,N,G,bvar", [(70, 45, 16, 1, True), (12, 33, 8, 2, True), (6, 20, 4, 1, False), (130, 37, 24, 1, True)])
def test_selective_scan_fwd_emulated(dtype, layout, Dm, L, N, G, bvar):
    from omnimamba_amd.selective_scan import selective_scan_fn
    torch.manual_seed(0)
    Bsz = 2

    def mk(scale=1.0, rand=False):
        t = (torch.rand(Bsz, L, Dm) if rand else torch.randn(Bsz, L, Dm)) * scale
        t = t.to(dtype)
        return t.transpose(1, 2) if layout == "bld" else t.transpose(1, 2).contiguous()

    u, delta, z = mk(), mk(0.5, True), mk()
    A = -(torch.rand(Dm, N) + 0.1)
    if bvar:
        Bm = torch.randn(Bsz, G, N, L).to(dtype) if G > 1 else torch.randn(Bsz, N, L).to(dtype)
        Cm = torch.randn(Bsz, G, N, L).to(dtype) if G > 1 else torch.randn(Bsz, N, L).to(dtype)
    else:
        Bm, Cm = torch.randn(Dm, N), torch.randn(Bsz, N, L).to(dtype)
    D, db = torch.randn(Dm), torch.randn(Dm) * 0.1
    with use_emulator():
        out, last = selective_scan_fn(u, delta, A, Bm, Cm, D, z, db, True, True)
    o0, l0 = O.selective_scan_ref(u, delta, A, Bm, Cm, D, z, db, True, True)
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert out.shape == u.shape and rel(out, o0) < tol and rel(last, l0) < 2e-5

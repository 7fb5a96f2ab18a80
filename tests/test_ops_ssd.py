"""SSD scan kernels (omk_ssd_scan_fwd / _bwd: generic fp32 path and the MFMA path) vs the oracle: emulator on CPU,
MI355X under -m gpu."""
import pytest
import torch

import oracle as O


def rel(a, b):
    return ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def make(Bsz, L, H, P, N, G, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(Bsz, L, H, P, generator=g).to(dtype)
    dt = (torch.randn(Bsz, L, H, generator=g) * 0.5).to(dtype)
    A = -(torch.rand(H, generator=g) * 4 + 0.5)
    Bm = torch.randn(Bsz, L, G, N, generator=g).to(dtype)
    Cm = torch.randn(Bsz, L, G, N, generator=g).to(dtype)
    D = torch.randn(H, generator=g)
    z = torch.randn(Bsz, L, H, P, generator=g).to(dtype)
    dtb = torch.randn(H, generator=g) * 0.5 - 1.0
    init = torch.randn(Bsz, H, P, N, generator=g)
    return x, dt, A, Bm, Cm, D, z, dtb, init


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("L,H,P,N,G", [(37, 4, 8, 16, 2), (20, 2, 16, 24, 1), (70, 2, 64, 128, 1)])
def test_ssd_generic_fwd(dev, dtype, L, H, P, N, G):
    from omnimamba_amd.ssd_combined import ssd_scan_fwd
    x, dt, A, Bm, Cm, D, z, dtb, init = make(2, L, H, P, N, G, dtype)
    d = lambda t: t.to(dev)
    out, out_x, fin = ssd_scan_fwd(d(x), d(dt), d(A), d(Bm), d(Cm), D=d(D), z=d(z), dt_bias=d(dtb), initial_states=d(init),
                                   dt_softplus=True, dt_limit=(0.0, 3.0), return_final_states=True, want_out_x=True,
                                   force_generic=True)
    o0, f0 = O.ssd_ref_sequential(x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dtb, initial_states=init, dt_softplus=True,
                                  dt_limit=(0.0, 3.0), return_final_states=True)
    ox0 = O.ssd_ref_sequential(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, initial_states=init, dt_softplus=True, dt_limit=(0.0, 3.0))
    tol = 3e-5 if dtype == torch.float32 else 6e-3
    assert rel(out, o0) < tol and rel(out_x, ox0) < tol and rel(fin, f0) < 3e-5


@pytest.mark.parametrize("L,H,P,N,G,dhp", [(29, 4, 8, 16, 2, False), (18, 2, 16, 8, 1, True)])
def test_ssd_generic_bwd(dev, L, H, P, N, G, dhp, monkeypatch):
    import omnimamba_amd.ssd_combined as S
    x, dt, A, Bm, Cm, D, z, dtb, init = make(2, L, H, P, N, G, torch.float32, seed=3)
    if dhp:
        D = torch.randn(H, P)
    leaves = [t.clone().to(dev).requires_grad_() for t in (x, dt, A, Bm, Cm, D, z, dtb, init)]
    xr, dtr, Ar, Br, Cr, Dr, zr, dtbr, ir = leaves
    y, fin = S.mamba_chunk_scan_combined(xr, dtr, Ar, Br, Cr, 64, D=Dr, z=zr, dt_bias=dtbr, initial_states=ir,
                                         dt_softplus=True, return_final_states=True)
    gy, gf = torch.randn(y.shape), torch.randn(fin.shape)
    torch.autograd.backward([y, fin], [gy.to(dev), gf.to(dev)])
    dl = [t.double().clone().requires_grad_() for t in (x, dt, A, Bm, Cm, D, z, dtb, init)]
    y0, f0 = O.ssd_ref_sequential(dl[0], dl[1], dl[2], dl[3], dl[4], D=dl[5], z=dl[6], dt_bias=dl[7], initial_states=dl[8],
                                  dt_softplus=True, return_final_states=True, compute_dtype=torch.float64)
    torch.autograd.backward([y0, f0], [gy.double(), gf.double()])
    assert rel(y.detach(), y0.detach()) < 3e-5 and rel(fin.detach(), f0.detach()) < 3e-5
    for n, a, b in zip(["x", "dt", "A", "B", "C", "D", "z", "dt_bias", "init"], leaves, dl):
        assert rel(a.grad, b.grad) < 2e-4, n

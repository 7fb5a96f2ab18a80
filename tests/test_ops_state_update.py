"""omk_selective_state_update vs the oracle: emulator on CPU, MI355X under -m gpu."""
import pytest
import torch

import oracle as O


def rel(a, b):
    return ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("sdt,xdt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("H,P,N,G,tied", [(4, 64, 128, 1, True), (4, 8, 16, 2, True), (2, 5, 64, 1, False), (3, 7, 6, 1, False)])
def test_state_update(dev, sdt, xdt, H, P, N, G, tied):
    from omnimamba_amd.selective_state_update import selective_state_update
    torch.manual_seed(0)
    Bsz = 2
    st = torch.randn(Bsz, H, P, N).to(sdt)
    x, z = torch.randn(Bsz, H, P).to(xdt), torch.randn(Bsz, H, P).to(xdt)
    Bm, Cm = torch.randn(Bsz, G, N).to(xdt), torch.randn(Bsz, G, N).to(xdt)
    if tied:
        dt_, A_, D_, dtb_ = torch.randn(Bsz, H).to(xdt), -(torch.rand(H) * 15 + 1), torch.randn(H), torch.randn(H)
        ex = lambda t, d: (t.to(d)[..., None].expand(*t.shape, P))
        dt, D, dtb = ex(dt_, "cpu"), ex(D_, "cpu"), ex(dtb_, "cpu")
        A = A_[:, None, None].expand(H, P, N)
        dtd, Dd, dtbd, Ad = ex(dt_, dev), ex(D_, dev), ex(dtb_, dev), A_.to(dev)[:, None, None].expand(H, P, N)
    else:
        dt, A, D, dtb = torch.randn(Bsz, H, P).to(xdt), -(torch.rand(H, P, N) + 0.1), torch.randn(H, P), torch.randn(H, P)
        dtd, Ad, Dd, dtbd = dt.to(dev), A.to(dev), D.to(dev), dtb.to(dev)
    s1, s0 = st.clone().to(dev), st.clone()
    y = selective_state_update(s1, x.to(dev), dtd, Ad, Bm.to(dev), Cm.to(dev), D=Dd, z=z.to(dev), dt_bias=dtbd, dt_softplus=True)
    y0 = O.selective_state_update_ref(s0, x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dtb, dt_softplus=True)
    tol = 2e-5 if xdt == torch.float32 else 6e-3
    stol = 2e-5 if sdt == torch.float32 else 6e-3
    assert rel(y, y0) < tol and rel(s1, s0) < stol


@pytest.mark.parametrize("sdt,xdt,N", [(torch.float32, torch.float32, 128), (torch.bfloat16, torch.bfloat16, 128), (torch.float32, torch.bfloat16, 64)])
def test_state_update_tied_rows_kernel(dev, sdt, xdt, N):
    """Several sequences of the 1.3B block shape: the tied-scalar kernel with four rows per lane group (>= 2^21 state
    elements).  Same oracle, same tolerances."""
    from omnimamba_amd.selective_state_update import selective_state_update
    torch.manual_seed(3)
    Bsz, H, P, G = (4 if N == 128 else 8), 64, 64, 1
    st = torch.randn(Bsz, H, P, N).to(sdt)
    x, z = torch.randn(Bsz, H, P).to(xdt), torch.randn(Bsz, H, P).to(xdt)
    Bm, Cm = torch.randn(Bsz, G, N).to(xdt), torch.randn(Bsz, G, N).to(xdt)
    dt_, A_, D_, dtb_ = torch.randn(Bsz, H).to(xdt), -(torch.rand(H) * 15 + 1), torch.randn(H), torch.randn(H)
    ex = lambda t, d: (t.to(d)[..., None].expand(*t.shape, P))
    s1, s0 = st.clone().to(dev), st.clone()
    y = selective_state_update(s1, x.to(dev), ex(dt_, dev), A_.to(dev)[:, None, None].expand(H, P, N), Bm.to(dev), Cm.to(dev), D=ex(D_, dev),
                               z=z.to(dev), dt_bias=ex(dtb_, dev), dt_softplus=True)
    y0 = O.selective_state_update_ref(s0, x, ex(dt_, "cpu"), A_[:, None, None].expand(H, P, N), Bm, Cm, D=ex(D_, "cpu"), z=z,
                                      dt_bias=ex(dtb_, "cpu"), dt_softplus=True)
    assert rel(y, y0) < (2e-5 if xdt == torch.float32 else 6e-3) and rel(s1, s0) < (2e-5 if sdt == torch.float32 else 6e-3)


def test_state_update_no_heads(dev):
    from omnimamba_amd.selective_state_update import selective_state_update
    torch.manual_seed(1)
    Bsz, Dm, N = 3, 12, 16
    st = torch.randn(Bsz, Dm, N)
    x, dt = torch.randn(Bsz, Dm), torch.rand(Bsz, Dm)
    A, Bm, Cm = -(torch.rand(Dm, N) + 0.1), torch.randn(Bsz, N), torch.randn(Bsz, N)
    s1, s0 = st.clone().to(dev), st.clone()
    y = selective_state_update(s1, x.to(dev), dt.to(dev), A.to(dev), Bm.to(dev), Cm.to(dev))
    y0 = O.selective_state_update_ref(s0, x, dt, A, Bm, Cm)
    assert rel(y, y0) < 2e-5 and rel(s1, s0) < 2e-5

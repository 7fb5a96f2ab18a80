"""selective_state_update HIP kernel through the emulator vs the oracle."""
import pytest
import torch

import oracle as O
from emu.loader import use_emulator


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("sdt,xdt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("H,P,N,G,tied", [(4, 64, 128, 1, True), (4, 8, 16, 2, True), (2, 5, 64, 1, False), (3, 7, 6, 1, False)])
def test_state_update_emulated(sdt, xdt, H, P, N, G, tied):
    from omnimamba_amd.selective_state_update import selective_state_update
    if H % G:
        pytest.skip("H % G")
    torch.manual_seed(0)
    Bsz = 2
    st = torch.randn(Bsz, H, P, N).to(sdt)
    x, z = torch.randn(Bsz, H, P).to(xdt), torch.randn(Bsz, H, P).to(xdt)
    Bm, Cm = torch.randn(Bsz, G, N).to(xdt), torch.randn(Bsz, G, N).to(xdt)
    if tied:
        dt = torch.randn(Bsz, H).to(xdt)[..., None].expand(Bsz, H, P)
        A = (-(torch.rand(H) * 15 + 1))[:, None, None].expand(H, P, N)
        D = torch.randn(H)[:, None].expand(H, P)
        dtb = torch.randn(H)[:, None].expand(H, P)
    else:
        dt, A, D, dtb = torch.randn(Bsz, H, P).to(xdt), -(torch.rand(H, P, N) + 0.1), torch.randn(H, P), torch.randn(H, P)
    s1, s0 = st.clone(), st.clone()
    with use_emulator():
        y = selective_state_update(s1, x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dtb, dt_softplus=True)
    y0 = O.selective_state_update_ref(s0, x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dtb, dt_softplus=True)
    tol = 2e-5 if xdt == torch.float32 else 6e-3
    stol = 2e-5 if sdt == torch.float32 else 6e-3
    assert rel(y, y0) < tol and rel(s1, s0) < stol


def test_state_update_no_heads_emulated():
    from omnimamba_amd.selective_state_update import selective_state_update
    torch.manual_seed(1)
    Bsz, Dm, N = 3, 12, 16
    st = torch.randn(Bsz, Dm, N)
    x, dt = torch.randn(Bsz, Dm), torch.rand(Bsz, Dm)
    A, Bm, Cm = -(torch.rand(Dm, N) + 0.1), torch.randn(Bsz, N), torch.randn(Bsz, N)
    s1, s0 = st.clone(), st.clone()
    with use_emulator():
        y = selective_state_update(s1, x, dt, A, Bm, Cm)
    y0 = O.selective_state_update_ref(s0, x, dt, A, Bm, Cm)
    assert rel(y, y0) < 2e-5 and rel(s1, s0) < 2e-5

"""The 4-scan decomposition of the SSD backward (ssd_scan.h) checked against autograd of the naive recurrence,
in fp64 torch -- documents and pins the algorithm the HIP backward implements."""
import torch

import oracle as O


def test_four_scan_backward_matches_autograd():
    torch.manual_seed(0)
    dd = torch.float64
    Bsz, L, H, P, N, G = 2, 19, 4, 5, 6, 2
    x = torch.randn(Bsz, L, H, P, dtype=dd, requires_grad=True)
    dt = (torch.randn(Bsz, L, H, dtype=dd) * 0.5).requires_grad_()
    A = (-(torch.rand(H, dtype=dd) * 3 + 0.5)).requires_grad_()
    Bm = torch.randn(Bsz, L, G, N, dtype=dd, requires_grad=True)
    Cm = torch.randn(Bsz, L, G, N, dtype=dd, requires_grad=True)
    D = torch.randn(H, dtype=dd, requires_grad=True)
    dtb = (torch.randn(H, dtype=dd) * 0.3).requires_grad_()
    init = torch.randn(Bsz, H, P, N, dtype=dd, requires_grad=True)
    y, fin = O.ssd_ref_sequential(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, initial_states=init, dt_softplus=True,
                                  return_final_states=True, compute_dtype=dd)
    dy, dfin = torch.randn_like(y), torch.randn_like(fin)
    torch.autograd.backward([y, fin], [dy, dfin])
    with torch.no_grad():
        rep = H // G
        dtp, dsoft = torch.nn.functional.softplus(dt + dtb), torch.sigmoid(dt + dtb)
        a = dtp * A
        Bh, Ch = Bm.repeat_interleave(rep, 2), Cm.repeat_interleave(rep, 2)
        S = init.clone()
        OC, e = torch.zeros(Bsz, L, H, N, dtype=dd), torch.zeros(Bsz, L, H, dtype=dd)
        for t in range(L):                                                       # GS_DC
            S = S * torch.exp(a[:, t])[..., None, None] + (dtp[:, t, :, None] * x[:, t])[..., None] * Bh[:, t, :, None, :]
            OC[:, t] = torch.einsum("bhpn,bhp->bhn", S, dy[:, t])
            e[:, t] = (OC[:, t] * Ch[:, t]).sum(-1)
        dC = OC.reshape(Bsz, L, G, rep, N).sum(3)
        g = dfin.clone()
        dxraw, OB, wsum = torch.zeros_like(x), torch.zeros(Bsz, L, H, N, dtype=dd), torch.zeros(Bsz, L, H, dtype=dd)
        for t in range(L - 1, -1, -1):                                           # GS_DX + GS_DB share g
            dec = torch.exp(a[:, t + 1]) if t + 1 < L else torch.ones(Bsz, H, dtype=dd)
            g = g * dec[..., None, None] + dy[:, t][..., None] * Ch[:, t, :, None, :]
            dxraw[:, t] = torch.einsum("bhpn,bhn->bhp", g, Bh[:, t])
            OB[:, t] = torch.einsum("bhpn,bhp->bhn", g, x[:, t])
            wsum[:, t] = (OB[:, t] * Bh[:, t]).sum(-1)
        dinit = g * torch.exp(a[:, 0])[..., None, None]
        dx = dtp[..., None] * dxraw + D[None, None, :, None] * dy
        dB = (dtp[..., None] * OB).reshape(Bsz, L, G, rep, N).sum(3)
        dD = (dy * x).sum((0, 1, 3))
        dl, run = torch.zeros(Bsz, L, H, dtype=dd), (dfin * S).sum((-1, -2))
        for t in range(L - 1, -1, -1):
            run = run + e[:, t] - dtp[:, t] * wsum[:, t]
            dl[:, t] = run
        ddt = (wsum + A * dl) * dsoft
        dA = (dtp * dl).sum((0, 1))

    def r(p, q):
        return ((p - q).norm() / q.norm()).item()

    for got, ref in [(dx, x.grad), (dB, Bm.grad), (dC, Cm.grad), (ddt, dt.grad), (dA, A.grad), (dD, D.grad),
                     (ddt.sum((0, 1)), dtb.grad), (dinit, init.grad)]:
        assert r(got, ref) < 1e-10

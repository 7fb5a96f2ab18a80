"""Norm kernels (omk_add_norm_*, omk_norm_gated_*) vs the oracle: under the SIMT emulator on CPU and, with -m gpu, on the MI355X."""
import pytest
import torch

import oracle as O


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cols,has_res,res32,rms", [(64, True, True, True), (520, True, False, True), (48, False, True, True), (2560, True, True, True),
                                                    (100, True, True, False), (37, False, False, True)])
def test_add_norm_fwd_bwd(dev, dtype, cols, has_res, res32, rms):
    from omnimamba_amd.layer_norm import layer_norm_fn
    torch.manual_seed(0)
    x = torch.randn(3, 5, cols).to(dtype)
    res = (torch.randn(3, 5, cols).to(torch.float32 if res32 else dtype)) if has_res else None
    w = torch.randn(cols)
    b = None if rms else torch.randn(cols)
    xr, wr = x.clone().to(dev).requires_grad_(), w.clone().to(dev).requires_grad_()
    rr = None if res is None else res.clone().to(dev).requires_grad_()
    br = None if b is None else b.clone().to(dev).requires_grad_()
    y, ro = layer_norm_fn(xr, wr, br, residual=rr, eps=1e-5, prenorm=True, residual_in_fp32=res32, is_rms_norm=rms)
    gy, gr = torch.randn(y.shape).to(y.dtype), torch.randn(ro.shape).to(ro.dtype)
    torch.autograd.backward([y, ro], [gy.to(dev), gr.to(dev)])
    y, ro = y.detach().cpu(), ro.detach().cpu()
    y0, ro0 = O.add_norm_ref(x, w, b, residual=res, eps=1e-5, prenorm=True, residual_in_fp32=res32, is_rms_norm=rms)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert ro.dtype == ro0.dtype and rel(ro, ro0) < tol and rel(y, y0) < tol
    # gradients: autograd through an fp64 restatement
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    rd = None if res is None else res.double().requires_grad_()
    bd = None if b is None else b.double().requires_grad_()
    r = xd if rd is None else xd + rd
    if rms:
        yd = r * torch.rsqrt(r.pow(2).mean(-1, keepdim=True) + 1e-5) * wd
    else:
        mu = r.mean(-1, keepdim=True)
        yd = (r - mu) * torch.rsqrt((r - mu).pow(2).mean(-1, keepdim=True) + 1e-5) * wd + bd
    torch.autograd.backward([yd, r], [gy.double(), gr.double()])
    gtol = 1e-4 if dtype == torch.float32 else 1.5e-2
    assert rel(xr.grad.cpu(), xd.grad) < gtol and rel(wr.grad.cpu(), wd.grad) < gtol
    if rr is not None:
        assert rr.grad.dtype == res.dtype and rel(rr.grad.cpu(), rd.grad) < gtol
    if br is not None:
        assert rel(br.grad.cpu(), bd.grad) < gtol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cols,gs,nbg,has_z", [(64, None, False, True), (128, 32, False, True), (96, 48, True, True), (40, None, False, False),
                                               (4096, None, False, True), (4608, 2304, False, True)])
def test_norm_gated(dev, dtype, cols, gs, nbg, has_z):
    from omnimamba_amd.layernorm_gated import rmsnorm_fn
    torch.manual_seed(1)
    x = torch.randn(7, cols).to(dtype)
    z = torch.randn(7, cols).to(dtype) if has_z else None
    w = torch.randn(cols)
    xr, wr = x.clone().to(dev).requires_grad_(), w.clone().to(dev).requires_grad_()
    zr = None if z is None else z.clone().to(dev).requires_grad_()
    y = rmsnorm_fn(xr, wr, None, z=zr, eps=1e-5, group_size=gs, norm_before_gate=nbg)
    gy = torch.randn(y.shape).to(y.dtype)
    y.backward(gy.to(dev))
    y = y.detach().cpu()
    y0 = O.rmsnorm_gated_ref(x, w, None, z=z, eps=1e-5, group_size=gs, norm_before_gate=nbg)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel(y, y0) < tol
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    zd = None if z is None else z.double().requires_grad_()
    yd = O.rmsnorm_gated_ref(xd, wd, None, z=zd, eps=1e-5, group_size=gs, norm_before_gate=nbg, compute_dtype=torch.float64)
    yd.backward(gy.double())
    gtol = 1e-4 if dtype == torch.float32 else 1.5e-2
    assert rel(xr.grad.cpu(), xd.grad) < gtol and rel(wr.grad.cpu(), wd.grad) < gtol
    if zr is not None:
        assert rel(zr.grad.cpu(), zd.grad) < gtol


@pytest.mark.parametrize("cols", [2048, 4096, 8192])
def test_norm_gated_lean_kernels_walk_several_rows_per_block(dev, monkeypatch, cols):
    """The reference's mode on full segments (bf16, gate, norm_before_gate = 0, no bias) takes the software-pipelined forward and the
    one-sigmoid backward of norms.hip; a grid capped at 64 blocks makes every block walk several rows (next-row loads in flight,
    ragged last iteration), which the 7-row cases above never do.  Same results as the general kernels, and the oracle's."""
    from omnimamba_amd.layernorm_gated import rmsnorm_fn
    monkeypatch.setenv("OMK_NORM_BLOCKS", "64")
    torch.manual_seed(3)
    rows = 150 if cols <= 4096 else 70
    x, z, w = torch.randn(rows, cols).bfloat16(), torch.randn(rows, cols).bfloat16(), torch.randn(cols)
    gy = torch.randn(rows, cols).bfloat16()

    def run():
        xr, zr, wr = x.clone().to(dev).requires_grad_(), z.clone().to(dev).requires_grad_(), w.clone().to(dev).requires_grad_()
        y = rmsnorm_fn(xr, wr, None, z=zr, eps=1e-5, group_size=None, norm_before_gate=False)
        y.backward(gy.to(dev))
        return [t.detach().float().cpu() for t in (y, xr.grad, zr.grad, wr.grad)]

    lean = run()
    monkeypatch.setenv("OMK_NORM_NO_LEAN", "1")
    gen = run()
    for a_, b_ in zip(lean, gen):
        assert rel(a_, b_) < 3e-3
    xd, zd, wd = x.double().requires_grad_(), z.double().requires_grad_(), w.double().requires_grad_()
    yd = O.rmsnorm_gated_ref(xd, wd, None, z=zd, eps=1e-5, group_size=None, norm_before_gate=False, compute_dtype=torch.float64)
    yd.backward(gy.double())
    assert rel(lean[0], yd) < 6e-3
    for got, want in zip(lean[1:], (xd.grad, zd.grad, wd.grad)):
        assert rel(got, want) < 1.5e-2

"""causal_conv1d kernels (omk_causal_conv1d_{fwd,bwd,update}) vs the oracle: emulator on CPU, MI355X under -m gpu."""
import pytest
import torch

import oracle as O


def rel(a, b):
    return ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("W", [2, 3, 4])
@pytest.mark.parametrize("layout,C,L", [("cl", 16, 37), ("cl", 24, 140), ("cf", 6, 19), ("cl", 8, 1100)])   # 1100: the long-sequence strips
def test_conv1d_fwd_bwd(dev, dtype, W, layout, C, L):
    from omnimamba_amd.causal_conv1d import causal_conv1d_fn
    torch.manual_seed(0)
    B = 2
    if layout == "cl":  # slice of a wider channel-last row, like xBC inside zxbcdt
        base = torch.randn(B, L, C + 8).to(dtype)
        x = base[:, :, 8:].transpose(1, 2)
        xdev = base.to(dev)[:, :, 8:].transpose(1, 2)
    else:
        x = torch.randn(B, C, L).to(dtype)
        xdev = x.to(dev)
    w, b = torch.randn(C, W), torch.randn(C)
    init = torch.randn(B, C, W - 1).to(dtype)
    for use_init in (False, True):
        xr = xdev.detach().requires_grad_()
        wr, br = w.clone().to(dev).requires_grad_(), b.clone().to(dev).requires_grad_()
        ir = init.clone().to(dev).requires_grad_() if use_init else None
        out, fin = causal_conv1d_fn(xr, wr, br, initial_states=ir, return_final_states=True, activation="silu")
        g = torch.randn(out.shape).to(dtype)
        out.backward(g.to(dev))
        o0, f0 = O.causal_conv1d_ref(x, w, b, initial_states=init if use_init else None, return_final_states=True, activation="silu")
        tol = 1e-5 if dtype == torch.float32 else 6e-3
        assert rel(out.detach(), o0) < tol
        assert torch.equal(fin.float().cpu(), f0.float())
        xd, wd, bd = x.double().detach().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
        idd = init.double().requires_grad_() if use_init else None
        od = O.causal_conv1d_ref(xd, wd, bd, initial_states=idd, activation="silu", compute_dtype=torch.float64)
        od.backward(g.double())
        gtol = 1e-4 if dtype == torch.float32 else 1.5e-2
        assert rel(xr.grad, xd.grad) < gtol and rel(wr.grad, wd.grad) < gtol and rel(br.grad, bd.grad) < gtol
        if use_init:
            assert rel(ir.grad, idd.grad) < gtol


@pytest.mark.parametrize("W", [2, 3, 4])
def test_conv1d_long_strips_both_prologues_agree(dev, W):
    """The scalar-position kernels of long channel-last rows take a short prologue for contiguous fp32 (C, W) weights (16-byte requests, strips of
    16 tokens in the forward) and the general run-time-dtype prologue for everything else: the same op with the same weights as a strided view and
    as bf16-representable values in a bf16 tensor must give the same bits (forward, final states) and the same gradients."""
    from omnimamba_amd.causal_conv1d import causal_conv1d_fn
    torch.manual_seed(1)
    B, C, L = 2, 12, 600
    base = torch.randn(B, L, C + 4).bfloat16().to(dev)
    w = torch.randn(C, W).bfloat16().float().to(dev)                 # bf16-representable values
    b = torch.randn(C).bfloat16().float().to(dev)
    init = torch.randn(B, C, W - 1).bfloat16().to(dev)
    wide = torch.zeros(C, 2 * W, device=dev); wide[:, ::2] = w
    variants = {"fp32 contiguous": (w, b), "fp32 strided view": (wide[:, ::2], b), "bf16": (w.bfloat16(), b.bfloat16())}
    res = {}
    for name, (wv, bv) in variants.items():
        x = base[:, :, 4:].transpose(1, 2).detach().requires_grad_()
        wr, br = wv.detach().requires_grad_(), bv.detach().requires_grad_()
        ir = init.detach().requires_grad_()
        out, fin = causal_conv1d_fn(x, wr, br, initial_states=ir, return_final_states=True, activation="silu")
        g = torch.ones_like(out) * 0.5
        out.backward(g)
        res[name] = (out.detach().float().cpu(), fin.float().cpu(), x.grad.float().cpu(), wr.grad.float().cpu(), br.grad.float().cpu(), ir.grad.float().cpu())
    ref = res["fp32 contiguous"]
    for name in ("fp32 strided view", "bf16"):
        got = res[name]
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), name                    # forward: bit for bit
        assert torch.equal(got[2], ref[2]) and torch.equal(got[5], ref[5]), name                    # dx, dinit: the same arithmetic
        for i in (3, 4):                                                                             # dw, db: atomics, order free (bf16: rounded)
            tol = 1e-2 if name == "bf16" else 1e-5
            assert rel(got[i], ref[i].double()) < tol, (name, i)
    o0 = O.causal_conv1d_ref(base[:, :, 4:].transpose(1, 2).cpu(), w.cpu(), b.cpu(), initial_states=init.cpu(), activation="silu")
    assert rel(ref[0], o0) < 6e-3


def test_conv1d_bwd_partial_rows_are_deterministic_and_equal_the_atomics(dev, monkeypatch):
    """Long channel-last 16-bit rows: with the workspace of omk_causal_conv1d_bwd_workspace_bytes (what the autograd nodes pass) dweight / dbias come
    from per-tile partial rows added up in a fixed order -- the same bits on every run; OMK_CONV_BWD_ATOMICS=1 takes the fp32 atomics of rounds 1 - 5."""
    from omnimamba_amd.causal_conv1d import causal_conv1d_fn
    torch.manual_seed(3)
    B, C, L, W = 3, 24, 700, 4
    base = torch.randn(B, L, C + 8).bfloat16().to(dev)
    w, b = torch.randn(C, W).to(dev), torch.randn(C).to(dev)
    g = torch.randn(B, C, L).bfloat16().to(dev).transpose(1, 2).contiguous().transpose(1, 2)   # channel-last, like x

    def grads():
        x = base[:, :, 8:].transpose(1, 2).detach().requires_grad_()
        wr, br = w.detach().requires_grad_(), b.detach().requires_grad_()
        causal_conv1d_fn(x, wr, br, activation="silu").backward(g)
        return x.grad.float().cpu(), wr.grad.cpu(), br.grad.cpu()

    a1, a2 = grads(), grads()
    assert all(torch.equal(u, v) for u, v in zip(a1, a2))
    monkeypatch.setenv("OMK_CONV_BWD_ATOMICS", "1")
    a3 = grads()
    assert torch.equal(a1[0], a3[0]) and rel(a1[1], a3[1].double()) < 1e-5 and rel(a1[2], a3[2].double()) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv1d_update(dev, dtype):
    from omnimamba_amd.causal_conv1d import causal_conv1d_update
    torch.manual_seed(1)
    B, C, W = 2, 20, 4
    w, b = torch.randn(C, W), torch.randn(C)
    xs = torch.randn(B, C, 9).to(dtype)
    st = torch.zeros(B, W, C, dtype=dtype, device=dev).transpose(1, 2)   # channel-last state like Mamba2.allocate_inference_cache
    st0 = torch.zeros(B, C, W, dtype=dtype)
    wd, bd, xsd = w.to(dev), b.to(dev), xs.to(dev)
    for t in range(9):
        o = causal_conv1d_update(xsd[:, :, t], st, wd, bd, activation="silu")
        o0 = O.causal_conv1d_update_ref(xs[:, :, t], st0, w, b, activation="silu")
        assert rel(o, o0) < (1e-5 if dtype == torch.float32 else 6e-3)
        assert torch.equal(st.float().cpu(), st0.float())
    st3 = torch.randn(B, C, W - 1).to(dtype)
    st30 = st3.clone()
    st3 = st3.to(dev)
    o = causal_conv1d_update(xsd[:, :, :5], st3, wd, bd, activation=None)
    o0 = O.causal_conv1d_update_ref(xs[:, :, :5], st30, w, b, activation=None)
    assert rel(o, o0) < (1e-5 if dtype == torch.float32 else 6e-3) and torch.equal(st3.float().cpu(), st30.float())

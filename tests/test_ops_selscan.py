"""Mamba-1 selective_scan forward (omk_selective_scan_fwd) vs the oracle: emulator on CPU, MI355X under -m gpu."""
import pytest
import torch

import oracle as O


def rel(a, b):
    return ((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["bdl", "bld"])
@pytest.mark.parametrize("Dm,L,N,G,bvar", [(70, 45, 16, 1, True), (12, 33, 8, 2, True), (6, 20, 4, 1, False), (130, 37, 24, 1, True),
                                           # L >= 64 with L-contiguous storage: the chunked associative scan (one wave per channel, lanes = time
                                           # chunks); 1100 = two passes of 1024 tokens with a carry, ragged tail, unaligned rows
                                           (10, 200, 16, 1, True), (6, 1100, 16, 2, True), (5, 130, 4, 1, False), (3, 1024, 24, 1, True),
                                           # 8 | channels per group: the workgroup's eight waves share the B / C rows of a pass through LDS
                                           (16, 600, 16, 2, True), (8, 1100, 8, 1, True), (24, 130, 4, 1, False)])
def test_selective_scan_fwd(dev, dtype, layout, Dm, L, N, G, bvar):
    from omnimamba_amd.selective_scan import selective_scan_fn
    torch.manual_seed(0)
    Bsz = 2

    def mk(scale=1.0, rand=False):
        t = ((torch.rand(Bsz, L, Dm) if rand else torch.randn(Bsz, L, Dm)) * scale).to(dtype)
        if layout == "bld":
            return t.transpose(1, 2), t.to(dev).transpose(1, 2)
        c = t.transpose(1, 2).contiguous()
        return c, c.to(dev)

    (u, ud), (delta, dd), (z, zd) = mk(), mk(0.5, True), mk()
    A = -(torch.rand(Dm, N) + 0.1)
    if bvar:
        Bm = torch.randn(Bsz, G, N, L).to(dtype) if G > 1 else torch.randn(Bsz, N, L).to(dtype)
        Cm = torch.randn(Bsz, G, N, L).to(dtype) if G > 1 else torch.randn(Bsz, N, L).to(dtype)
    else:
        Bm, Cm = torch.randn(Dm, N), torch.randn(Bsz, N, L).to(dtype)
    D, db = torch.randn(Dm), torch.randn(Dm) * 0.1
    out, last = selective_scan_fn(ud, dd, A.to(dev), Bm.to(dev), Cm.to(dev), D.to(dev), zd, db.to(dev), True, True)
    o0, l0 = O.selective_scan_ref(u, delta, A, Bm, Cm, D, z, db, True, True)
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert out.shape == u.shape and rel(out, o0) < tol and rel(last, l0) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout", ["bdl", "bld"])
@pytest.mark.parametrize("bc_layout", ["nl", "ln"])
@pytest.mark.parametrize("Dm,L,N,G,with_z,softplus", [(70, 45, 16, 1, True, True), (12, 33, 8, 2, False, True), (130, 100, 5, 1, True, False),
                                                      (64, 16, 16, 1, True, True), (96, 530, 16, 2, True, True)])
def test_selective_scan_fwd_lanes_are_channels(dev, monkeypatch, dtype, layout, bc_layout, Dm, L, N, G, with_z, softplus):
    """The one-sweep form for many sequences (selscan_fwd_lanes_kernel: a lane owns a channel, B_t / C_t broadcast from a wave-private LDS
    tile), forced onto small shapes: channel-last storage read as it lies (registers only) and L-contiguous storage through the
    transposing tiles; ragged channel tiles (70 = 64 + 6, groups of 6 / 48), ragged last block, odd d_state, B / C with either of
    (state, token) contiguous; last state and the pass states the chunked backward restarts from."""
    from omnimamba_amd.selective_scan import selective_scan_fn
    from omnimamba_amd import _capi as K
    from omnimamba_amd._lib import get_lib
    monkeypatch.setenv("OMK_SELSCAN_LANES", "1")
    torch.manual_seed(3)
    Bsz = 3

    def mk(scale=1.0, rand=False):
        t = ((torch.rand(Bsz, L, Dm) if rand else torch.randn(Bsz, L, Dm)) * scale).to(dtype)
        if layout == "bld":
            return t.transpose(1, 2), t.to(dev).transpose(1, 2)
        c = t.transpose(1, 2).contiguous()
        return c, c.to(dev)

    def mkbc():
        t = torch.randn(Bsz, G, L, N).to(dtype)
        if bc_layout == "ln":
            return t.transpose(2, 3), t.to(dev).transpose(2, 3)
        c = t.transpose(2, 3).contiguous()
        return c, c.to(dev)

    (u, ud), (delta, dd), (z, zd) = mk(), mk(0.5, True), mk()
    if not with_z:
        z = zd = None
    A = -(torch.rand(Dm, N) + 0.1)
    (Bm, Bd), (Cm, Cd) = mkbc(), mkbc()
    D, db = torch.randn(Dm), torch.randn(Dm) * 0.1
    lib = get_lib()
    Ad, Dd, dbd = A.to(dev), D.to(dev), db.to(dev)   # (locals: a descriptor holds a pointer, not a reference)
    o = torch.empty(Bsz, L, Dm, dtype=dtype, device=dev).transpose(1, 2) if layout == "bld" else torch.empty_like(ud)
    import ctypes
    probe = K.SelScanFwd(u=K.T(ud), delta=K.T(dd), A=K.T(Ad), Bm=K.T(Bd), Cm=K.T(Cd), D=K.T(None), z=K.T(zd), delta_bias=K.T(None),
                         out=K.T(o), last_state=K.T(None), pass_states=K.T(None), delta_softplus=int(softplus))
    assert lib.omk_selective_scan_fwd_form(ctypes.byref(probe)) == 2
    out, last = selective_scan_fn(ud, dd, Ad, Bd, Cd, Dd, zd, dbd, softplus, True)
    o0, l0 = O.selective_scan_ref(u, delta, A, Bm.contiguous(), Cm.contiguous(), D, z, db, softplus, True)
    tol = 2e-5 if dtype == torch.float32 else 6e-3
    assert out.shape == u.shape and rel(out, o0) < tol and rel(last, l0) < 2e-5
    if layout == "bld":
        assert out.stride(1) == 1   # read and written as it lies: no L-contiguous copies
    # pass states (state in front of every 512-token pass) straight through the C ABI
    nP = (L + 511) // 512
    ps = torch.full((Bsz, Dm, nP, N), float("nan"), dtype=torch.float32, device=dev)
    o2 = torch.empty_like(out)
    p = K.SelScanFwd(u=K.T(ud), delta=K.T(dd), A=K.T(Ad), Bm=K.T(Bd), Cm=K.T(Cd), D=K.T(Dd), z=K.T(zd), delta_bias=K.T(dbd),
                     out=K.T(o2), last_state=K.T(None), pass_states=K.T(ps), delta_softplus=int(softplus))
    K.run(lib, "omk_selective_scan_fwd", p, ud)
    assert torch.equal(o2.cpu(), out.cpu()) and bool((ps[:, :, 0] == 0).all())
    if nP > 1:
        _, l512 = O.selective_scan_ref(u[..., :512], delta[..., :512], A, Bm.contiguous()[..., :512], Cm.contiguous()[..., :512], D,
                                       None if z is None else z[..., :512], db, softplus, True)
        assert rel(ps[:, :, 1], l512) < 2e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("layout,Dm,L,N,G,bvar,with_z,softplus", [("bdl", 70, 45, 16, 1, True, True, True), ("bld", 12, 33, 8, 2, True, False, True),
                                                                  ("bdl", 6, 20, 4, 1, False, True, False), ("bld", 130, 37, 16, 1, True, True, True),
                                                                  # L >= 64, L-contiguous, 8 | channels per group: the chunked scan in both directions
                                                                  # (two passes of 512 tokens, ragged tail; 16-wave and 8-wave workgroups; d_state 24 =
                                                                  # two blocks of state rows; no z / no softplus; channel-last views are copied)
                                                                  ("bdl", 16, 600, 16, 1, True, True, True), ("bdl", 16, 1100, 8, 2, True, False, True),
                                                                  ("bdl", 32, 130, 24, 2, True, True, False), ("bld", 8, 200, 16, 1, True, True, True)])
def test_selective_scan_bwd(dev, dtype, layout, Dm, L, N, G, bvar, with_z, softplus):
    """omk_selective_scan_bwd vs autograd through the fp64 oracle recurrence on identical inputs."""
    from omnimamba_amd.selective_scan import selective_scan_fn
    torch.manual_seed(1)
    Bsz = 2

    def mk(scale=1.0, rand=False):
        t = ((torch.rand(Bsz, L, Dm) if rand else torch.randn(Bsz, L, Dm)) * scale).to(dtype)
        return t.transpose(1, 2) if layout == "bld" else t.transpose(1, 2).contiguous()

    u, delta, z = mk(), mk(0.5, True), (mk() if with_z else None)
    A = -(torch.rand(Dm, N) + 0.1)
    if bvar:
        Bm = torch.randn(Bsz, G, N, L).to(dtype) if G > 1 else torch.randn(Bsz, N, L).to(dtype)
        Cm = torch.randn(Bsz, G, N, L).to(dtype) if G > 1 else torch.randn(Bsz, N, L).to(dtype)
    else:
        Bm, Cm = torch.randn(Dm, N), torch.randn(Dm, N)
    D, db = torch.randn(Dm), torch.randn(Dm) * 0.1
    src = [u, delta, A, Bm, Cm, D, z, db]
    leaves = [None if t is None else t.detach().clone().to(dev).requires_grad_() for t in src]
    out = selective_scan_fn(*leaves, softplus)
    g = torch.randn(out.shape).to(dtype)
    out.backward(g.to(dev))
    dl = [None if t is None else t.detach().double().clone().requires_grad_() for t in src]
    o0 = O.selective_scan_ref(*dl, softplus, compute_dtype=torch.float64)
    o0.backward(g.double())
    assert rel(out.detach(), o0.detach()) < (2e-5 if dtype == torch.float32 else 6e-3)
    # bf16: du / ddelta / dz / dB / dC carry one bf16 output rounding; the fp32 accumulators (dA, dD, ddelta_bias) do not
    tol = 2e-4 if dtype == torch.float32 else 8e-3
    for name, a, b in zip(["u", "delta", "A", "B", "C", "D", "z", "delta_bias"], leaves, dl):
        if a is not None:
            assert a.grad is not None and a.grad.shape == a.shape, name
            assert rel(a.grad, b.grad) < tol, (name, rel(a.grad, b.grad))


def test_selective_scan_bwd_channel_tiles(dev, monkeypatch):
    """Chunked backward with two channel tiles per workgroup (the dB / dC rows of a pass collect both before they go to HBM) ==
    one tile per workgroup == the fp64 oracle."""
    from omnimamba_amd.selective_scan import selective_scan_fn
    torch.manual_seed(3)
    Bsz, Dm, L, N = 1, 32, 600, 16
    u, delta, z = torch.randn(Bsz, Dm, L), 0.5 * torch.rand(Bsz, Dm, L), torch.randn(Bsz, Dm, L)
    A, Bm, Cm = -(torch.rand(Dm, N) + 0.1), torch.randn(Bsz, N, L), torch.randn(Bsz, N, L)
    D, db = torch.randn(Dm), 0.1 * torch.randn(Dm)
    src = [u, delta, A, Bm, Cm, D, z, db]
    g = torch.randn(Bsz, Dm, L)
    grads = {}
    for oct_ in ("1", "2"):
        monkeypatch.setenv("OMK_SELSCAN_BWD_OCT", oct_)
        if oct_ == "2":
            monkeypatch.setenv("OMK_SELSCAN_NO_PASS_STATES", "1")   # the backward's own state-only forward pass instead of the saved states
        leaves = [t.clone().to(dev).requires_grad_() for t in src]
        selective_scan_fn(*leaves, True).backward(g.to(dev))
        grads[oct_] = [t.grad.cpu() for t in leaves]
    dl = [t.double().clone().requires_grad_() for t in src]
    O.selective_scan_ref(*dl, True, compute_dtype=torch.float64).backward(g.double())
    for a, b, c in zip(grads["1"], grads["2"], dl):
        assert rel(a, c.grad) < 2e-4 and rel(b, c.grad) < 2e-4 and rel(a, b) < 1e-5


def test_selective_scan_bwd_delta_underflow(dev):
    """Chunked backward: a token whose softplus(delta) underflows to exactly 0 has lost u = (delta u) / delta in registers -- the
    kernel then reads the row again.  Gradients of such a batch against the fp64 oracle (ddelta there is gB u sigmoid(raw) ~ 0,
    du = D dy)."""
    from omnimamba_amd.selective_scan import selective_scan_fn
    torch.manual_seed(5)
    Bsz, Dm, L, N = 1, 8, 128, 8
    u, z = torch.randn(Bsz, Dm, L), torch.randn(Bsz, Dm, L)
    delta = torch.randn(Bsz, Dm, L) * 0.5
    delta[:, :, 10:14] = -200.0                      # softplus -> 0 exactly in fp32
    A, Bm, Cm = -(torch.rand(Dm, N) + 0.1), torch.randn(Bsz, N, L), torch.randn(Bsz, N, L)
    D = torch.randn(Dm)
    src = [u, delta, A, Bm, Cm, D, z, None]
    leaves = [None if t is None else t.clone().to(dev).requires_grad_() for t in src]
    g = torch.randn(Bsz, Dm, L)
    selective_scan_fn(*leaves, True).backward(g.to(dev))
    dl = [None if t is None else t.double().clone().requires_grad_() for t in src]
    O.selective_scan_ref(*dl, True, compute_dtype=torch.float64).backward(g.double())
    for name, a, b in zip(["u", "delta", "A", "B", "C", "D", "z"], leaves, dl):
        if a is not None:
            assert torch.isfinite(a.grad).all(), name
            assert rel(a.grad, b.grad) < 2e-4, (name, rel(a.grad, b.grad))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("bc_layout", ["nl", "ln"])
@pytest.mark.parametrize("Dm,L,N,G,with_z,softplus", [(70, 45, 16, 1, True, True), (12, 33, 8, 2, False, True), (130, 100, 5, 1, True, False),
                                                      (64, 16, 16, 1, True, True), (96, 200, 16, 2, True, True)])
def test_selective_scan_bwd_lanes_are_channels(dev, monkeypatch, dtype, bc_layout, Dm, L, N, G, with_z, softplus):
    """Round 6: the backward on CHANNEL-LAST views as they lie (selscan_bwd_lanes_kernel: a lane owns a channel and walks the sequence
    backwards in 16-token tiles, forward states re-run from the checkpoints of a forward pass, dB / dC through a wave butterfly + float
    atomics) -- forced onto small shapes: ragged channel tiles (70 = 64 + 6, groups of 6 / 48), ragged last tile, odd d_state, B / C with either
    of (state, token) contiguous, with and without gate / softplus.  Every gradient against autograd of the fp64 recurrence, and the
    gradient tensors of u / delta / z come back channel-last (no L-contiguous copies anywhere)."""
    from omnimamba_amd.selective_scan import selective_scan_fn
    monkeypatch.setenv("OMK_SELSCAN_LANES", "1")
    torch.manual_seed(5)
    Bsz = 3
    mk = lambda scale=1.0, rand=False: ((torch.rand(Bsz, L, Dm) if rand else torch.randn(Bsz, L, Dm)) * scale).to(dtype)
    u, delta, z = mk(), mk(0.5, True), mk()
    Bm, Cm = torch.randn(Bsz, G, L, N).to(dtype), torch.randn(Bsz, G, L, N).to(dtype)
    A, D, db = -(torch.rand(Dm, N) + 0.1), torch.randn(Dm), torch.randn(Dm) * 0.1
    gy = torch.randn(Bsz, L, Dm).to(dtype)

    def leaves(device, f64):
        cv = (lambda t: t.detach().double()) if f64 else (lambda t: t.detach().clone())
        return [cv(t).to(device).requires_grad_() for t in (u, delta, A, Bm, Cm, D, z, db)]

    lv = leaves(dev, False)
    ud, dd, zd = lv[0].transpose(1, 2), lv[1].transpose(1, 2), lv[6].transpose(1, 2)        # channel-last views (B, D, L) of (B, L, D) storage
    Bd = lv[3].transpose(2, 3) if bc_layout == "nl" else lv[3].transpose(2, 3).contiguous()   # (B, G, N, L): token or state contiguous
    Cd = lv[4].transpose(2, 3) if bc_layout == "nl" else lv[4].transpose(2, 3).contiguous()
    out = selective_scan_fn(ud, dd, lv[2], Bd, Cd, lv[5], zd if with_z else None, lv[7], softplus)
    assert out.stride(1) == 1
    out.backward(gy.to(dev).transpose(1, 2))
    r = leaves("cpu", True)
    o0 = O.selective_scan_ref(r[0].transpose(1, 2), r[1].transpose(1, 2), r[2], r[3].transpose(2, 3), r[4].transpose(2, 3), r[5],
                              r[6].transpose(1, 2) if with_z else None, r[7], softplus)
    o0.backward(gy.double().transpose(1, 2))
    tol = 2e-4 if dtype == torch.float32 else 1.2e-2
    for nm, a_, b_ in zip(["u", "delta", "A", "B", "C", "D", "z", "delta_bias"], lv, r):
        if nm == "z" and not with_z:
            continue
        e = rel(a_.grad, b_.grad)
        assert e < (tol if nm not in ("A", "delta_bias", "D") or dtype == torch.float32 else 3e-2), (nm, e)
    assert lv[0].grad.is_contiguous() and lv[1].grad.is_contiguous()      # (B, L, D) gradients written in place: channel-last all the way

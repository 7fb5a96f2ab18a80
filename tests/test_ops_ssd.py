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


@pytest.mark.parametrize("L,H,P,G,with_z,with_init,dhp", [(150, 2, 64, 1, True, True, False), (16, 4, 64, 2, False, False, False),
                                                          (37, 2, 32, 1, True, False, True), (1, 2, 64, 1, False, True, False)])
def test_ssd_f32_mfma_fwd(dev, monkeypatch, L, H, P, G, with_z, with_init, dhp):
    """fp32 activations (the reference's inference default) take the fp32 matrix-instruction kernel (csrc/ssd_f32.hip: chunks of 16
    tokens, a head cut into 16-column workgroups, no operand rounding): same accuracy class as the token-by-token generic kernel --
    y, the pre-gate copy and the final state against the fp64 recurrence at 3e-5; ragged lengths, one token, gate, initial state,
    D per head and per (head, column), two groups.  OMK_SSD_F32_MFMA=0 gives the generic kernel: the two agree to 2e-5."""
    from omnimamba_amd.ssd_combined import ssd_scan_fwd
    N = 128
    x, dt, A, Bm, Cm, D, z, dtb, init = make(2, L, H, P, N, G, torch.float32, seed=9)
    if dhp:
        D = torch.randn(H, P, generator=torch.Generator().manual_seed(2))
    if not with_z:
        z = None
    if not with_init:
        init = None
    d = lambda t: None if t is None else t.to(dev)
    kw = dict(D=d(D), z=d(z), dt_bias=d(dtb), initial_states=d(init), dt_softplus=True, dt_limit=(0.0, 3.0), return_final_states=True, want_out_x=True)
    out, out_x, fin = ssd_scan_fwd(d(x), d(dt), d(A), d(Bm), d(Cm), **kw)
    o0, f0 = O.ssd_ref_sequential(x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dtb, initial_states=init, dt_softplus=True, dt_limit=(0.0, 3.0),
                                  return_final_states=True, compute_dtype=torch.float64)
    assert out.dtype == torch.float32 and rel(out, o0) < 3e-5 and rel(fin, f0) < 3e-5
    if with_z:
        ox0 = O.ssd_ref_sequential(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, initial_states=init, dt_softplus=True, dt_limit=(0.0, 3.0), compute_dtype=torch.float64)
        assert rel(out_x, ox0) < 3e-5
    monkeypatch.setenv("OMK_SSD_F32_MFMA", "0")
    out_g, _, fin_g = ssd_scan_fwd(d(x), d(dt), d(A), d(Bm), d(Cm), **kw)
    assert rel(out, out_g.cpu()) < 2e-5 and rel(fin, fin_g.cpu()) < 2e-5 and not torch.equal(out.cpu(), out_g.cpu())


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


def _mfma_case(dev, Bsz, L, H, G, with_z, with_init, seed=11):
    from omnimamba_amd.ssd_combined import ssd_scan_fwd
    P, N = 64, 128
    x, dt, A, Bm, Cm, D, z, dtb, init = make(Bsz, L, H, P, N, G, torch.bfloat16, seed=seed)
    A = -(torch.rand(H) * 15 + 1)          # module-default range A ~ U(1, 16)
    dtb = torch.randn(H) * 0.5 - 3.0        # dt' around softplus(-3) ~ 0.05 like the module's dt init
    d = lambda t: None if t is None else t.to(dev)
    zz, ii = (z if with_z else None), (init if with_init else None)
    out, out_x, fin = ssd_scan_fwd(d(x), d(dt), d(A), d(Bm), d(Cm), D=d(D), z=d(zz), dt_bias=d(dtb), initial_states=d(ii),
                                   dt_softplus=True, return_final_states=True, want_out_x=True)
    outg, _, fing = ssd_scan_fwd(d(x), d(dt), d(A), d(Bm), d(Cm), D=d(D), z=d(zz), dt_bias=d(dtb), initial_states=d(ii),
                                 dt_softplus=True, return_final_states=True, force_generic=True)
    o0, f0 = O.ssd_ref_sequential(x, dt, A, Bm, Cm, D=D, z=zz, dt_bias=dtb, initial_states=ii, dt_softplus=True,
                                  return_final_states=True)       # fp32 math on the same bf16 inputs, rounded once
    o32 = O.ssd_ref_sequential(x.float(), dt.float(), A, Bm.float(), Cm.float(), D=D, z=None if zz is None else zz.float(),
                               dt_bias=dtb, initial_states=ii, dt_softplus=True)
    return out, out_x, fin, outg, fing, o0, f0, o32


def _budgets(L, H, G, with_z, with_init, seed=11):
    """Per-case bounds under the rule of tests/tolerances.py (same generator draws as _mfma_case)."""
    from tolerances import forward_budget
    P, N = 64, 128
    x, dt, A, Bm, Cm, D, z, dtb, init = make(1, L, H, P, N, G, torch.bfloat16, seed=seed)
    A = -(torch.rand(H) * 15 + 1)
    dtb = torch.randn(H) * 0.5 - 3.0
    _, _, by, bf, up = forward_budget(x, dt, A, Bm, Cm, D=D, z=z if with_z else None, dt_bias=dtb, initial_states=init if with_init else None,
                                      dt_softplus=True)
    return by, bf, up


@pytest.mark.parametrize("L,H,G,with_z,with_init", [(150, 2, 1, False, False), (64, 4, 2, True, True), (200, 2, 1, True, True)])
def test_ssd_mfma_fwd(dev, L, H, G, with_z, with_init):
    """bf16 MFMA path vs the fp32 / fp64 oracle on identical bf16 inputs.  Tolerance (tests/tolerances.py): rel-L2 <=
    sqrt(arith^2 + q^2), q the unavoidable bf16 quantisation of the output itself (measured on the oracle), arith =
    max(1e-3 north star, the arithmetic error of the reference pipeline's own rounding points on the same inputs): from an O(1)
    random initial_states the bf16 copy of S_in fed to the C . S MFMA -- upstream rounds there too -- costs both ~1.2e-3."""
    st = torch.random.get_rng_state()
    by, bf, up = _budgets(L, H, G, with_z, with_init)
    torch.random.set_rng_state(st)
    out, out_x, fin, outg, fing, o0, f0, o32 = _mfma_case(dev, 1, L, H, G, with_z, with_init)
    q = rel(o32.bfloat16().float(), o32)
    tol = (by ** 2 + q ** 2) ** 0.5
    e = rel(out.float(), o32)
    assert e < tol, (e, q, tol, up)
    # final state: the w_l K_l operand of the state update is rounded to bf16 once (as upstream's chunk-state kernel does) unless
    # OMK_SSD_KHILO / OMK_SSD_PRECISE ask for the hi + lo pair
    # round 4: a kept final state is carried with the hi + lo operand (ssd_a6.hip) -- the bare north-star 1e-3, not the upstream-rounding budget
    assert rel(fin, f0) < 1e-3, (rel(fin, f0), bf, up)
    assert rel(outg.float(), o32) < tol and rel(fing, f0) < 1e-4
    if with_z:
        assert out_x is not None


def test_ssd_khilo_keeps_the_carried_state_exact(dev, monkeypatch):
    """OmkSsdFwd.flags & OMK_SSD_KHILO (scan_options(khilo=True)): the w_l K_l operand of the state update as a bf16 hi + lo pair -- the carried state (and final_states) no
    longer carries one bf16 rounding per chunk: 1.7e-3 -> ~1e-5 (+ 10 % scan time on the MI355X, hence opt-in)."""
    import omnimamba_amd.ssd_combined as S
    with S.scan_options(khilo=True):
        out, _, fin, _, _, o0, f0, o32 = _mfma_case(dev, 1, 200, 2, 1, False, True, seed=23)
    assert rel(fin, f0) < 1e-4, rel(fin, f0)
    q = rel(o32.bfloat16().float(), o32)
    assert rel(out.float(), o32) < (1.2e-3 ** 2 + q ** 2) ** 0.5


@pytest.mark.parametrize("L,H,G,with_z,with_init,minc", [(200, 2, 1, False, False, 1), (330, 2, 2, True, True, 2)])
def test_ssd_mfma_fwd_split_sequence(dev, monkeypatch, L, H, G, with_z, with_init, minc):
    """Few (batch, head) pairs: the class A scans cut the sequence into segments (state-only pass + scan proper from the
    folded segment states, ssd_scan.h).  Same tolerances as the unsplit kernel, and both agree with each other."""
    monkeypatch.setenv("OMK_SSD_SEG_CHUNKS", str(minc))   # production: >= 8 chunks per segment
    torch.manual_seed(21)
    out, out_x, fin, outg, fing, o0, f0, o32 = _mfma_case(dev, 1, L, H, G, with_z, with_init)
    import omnimamba_amd.ssd_combined as S
    torch.manual_seed(21)
    with S.scan_options(no_split=True):
        out1, _, fin1, _, _, _, _, _ = _mfma_case(dev, 1, L, H, G, with_z, with_init)
    torch.manual_seed(21)
    by, bf, up = _budgets(L, H, G, with_z, with_init)
    q = rel(o32.bfloat16().float(), o32)
    tol = (by ** 2 + q ** 2) ** 0.5
    # (the segment start states come from the row-strip kernel's zero-start state pass: one bf16 operand rounding per chunk, the upstream budget)
    assert rel(out.float(), o32) < tol and rel(fin, f0) < bf, (rel(out.float(), o32), tol, rel(fin, f0), bf, up)
    assert rel(out.float(), out1.float().cpu()) < 2e-3 and rel(fin, fin1.cpu()) < 1e-5


def test_ssd_mfma_split_sequence_long_memory(dev, monkeypatch):
    """Slow decay (|A| ~ 0.1): the state entering a segment is dominated by what the earlier segments left, so this
    pins the fold of the segment states (coefficients = products of the later segments' decays, initial_states
    included) against the fp32 oracle, and the reverse scan (dx) through the gradient."""
    import omnimamba_amd.ssd_combined as S
    monkeypatch.setenv("OMK_SSD_SEG_CHUNKS", "1")
    L, H, G, P, N = 300, 2, 1, 64, 128
    x, dt, A, Bm, Cm, D, z, dtb, init = make(1, L, H, P, N, G, torch.bfloat16, seed=3)
    A = -(torch.rand(H) * 0.1 + 0.05)
    dtb = torch.randn(H) * 0.3 - 2.0
    src = [x, dt, A, Bm, Cm, D, dtb, init]
    lv = [t.clone().to(dev).requires_grad_() for t in src]
    y, fin = S.mamba_chunk_scan_combined(lv[0], lv[1], lv[2], lv[3], lv[4], 256, D=lv[5], dt_bias=lv[6], initial_states=lv[7],
                                         dt_softplus=True, return_final_states=True)
    gy, gf = torch.randn(y.shape).bfloat16(), torch.randn(fin.shape)
    torch.autograd.backward([y, fin], [gy.to(dev), gf.to(dev)])
    dl = [t.double().clone().requires_grad_() for t in src]
    y0, f0 = O.ssd_ref_sequential(dl[0], dl[1], dl[2], dl[3], dl[4], D=dl[5], dt_bias=dl[6], initial_states=dl[7],
                                  dt_softplus=True, return_final_states=True, compute_dtype=torch.float64)
    torch.autograd.backward([y0, f0], [gy.double(), gf.double()])
    from tolerances import forward_budget
    _, _, by, bf, up = forward_budget(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, initial_states=init, dt_softplus=True)
    q = rel(y0.float().bfloat16().float(), y0.float())
    # slow decay: nearly all of y is the inter-chunk term, which sees the bf16 copy of the carried state -- in upstream's pipeline
    # (the rule of tests/tolerances.py) as much as here
    assert rel(y.float().cpu(), y0.float()) < (by ** 2 + q ** 2) ** 0.5, (rel(y.float().cpu(), y0.float()), by, q, up)
    assert rel(fin.cpu(), f0.float()) < bf, (rel(fin.cpu(), f0.float()), bf, up)
    assert rel(lv[0].grad.float().cpu(), dl[0].grad.float()) < 6e-3          # dx: reverse scan, split the same way
    assert rel(lv[7].grad.float().cpu(), dl[7].grad.float()) < 6e-3          # d initial_states: its final state
    # the dC / dB scans cut the sequence the same way (their start states: the folded forward / adjoint segment states)
    assert rel(lv[3].grad.float().cpu(), dl[3].grad.float()) < 6e-3 and rel(lv[4].grad.float().cpu(), dl[4].grad.float()) < 6e-3
    assert rel(lv[1].grad.float().cpu(), dl[1].grad.float()) < 8e-3          # d(dt): token scalars + boundary restarts
    assert rel(lv[2].grad.float().cpu(), dl[2].grad.float()) < 3e-2          # dA: a signed sum of d(dt) over 300 tokens


@pytest.mark.parametrize("cp", ["1", "0"])
@pytest.mark.parametrize("L,H,G,with_z,with_init,minc", [(130, 2, 1, False, False, None), (70, 4, 2, True, True, None),
                                                                (200, 2, 1, False, True, 1), (330, 4, 2, False, True, 2),
                                                                (300, 8, 1, False, True, None)])
def test_ssd_mfma_bwd(dev, monkeypatch, request, L, H, G, with_z, with_init, minc, cp):
    """bf16 MFMA backward vs autograd of the fp64 oracle on identical bf16 inputs, both forms: cp = 1 the chunk-parallel backward
    of round 3 (dx scan + state-only forward pass dump the window states, ssd_cp.hip forms dB / dC / token scalars / dD per
    128-token window with the head sum on chip; the 8-head case splits the heads of the group over two workgroups), cp = 0 the
    three sequential scans of rounds 1 / 2 (still the path for D per (head, column)).  minc: split the sequence of the class A
    scans into segments of that many chunks (see test_ssd_mfma_fwd_split_sequence)."""
    import contextlib
    import omnimamba_amd.ssd_combined as S
    es = contextlib.ExitStack()
    es.enter_context(S.scan_options(sequential_bwd=(cp == "0")))
    request.addfinalizer(es.close)
    if minc:
        monkeypatch.setenv("OMK_SSD_SEG_CHUNKS", str(minc))
    import omnimamba_amd.ssd_combined as S
    P, N = 64, 128
    x, dt, A, Bm, Cm, D, z, dtb, init = make(1, L, H, P, N, G, torch.bfloat16, seed=5)
    torch.manual_seed(7)
    A = -(torch.rand(H) * 15 + 1)
    dtb = torch.randn(H) * 0.5 - 3.0
    if not with_z:
        z = None
    if not with_init:
        init = None
    src = [x, dt, A, Bm, Cm, D, z, dtb, init]
    leaves = [None if t is None else t.clone().to(dev).requires_grad_() for t in src]
    xr, dtr, Ar, Br, Cr, Dr, zr, dtbr, ir = leaves
    y, fin = S.mamba_chunk_scan_combined(xr, dtr, Ar, Br, Cr, 256, D=Dr, z=zr, dt_bias=dtbr, initial_states=ir,
                                         dt_softplus=True, return_final_states=True)
    gy, gf = torch.randn(y.shape).bfloat16(), torch.randn(fin.shape) * (1.0 if with_init else 0.0)
    torch.autograd.backward([y, fin], [gy.to(dev), gf.to(dev)])
    dl = [None if t is None else t.double().clone().requires_grad_() for t in src]
    y0, f0 = O.ssd_ref_sequential(dl[0], dl[1], dl[2], dl[3], dl[4], D=dl[5], z=dl[6], dt_bias=dl[7], initial_states=dl[8],
                                  dt_softplus=True, return_final_states=True, compute_dtype=torch.float64)
    torch.autograd.backward([y0, f0], [gy.double(), gf.double()])
    # bf16 outputs (dx, dB, dC, dz) carry one output rounding (1.65e-3) on top of the arithmetic error; fp32 outputs do not
    # bf16 outputs (dx, dB, dC, dz) carry one output rounding (1.65e-3) on top of the arithmetic error.  d(dt) is
    # accurate per token (~2e-3); dA and d(dt_bias) are signed SUMS of it over every token, so at this tiny size
    # (130 tokens, 1 sequence) cancellation amplifies the same bf16-level noise: 8e-2 of the vector norm here, shrinking
    # ~1/sqrt(tokens) at training sizes.  The fp32 generic path (test_ssd_generic_bwd) is exact to 2e-4.
    tol = {"x": 5e-3, "dt": 6e-3, "A": 8e-2, "B": 5e-3, "C": 5e-3, "D": 5e-3, "z": 5e-3, "dt_bias": 8e-2, "init": 5e-3}
    if cp == "1":
        # the chunk-parallel form builds the token scalars in fp32 from exact bf16 products (no bf16 M operand in that chain) and
        # restarts the decay-gradient prefix from exact window values: d(dt) 1.4e-3 .. 2e-3, dA / d(dt_bias) 3e-4 .. 5e-3 measured
        tol.update({"dt": 4e-3, "A": 1.5e-2, "dt_bias": 1.5e-2, "B": 4e-3, "C": 4e-3})
    for n, a, b in zip(["x", "dt", "A", "B", "C", "D", "z", "dt_bias", "init"], leaves, dl):
        if a is not None:
            e = rel(a.grad, b.grad)
            assert e < tol[n], (n, e, cp)


@pytest.mark.parametrize("L,H,G,with_init,minc", [(130, 2, 1, False, None), (300, 4, 2, True, None), (330, 2, 1, True, 2)])
def test_ssd_bwd_with_window_states_saved_by_the_forward(dev, monkeypatch, L, H, G, with_init, minc):
    """A training forward leaves the carried state in front of every 128-token window behind (OmkSsdFwd.window_states); the backward
    that receives it skips its own state pass over x.  Same images from the same code: the gradients are IDENTICAL to the ones of
    the recomputing backward (OMK_SSD_SAVE_WINDOW_STATES=0), the forward output identical to the forward without the dumps, and a
    forward that cannot save them (gate requested) reports 0 bytes."""
    if minc:
        monkeypatch.setenv("OMK_SSD_SEG_CHUNKS", str(minc))
    import omnimamba_amd.ssd_combined as S
    P, N = 64, 128
    x, dt, A, Bm, Cm, D, z, dtb, init = make(1, L, H, P, N, G, torch.bfloat16, seed=11)
    A = -(torch.rand(H, generator=torch.Generator().manual_seed(3)) * 15 + 1)
    if not with_init:
        init = None
    src = [x, dt, A, Bm, Cm, D, dtb, init]
    gy = torch.randn(1, L, H, P, generator=torch.Generator().manual_seed(4)).bfloat16()
    grads, outs = {}, {}
    def run(mode):
        monkeypatch.setenv("OMK_SSD_SAVE_WINDOW_STATES", mode)
        leaves = [None if t is None else t.clone().to(dev).requires_grad_() for t in src]
        xr, dtr, Ar, Br, Cr, Dr, dtbr, ir = leaves
        y = S.mamba_chunk_scan_combined(xr, dtr, Ar, Br, Cr, 256, D=Dr, dt_bias=dtbr, initial_states=ir, dt_softplus=True)
        y.backward(gy.to(dev))
        return [None if t is None else t.grad.detach().cpu() for t in leaves], y.detach().cpu()
    # bit for bit with the basis of the carried state moved at every chunk (the state pass of the recomputing backward is the column-slice
    # kernel, which does): identical images, identical gradients
    with S.scan_options(every_chunk=True):
        for mode in ("1", "0"):
            grads[mode], outs[mode] = run(mode)
    assert torch.equal(outs["1"], outs["0"])
    for n, a, b in zip(["x", "dt", "A", "B", "C", "D", "dt_bias", "init"], grads["1"], grads["0"]):
        if a is not None:
            assert torch.equal(a, b), n
    # the default (lazy basis, round 6): the forward that dumps pulls the basis up in front of every image, the plain one only where it
    # drifts -- equal to a rounding of the fp32 state, not to the bit
    for mode in ("1", "0"):
        grads["d" + mode], outs["d" + mode] = run(mode)
    # (the bf16 copy of the state that meets C is the rounding of S 2^-drift instead of S: another draw of the same 2^-9 rounding error,
    # so the two outputs differ by about sqrt(2) x the arithmetic error of either -- test_ssd_lazy_basis_of_the_carried_state compares both
    # with the fp64 recurrence)
    assert rel(outs["d1"].float(), outs["1"].float()) < 2.5e-3 and rel(outs["d0"].float(), outs["1"].float()) < 2.5e-3
    for n, a, b in zip(["x", "dt", "A", "B", "C", "D", "dt_bias", "init"], grads["d1"], grads["1"]):
        if a is not None:
            assert rel(a.float(), b.float()) < (6e-3 if n in ("A", "dt_bias", "dt") else 3e-3), (n, rel(a.float(), b.float()))
    # the raw call: the window-state tensor exists for the plain scan and not when a gate is asked for
    d = lambda t: None if t is None else t.to(dev)
    r = S.ssd_scan_fwd(d(x), d(dt), d(A), d(Bm), d(Cm), D=d(D), dt_bias=d(dtb), dt_softplus=True, save_window_states=True)
    assert r[3] is not None and r[3].numel() == ((L + 127) // 128) * H * 8192
    r = S.ssd_scan_fwd(d(x), d(dt), d(A), d(Bm), d(Cm), D=d(D), z=d(z), dt_bias=d(dtb), dt_softplus=True, save_window_states=True)
    assert r[3] is None


@pytest.mark.parametrize("L,H,G", [(200, 2, 1), (330, 4, 2)])
def test_ssd_precise_forward_meets_the_1e3_budget_with_initial_states(dev, monkeypatch, L, H, G):
    """OmkSsdFwd.flags & OMK_SSD_PRECISE (scan_options(precise=True); the PRECISE instantiation of ssd_a8.hip): the bf16 copy of the carried state and the w_l K_l operand of the state update
    as hi + lo pairs.  From an O(1) random initial state -- the stress case where the default path (and upstream's kernels, which
    round the same two operands to bf16) needs a 1.5e-3 budget for y and 2.5e-3 for the final state -- y stays inside the
    north-star 1e-3 (on top of the output's own bf16 quantisation) and the final state inside 1e-3 of the fp32 recurrence."""
    import omnimamba_amd.ssd_combined as S
    res = {}
    for mode in ("0", "1"):
        with S.scan_options(precise=(mode == "1")):
            out, _, fin, _, _, o0, f0, o32 = _mfma_case(dev, 1, L, H, G, False, True, seed=21)
        q = rel(o32.bfloat16().float(), o32)
        e = rel(out.float(), o32)
        res[mode] = (max(e * e - q * q, 0.0) ** 0.5, rel(fin, f0))       # arithmetic part of the error of y, error of the final state
    assert res["1"][0] < 1e-3 and res["1"][1] < 1e-3, res
    # round 4: the default kernel (ssd_a6.hip) carries its state in fp32 accumulators and enters the scaled operand of the state update
    # as hi + lo whenever the final state is kept, so the default final state is already exact to 1e-5; what the precise path still
    # buys is the bf16 rounding of the state copy that feeds y
    assert res["0"][1] < 1e-3, res
    assert res["1"][0] < 0.6 * res["0"][0], res


def test_training_forward_on_a_layout_the_mfma_kernel_cannot_take_falls_back(dev):
    """Advisor finding (round 3): bf16 (1, 130, 2, 64) x at a 2-byte storage offset with requires_grad.  The window-state query must
    answer 0 for a forward the MFMA kernel cannot take (it makes the same dry check as the launch), so the call takes the generic
    fall-back instead of failing with 'window_states asked for on a shape outside the MFMA kernel'."""
    import omnimamba_amd.ssd_combined as S
    L, H, P, N = 130, 2, 64, 128
    x, dt, A, Bm, Cm, D, z, dtb, init = make(1, L, H, P, N, 1, torch.bfloat16, seed=5)
    buf = torch.zeros(x.numel() + 1, dtype=torch.bfloat16)
    buf[1:] = x.reshape(-1)
    xo = buf[1:].view(1, L, H, P).to(dev) if dev.type == "cpu" else None
    if xo is None:   # keep the odd storage offset on the device
        db = buf.to(dev)
        xo = db[1:].view(1, L, H, P)
    assert xo.storage_offset() == 1
    xo = xo.detach().requires_grad_()
    d = lambda t: None if t is None else t.to(dev)
    y = S.mamba_chunk_scan_combined(xo, d(dt), d(A), d(Bm), d(Cm), 256, D=d(D), dt_bias=d(dtb), dt_softplus=True)
    y.float().sum().backward()
    y0 = S.mamba_chunk_scan_combined(d(x), d(dt), d(A), d(Bm), d(Cm), 256, D=d(D), dt_bias=d(dtb), dt_softplus=True)
    assert rel(y.float().cpu(), y0.float().cpu()) < 6e-3 and xo.grad is not None and torch.isfinite(xo.grad.float()).all()


@pytest.mark.parametrize("L", [1, 5, 31, 33, 63, 65, 127, 129, 191])
def test_ssd_column_slice_kernel_short_and_ragged_sequences(dev, L):
    """ssd_a6.hip stages two chunks ahead and builds its intra tiles a chunk ahead: sequences shorter than one chunk, ragged last
    chunks and lengths around the 32-token sub-chunk and 128-token window boundaries (rows behind the end arrive as zeros, their
    weights are zero, their stores are dropped).  Forward with final state and backward (dx scan, window-state dumps) vs the oracle."""
    import omnimamba_amd.ssd_combined as S
    H, P, N, G = 4, 64, 128, 2
    x, dt, A, Bm, Cm, D, z, dtb, init = make(2, L, H, P, N, G, torch.bfloat16, seed=100 + L)
    A = -(torch.rand(H, generator=torch.Generator().manual_seed(L)) * 15 + 1)
    d = lambda t: None if t is None else t.to(dev)
    leaves = [t.clone().to(dev).requires_grad_() for t in (x, dt, A, Bm, Cm, D, dtb, init)]
    y, fin = S.mamba_chunk_scan_combined(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], 256, D=leaves[5], dt_bias=leaves[6],
                                         initial_states=leaves[7], dt_softplus=True, return_final_states=True)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).bfloat16()
    y.backward(gy.to(dev))
    ref = [t.clone().float().requires_grad_() for t in (x, dt, A, Bm, Cm, D, dtb, init)]
    y0, f0 = O.ssd_ref_chunked(ref[0], ref[1], ref[2], ref[3], ref[4], 64, D=ref[5], dt_bias=ref[6], initial_states=ref[7], dt_softplus=True,
                               return_final_states=True)
    y0.backward(gy.float())
    assert rel(y.float().cpu(), y0) < 6e-3 and rel(fin.cpu(), f0) < 1e-3
    assert rel(leaves[0].grad.float().cpu(), ref[0].grad) < 8e-3                      # dx
    assert rel(leaves[3].grad.float().cpu(), ref[3].grad) < 8e-3 and rel(leaves[4].grad.float().cpu(), ref[4].grad) < 8e-3   # dB, dC
    assert rel(leaves[7].grad.float().cpu(), ref[7].grad) < 8e-3                      # d(initial_states)
    for t in leaves:
        assert torch.isfinite(t.grad.float()).all()


def test_ssd_column_slice_kernel_extreme_decays_take_the_exponent_path(dev):
    """ssd_a6.hip builds the decay of an intra tile as (row factor) x (column factor) around a reference inside the tile; a head whose
    16-token blocks decay by more than 2^-90 would leave the fp32 range with that, so its scalar wave hands the builders exponents
    instead (SmemA6::wide) -- per head and chunk.  Heads here: moderate, extreme (dt |A| ~ 16 per token), around the threshold, and
    one that switches inside the sequence; the forward, the final state and dx against the fp64 oracle."""
    import omnimamba_amd.ssd_combined as S
    Bsz, L, H, P, N, G = 1, 200, 4, 64, 128, 1
    x, dt, A, Bm, Cm, D, z, dtb, init = make(Bsz, L, H, P, N, G, torch.bfloat16, seed=77)
    A = torch.tensor([-2.0, -16.0, -7.8, -12.0])
    dt = torch.ones(Bsz, L, H) + 0.02 * torch.randn(Bsz, L, H, generator=torch.Generator().manual_seed(3))
    dt[:, :, 0] *= 0.05
    dt[:, 100:, 3] *= 0.01            # head 3: extreme for the first 100 tokens, slow behind
    dt = dt.bfloat16()
    leaves = [t.clone().to(dev).requires_grad_() for t in (x, dt, A, Bm, Cm, D, init)]
    y, fin = S.mamba_chunk_scan_combined(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], 256, D=leaves[5], initial_states=leaves[6],
                                         return_final_states=True)
    gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).bfloat16()
    y.backward(gy.to(dev))
    ref = [t.clone().double().requires_grad_() for t in (x, dt, A, Bm, Cm, D, init)]
    y0, f0 = O.ssd_ref_chunked(ref[0], ref[1], ref[2], ref[3], ref[4], 64, D=ref[5], initial_states=ref[6], return_final_states=True,
                               compute_dtype=torch.float64)
    y0.backward(gy.double())
    assert torch.isfinite(y.float()).all() and torch.isfinite(fin).all()
    for h in range(H):
        assert rel(y[:, :, h].float().cpu(), y0[:, :, h]) < 6e-3, h
        assert rel(fin[:, h].cpu(), f0[:, h]) < 1e-3, h
        assert rel(leaves[0].grad[:, :, h].float().cpu(), ref[0].grad[:, :, h]) < 8e-3, h


@pytest.mark.parametrize("regime", ["slow", "mixed", "fast", "extreme"])
def test_ssd_lazy_basis_of_the_carried_state(dev, regime):
    """Round 6: ssd_a8.hip moves the basis of the carried state (the decay multiply of its 64 accumulator registers) only where the
    basis drifts by 2^60, where an image or the final state needs the true state, never otherwise.  Against the every-chunk arithmetic
    (OMK_SSD_EVERY_CHUNK, bit-equal to ssd_a6.hip) the result must agree to a rounding of the fp32 state -- across decay regimes: slow
    heads (the basis never moves before the last chunk), heads that cross the 2^-60 drift every few chunks, heads whose every 32-token
    sub-chunk decays by more than 2^-60, and both in one group; forward with and without final state, the training forward (images), the
    dx scan of the backward; and against the fp64 recurrence directly."""
    import omnimamba_amd.ssd_combined as S
    L, H, P, N, G = 650, 4, 64, 128, 1
    x, dt, A, Bm, Cm, D, z, dtb, init = make(1, L, H, P, N, G, torch.bfloat16, seed=31)
    if regime == "slow":
        A, dtb = -torch.tensor([1.0, 1.5, 2.0, 1.2]), torch.full((H,), -6.0)          # ~0.005 per token: 2^-0.3 per chunk
    elif regime == "mixed":
        A, dtb = -torch.tensor([1.0, 16.0, 6.0, 11.0]), torch.tensor([-6.0, -1.0, -2.5, -2.0])   # 2^-0.3 .. 2^-400 per chunk
    elif regime == "fast":
        A, dtb = -torch.tensor([16.0, 12.0, 14.0, 9.0]), torch.full((H,), 0.5)        # 2^-70 .. 2^-130 per 32 tokens
    else:
        A, dtb = -torch.tensor([16.0, 1.0, 16.0, 3.0]), torch.tensor([3.0, -7.0, 2.0, -1.0])     # 2^-2000 per sub-chunk next to 2^-0.1 per chunk
    d = lambda t: t.to(dev)

    def run(keep_final, train):
        leaves = [t.clone().to(dev).requires_grad_(train) for t in (x, dt, A, Bm, Cm, D, dtb, init)]
        r = S.mamba_chunk_scan_combined(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], 256, D=leaves[5], dt_bias=leaves[6],
                                        initial_states=leaves[7], dt_softplus=True, return_final_states=keep_final)
        y, fin = r if keep_final else (r, None)
        g = []
        if train:
            y.backward(torch.ones_like(y))
            g = [t.grad.float().cpu() for t in leaves]
        return [y.detach().float().cpu(), None if fin is None else fin.detach().cpu()] + g

    y64, f64 = O.ssd_ref_chunked(x, dt, A, Bm, Cm, 64, D=D, dt_bias=dtb, initial_states=init, dt_softplus=True, return_final_states=True,
                                 round_output=False, compute_dtype=torch.float64)
    for keep_final, train in ((True, False), (False, False), (False, True)):
        with S.scan_options(every_chunk=True):
            ref = run(keep_final, train)
        got = run(keep_final, train)
        # y: the bf16 copy of the state that meets C is the rounding of S 2^-drift instead of S -- another draw of the same 2^-9 rounding
        # error, so the two outputs differ by about sqrt(2) x the arithmetic error of either; against the exact result neither is worse
        assert rel(got[0], ref[0]) < 2.5e-3, (regime, keep_final, train, rel(got[0], ref[0]))
        assert rel(got[0], y64) <= rel(ref[0], y64) * 1.1 + 1e-5, (regime, rel(got[0], y64), rel(ref[0], y64))
        if keep_final:   # (fp32 accumulators, hi + lo operand: the state itself is exact either way)
            assert rel(got[1], ref[1]) < 5e-6 and rel(got[1], f64) < 2e-5, (regime, rel(got[1], ref[1]), rel(got[1], f64))
        for nm, a_, b_ in zip(["dx", "d dt", "dA", "dB", "dC", "dD", "d dt_bias", "d init"], got[2:], ref[2:]):
            assert rel(a_, b_) < (6e-3 if nm in ("dA", "d dt_bias", "d dt") else 3e-3), (regime, nm, rel(a_, b_))


@pytest.mark.parametrize("L,minc", [(200, None), (330, 2), (64, None), (97, None)])
def test_ssd_specialised_wave_kernel_matches_the_column_slice_kernel_bitwise(dev, monkeypatch, L, minc):
    """ssd_a8.hip (four compute waves of 32 state columns + four helper waves per head pair) is the column-slice scan of ssd_a6.hip with
    another work split and instruction order and -- with OMK_SSD_EVERY_CHUNK -- the SAME arithmetic: output, final state (hi + lo operand), and the gradients of a
    backward that runs its dx scan and its window-state images through it must equal ssd_a6.hip's bit for bit -- unsplit and split
    sequences (minc: chunks per segment), a ragged last chunk, a single chunk."""
    import omnimamba_amd.ssd_combined as S
    H, P, N, G = 4, 64, 128, 2
    x, dt, A, Bm, Cm, D, z, dtb, init = make(2, L, H, P, N, G, torch.bfloat16, seed=11 + L)
    if minc:
        monkeypatch.setenv("OMK_SSD_SEG_CHUNKS", str(minc))

    def run(keep_final):
        leaves = [t.clone().to(dev).requires_grad_() for t in (x, dt, A, Bm, Cm, D, dtb, init)]
        r = S.mamba_chunk_scan_combined(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], 256, D=leaves[5], dt_bias=leaves[6],
                                        initial_states=leaves[7], dt_softplus=True, return_final_states=keep_final)
        y, fin = r if keep_final else (r, None)
        y.backward(torch.ones_like(y))
        return [y.detach().float().cpu(), None if fin is None else fin.detach().cpu()] + [t.grad.float().cpu() for t in leaves]

    names = ["y", "final state", "dx", "d dt", "dA", "dB", "dC", "dD", "d dt_bias", "d initial_states"]
    for keep_final in (True, False):
        with S.scan_options(column_slice=True):
            ref = run(keep_final)
        with S.scan_options(every_chunk=True):    # (the default moves the basis of the carried state lazily: next test)
            got = run(keep_final)
        for nm, r, g_ in zip(names, ref, got):
            if r is None:
                continue
            if nm in ("dA", "dD", "d dt_bias", "d dt"):   # (sums formed with float atomics / by another launch order: equal to rounding, not to the bit)
                assert rel(g_, r) < 1e-5, (nm, keep_final)
            else:
                assert torch.equal(r, g_), (nm, keep_final)


@pytest.mark.parametrize("L,H,G,W", [(200, 8, 1, 4), (64, 8, 2, 4), (333, 8, 2, 3), (2, 8, 1, 4), (130, 8, 1, 2)])   # (H = 8: zxbcdt rows of a multiple of 16 bytes, like the 8512-wide rows of the model)
def test_k2_fused_conv_scan_forward_equals_the_separate_ops(dev, monkeypatch, L, H, G, W):
    """K2 fusion, forward-only path (models/stage2/generation.py:195-211 prefill): under no_grad the fused node
    hands the PRE-conv x columns to the scan (OmkSsdFwd.conv_weight), which applies conv1d + SiLU while staging them, and only the 2 G N
    B / C channels take a conv launch.  Same fp32 taps in the same order and one bf16 rounding: output, final state and the conv state
    left for the decode step must equal the separate ops BIT FOR BIT -- and both must meet the oracle composed from conv + scan + norm."""
    import omnimamba_amd.ssd_combined as S
    from omnimamba_amd._lib import get_lib
    P, N, Bsz = 64, 128, 2
    d_ssm = H * P
    Ct = d_ssm + 2 * G * N
    g = torch.Generator().manual_seed(100 + L)
    zxbcdt = (torch.randn(Bsz, L, 2 * d_ssm + 2 * G * N + H, generator=g) * 0.8).bfloat16()
    cw, cb = torch.randn(Ct, W, generator=g) * 0.4, torch.randn(Ct, generator=g) * 0.2
    dtb, A, D = torch.randn(H, generator=g) * 0.5 - 2.0, -(torch.rand(H, generator=g) * 8 + 0.5), torch.randn(H, generator=g)
    nw = torch.rand(d_ssm, generator=g) + 0.5
    d = lambda t: t.to(dev)
    res = {}
    monkeypatch.setenv("OMK_K2_MIN_WGS", "1")     # (production takes the fused form only when the scan's workgroups fill the chip)
    for mode in ("1", "0"):
        monkeypatch.setenv("OMK_K2_FUSED", mode)
        cs = torch.full((Bsz, Ct, 4), 7.0, dtype=torch.bfloat16).to(dev)      # (a state_len of 4 >= width - 1: left zero padded / oldest column first)
        with torch.no_grad():
            out, fin = S.mamba_split_conv1d_scan_combined(d(zxbcdt), d(cw), d(cb), d(dtb), d(A), d(D), 256, return_final_states=True,
                                                          rmsnorm_weight=d(nw), rmsnorm_eps=1e-5, headdim=P, ngroups=G, norm_before_gate=False,
                                                          conv_state_out=cs)
        res[mode] = (out.float().cpu(), fin.cpu(), cs.float().cpu(), get_lib().omk_ssd_last_kernels().decode())
    assert "conv=1" in res["1"][3] and "conv=1" not in res["0"][3], (res["1"][3], res["0"][3])
    for i, nm in enumerate(["out", "final state", "conv state"]):
        assert torch.equal(res["1"][i], res["0"][i]), (nm, rel(res["1"][i], res["0"][i]))
    # against the oracle: conv + SiLU (rounded to bf16 like the conv kernel's output), scan, gated norm
    z, xBC, dt = torch.split(zxbcdt, [d_ssm, Ct, H], dim=-1)
    xc = O.causal_conv1d_ref(xBC.transpose(1, 2).float(), cw, cb, activation="silu").transpose(1, 2).bfloat16()
    x, Bm, Cm = torch.split(xc, [d_ssm, G * N, G * N], dim=-1)
    y, f0 = O.ssd_ref_sequential(x.unflatten(-1, (H, P)), dt, A, Bm.unflatten(-1, (G, N)), Cm.unflatten(-1, (G, N)), D=D, dt_bias=dtb, dt_softplus=True,
                                 return_final_states=True)
    ref = O.rmsnorm_gated_ref(y.flatten(-2).float(), nw, z=z.float(), eps=1e-5, group_size=d_ssm // G, norm_before_gate=False)
    assert rel(res["1"][0], ref) < 8e-3 and rel(res["1"][1], f0) < 1e-3, (rel(res["1"][0], ref), rel(res["1"][1], f0))
    want_cs = torch.nn.functional.pad(xBC.transpose(1, 2).float(), (max(4 - L, 0), 0))[..., -4:]
    assert torch.equal(res["1"][2], want_cs)

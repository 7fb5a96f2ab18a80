"""Committed golden vectors (tests/golden/oracle_ops.npz, made by make_golden.py): (1) the oracle still reproduces them
(pins it against drift), (2) the HIP kernels reproduce them (emulator on CPU, MI355X under -m gpu)."""
import os

import numpy as np
import torch

import oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "oracle_ops.npz"))
T = lambda k: torch.from_numpy(G[k])


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def test_oracle_reproduces_golden():
    H, P, N = 4, 8, 16
    y, fin = O.ssd_ref_chunked(T("ssd.x"), T("ssd.dt"), T("ssd.A"), T("ssd.B"), T("ssd.C"), 32, D=T("ssd.D"), z=T("ssd.z"),
                               dt_bias=T("ssd.dt_bias"), initial_states=T("ssd.init"), dt_softplus=True, return_final_states=True)
    assert rel(y, T("ssd.y")) < 2e-5 and rel(fin, T("ssd.fin")) < 2e-5
    assert rel(O.causal_conv1d_ref(T("conv.x"), T("conv.w"), T("conv.b"), activation="silu"), T("conv.y")) < 1e-6
    o, last = O.selective_scan_ref(T("ss.u"), T("ss.delta"), T("ss.A"), T("ss.B"), T("ss.C"), T("ss.D"), T("ss.z"), T("ss.db"), True, True)
    assert rel(o, T("ss.out")) < 1e-6 and rel(last, T("ss.last")) < 1e-6
    assert rel(O.rmsnorm_gated_ref(T("norm.x"), T("norm.w"), None, T("norm.z"), eps=1e-5, group_size=32, norm_before_gate=False), T("norm.gated")) < 1e-6


def test_kernels_reproduce_golden(dev):
    from omnimamba_amd.causal_conv1d import causal_conv1d_fn
    from omnimamba_amd.layer_norm import rms_norm_fn
    from omnimamba_amd.layernorm_gated import rmsnorm_fn
    from omnimamba_amd.selective_scan import selective_scan_fn
    from omnimamba_amd.selective_state_update import selective_state_update
    from omnimamba_amd.ssd_combined import ssd_scan_fwd
    d = lambda k: T(k).to(dev)
    H, P, N = 4, 8, 16
    y, _, fin = ssd_scan_fwd(d("ssd.x"), d("ssd.dt"), d("ssd.A"), d("ssd.B"), d("ssd.C"), D=d("ssd.D"), z=d("ssd.z"), dt_bias=d("ssd.dt_bias"),
                             initial_states=d("ssd.init"), dt_softplus=True, return_final_states=True)
    assert rel(y, T("ssd.y")) < 3e-5 and rel(fin, T("ssd.fin")) < 3e-5
    st = fin.clone()
    ys = selective_state_update(st, d("su.x"), d("su.dt")[..., None].expand(1, H, P), d("ssd.A")[:, None, None].expand(H, P, N), d("su.B"), d("su.C"),
                                D=d("ssd.D")[:, None].expand(H, P), dt_bias=d("ssd.dt_bias")[:, None].expand(H, P), dt_softplus=True)
    assert rel(ys, T("su.y")) < 3e-5 and rel(st, T("su.state")) < 3e-5
    assert rel(causal_conv1d_fn(d("conv.x"), d("conv.w"), d("conv.b"), activation="silu"), T("conv.y")) < 1e-5
    o, last = selective_scan_fn(d("ss.u"), d("ss.delta"), d("ss.A"), d("ss.B"), d("ss.C"), d("ss.D"), d("ss.z"), d("ss.db"), True, True)
    assert rel(o, T("ss.out")) < 2e-5 and rel(last, T("ss.last")) < 2e-5
    assert rel(rmsnorm_fn(d("norm.x"), d("norm.w"), None, z=d("norm.z"), eps=1e-5, group_size=32, norm_before_gate=False), T("norm.gated")) < 1e-5
    assert rel(rms_norm_fn(d("norm.x"), d("norm.w"), None, residual=d("norm.res"), eps=1e-5), T("norm.add")) < 1e-5

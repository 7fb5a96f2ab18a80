"""The Mamba-1 mixer module (mamba_ssm.modules.mamba_simple.Mamba drop-in), mamba_inner_fn and selective_scan_ref of the
facade: forward + every parameter gradient against a composition of the CPU oracle's ops; prefill + steps against the
no-cache forward; the torch-form selective_scan_ref against the oracle's; and the independent restatement shipped in
`transformers` (MambaMixer.slow_forward) as a cross-check of the wiring."""
import pytest
import torch
import torch.nn.functional as F

import oracle as O


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def oracle_forward(m, h):
    """in_proj -> conv1d + SiLU -> x_proj -> dt_proj -> selective_scan_ref -> out_proj with the oracle's ops."""
    B, L, _ = h.shape
    xz = F.linear(h, m.in_proj.weight).transpose(1, 2)
    x, z = xz.chunk(2, dim=1)
    x = O.causal_conv1d_ref(x, m.conv1d.weight.squeeze(1), m.conv1d.bias, activation="silu")
    x_dbl = F.linear(x.transpose(1, 2), m.x_proj.weight)
    dt, Bm, Cm = torch.split(x_dbl, [m.dt_rank, m.d_state, m.d_state], dim=-1)
    dt = F.linear(dt, m.dt_proj.weight).transpose(1, 2)
    y = O.selective_scan_ref(x, dt, -torch.exp(m.A_log.float()), Bm.transpose(1, 2), Cm.transpose(1, 2), m.D.float(), z,
                             m.dt_proj.bias.float(), True)
    return F.linear(y.transpose(1, 2), m.out_proj.weight)


def test_mamba1_forward_backward_vs_oracle(dev):
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(0)
    m = Mamba(32, d_state=8, layer_idx=0)
    h = torch.randn(2, 70, 32)
    ref_m = Mamba(32, d_state=8, layer_idx=0)
    ref_m.load_state_dict(m.state_dict())
    m = m.to(dev)
    hd = h.clone().to(dev).requires_grad_()
    out = m(hd)
    g = torch.randn(out.shape)
    out.backward(g.to(dev))
    hr = h.clone().requires_grad_()
    o0 = oracle_forward(ref_m, hr)
    o0.backward(g)
    assert rel(out.detach(), o0.detach()) < 2e-5 and rel(hd.grad, hr.grad) < 3e-4
    for (n, p), (_, q) in zip(m.named_parameters(), ref_m.named_parameters()):
        assert p.grad is not None and rel(p.grad, q.grad) < 3e-4, n


def test_mamba1_prefill_then_steps_equal_full_forward(dev):
    from mamba_ssm.modules.mamba_simple import Mamba
    from omnimamba_amd.generation import InferenceParams
    torch.manual_seed(1)
    m = Mamba(32, d_state=16, layer_idx=0).to(dev).eval()
    h = torch.randn(2, 13, 32).to(dev)
    with torch.no_grad():
        full = m(h)
        ip = InferenceParams(max_seqlen=32, max_batch_size=2)
        pre = m(h[:, :9], inference_params=ip)
        outs = [pre]
        for t in range(9, 13):
            ip.seqlen_offset = t
            outs.append(m(h[:, t:t + 1], inference_params=ip))
    assert rel(torch.cat(outs, 1), full) < 2e-5


def test_selective_scan_ref_of_the_facade_matches_oracle():
    from mamba_ssm.ops.selective_scan_interface import selective_scan_ref
    torch.manual_seed(2)
    u, dl, A = torch.randn(2, 6, 33), torch.rand(2, 6, 33) * 0.5, -(torch.rand(6, 4) + 0.1)
    Bg, Cg, D, z, db = torch.randn(2, 2, 4, 33), torch.randn(2, 2, 4, 33), torch.randn(6), torch.randn(2, 6, 33), torch.randn(6) * 0.1
    a, la = selective_scan_ref(u, dl, A, Bg, Cg, D, z, db, True, True)
    b, lb = O.selective_scan_ref(u, dl, A, Bg, Cg, D, z, db, True, True)
    assert rel(a, b) < 1e-6 and rel(la, lb) < 1e-6
    a2 = selective_scan_ref(u, dl, A, torch.randn(6, 4), Cg[:, 0], None, None, None, False)
    assert a2.shape == u.shape and torch.isfinite(a2).all()


def test_mamba1_matches_transformers_restatement(dev):
    """Cross-check of the mixer wiring against the pure-PyTorch MambaMixer.slow_forward in `transformers` (an unrelated
    third-party restatement of the same upstream module)."""
    tm = pytest.importorskip("transformers.models.mamba.modeling_mamba")
    from transformers import MambaConfig
    from mamba_ssm.modules.mamba_simple import Mamba
    torch.manual_seed(3)
    cfg = MambaConfig(hidden_size=32, state_size=8, conv_kernel=4, expand=2, time_step_rank=2, use_bias=False, use_conv_bias=True,
                      num_hidden_layers=1, vocab_size=8)
    try:
        hf = tm.MambaMixer(cfg, layer_idx=0).eval()
    except Exception as e:        # constructor signatures move between transformers versions
        pytest.skip(f"MambaMixer not constructible here: {e}")
    m = Mamba(32, d_state=8, dt_rank=2, layer_idx=0)
    sd = {k: v for k, v in hf.state_dict().items() if k in m.state_dict()}
    assert set(sd) == set(m.state_dict())
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    h = torch.randn(2, 21, 32)
    with torch.no_grad():
        try:
            want = (hf._slow_forward if hasattr(hf, "_slow_forward") else hf.slow_forward)(h)
        except Exception as e:
            pytest.skip(f"slow_forward signature differs: {e}")
        got = m(h.to(dev))
    assert rel(got, want) < 2e-5

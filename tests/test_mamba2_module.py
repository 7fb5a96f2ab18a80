"""Mamba2 module (drop-in for mamba_ssm.modules.mamba2.Mamba2) vs the oracle block: fused training path, prefill with
cache + decode steps, gradients of every parameter.  Emulator on CPU; MI355X under -m gpu."""
import pytest
import torch

import oracle as O


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def build(dev, d_model=32, headdim=8, d_state=16, ngroups=1, dtype=torch.float32, **kw):
    from mamba_ssm.modules.mamba2 import Mamba2   # through the facade, like the reference imports it
    torch.manual_seed(0)
    m = Mamba2(d_model, d_state=d_state, headdim=headdim, ngroups=ngroups, chunk_size=16, layer_idx=0, **kw)
    with torch.no_grad():
        m.D.copy_(torch.randn_like(m.D))
        m.norm.weight.copy_(torch.randn_like(m.norm.weight) * 0.2 + 1.0)
    p = O.Mamba2RefParams(in_proj_weight=m.in_proj.weight.detach().clone(), conv_weight=m.conv1d.weight.detach().squeeze(1).clone(),
                          conv_bias=m.conv1d.bias.detach().clone(), dt_bias=m.dt_bias.detach().clone(), A_log=m.A_log.detach().clone(),
                          D=m.D.detach().clone(), norm_weight=m.norm.weight.detach().clone(), out_proj_weight=m.out_proj.weight.detach().clone(),
                          headdim=headdim, d_state=d_state, ngroups=ngroups, chunk_size=16)
    return m.to(dev), p


def test_state_dict_keys():
    from mamba_ssm.modules.mamba2 import Mamba2
    m = Mamba2(64)
    assert sorted(m.state_dict().keys()) == sorted(["in_proj.weight", "conv1d.weight", "conv1d.bias", "dt_bias", "A_log", "D",
                                                    "norm.weight", "out_proj.weight"])
    assert m.conv1d.weight.shape == (2 * 64 + 2 * 128, 1, 4) and m.in_proj.weight.shape == (2 * 128 + 2 * 128 + 2, 64)
    assert all(getattr(getattr(m, n), "_no_weight_decay", False) for n in ("dt_bias", "A_log", "D"))


@pytest.mark.parametrize("ngroups", [1, 2])
def test_forward_backward_vs_oracle(dev, ngroups):
    m, p = build(dev, ngroups=ngroups)
    torch.manual_seed(1)
    u = torch.randn(2, 37, 32)
    ur = u.clone().to(dev).requires_grad_()
    y = m(ur)
    g = torch.randn(y.shape)
    y.backward(g.to(dev))
    # oracle with autograd in fp64
    pd = O.Mamba2RefParams(**{k: (v.double().requires_grad_() if torch.is_tensor(v) else v) for k, v in p.__dict__.items()})
    ud = u.double().requires_grad_()
    y0 = O.mamba2_forward_ref(pd, ud, compute_dtype=torch.float64)
    y0.backward(g.double())
    assert rel(y.detach(), y0.detach()) < 1e-4
    assert rel(ur.grad, ud.grad) < 1e-3
    pairs = [(m.in_proj.weight, pd.in_proj_weight), (m.conv1d.weight.squeeze(1), pd.conv_weight), (m.conv1d.bias, pd.conv_bias),
             (m.dt_bias, pd.dt_bias), (m.A_log, pd.A_log), (m.D, pd.D), (m.norm.weight, pd.norm_weight), (m.out_proj.weight, pd.out_proj_weight)]
    names = ["in_proj", "conv_w", "conv_b", "dt_bias", "A_log", "D", "norm_w", "out_proj"]
    for n, (a, b) in zip(names, pairs):
        ga = a.grad if a.grad is not None else m.conv1d.weight.grad.squeeze(1)
        assert rel(ga, b.grad) < 1e-3, n


def test_prefill_then_decode_matches_full(dev):
    from types import SimpleNamespace
    m, p = build(dev)
    torch.manual_seed(2)
    u = torch.randn(2, 20, 32)
    with torch.no_grad():
        full = m(u.to(dev))
        ip = SimpleNamespace(key_value_memory_dict={}, seqlen_offset=0, max_seqlen=64, max_batch_size=2, lengths_per_sample=None)
        # garbage in the caches first: prefill must fully overwrite them (generation.py captures the graph before prefill)
        cs, ss = m.allocate_inference_cache(2, 64)
        cs.fill_(7.0)
        ss.fill_(-3.0)
        ip.key_value_memory_dict[0] = (cs, ss)
        outs = [m(u[:, :13].to(dev), inference_params=ip)]
        for t in range(13, 20):
            ip.seqlen_offset = t
            outs.append(m(u[:, t:t + 1].to(dev), inference_params=ip))
        got = torch.cat(outs, 1)
    y0 = O.mamba2_forward_ref(p, u)
    assert rel(full, y0) < 1e-4 and rel(got, y0) < 1e-4


def test_recompute_flag_gives_identical_gradients(dev, monkeypatch):
    """OMK_RECOMPUTE=1 (upstream's memory behaviour: conv and norm outputs rebuilt in backward) and the default (kept)
    must be the same arithmetic: bit-identical outputs and gradients."""
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("OMK_RECOMPUTE", flag)
        m, _ = build(dev)
        torch.manual_seed(3)
        u = torch.randn(2, 37, 32).to(dev).requires_grad_()
        y = m(u)
        y.backward(torch.ones_like(y))
        res.append([y.detach().clone(), u.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    for i, (a, b) in enumerate(zip(*res)):
        if dev.type == "cpu" or i == 0:
            assert torch.equal(a, b)          # same kernels on the same values
        else:
            assert rel(a, b) < 1e-5           # GPU: weight gradients fold through fp32 atomics (order varies run to run)


def test_frozen_out_proj_forms_no_weight_gradient(dev):
    """A frozen out_proj (the reference's 'align' stage freezes every base projection) gets no gradient and keeps no norm
    output; every other gradient is the one of the unfrozen module."""
    m, _ = build(dev)
    torch.manual_seed(2)
    u = torch.randn(2, 37, 32)
    g = torch.randn(2, 37, 32)
    grads = {}
    for frozen in (False, True):
        m.zero_grad(set_to_none=True)
        m.out_proj.weight.requires_grad_(not frozen)
        ud = u.clone().to(dev).requires_grad_()
        m(ud).backward(g.to(dev))
        grads[frozen] = {n: (None if p.grad is None else p.grad.detach().cpu().clone()) for n, p in m.named_parameters()}
        grads[frozen]["u"] = ud.grad.cpu()
    assert grads[True]["out_proj.weight"] is None and grads[False]["out_proj.weight"] is not None
    for n, v in grads[False].items():
        if n != "out_proj.weight":
            assert rel(grads[True][n], v) < 1e-5, n      # (not bit-equal on the GPU: the conv weight gradient is summed with atomics)


@pytest.mark.parametrize("L", [13, 2])
def test_fused_prefill_fills_both_states_like_the_unfused_branch(dev, monkeypatch, L):
    """the prefill of a cached decode through the fused node (conv1d + SiLU with the conv_state fill in
    its epilogue -> scan with the final state -> gated norm -> out_proj) == upstream's unfused prefill branch (A.2 of the survey):
    same output, same ssm_state, and the same conv_state in ALL d_conv columns (left zero padded when L < d_conv)."""
    from types import SimpleNamespace
    m, p = build(dev)
    torch.manual_seed(4)
    u = torch.randn(2, L, 32).to(dev)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("OMK_FUSED_PREFILL", flag)
        with torch.no_grad():
            ip = SimpleNamespace(key_value_memory_dict={}, seqlen_offset=0, max_seqlen=64, max_batch_size=2, lengths_per_sample=None)
            cs, ss = m.allocate_inference_cache(2, 64)
            cs.fill_(7.0)
            ss.fill_(-3.0)
            ip.key_value_memory_dict[0] = (cs, ss)
            out = m(u, inference_params=ip)
        res[flag] = (out.cpu(), cs.cpu().clone(), ss.cpu().clone())
    assert rel(res["1"][0], res["0"][0]) < 1e-6
    assert torch.equal(res["1"][1], res["0"][1])          # copies of the same inputs: bit-equal, zero padding included
    assert rel(res["1"][2], res["0"][2]) < 1e-6
    if L < 4:
        assert (res["1"][1][:, :, : 4 - L] == 0).all()

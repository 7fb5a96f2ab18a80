"""omk_norm_linear (fused norm + projection + LoRA of the decode step) vs the plain PyTorch composition."""
import pytest
import torch
import torch.nn.functional as F

import oracle as O


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,In,Out,wdtype,xdtype", [(1, 1024, 40, torch.float32, torch.float32), (1, 2048, 70, torch.float32, torch.float32),
                                                    (1, 1024, 33, torch.bfloat16, torch.bfloat16), (1, 1024, 600, torch.float32, torch.bfloat16)])
def test_prenorm_linear_lora(dev, B, In, Out, wdtype, xdtype):
    from omnimamba_amd.norm_linear import norm_linear
    x, res = torch.randn(B, In).to(xdtype), torch.randn(B, In)
    nw, W, bias = torch.rand(In) + 0.5, (torch.randn(Out, In) * 0.05).to(wdtype), torch.randn(Out)
    la, lb = torch.randn(8, In) * 0.05, torch.randn(Out, 8) * 0.05
    out, ro = norm_linear(x.to(dev), W.to(dev), bias.to(dev), norm_weight=nw.to(dev), eps=1e-5, residual=res.to(dev),
                          residual_out_dtype=torch.float32, lora_a=la.to(dev), lora_b=lb.to(dev), lora_scale=4.0)
    n0, r0 = O.add_norm_ref(x, nw, None, residual=res, eps=1e-5, prenorm=True, residual_in_fp32=True, is_rms_norm=True,
                            compute_dtype=torch.float64)
    n0 = n0.double()
    y0 = n0 @ W.double().t() + bias.double() + 4.0 * (n0 @ la.double().t()) @ lb.double().t()
    tol = 2e-5 if xdtype == torch.float32 and wdtype == torch.float32 else 6e-3
    assert out.dtype == xdtype and rel(ro, r0) < 1e-6
    # the oracle rounds n to the activation dtype (a separate kernel would); the fused kernel keeps it in fp32
    assert rel(out, y0) < (tol if xdtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("B,In,Out,G,nbg", [(1, 1024, 48, 1, False), (1, 2048, 48, 2, False), (1, 1024, 20, 1, True)])
def test_gated_norm_linear(dev, B, In, Out, G, nbg):
    from omnimamba_amd.norm_linear import norm_linear
    x, z = torch.randn(B, In), torch.randn(B, In)
    nw, W = torch.rand(In) + 0.5, torch.randn(Out, In) * 0.05
    out = norm_linear(x.to(dev), W.to(dev), None, norm_weight=nw.to(dev), eps=1e-5, z=z.to(dev), group_size=In // G,
                      norm_before_gate=nbg)
    n0 = O.rmsnorm_gated_ref(x, nw, z=z, eps=1e-5, group_size=In // G, norm_before_gate=nbg, compute_dtype=torch.float64)
    assert rel(out, n0.double() @ W.double().t()) < 2e-5


@pytest.mark.parametrize("In,Out,dtype,rdtype,with_lora,with_z", [
    (2048, 333, torch.float32, torch.float32, True, False), (4096, 77, torch.float32, torch.float32, False, True),
    (2048, 517, torch.bfloat16, torch.float32, True, False), (2048, 64, torch.bfloat16, torch.bfloat16, True, False),
    (4096, 130, torch.bfloat16, torch.float32, False, True), (1024, 1000, torch.bfloat16, torch.float32, True, True)])
def test_uniform_dtype_variant(dev, monkeypatch, In, Out, dtype, rdtype, with_lora, with_z):
    """The templated kernel the 1.3B decode takes (one dtype for activations, weights, norm weight and LoRA; fp32 or that
    dtype for the residual): against the fp64 composition, and against the generic kernel on the same inputs."""
    from omnimamba_amd.norm_linear import norm_linear
    x, res, z = torch.randn(1, In).to(dtype), torch.randn(1, In).to(rdtype), torch.randn(1, In).to(dtype)
    nw, W, bias = (torch.rand(In) + 0.5).to(dtype), (torch.randn(Out, In) * 0.05).to(dtype), torch.randn(Out).to(dtype)
    la, lb = (torch.randn(8, In) * 0.05).to(dtype), (torch.randn(Out, 8) * 0.05).to(dtype)
    kw = dict(norm_weight=nw.to(dev), eps=1e-5)
    if with_z:
        kw.update(z=z.to(dev))
    else:
        kw.update(residual=res.to(dev), residual_out_dtype=rdtype)
    if with_lora:
        kw.update(lora_a=la.to(dev), lora_b=lb.to(dev), lora_scale=4.0)
    r = norm_linear(x.to(dev), W.to(dev), bias.to(dev), **kw)
    monkeypatch.setenv("OMK_NORM_LINEAR_GENERIC", "1")
    g = norm_linear(x.to(dev), W.to(dev), bias.to(dev), **kw)
    out, outg = (r, g) if with_z else (r[0], g[0])
    xd = x.double()
    if with_z:
        q = xd * F.silu(z.double())
    else:
        q = xd + res.double()
        assert torch.equal(r[1].cpu(), g[1].cpu()) and rel(r[1], q) < (1e-6 if rdtype == torch.float32 else 5e-3)
    n0 = q * torch.rsqrt((q * q).mean() + 1e-5) * nw.double()
    y0 = n0 @ W.double().t() + bias.double()
    if with_lora:
        y0 = y0 + 4.0 * (n0 @ la.double().t()) @ lb.double().t()
    tol = 2e-5 if dtype == torch.float32 else 5e-3      # bf16: one output rounding
    assert out.dtype == dtype and rel(out, y0) < tol and rel(out, outg.double()) < tol


@pytest.mark.parametrize("dtype,W,S", [(torch.float32, 4, 4), (torch.bfloat16, 4, 4), (torch.float32, 3, 3), (torch.float32, 2, 1), (torch.bfloat16, 4, 3)])
def test_conv_tail(dev, dtype, W, S):
    """in_proj of Mamba2.step with the convolution update riding on it: output columns [off, off + C) = silu(conv) of the
    projected values, conv_state rolled in place -- three consecutive steps against projection + causal_conv1d_update_ref."""
    from omnimamba_amd.norm_linear import norm_linear, conv_tail_applies
    In, Out, C, off = 1024, 200, 120, 48
    nw, Wt = (torch.rand(In) + 0.5).to(dtype), (torch.randn(Out, In) * 0.05).to(dtype)
    cw, cb = (torch.randn(C, W) * 0.5).to(dtype), (torch.randn(C) * 0.2).to(dtype)
    cst = torch.randn(1, S, C).to(dtype).transpose(1, 2)             # channel-contiguous like the module's cache
    cst_d = cst.transpose(1, 2).contiguous().to(dev).transpose(1, 2)
    cst0 = cst.clone()
    for step in range(3):
        x, res = torch.randn(1, In).to(dtype), torch.randn(1, In)
        assert conv_tail_applies(x, Wt, nw, cst_d, cw, cb, residual=res)
        out, ro = norm_linear(x.to(dev), Wt.to(dev), None, norm_weight=nw.to(dev), eps=1e-5, residual=res.to(dev), residual_out_dtype=torch.float32,
                              conv_state=cst_d, conv_weight=cw.to(dev), conv_bias=cb.to(dev), conv_offset=off)
        q = x.double() + res.double()
        y0 = ((q * torch.rsqrt((q * q).mean() + 1e-5) * nw.double()) @ Wt.double().t()).to(dtype)     # zxbcdt as upstream stores it
        y0c = y0.clone()
        y0c[:, off:off + C] = O.causal_conv1d_update_ref(y0[:, off:off + C], cst0, cw, cb, activation="silu")
        tol = 3e-5 if dtype == torch.float32 else 1e-2
        assert rel(out, y0c.double()) < tol, step
        assert rel(cst_d, cst0.double()) < (1e-6 if dtype == torch.float32 else 6e-3), step   # bf16: the stored input may round differently by an ulp


@pytest.mark.parametrize("B,In,Out,dtype,rdtype,mode", [
    (2, 2048, 333, torch.float32, torch.float32, "lora"), (3, 2048, 200, torch.bfloat16, torch.float32, "lora+conv"),
    (8, 2048, 517, torch.float32, torch.float32, "lora+conv"), (8, 4096, 130, torch.bfloat16, torch.bfloat16, "gate"),
    (4, 4096, 77, torch.float32, torch.float32, "gate"), (5, 1024, 90, torch.bfloat16, torch.float32, "lora")])
def test_batched_variant(dev, B, In, Out, dtype, rdtype, mode):
    """Two to eight sequences per call (decode batches): every weight vector is used for all sequences while it is in
    registers.  Against the fp64 composition per sequence; with the conv tail, two consecutive steps."""
    from omnimamba_amd.norm_linear import norm_linear, applies
    C, off, W, S = 64, 16, 4, 4
    nw, Wt, bias = (torch.rand(In) + 0.5).to(dtype), (torch.randn(Out, In) * 0.05).to(dtype), torch.randn(Out).to(dtype)
    la, lb = (torch.randn(8, In) * 0.05).to(dtype), (torch.randn(Out, 8) * 0.05).to(dtype)
    cw, cb = (torch.randn(C, W) * 0.5).to(dtype), (torch.randn(C) * 0.2).to(dtype)
    cst = torch.randn(B, S, C).to(dtype).transpose(1, 2)
    cst_d, cst0 = cst.transpose(1, 2).contiguous().to(dev).transpose(1, 2), cst.clone()
    for step in range(2 if "conv" in mode else 1):
        x, res, z = torch.randn(B, In).to(dtype), torch.randn(B, In).to(rdtype), torch.randn(B, In).to(dtype)
        kw = dict(norm_weight=nw.to(dev), eps=1e-5)
        if mode == "gate":
            kw.update(z=z.to(dev))
            q = x.double() * F.silu(z.double())
        else:
            kw.update(residual=res.to(dev), residual_out_dtype=rdtype)
            q = x.double() + res.double()
        if "lora" in mode:
            kw.update(lora_a=la.to(dev), lora_b=lb.to(dev), lora_scale=4.0)
        if "conv" in mode:
            kw.update(conv_state=cst_d, conv_weight=cw.to(dev), conv_bias=cb.to(dev), conv_offset=off)
        assert applies(x.to(dev), Wt.to(dev), nw.to(dev), la.to(dev) if "lora" in mode else None, bias.to(dev))
        r = norm_linear(x.to(dev), Wt.to(dev), bias.to(dev), **kw)
        out = r if mode == "gate" else r[0]
        n0 = q * torch.rsqrt((q * q).mean(-1, keepdim=True) + 1e-5) * nw.double()
        if dtype == torch.bfloat16:
            n0 = n0.to(dtype).double()                      # the batched kernel keeps u in bf16 (upstream's rounding point)
        y0 = n0 @ Wt.double().t() + bias.double()
        if "lora" in mode:
            y0 = y0 + 4.0 * (n0 @ la.double().t()) @ lb.double().t()
        if mode != "gate":
            assert rel(r[1], q) < (1e-6 if rdtype == torch.float32 else 5e-3)
        if "conv" in mode:
            y0 = y0.to(dtype)
            y0[:, off:off + C] = O.causal_conv1d_update_ref(y0[:, off:off + C].clone(), cst0, cw, cb, activation="silu")
            assert rel(cst_d, cst0.double()) < (1e-6 if dtype == torch.float32 else 6e-3), step
        tol = 3e-5 if dtype == torch.float32 else 1e-2
        assert out.shape == (B, Out) and out.dtype == dtype and rel(out, y0.double()) < tol, step


@pytest.mark.parametrize("wgs,dtype", [(1, torch.bfloat16), (3, torch.bfloat16), (4, torch.bfloat16), (3, torch.float32), (4, torch.float32)])
def test_batched_matrix_form_walks_several_tiles_per_workgroup(dev, wgs, dtype, monkeypatch):
    """Two to eight sequences (bf16 weights; fp32 weights with rows of up to 2048 features): a workgroup of norm_linear_mfma_kernel walks the tiles blockIdx.x, + gridDim.x, ... with two
    register sets of weights and finish operands in flight (the 1.3B in_proj: 532 tiles on 256 workgroups).  Forced here on a small matrix
    (OMK_NL_MFMA_WGS): 13 tiles on 1 / 3 / 4 workgroups -- odd and even tile counts per workgroup, a ragged last tile, LoRA, conv tail over
    two steps; and the same call through the vector form (OMK_NL_MFMA=0 is read once per process, so the comparison is with the composition)."""
    from omnimamba_amd.norm_linear import norm_linear
    monkeypatch.setenv("OMK_NL_MFMA_WGS", str(wgs))
    monkeypatch.setenv("OMK_NL_MFMA_F32", "2")        # fp32 weights: the matrix form at any batch (by default only where it is ahead: 8 sequences + LoRA)
    torch.manual_seed(3)
    B, In, Out, C, off, W, S = 5, 2048, 200, 100, 60, 4, 3
    f32 = dtype == torch.float32
    nw, Wt, bias = (torch.rand(In) + 0.5).to(dtype), (torch.randn(Out, In) * 0.05).to(dtype), torch.randn(Out).to(dtype)
    la, lb = (torch.randn(8, In) * 0.05).to(dtype), (torch.randn(Out, 8) * 0.05).to(dtype)
    cw, cb = (torch.randn(C, W) * 0.5).to(dtype), (torch.randn(C) * 0.2).to(dtype)
    cst = torch.randn(B, S, C).to(dtype).transpose(1, 2)
    cst_d, cst0 = cst.transpose(1, 2).contiguous().to(dev).transpose(1, 2), cst.clone()
    for step in range(2):
        x, res = torch.randn(B, In).to(dtype), torch.randn(B, In)
        out, ro = norm_linear(x.to(dev), Wt.to(dev), bias.to(dev), norm_weight=nw.to(dev), eps=1e-5, residual=res.to(dev), residual_out_dtype=torch.float32,
                              lora_a=la.to(dev), lora_b=lb.to(dev), lora_scale=4.0, conv_state=cst_d, conv_weight=cw.to(dev), conv_bias=cb.to(dev), conv_offset=off)
        q = x.double() + res.double()
        n0 = (q * torch.rsqrt((q * q).mean(-1, keepdim=True) + 1e-5) * nw.double()).to(dtype).double()
        y0 = (n0 @ Wt.double().t() + bias.double() + 4.0 * (n0 @ la.double().t()) @ lb.double().t()).to(dtype)
        y0[:, off:off + C] = O.causal_conv1d_update_ref(y0[:, off:off + C].clone(), cst0, cw, cb, activation="silu")
        assert rel(ro, q) < 1e-6 and rel(cst_d, cst0.double()) < (1e-6 if f32 else 6e-3), step
        assert rel(out, y0.double()) < (3e-5 if f32 else 1e-2), step
        assert float((out.double().cpu() - y0.double()).abs().max()) < (2e-3 if f32 else 0.08), step       # no row or sequence left out / taken twice


def test_residual_out_without_incoming_residual(dev):
    """First block of a stack: no residual yet, residual_out must still be x (in the requested dtype)."""
    from omnimamba_amd.norm_linear import norm_linear
    x, nw, W = torch.randn(1, 1024), torch.rand(1024) + 0.5, torch.randn(16, 1024) * 0.05
    out, ro = norm_linear(x.to(dev), W.to(dev), None, norm_weight=nw.to(dev), eps=1e-5, residual_out_dtype=torch.float32)
    n0, r0 = O.add_norm_ref(x, nw, None, residual=None, eps=1e-5, prenorm=True, residual_in_fp32=True, is_rms_norm=True,
                            compute_dtype=torch.float64)
    assert rel(ro, x) < 1e-7 and rel(out, n0.double() @ W.double().t()) < 2e-5


def test_plain_linear_and_limits(dev):
    from omnimamba_amd import norm_linear as NL
    x, W = torch.randn(1, 1024), torch.randn(10, 1024)
    out = NL.norm_linear(x.to(dev), W.to(dev))
    assert rel(out, F.linear(x, W)) < 2e-5
    assert NL.applies(x, W) and not NL.applies(torch.randn(9, 1024), W) and not NL.applies(torch.randn(1, 256), torch.randn(10, 256))
    assert not NL.applies(torch.randn(2, 1024), W) and NL.applies(torch.randn(2, 1024), W, torch.ones(1024))     # batches need the norm weight
    assert not NL.applies(x, W.requires_grad_())

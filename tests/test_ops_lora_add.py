"""omk_lora_add (result += scaling * h @ lora_B^T, in place) vs the plain composition, forward and gradients."""
import pytest
import torch


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("T,N,R,dtype", [(300, 8512, 8, torch.bfloat16), (70, 264, 16, torch.bfloat16), (33, 96, 8, torch.float32)])
def test_lora_add(dev, T, N, R, dtype):
    from omnimamba_amd import lora_add as LA
    torch.manual_seed(0)
    res, h = torch.randn(T, N).to(dtype), torch.randn(T, R).to(dtype)
    Bw = torch.randn(N, R) * 0.1
    resd = res.clone().to(dev).requires_grad_()
    hd, Bd = h.clone().to(dev).requires_grad_(), Bw.clone().to(dev).requires_grad_()
    work = resd * 1.0                                    # a non-leaf buffer the op may overwrite
    assert LA.applies(work, hd, Bd)
    out = LA.lora_add(work, hd, Bd, 4.0)
    ref = res.double() + 4.0 * h.double() @ Bw.double().t()
    tol = 2e-6 if dtype == torch.float32 else 6e-3
    assert out.dtype == dtype and rel(out, ref) < tol
    g = torch.randn(T, N).to(dtype)
    out.backward(g.to(dev))
    assert rel(resd.grad, g.double()) < 1e-6
    assert rel(hd.grad, 4.0 * g.double() @ Bw.double()) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert rel(Bd.grad, 4.0 * g.double().t() @ h.double()) < (1e-5 if dtype == torch.float32 else 1e-2) and Bd.grad.dtype == torch.float32


def test_lora_add_limits(dev):
    from omnimamba_amd import lora_add as LA
    assert not LA.applies(torch.randn(4, 96), torch.randn(4, 4), torch.randn(96, 4))       # rank 4: addmm
    assert not LA.applies(torch.randn(4, 98).bfloat16(), torch.randn(4, 8).bfloat16(), torch.randn(98, 8))


@pytest.mark.parametrize("T,N,R,dtype", [(300, 2048, 8, torch.bfloat16), (70, 264, 16, torch.bfloat16), (33, 96, 8, torch.float32)])
def test_lora_add_masked(dev, T, N, R, dtype):
    """The masked form (dropout backward of the LoRA A branch): only elements whose mask byte is set receive the update."""
    from omnimamba_amd import lora_add as LA
    torch.manual_seed(1)
    out, h = torch.randn(T, N).to(dtype), torch.randn(T, R).to(dtype)
    Bw = (torch.randn(N, R) * 0.1).to(dtype)
    mask = torch.rand(T, N) < 0.8
    od = out.clone().to(dev)
    LA.lora_add_(od, h.to(dev), Bw.to(dev), 1.25, mask.to(dev))
    ref = out.double() + mask.double() * 1.25 * (h.double() @ Bw.double().t())
    assert rel(od, ref) < (2e-6 if dtype == torch.float32 else 6e-3)
    assert torch.equal(od.cpu()[~mask], out[~mask])          # untouched elements are bit-identical


@pytest.mark.parametrize("T,N", [(300, 8512), (64, 256), (1000, 520)])
def test_lora_up_bwd(dev, T, N):
    """omk_lora_up_bwd: dh = dy B and dB = dy^T h from one pass over dy, vs fp64 (ragged token tail, partial last column block)."""
    from omnimamba_amd import lora_add as LA
    torch.manual_seed(3)
    dy, h = torch.randn(T, N).bfloat16(), torch.randn(T, 8).bfloat16()
    Bw = torch.randn(N, 8) * 0.1
    assert LA.up_bwd_applies(dy.to(dev), h.to(dev), Bw.to(dev))
    dh, db = LA.lora_up_bwd(dy.to(dev), h.to(dev), Bw.to(dev))
    Bq = Bw.bfloat16().double()                      # the kernel holds lora_b as bf16 MFMA fragments
    assert dh.dtype == torch.float32 and rel(dh, dy.double() @ Bq) < 1e-5
    assert db.dtype == torch.float32 and rel(db, dy.double().t() @ h.double()) < 1e-5

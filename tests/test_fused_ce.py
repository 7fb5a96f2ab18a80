"""Row f1 of SURVEY.md section 8: omk_cross_entropy (loss + in-place gradient over a block of logits) and the chunked fused
linear + cross-entropy built on it, against torch's own cross_entropy / autograd on identical inputs."""
import pytest
import torch
import torch.nn.functional as F


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype,V", [(torch.float32, 100), (torch.float32, 50288), (torch.bfloat16, 16384), (torch.bfloat16, 136)])
def test_cross_entropy_kernel(dev, dtype, V):
    from omnimamba_amd.fused_ce import cross_entropy_inplace
    torch.manual_seed(0)
    T = 7
    logits = (torch.randn(T, V) * 3).to(dtype)
    labels = torch.randint(0, V, (T,))
    labels[2] = -100
    scale = torch.tensor([0.25])
    ref = logits.float().clone().requires_grad_()
    l0 = F.cross_entropy(ref, labels, ignore_index=-100, reduction="none")
    (l0.sum() * 0.25).backward()
    work = logits.clone().to(dev)
    losses = cross_entropy_inplace(work, labels.to(dev), scale.to(dev))
    assert rel(losses, l0.detach()) < 1e-5 and float(losses[2]) == 0.0
    assert rel(work.float(), ref.grad) < (1e-5 if dtype == torch.float32 else 4e-3)
    assert float(work[2].float().abs().max()) == 0.0                    # ignored row: zero gradient
    only = cross_entropy_inplace(logits.clone().to(dev), labels.to(dev), None, write_grad=False)
    assert rel(only, l0.detach()) < 1e-5


@pytest.mark.parametrize("train_w", [True, False])
def test_fused_linear_cross_entropy_matches_materialised(dev, train_w, monkeypatch):
    """Several token blocks (the block size is forced down), ignored labels, weight gradient on / off."""
    from omnimamba_amd import fused_ce
    monkeypatch.setattr(fused_ce, "_block_tokens", lambda vocab, elem: 256)
    torch.manual_seed(1)
    T, d, V = 600, 16, 96
    h0, w0 = torch.randn(T, d), torch.randn(V, d) * 0.5
    labels = torch.randint(0, V, (T,))
    labels[::7] = -100
    h, w = h0.clone().to(dev).requires_grad_(), w0.clone().to(dev).requires_grad_(train_w)
    loss = fused_ce.fused_linear_cross_entropy(h, w, labels.to(dev))
    (loss * 3.0).backward()
    hr, wr = h0.clone().requires_grad_(), w0.clone().requires_grad_(train_w)
    l0 = F.cross_entropy(F.linear(hr, wr), labels, ignore_index=-100)
    (l0 * 3.0).backward()
    assert rel(loss.detach(), l0.detach()) < 1e-5 and rel(h.grad, hr.grad) < 1e-4
    if train_w:
        assert rel(w.grad, wr.grad) < 1e-4
    else:
        assert w.grad is None


def test_fused_linear_cross_entropy_edges(dev):
    """Ids outside [0, V) are scored like ignored ones AND left out of the mean (torch asserts on them); nothing to score: loss 0."""
    from omnimamba_amd import fused_ce
    torch.manual_seed(2)
    T, d, V = 40, 16, 96
    h0, w0 = torch.randn(T, d), torch.randn(V, d) * 0.5
    labels = torch.randint(0, V, (T,))
    labels[3], labels[9], labels[11] = V + 5, -3, -100
    h = h0.clone().to(dev).requires_grad_()
    loss = fused_ce.fused_linear_cross_entropy(h, w0.to(dev), labels.to(dev))
    loss.backward()
    clean = labels.clone()
    clean[3] = clean[9] = -100
    hr = h0.clone().requires_grad_()
    l0 = F.cross_entropy(F.linear(hr, w0), clean, ignore_index=-100)
    l0.backward()
    assert rel(loss.detach(), l0.detach()) < 1e-5 and rel(h.grad, hr.grad) < 1e-4
    assert float(h.grad[[3, 9, 11]].abs().max()) == 0.0
    none = fused_ce.fused_linear_cross_entropy(h0.to(dev), w0.to(dev), torch.full((T,), -100).to(dev))
    assert float(none) == 0.0
    with pytest.raises(TypeError):
        fused_ce.fused_linear_cross_entropy(h0.to(dev), w0.to(dev), labels.to(dev).int())


@pytest.mark.gpu
def test_fused_linear_cross_entropy_bf16_autocast_lm_head_size():
    """The MMU head at its real width (50 288) under bf16 autocast, two blocks: loss and gradients vs the materialised form."""
    from omnimamba_amd import fused_ce
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    T, d, V = 6144, 256, 50288
    h0 = torch.randn(T, d, device=dev)
    w0 = torch.randn(V, d, device=dev) * 0.05
    labels = torch.randint(0, 50277, (T,), device=dev)
    labels[:100] = -100
    h, w = h0.clone().requires_grad_(), w0.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = fused_ce.fused_linear_cross_entropy(h, w, labels)
    loss.backward()
    hr, wr = h0.clone().requires_grad_(), w0.clone().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        l0 = F.cross_entropy(F.linear(hr, wr).float(), labels, ignore_index=-100)
    l0.backward()
    assert abs(float(loss) - float(l0)) < 2e-3 * float(l0)
    assert rel(h.grad, hr.grad) < 1e-2 and rel(w.grad, wr.grad) < 1e-2

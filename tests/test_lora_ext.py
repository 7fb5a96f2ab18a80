"""Base + task LoRA as one GEMM over the extended contraction dimension (omnimamba_amd/lora_ext.py) == the reference formula
(models/stage2/lora.py:263-279: result + B(A(dropout(x))) * scaling), forward and every gradient; two task forwards before one
backward (the reference's compute_loss order) must not disturb each other through the shared [W | B | 0] buffer."""
import pytest
import torch

from omnimamba_amd.stack import TaskLoRALinear


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _module(train_base):
    torch.manual_seed(0)
    m = TaskLoRALinear(64, 96, r=8, lora_dropout=0.0, dtype=torch.bfloat16)
    for t in ("mmu", "t2i"):
        torch.nn.init.normal_(getattr(m, f"{t}_lora_B0").weight, std=0.05)
    m.weight.requires_grad_(train_base)
    return m


def _formula(m, x, task):
    W, A, B = (t.detach().double().requires_grad_() for t in (m.weight, getattr(m, f"{task}_lora_A0").weight, getattr(m, f"{task}_lora_B0").weight))
    xd = x.detach().double().requires_grad_()
    return xd @ W.t() + m.scaling * (xd @ A.t()) @ B.t(), (xd, W, A, B)


@pytest.mark.parametrize("train_base", [False, True])
def test_lora_ext_matches_formula_two_tasks(train_base, monkeypatch):
    m = _module(train_base)
    x1 = torch.randn(4, 160, 64, dtype=torch.bfloat16, requires_grad=True)
    x2 = torch.randn(2, 300, 64, dtype=torch.bfloat16, requires_grad=True)
    g1, g2 = torch.randn(4, 160, 96, dtype=torch.bfloat16), torch.randn(2, 300, 96, dtype=torch.bfloat16)
    monkeypatch.setenv("OMK_LORA_EXT", "1")
    m.task_types = "t2i"
    y1 = m(x1)
    assert type(y1.grad_fn).__name__ in ("ViewBackward0", "_WGradFnBackward", "UnsafeViewBackward0", "_LoraExtFnBackward")
    m.task_types = "mmu"
    y2 = m(x2)                                   # rewrites the B columns of the shared buffer before y1's backward
    ((y1.float() * g1.float()).sum() + (y2.float() * g2.float()).sum()).backward()
    r1, l1 = _formula(m, x1, "t2i")
    r2, l2 = _formula(m, x2, "mmu")
    ((r1 * g1.double()).sum() + (r2 * g2.double()).sum()).backward()
    assert rel(y1, r1) < 4e-3 and rel(y2, r2) < 4e-3
    assert rel(x1.grad, l1[0].grad) < 6e-3 and rel(x2.grad, l2[0].grad) < 6e-3
    for task, leaves in (("t2i", l1), ("mmu", l2)):
        assert rel(getattr(m, f"{task}_lora_A0").weight.grad, leaves[2].grad) < 1.5e-2
        assert rel(getattr(m, f"{task}_lora_B0").weight.grad, leaves[3].grad) < 1.5e-2
    if train_base:
        assert rel(m.weight.grad, l1[1].grad + l2[1].grad) < 6e-3
    else:
        assert m.weight.grad is None
    # the buffer follows the master weight
    with torch.no_grad():
        m.weight.mul_(0.5)
    m.task_types = "t2i"
    y3 = m(x1.detach())
    assert rel(y3, _formula(m, x1, "t2i")[0]) < 4e-3
    assert "_omk_we" not in m.state_dict() and all("_omk" not in k for k in m.state_dict())


def test_lora_ext_equals_streaming_add_path(monkeypatch):
    m = _module(False)
    m.task_types = "mmu"
    x = torch.randn(1, 600, 64, dtype=torch.bfloat16)
    monkeypatch.setenv("OMK_LORA_EXT", "1")
    a = m(x)
    monkeypatch.setenv("OMK_LORA_EXT", "0")
    b = m(x)
    assert rel(a, b) < 5e-3


def test_lora_ext_dropout_backward_uses_the_mask():
    """With lora_dropout > 0 the A branch sees dropout(x): y and dx / dA / dB against the formula evaluated with the SAME mask
    (same generator state), the masked in-place update of dx (omk_lora_add with a mask) included."""
    torch.manual_seed(0)
    m = TaskLoRALinear(64, 96, r=8, lora_dropout=0.25, dtype=torch.bfloat16)
    torch.nn.init.normal_(m.mmu_lora_B0.weight, std=0.05)
    m.task_types = "mmu"
    m.train()
    x = torch.randn(3, 200, 64, dtype=torch.bfloat16, requires_grad=True)
    g = torch.randn(3, 200, 96, dtype=torch.bfloat16)
    torch.manual_seed(123)
    y = m(x)
    y.backward(g)
    torch.manual_seed(123)
    xd, mask = torch.ops.aten.native_dropout(x.detach().reshape(-1, 64), 0.25, True)
    assert 0.6 < mask.float().mean() < 0.9
    W, A, B = (t.detach().double().requires_grad_() for t in (m.weight, m.mmu_lora_A0.weight, m.mmu_lora_B0.weight))
    x64 = x.detach().double().reshape(-1, 64).requires_grad_()
    ref = x64 @ W.t() + m.scaling * ((x64 * mask.double() / 0.75) @ A.t()) @ B.t()
    ref.backward(g.double().reshape(-1, 96))
    assert rel(y.reshape(-1, 96), ref) < 4e-3
    assert rel(x.grad.reshape(-1, 64), x64.grad) < 6e-3
    assert rel(m.mmu_lora_A0.weight.grad, A.grad) < 1.5e-2 and rel(m.mmu_lora_B0.weight.grad, B.grad) < 1.5e-2


def test_align_stage_applies_lora_dropout_on_both_paths(monkeypatch):
    """Stage 'align' leaves the in_proj module in eval() and re-enables train() only on the modules named *lora*
    (reference omnimamba.py:137-151), so the dropout module's own mode -- not in_proj.training -- must decide: the reference
    (lora.py:270-274) calls self.lora_dropout(x) unconditionally.  Both the one-GEMM path and the streaming path drop."""
    torch.manual_seed(0)
    m = TaskLoRALinear(64, 96, r=8, lora_dropout=0.5, dtype=torch.bfloat16)
    torch.nn.init.normal_(m.mmu_lora_B0.weight, std=0.05)
    m.task_types = "mmu"
    m.eval()
    for n, sub in m.named_modules():
        if "lora" in n.lower():
            sub.train()
    assert not m.training and m.lora_dropout.training
    x = torch.randn(1, 600, 64, dtype=torch.bfloat16)
    for ext in ("1", "0"):
        monkeypatch.setenv("OMK_LORA_EXT", ext)
        torch.manual_seed(5)
        a = m(x)
        torch.manual_seed(6)
        b = m(x)
        assert rel(a, b) > 1e-3, f"OMK_LORA_EXT={ext}: two different dropout masks must give different outputs"
    m.lora_dropout.eval()
    for ext in ("1", "0"):
        monkeypatch.setenv("OMK_LORA_EXT", ext)
        assert rel(m(x), m(x)) == 0.0

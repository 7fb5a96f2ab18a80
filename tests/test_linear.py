"""omnimamba_amd.linear: F.linear with a token-split weight gradient (fills the CU rounds of the dW GEMM)."""
import pytest
import torch
import torch.nn.functional as F

from omnimamba_amd import linear as L


def test_split_factor_heuristic():
    dev = torch.device("cpu")   # no GPU: _n_cu falls back to 256 (MI355X)
    assert L.split_factor(8512, 2048, 32768, dev) == 4      # in_proj: 272 tiles -> 53 % of two rounds
    assert L.split_factor(2048, 4096, 32768, dev) == 2      # out_proj: 128 tiles = half a round
    assert L.split_factor(4096, 4096, 32768, dev) == 1      # 256 tiles: exactly one round
    assert L.split_factor(8512, 2048, 4096, dev) == 2       # slices never drop below 2048 tokens
    assert L.split_factor(8512, 2048, 2048, dev) == 1


def test_cpu_path_is_plain_linear():
    torch.manual_seed(0)
    x, w, b = torch.randn(5, 7, requires_grad=True), torch.randn(3, 7, requires_grad=True), torch.randn(3, requires_grad=True)
    y = L.linear(x, w, b)
    assert torch.equal(y, F.linear(x, w, b))
    dw = L.weight_grad(torch.randn(11, 3), torch.randn(11, 7), torch.float32)
    assert dw.shape == (3, 7)


@pytest.mark.gpu
@pytest.mark.parametrize("tokens,out_f,in_f", [(8192, 8512, 2048), (4096, 2048, 4096), (300, 96, 64)])
def test_gradients_match_autograd(tokens, out_f, in_f):
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(2, tokens // 2, in_f, device=dev, dtype=torch.bfloat16)
    w = torch.randn(out_f, in_f, device=dev) * 0.02
    b = torch.randn(out_f, device=dev)
    g = torch.randn(2, tokens // 2, out_f, device=dev, dtype=torch.bfloat16)
    res = []
    for fn in (L.linear, F.linear):
        xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn(xr, wr, br)
        y.backward(g)
        res.append((y.detach().float(), xr.grad.float(), wr.grad.float(), br.grad.float()))
        assert wr.grad.dtype == torch.float32 and xr.grad.dtype == torch.bfloat16
    exact = g.reshape(-1, out_f).double().t() @ x.reshape(-1, in_f).double()
    for a, r in zip(res[0][:2], res[1][:2]):
        assert torch.equal(a, r)                                   # same forward / dgrad GEMMs
    rel = lambda a: ((a.double() - exact).norm() / exact.norm()).item()
    assert rel(res[0][2]) <= rel(res[1][2]) * 1.05 + 1e-6          # fp32 partial sums: at least as accurate as one bf16-output GEMM
    assert rel(res[0][2]) < 3e-3
    assert ((res[0][3] - res[1][3]).norm() / res[1][3].norm()).item() < 1e-2


@pytest.mark.gpu
def test_weight_gradient_is_ready_before_the_input_gradient():
    """Two autograd nodes: the weight's AccumulateGrad (where DDP starts the bucket's all-reduce) runs before the dx GEMM,
    so the last weight gradient of backward still overlaps something."""
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    x = torch.randn(4, 512, 256, device=dev, dtype=torch.bfloat16, requires_grad=True)
    w = (torch.randn(384, 256, device=dev) * 0.02).requires_grad_()
    order = []
    w.register_post_accumulate_grad_hook(lambda p: order.append("w"))
    x.register_hook(lambda g: order.append("x"))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = L.linear(x, w)
    y.float().sum().backward()
    assert order == ["w", "x"], order
    # frozen weight (LoRA base in stage 1): no second node, plain dgrad
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = L.linear(x, w.detach())
        assert y2.grad_fn is not None and "XGrad" in type(y2.grad_fn).__name__
        # input without gradient (first layer on fixed embeddings): the weight still gets its gradient
        w3 = w.detach().clone().requires_grad_()
        y3 = L.linear(x.detach(), w3)
    y3.float().sum().backward()
    assert w3.grad is not None and w3.grad.shape == w3.shape

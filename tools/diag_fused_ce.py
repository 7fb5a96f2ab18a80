import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from omnimamba_amd import fused_ce
dev = torch.device("cuda:0")
for it in range(12):
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
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    print(it, float(loss), float(l0), abs(float(loss) - float(l0)) / float(l0), rel(h.grad, hr.grad), rel(w.grad, wr.grad), flush=True)

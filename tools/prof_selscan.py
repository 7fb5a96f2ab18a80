import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from omnimamba_amd.selective_scan import selective_scan_fn
dev = torch.device("cuda:0")
Bsz, Dm, L, N = 64, 768, 1024, 16
torch.manual_seed(0)
u, delta, z = (torch.randn(Bsz, Dm, L, device=dev) for _ in range(3))
A = -(torch.rand(Dm, N, device=dev) + 0.1)
Bm, Cm = torch.randn(Bsz, N, L, device=dev), torch.randn(Bsz, N, L, device=dev)
D, db = torch.randn(Dm, device=dev), 0.1 * torch.randn(Dm, device=dev)
leaves = [t.requires_grad_() for t in (u, delta, A, Bm, Cm, D, z, db)]
g = torch.randn_like(u)
for _ in range(4):
    for t in leaves: t.grad = None
    selective_scan_fn(*leaves, True).backward(g)
torch.cuda.synchronize()

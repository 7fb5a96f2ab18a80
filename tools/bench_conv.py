"""Developer tool: causal conv1d fwd / bwd timing at the 1.3B block shape (channel-last xBC slice of zxbcdt)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.causal_conv1d import causal_conv1d_fn  # noqa: E402

dev = torch.device("cuda:0")
B, L, C, W = 8, 4096, 4352, 4
zx = torch.randn(B, L, 8512, device=dev).bfloat16()
x = zx[..., 4096:4096 + C].transpose(1, 2).requires_grad_(True)   # (B, C, L) view, channel-last storage
w = torch.randn(C, W, device=dev, requires_grad=True)
b = torch.randn(C, device=dev, requires_grad=True)


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


f = lambda: causal_conv1d_fn(x, w, b, activation="silu")
tf = timeit(f)
out = f()
g = torch.randn_like(out)
tb = timeit(lambda: torch.autograd.grad(out, (x, w, b), g, retain_graph=True))
byt = B * L * C * 2
print(f"conv fwd {tf:7.1f} us ({2*byt/tf/1e6:6.0f} GB/s)   bwd {tb:7.1f} us ({3*byt/tb/1e6:6.0f} GB/s)")

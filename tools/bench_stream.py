"""Developer tool: the four streaming helpers of the block step (conv1d fwd / bwd on the channel-last xBC slice, gated RMSNorm fwd / bwd) at the
configs[1] shape, next to what plain torch element-wise kernels reach on the same access patterns (the achievable ceiling on this box)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.causal_conv1d import causal_conv1d_fn  # noqa: E402
from omnimamba_amd.layernorm_gated import rmsnorm_fn  # noqa: E402

dev = torch.device("cuda:0")
B, L, C, W = 8, 4096, 4352, 4


def timeit(fn, iters=20, warm=3):
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


def best(fn):
    return min(timeit(fn) for _ in range(3))


zx = torch.randn(B, L, 8512, device=dev).bfloat16()
xv = zx[..., 4096:4096 + C]
x = xv.transpose(1, 2).requires_grad_(True)   # (B, C, L) view, channel-last storage
w = torch.randn(C, W, device=dev, requires_grad=True)
b = torch.randn(C, device=dev, requires_grad=True)
byt = B * L * C * 2
cont = torch.empty(B, L, C, device=dev, dtype=torch.bfloat16)
a1, a2 = torch.randn(B * L * C, device=dev).bfloat16(), torch.empty(B * L * C, device=dev, dtype=torch.bfloat16)
t = best(lambda: torch.add(a1, a1, out=a2)); print(f"torch add contiguous (1 read + 1 write)        {t:7.1f} us  {2*byt/t/1e3:6.0f} GB/s")
t = best(lambda: cont.copy_(xv)); print(f"torch copy xBC slice -> contiguous             {t:7.1f} us  {2*byt/t/1e3:6.0f} GB/s")
t = best(lambda: torch.nn.functional.silu(a1, inplace=False)); print(f"torch silu contiguous                          {t:7.1f} us  {2*byt/t/1e3:6.0f} GB/s")
f = lambda: causal_conv1d_fn(x, w, b, activation="silu")
t = best(f); print(f"conv1d fwd + silu                              {t:7.1f} us  {2*byt/t/1e3:6.0f} GB/s")
t = best(lambda: causal_conv1d_fn(x, w, b, activation=None)); print(f"conv1d fwd, no activation                      {t:7.1f} us  {2*byt/t/1e3:6.0f} GB/s")
out = f()
g = torch.randn_like(out)
t = best(lambda: torch.autograd.grad(out, (x, w, b), g, retain_graph=True)); print(f"conv1d bwd (x, dout -> dx, dw, db)             {t:7.1f} us  {3*byt/t/1e3:6.0f} GB/s")
Dn = 4096
y = torch.randn(B * L, Dn, device=dev).bfloat16().requires_grad_(True)
z = zx.reshape(-1, 8512)[:, :Dn].detach().requires_grad_(True)
wn = torch.randn(Dn, device=dev, requires_grad=True)
nb = B * L * Dn * 2
fn = lambda: rmsnorm_fn(y, wn, None, z=z, eps=1e-5, group_size=Dn, norm_before_gate=False)
t = best(fn); print(f"gated rmsnorm fwd (y, z -> out)                {t:7.1f} us  {3*nb/t/1e3:6.0f} GB/s")
o = fn(); go = torch.randn_like(o)
t = best(lambda: torch.autograd.grad(o, (y, z, wn), go, retain_graph=True)); print(f"gated rmsnorm bwd (y, z, dout -> dy, dz, dw)    {t:7.1f} us  {5*nb/t/1e3:6.0f} GB/s")
y2 = torch.randn(B * L, Dn, device=dev).bfloat16(); z2 = torch.randn(B * L, Dn, device=dev).bfloat16(); o2 = torch.empty_like(y2)
t = best(lambda: torch.mul(y2, z2, out=o2)); print(f"torch mul contiguous (2 reads + 1 write)       {t:7.1f} us  {3*nb/t/1e3:6.0f} GB/s")
zs = zx.reshape(-1, 8512)[:, :Dn]
t = best(lambda: torch.mul(y2, zs, out=o2)); print(f"torch mul, z = strided slice of zxbcdt         {t:7.1f} us  {3*nb/t/1e3:6.0f} GB/s")

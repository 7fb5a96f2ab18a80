"""Developer diagnostic: time the backward scans with / without D (dD atomics) and with / without y (token scalars)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_fwd, ssd_scan_bwd


def timeit(fn, iters=3, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


dev = torch.device("cuda:0")
B, L, H, P, N, G = 8, 4096, 64, 64, 128, 1
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P)
Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N)
Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16()
A = -(torch.rand(H, device=dev) * 15 + 1)
D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
out, _, _ = ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)
dout = torch.randn_like(out)
for name, kw in [("y + D", dict(D=D, y=out)), ("y, no D", dict(D=None, y=out)), ("generic (no y)", dict(D=D))]:
    if name.startswith("generic") and "--generic" not in sys.argv:
        continue
    ms = timeit(lambda: ssd_scan_bwd(dout, x, dt, A, Bm, Cm, dt_bias=dtb, dt_softplus=True, **kw), 3, 1)
    print(f"bwd [{name}]: {ms*1e3:9.1f} us", flush=True)

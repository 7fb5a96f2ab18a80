"""Developer tool: backward scan time (B, L from argv) -- run through tools/with_lib.py for a same-box A/B."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_bwd  # noqa: E402
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 4096)
H, P, N, G = 64, 64, 128, 1
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
dout = torch.randn(B, L, H, P, device=dev).bfloat16()
ms = min(timeit(lambda: ssd_scan_bwd(dout, x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True), 10, 3) for _ in range(3))
print(f"bwd B={B} L={L}: {ms * 1e3:8.1f} us", flush=True)

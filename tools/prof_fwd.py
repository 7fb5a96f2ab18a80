"""Developer tool: launch only the SSD forward scan (configs[1] shape) a few times -- target for rocprofv3 --pmc."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_fwd

dev = torch.device("cuda:0")
B, L, H, P, N, G = int(os.environ.get("PB", "8")), int(os.environ.get("SEQ", "4096")), 64, 64, 128, 1
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P)
Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N)
Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16()
A = -(torch.rand(H, device=dev) * 15 + 1)
D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
for _ in range(int(os.environ.get("ITERS", "3"))):
    ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)
torch.cuda.synchronize()

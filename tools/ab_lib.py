"""Developer tool: same-box A/B of two builds of the library on the forward scan (interleaved repeats).
usage: python tools/ab_lib.py omnimamba_amd/lib/libomnimamba_hip_prev.so [B L]"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
import omnimamba_amd._lib as LB
if os.environ.get("AB_LIB"):
    LB.LIB_PATH = os.environ["AB_LIB"]
    LB._LIB = LB.load(LB.LIB_PATH)
from omnimamba_amd.ssd_combined import ssd_scan_fwd
from tools.bench_scan import timeit
dev = torch.device("cuda:0")
H, P, N, G = 64, 64, 128, 1
B, L = int(os.environ.get("AB_B", "8")), int(os.environ.get("AB_L", "4096"))
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
ms = min(timeit(lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True), 20, 5) for _ in range(3))
print(f"{os.environ.get('AB_TAG')}: B={B} L={L} fwd {ms*1e3:7.1f} us", flush=True)
''' % ROOT
other = os.path.abspath(sys.argv[1])
if len(sys.argv) > 3:
    os.environ["AB_B"], os.environ["AB_L"] = sys.argv[2], sys.argv[3]
for rep in range(3):
    for tag, lib in (("new ", ""), ("prev", other)):
        subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, AB_LIB=lib, AB_TAG=tag), check=True)

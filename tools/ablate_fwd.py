"""Developer tool (OMK_PHASE_PROF build): forward scan time with phases skipped (OMK_ABLATE bit mask; results wrong).
bit 0: Q.S   1: intra (G, M, M.U)   2: state update   3: publish S   4: epilogue"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
from omnimamba_amd.ssd_combined import ssd_scan_fwd
from tools.bench_scan import timeit
dev = torch.device("cuda:0")
H, P, N, G, L = 64, 64, 128, 1, 4096
for B in (4, 8):
    torch.manual_seed(0)
    xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
    x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
    dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
    dtb = torch.randn(H, device=dev) * 0.5 - 3
    ms = timeit(lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True), 20, 5)
    print(f"ablate={os.environ.get('OMK_ABLATE','0'):>3s} B={B}: {ms*1e3:7.1f} us", flush=True)
''' % ROOT
for m in [int(v) for v in os.environ.get('MASKS', '0,1,2,4,8,16,15,31').split(',')]:
    env = dict(os.environ, OMK_ABLATE=str(m))
    subprocess.run([sys.executable, "-c", CODE], env=env, check=True)

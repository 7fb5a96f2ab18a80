"""Developer tool: forward / backward scan at small batch with and without the sequence split (OMK_SSD_NO_SPLIT)."""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
from omnimamba_amd.ssd_combined import ssd_scan_fwd, ssd_scan_bwd
from tools.bench_scan import timeit
dev = torch.device("cuda:0")
H, P, N, G = 64, 64, 128, 1
tag = "unsplit" if os.environ.get("OMK_SSD_NO_SPLIT") else "split  "
for (B, L) in [(1, 2048), (1, 4096), (1, 8192), (1, 32768), (2, 8192), (4, 8192), (5, 4096), (8, 8192)]:
    torch.manual_seed(0)
    xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
    x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
    dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
    dtb = torch.randn(H, device=dev) * 0.5 - 3
    f = lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)
    ms = timeit(f, 20, 5)
    dout = torch.randn(B, L, H, P, device=dev).bfloat16()
    msb = timeit(lambda: ssd_scan_bwd(dout, x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True), 5, 2)
    print(f"{tag} B={B} L={L:6d}: fwd {ms*1e3:8.1f} us ({B*L*17024/ms/1e6/80:5.1f}%% of 8 TB/s)   bwd {msb*1e3:8.1f} us", flush=True)
''' % ROOT
for ns in ("", "1"):
    env = dict(os.environ)
    if ns:
        env["OMK_SSD_NO_SPLIT"] = ns
    subprocess.run([sys.executable, "-c", CODE], env=env, check=True)

"""Developer tool: same-box A/B of library builds on the scan AS A TRAINING STEP LAUNCHES IT (forward with window states, backward).
usage: python tools/ab_train_scan.py [--rounds N] tag=path_or_empty[:ENV=V,...] ..."""
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
from omnimamba_amd.ssd_combined import ssd_scan_fwd, ssd_scan_bwd
from tools.bench_scan import timeit
dev = torch.device("cuda:0")
H, P, N, G = 64, 64, 128, 1
B, L = 8, 4096
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
dout = torch.randn(B, L, H, P, device=dev).bfloat16()
f = lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, save_window_states=True, flags=int(os.environ.get('AB_FLAGS', '0')))
r = f()
b = lambda: ssd_scan_bwd(dout, x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, window_states=r[3], flags=int(os.environ.get('AB_FLAGS', '0')))
mf = min(timeit(f, 20, 5) for _ in range(3)); mb = min(timeit(b, 10, 3) for _ in range(3))
print(f"{os.environ.get('AB_TAG'):12s}: training fwd {mf*1e3:7.1f} us   bwd {mb*1e3:7.1f} us", flush=True)
''' % ROOT
args = sys.argv[1:]
rounds = 2
if args and args[0] == "--rounds":
    rounds = int(args[1]); args = args[2:]
specs = []
for a in args:
    tag, rest = a.split("=", 1)
    path, _, envs = rest.partition(":")
    env = dict(e.split("=", 1) for e in envs.split(",") if e)
    specs.append((tag, os.path.abspath(path) if path else "", env))
for rep in range(rounds):
    for tag, lib, env in specs:
        subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, AB_LIB=lib, AB_TAG=tag, **env), check=True)

"""Developer tool: when do the workgroups of ONE forward-scan launch (ssd_a8.hip) start and finish?  -DOMK_PHASE_PROF build, OMK_PROF_WG=1: every
workgroup leaves the 100 MHz wall clock of its first and last instruction and its XCC_ID.  A launch lasts as long as its slowest workgroup: with one
workgroup per CU and equal work per workgroup, the spread IS the loss.   usage: python tools/with_lib.py <prof lib> tools/wg_spread_a8.py   (PB, PL, PMODE=train)"""
import os, sys
os.environ["OMK_PROF_WG"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd import _capi as K
from omnimamba_amd._lib import get_lib

dev = torch.device("cuda:0")
B = int(os.environ.get("PB", "8"))
L, H, P, N, G = int(os.environ.get("PL", "4096")), 64, 64, 128, 1
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
lib = get_lib()
out = torch.empty(B, L, H, P, dtype=x.dtype, device=dev)
p = K.SsdFwd(x=K.T(x), dt=K.T(dt), A=K.T(A), Bm=K.T(Bm), Cm=K.T(Cm), D=K.T(D), z=K.T(None), dt_bias=K.T(dtb), initial_states=K.T(None), out=K.T(out),
             out_x=K.T(None), final_states=K.T(None), dt_min=0.0, dt_max=float("inf"), dt_softplus=1, chunk_size=256, force_generic=0, flags=K.SSD_NO_SPLIT | int(os.environ.get("AB_FLAGS", "0")))
ws = K.workspace(lib, "omk_ssd_scan_fwd_workspace_bytes", p, x)
nwg = B * H // 2
for rep in range(3):
    K.run(lib, "omk_ssd_scan_fwd", p, x)
    torch.cuda.synchronize()
    st = ws[-65536:].view(torch.int64).cpu()[128:128 + 4 * nwg].view(nwg, 4)
    mask = (1 << 60) - 1
    b0 = (st[:, 0] & mask).double(); e0 = (st[:, 1] & mask).double(); e1 = (st[:, 3] & mask).double()
    xcc = ((st[:, 1] >> 60) & 7)
    t0 = b0.min()
    beg, end = (b0 - t0) / 100.0, (torch.maximum(e0, e1) - t0) / 100.0
    q = lambda v, f: float(v.sort().values[min(len(v) - 1, int(f * len(v)))])
    print(f"launch {rep}: B={B} L={L}, {nwg} workgroups | start  median {q(beg, .5):.1f} max {float(beg.max()):.1f} us | finish  min {float(end.min()):.1f}  5% {q(end, .05):.1f}  median {q(end, .5):.1f}  "
          f"95% {q(end, .95):.1f}  max {float(end.max()):.1f} us | mean busy {float((end - beg).mean()):.1f} us = {float((end - beg).mean()) / float(end.max()):.2f} of the launch")
    if rep == 2:
        for xi in range(8):
            m = xcc == xi
            if m.any():
                print(f"   XCC {xi}: {int(m.sum()):3d} workgroups, finish median {q(end[m], .5):.1f} max {float(end[m].max()):.1f} us;  workgroup id mod 8 of its members: {sorted(set((torch.nonzero(m).flatten() % 8).tolist()))}")

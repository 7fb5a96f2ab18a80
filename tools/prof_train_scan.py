"""Developer tool: the SSD scan as a training step launches it -- forward that saves its window states, backward that takes them --
a few times (configs[1] shape); target for rocprofv3 (--kernel-trace --stats, or one --pmc counter group per pass)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_fwd, ssd_scan_bwd  # noqa: E402

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
dout = torch.randn(B, L, H, P, device=dev).bfloat16()
from omnimamba_amd._lib import get_lib  # noqa: E402
lib, ids = get_lib(), {}
for _ in range(int(os.environ.get("ITERS", "3"))):
    r = ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, save_window_states=True)
    ids["fwd_with_window_states"] = lib.omk_ssd_last_kernels().decode()
    ssd_scan_bwd(dout, x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, window_states=r[3])
    ids["bwd"] = lib.omk_ssd_last_kernels().decode()
    ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)   # the plain (inference) forward
    ids["fwd"] = lib.omk_ssd_last_kernels().decode()
torch.cuda.synchronize()
if os.environ.get("KERNEL_IDS_OUT"):      # what the library says it launched: tools/make_traffic_json.py copies it into the traffic files
    import json
    json.dump(ids, open(os.environ["KERNEL_IDS_OUT"], "w"), indent=1)

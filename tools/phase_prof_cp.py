import os, sys, torch
sys.path.insert(0, "/root/repo")
os.environ["OMK_CP_PROF"] = "1"
from omnimamba_amd.ssd_combined import ssd_scan_fwd, ssd_scan_bwd
dev = torch.device("cuda:0")
H, P, N, G = 64, 64, 128, 1
B, L = 8, 4096
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
dout = torch.randn(B, L, H, P, device=dev).bfloat16()
r = ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, save_window_states=True)
ssd_scan_bwd(dout, x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, window_states=r[3])
torch.cuda.synchronize()

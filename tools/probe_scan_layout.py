"""Developer probe: the forward scan on head-major storage ((B, H, L, P): a head's 64-token chunk is 8 KB of consecutive bytes) against the
reference's token-major storage ((B, L, H, P): 128 bytes per (token, head), 8 KB apart) -- is the access pattern of x / y what bounds the staging?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd import _capi as K
from omnimamba_amd._lib import get_lib
from tools.bench_scan import timeit

dev = torch.device("cuda:0")
B, L, H, P, N, G = 8, int(os.environ.get("SEQ", "4096")), 64, 64, 128, 1
torch.manual_seed(0)
lib = get_lib()
Bm = torch.randn(B, L, G, N, device=dev).bfloat16(); Cm = torch.randn(B, L, G, N, device=dev).bfloat16()
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
for name in ("token-major (B, L, H, P)", "head-major (B, H, L, P)"):
    if name.startswith("token"):
        x = torch.randn(B, L, H, P, device=dev).bfloat16(); out = torch.empty(B, L, H, P, device=dev, dtype=torch.bfloat16)
    else:
        x = torch.randn(B, H, L, P, device=dev).bfloat16().permute(0, 2, 1, 3); out = torch.empty(B, H, L, P, device=dev, dtype=torch.bfloat16).permute(0, 2, 1, 3)
    p = K.SsdFwd(x=K.T(x), dt=K.T(dt), A=K.T(A), Bm=K.T(Bm), Cm=K.T(Cm), D=K.T(D), z=K.T(None), dt_bias=K.T(dtb), initial_states=K.T(None), out=K.T(out),
                 out_x=K.T(None), final_states=K.T(None), dt_min=0.0, dt_max=float("inf"), dt_softplus=1, chunk_size=256, force_generic=0)
    ws = K.workspace(lib, "omk_ssd_scan_fwd_workspace_bytes", p, x)
    ms = min(timeit(lambda: K.run(lib, "omk_ssd_scan_fwd", p, x), 20, 5) for _ in range(3))
    print(f"{name:28s} {ms*1e3:7.1f} us")

"""Quick kernel timing helper (developer tool): SSD scan forward / backward at the BASELINE shapes."""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_fwd, ssd_scan_bwd  # noqa: E402


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    H, P, N, G = 64, 64, 128, 1
    for (B, L) in [(8, 4096), (8, 8192), (1, 8192)]:
        torch.manual_seed(0)
        # x, B, C as slices of one (B, L, conv_dim) buffer like the block does
        xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
        x = xBC[..., :H * P].view(B, L, H, P)
        Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N)
        Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
        dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16()
        A = -(torch.rand(H, device=dev) * 15 + 1)
        D = torch.ones(H, device=dev)
        dtb = torch.randn(H, device=dev) * 0.5 - 3
        tok = B * L
        f = lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)
        ms = timeit(f)
        byt = tok * 17024
        print(f"fwd  B={B} L={L}: {ms*1e3:8.1f} us  {byt/ms/1e6:7.1f} GB/s algorithmic  ({byt/ms/1e6/8000*100:5.1f}% of 8 TB/s)  "
              f"{tok*H*P/ms/1e3:9.1f} M-elem/s", flush=True)
        if "--generic" in sys.argv:
            fg = lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, force_generic=True)
            print(f"     generic fwd: {timeit(fg, 3, 1)*1e3:8.1f} us", flush=True)
        if "--bwd" in sys.argv:
            out, _, _ = f()
            dout = torch.randn_like(out)
            fb = lambda: ssd_scan_bwd(dout, x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)
            msb = timeit(fb, 3, 1)
            print(f"bwd  B={B} L={L}: {msb*1e3:8.1f} us", flush=True)


if __name__ == "__main__":
    main()

"""Developer probe: does a weight-gradient GEMM on a side stream hide behind the hand-written kernels of the block's backward?
main stream: [4 streaming elementwise passes over 268 MB tensors (stand-ins for the norm / conv backward)] + omk_ssd_scan_bwd;
side stream: dW = dy^T x of out_proj (2048 x 32768 x 4096, bf16).  Prints each alone, serial, and concurrent."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_bwd  # noqa: E402

dev = torch.device("cuda:0")
B, L, H, P, N, G = 8, 4096, 64, 64, 128, 1
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
dout = torch.randn(B, L, H, P, device=dev).bfloat16()
a1, a2, a3 = (torch.randn(B * L, H * P, device=dev).bfloat16() for _ in range(3))
dy = torch.randn(B * L, 2048, device=dev).bfloat16()
xn = torch.randn(B * L, 4096, device=dev).bfloat16()
side = torch.cuda.Stream()


def scans():
    ssd_scan_bwd(dout, x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)


def streams():
    for _ in range(2):
        torch.add(a1, a2, out=a3)
        torch.mul(a1, a3, out=a2)


def gemm():
    return dy.t() @ xn


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def both(main_fn):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gemm()
        main_fn()
        torch.cuda.current_stream().wait_stream(side)
    return f


tg, ts, tb = t(gemm), t(scans), t(streams)
print(f"alone: gemm {tg:7.1f} us, scan backward {ts:7.1f} us, 4 streaming passes {tb:7.1f} us")
print(f"gemm || scan backward     : {t(both(scans)):7.1f} us   (serial {tg + ts:7.1f})")
print(f"gemm || streaming passes  : {t(both(streams)):7.1f} us   (serial {tg + tb:7.1f})")
print(f"gemm || streaming + scans : {t(both(lambda: (streams(), scans()))):7.1f} us   (serial {tg + tb + ts:7.1f})")

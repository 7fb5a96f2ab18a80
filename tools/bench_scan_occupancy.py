"""Developer tool: forward scan time vs number of (batch, head) sequences at fixed L -- separates per-workgroup latency
from per-CU throughput (64 heads: B=1 -> 64 workgroups, B=4 -> one per CU, B=8 -> two per CU, B=16 -> two rounds)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_fwd  # noqa: E402
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
H, P, N, G = 64, 64, 128, 1
L = int(os.environ.get("SEQ", "4096"))
for B in [1, 2, 3, 4, 5, 6, 8, 12, 16]:
    torch.manual_seed(0)
    xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
    x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
    dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
    dtb = torch.randn(H, device=dev) * 0.5 - 3
    ms = timeit(lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True), 20, 5)
    nch = (L + 63) // 64
    print(f"B={B:2d} workgroups={B*H:5d}  {ms*1e3:7.1f} us  {ms*1e3/nch:6.3f} us/chunk/wg  {B*H*nch/ms/1e3:7.1f} chunk-heads/us  "
          f"{B*L*17024/ms/1e6/80:5.1f}% of 8 TB/s", flush=True)

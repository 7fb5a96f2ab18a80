"""Developer probe: the token-split weight gradient (linear.weight_grad) with fp32 vs bf16 partial outputs and different split
factors, in_proj and out_proj shapes of the benchmark block (32 768 tokens)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
T = 32768
for name, (m, n) in (("in_proj", (8512, 2048)), ("out_proj", (2048, 4096))):
    dy = torch.randn(T, m, device=dev).bfloat16()
    x = torch.randn(T, n, device=dev).bfloat16()
    flop = 2 * T * m * n
    t = min(timeit(lambda: dy.t() @ x, 10, 3) for _ in range(3))
    print(f"{name}: plain bf16 GEMM                         {t * 1e3:8.1f} us  {flop / t / 1e9:7.1f} TFLOP/s", flush=True)
    for s in (2, 4, 8):
        a, b = dy.view(s, T // s, m).transpose(1, 2), x.view(s, T // s, n)
        t32 = min(timeit(lambda: torch.bmm(a, b, out_dtype=torch.float32).sum(0), 10, 3) for _ in range(3))
        t16 = min(timeit(lambda: torch.bmm(a, b).float().sum(0), 10, 3) for _ in range(3))
        t16b = min(timeit(lambda: torch.bmm(a, b).sum(0, dtype=torch.float32), 10, 3) for _ in range(3))
        print(f"{name}: S = {s}  fp32 partials {t32 * 1e3:8.1f} us ({flop / t32 / 1e9:6.1f})   bf16 partials {t16 * 1e3:8.1f} us   bf16 partials, fused fp32 sum {t16b * 1e3:8.1f} us ({flop / t16b / 1e9:6.1f})", flush=True)

"""Developer probe: does the library GEMM keep its speed when the LoRA rank is appended to the contraction dimension (K = 2048 + 8,
row stride 2056) -- the 'K-extension' form of base + LoRA as ONE GEMM -- forward and input-gradient shapes of the 1.3B in_proj."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
T, K, N, r = 16384, 2048, 8512, 8
for Kx in (K, K + r, K + 64):
    x = torch.randn(T, Kx, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, Kx, device=dev, dtype=torch.bfloat16) * 0.02
    dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
    f = min(timeit(lambda: torch.nn.functional.linear(x, W), 20, 3) for _ in range(3))
    b = min(timeit(lambda: dy @ W, 20, 3) for _ in range(3))
    print(f"K = {Kx}: forward {f * 1e3:7.1f} us ({2 * T * Kx * N / f / 1e9:7.1f} TFLOP/s)   dgrad {b * 1e3:7.1f} us ({2 * T * Kx * N / b / 1e9:7.1f} TFLOP/s)", flush=True)
# out_proj shape: K = 4096 + 8
T, K, N = 16384, 4096, 2048
for Kx in (K, K + r):
    x = torch.randn(T, Kx, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, Kx, device=dev, dtype=torch.bfloat16) * 0.02
    dy = torch.randn(T, N, device=dev, dtype=torch.bfloat16)
    f = min(timeit(lambda: torch.nn.functional.linear(x, W), 20, 3) for _ in range(3))
    b = min(timeit(lambda: dy @ W, 20, 3) for _ in range(3))
    print(f"out_proj K = {Kx}: forward {f * 1e3:7.1f} us   dgrad {b * 1e3:7.1f} us", flush=True)

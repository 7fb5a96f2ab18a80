"""Developer tool: omk_lora_up_bwd (dh = dy B, dB = dy^T h in one pass over dy) against the two library GEMMs, 1.3B in_proj shape."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd import lora_add as LA  # noqa: E402
from omnimamba_amd.linear import weight_grad  # noqa: E402
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
T, N = 16384, 8512
dy, h = torch.randn(T, N, device=dev).bfloat16(), torch.randn(T, 8, device=dev).bfloat16()
B = torch.randn(N, 8, device=dev) * 0.1
t_f = min(timeit(lambda: LA.lora_up_bwd(dy, h, B), 20, 3) for _ in range(3))
t_g = min(timeit(lambda: (dy @ B.bfloat16(), weight_grad(dy, h, torch.float32)), 20, 3) for _ in range(3))
print(f"fused {t_f * 1e3:7.1f} us ({T * N * 2 / t_f / 1e6:6.0f} GB/s)   two GEMMs {t_g * 1e3:7.1f} us   (OMK_LORA_UP_DBG={os.environ.get('OMK_LORA_UP_DBG', '0')})")

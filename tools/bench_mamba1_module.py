"""Developer tool: forward of the Mamba-1 mixer module (d_model 384 -> d_inner 768, d_state 16) at L 1024 over a few batch sizes.
OMK_SELSCAN_LANES=0 keeps the lanes-are-channels scan out (L-contiguous copies of x, z, dt, B, C + chunked scan)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.mamba_simple import Mamba  # noqa: E402
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
for dtype in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    m = Mamba(384, d_state=16, expand=2, use_fast_path=False, device=dev, dtype=dtype).eval()
    for Bsz in (2, 32, 64, 128):
        h = torch.randn(Bsz, 1024, 384, device=dev, dtype=dtype)
        with torch.no_grad():
            ms = min(timeit(lambda: m(h), 20, 3) for _ in range(3))
        print(f"{str(dtype):15s} B={Bsz:3d}: module forward {ms * 1e3:8.1f} us  (LANES={os.environ.get('OMK_SELSCAN_LANES', 'auto')})", flush=True)

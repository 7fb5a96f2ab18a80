"""Developer probe: does touching a weight matrix shortly before the fused GEMV (so that it sits in the 256 MB memory-side
cache) shorten the GEMV?  Rotates over many copies so nothing is warm by accident."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.norm_linear import norm_linear  # noqa: E402

dev = torch.device("cuda:0")
for dtype in (torch.float32, torch.bfloat16):
    for (Out, In) in ((8512, 2048), (2048, 4096)):
        n = 24
        Ws = [torch.randn(Out, In, device=dev).to(dtype) * 0.02 for _ in range(n)]
        x = torch.randn(1, In, device=dev).to(dtype)
        nw = torch.ones(In, device=dev, dtype=dtype)
        big = torch.empty(1 << 28, device=dev)                          # 1 GB: flushes the caches between phases

        def run(warm):
            ts = []
            for i in range(n):
                big.add_(1.0)                                            # evict
                if warm:
                    Ws[i].sum()                                          # touch: HBM -> memory-side cache (and L2)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                norm_linear(x, Ws[i], None, norm_weight=nw, eps=1e-5)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            return ts[len(ts) // 2]
        run(False)
        c, w = run(False), run(True)
        mb = Out * In * Ws[0].element_size() / 1e6
        print(f"{str(dtype):15s} {Out}x{In} ({mb:5.1f} MB): cold {c:6.1f} us ({mb / c / 1e6 * 1e6 / 1e3:5.2f} TB/s)   touched first {w:6.1f} us ({mb / w / 1e3:5.2f} TB/s)", flush=True)

"""Developer tool: the fused decode-step projections (omk_norm_linear) at the 1.3B shapes vs the separate ops."""
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.layer_norm import layer_norm_fn  # noqa: E402
from omnimamba_amd.layernorm_gated import rmsnorm_fn  # noqa: E402
from omnimamba_amd.norm_linear import norm_linear  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for B in (1, 4):
    x, res = torch.randn(B, 2048, device=dev), torch.randn(B, 2048, device=dev)
    nw = torch.ones(2048, device=dev)
    W = torch.randn(8512, 2048, device=dev) * 0.02
    la, lb = torch.randn(8, 2048, device=dev) * 0.02, torch.randn(8512, 8, device=dev) * 0.02
    with torch.no_grad():
        t_f = timeit(lambda: norm_linear(x, W, None, norm_weight=nw, eps=1e-5, residual=res, residual_out_dtype=torch.float32,
                                         lora_a=la, lora_b=lb, lora_scale=4.0))

        def unfused():
            h, r = layer_norm_fn(x, nw, None, residual=res, prenorm=True, residual_in_fp32=True, eps=1e-5, is_rms_norm=True)
            y = F.linear(h, W)
            return torch.addmm(y, F.linear(h, la), lb.t(), alpha=4.0), r
        t_u = timeit(unfused)
        y, z = torch.randn(B, 4096, device=dev), torch.randn(B, 4096, device=dev)
        nw2, Wo = torch.ones(4096, device=dev), torch.randn(2048, 4096, device=dev) * 0.02
        t_f2 = timeit(lambda: norm_linear(y, Wo, None, norm_weight=nw2, eps=1e-5, z=z, group_size=4096, norm_before_gate=False))
        t_u2 = timeit(lambda: F.linear(rmsnorm_fn(y, nw2, None, z=z, eps=1e-5, group_size=4096, norm_before_gate=False), Wo))
    print(f"B={B}: pre-norm + in_proj + LoRA  fused {t_f:6.1f} us ({8512*2048*4/t_f/1e6:5.2f} TB/s of W)  separate {t_u:6.1f} us | "
          f"gated norm + out_proj  fused {t_f2:6.1f} us ({2048*4096*4/t_f2/1e6:5.2f} TB/s)  separate {t_u2:6.1f} us")

# preamble cost: the same call with a handful of output rows
x, res = torch.randn(1, 2048, device=dev), torch.randn(1, 2048, device=dev)
nw = torch.ones(2048, device=dev)
for out_rows in (8, 1024, 4096, 8512):
    W = torch.randn(out_rows, 2048, device=dev) * 0.02
    la, lb = torch.randn(8, 2048, device=dev) * 0.02, torch.randn(out_rows, 8, device=dev) * 0.02
    with torch.no_grad():
        t1 = timeit(lambda: norm_linear(x, W, None, norm_weight=nw, eps=1e-5, residual=res, residual_out_dtype=torch.float32,
                                        lora_a=la, lora_b=lb, lora_scale=4.0))
        t2 = timeit(lambda: norm_linear(x, W, None))
    print(f"out rows {out_rows:5d}: norm + LoRA + GEMV {t1:6.1f} us   plain GEMV {t2:6.1f} us")

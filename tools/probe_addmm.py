"""Developer probe: cost of forming result = x W^T + s h B^T in different ways (bf16, 16384 tokens, 2048 -> 8512, rank 8)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
T, K, N, R = 16384, 2048, 8512, 8
x = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
W = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
h = torch.randn(T, R, device=dev, dtype=torch.bfloat16)
Bm = torch.randn(N, R, device=dev, dtype=torch.bfloat16) * 0.02
Wt, Bt = W.t(), Bm.t()


def a_now():        # base GEMM, then out-of-place addmm (copies the result first)
    r = torch.nn.functional.linear(x, W)
    return torch.addmm(r, h, Bt, alpha=4.0)


def b_lora_first_inplace():   # LoRA GEMM into a fresh buffer, base GEMM accumulates in place (beta = 1)
    r = torch.mm(h, Bt)
    r.mul_(4.0) if False else None
    return r.addmm_(x, Wt)


def c_base_only():
    return torch.nn.functional.linear(x, W)


def d_inplace_k8():
    r = torch.nn.functional.linear(x, W)
    return r.addmm_(h, Bt, alpha=4.0)


def e_lora_add():
    from omnimamba_amd.lora_add import lora_add
    r = torch.nn.functional.linear(x, W)
    return lora_add(r, h, Bm.float(), 4.0)


Bf = Bm.float()


def f_lora_add_only(buf=[None]):
    from omnimamba_amd.lora_add import lora_add
    if buf[0] is None:
        buf[0] = torch.nn.functional.linear(x, W)
    return lora_add(buf[0], h, Bf, 4.0)


for name, fn in (("base GEMM only", c_base_only), ("linear + omk_lora_add", e_lora_add), ("omk_lora_add alone", f_lora_add_only), ("now: linear + addmm (copy)", a_now), ("LoRA GEMM first, base addmm_ in place", b_lora_first_inplace),
                 ("linear + addmm_ in place (K = 8)", d_inplace_k8)):
    print(f"{name:45s} {timeit(fn, 20, 5) * 1e3:8.1f} us", flush=True)

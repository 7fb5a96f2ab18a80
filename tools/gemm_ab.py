import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
K_, M, N = 32768, 8512, 2048
torch.manual_seed(0)
A = torch.randn(K_, M, device=dev, dtype=torch.bfloat16)
B = torch.randn(K_, N, device=dev, dtype=torch.bfloat16)
W = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
Y = torch.randn(K_, 4096, device=dev, dtype=torch.bfloat16); Wo = torch.randn(2048, 4096, device=dev, dtype=torch.bfloat16)
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
ops = {"in fwd": lambda: torch.nn.functional.linear(B, W), "in dgrad": lambda: A @ W, "in wgrad": lambda: A.t() @ B,
       "out fwd": lambda: torch.nn.functional.linear(Y, Wo), "out dgrad": lambda: B @ Wo, "out wgrad": lambda: B.t() @ Y}
base = {k: timeit(f) for k, f in ops.items()}
from omnimamba_amd.gemm_tuning import use_tuned_gemms
print("tuned file ok:", use_tuned_gemms())
tuned = {k: timeit(f) for k, f in ops.items()}
for k in ops: print(f"{k:10s} default {base[k]:8.1f} us   recorded {tuned[k]:8.1f} us")
print("sum", sum(base.values()), sum(tuned.values()))

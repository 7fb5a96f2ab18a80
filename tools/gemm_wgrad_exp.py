"""Developer experiment: the in_proj weight-gradient GEMM (8512 x 2048, K = 32768) has 264 256x256 output tiles on 256 CUs."""
import torch
dev = torch.device("cuda:0")
K_, M, N = 32768, 8512, 2048
A = torch.randn(K_, M, device=dev, dtype=torch.bfloat16)
B = torch.randn(K_, N, device=dev, dtype=torch.bfloat16)


def timeit(fn, iters=10, warm=3):
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


ref = (A.t() @ B).float()
print(f"A^T @ B              : {timeit(lambda: A.t() @ B):8.1f} us")
print(f"(B^T @ A)^T          : {timeit(lambda: (B.t() @ A)):8.1f} us")
for S in (2, 3, 4, 8):
    if K_ % S:
        continue
    A3, B3 = A.view(S, K_ // S, M), B.view(S, K_ // S, N)
    f = lambda: torch.bmm(A3.transpose(1, 2), B3).sum(0, dtype=torch.float32)
    out = f()
    err = ((out - ref).norm() / ref.norm()).item()
    print(f"bmm split-K S={S} + sum: {timeit(f):8.1f} us   rel diff vs single GEMM {err:.2e}")
    g = lambda: torch.bmm(A3.transpose(1, 2), B3)
    print(f"   bmm alone          : {timeit(g):8.1f} us")
# other GEMMs of the block for reference
X = torch.randn(K_, 2048, device=dev, dtype=torch.bfloat16); W = torch.randn(8512, 2048, device=dev, dtype=torch.bfloat16)
print(f"in_proj fwd  x W^T    : {timeit(lambda: X @ W.t()):8.1f} us")
print(f"in_proj dgrad dy W    : {timeit(lambda: A @ W):8.1f} us")
Y = torch.randn(K_, 4096, device=dev, dtype=torch.bfloat16); Wo = torch.randn(2048, 4096, device=dev, dtype=torch.bfloat16)
print(f"out_proj fwd          : {timeit(lambda: Y @ Wo.t()):8.1f} us")
print(f"out_proj dgrad        : {timeit(lambda: B @ Wo):8.1f} us")
print(f"out_proj wgrad        : {timeit(lambda: B.t() @ Y):8.1f} us")

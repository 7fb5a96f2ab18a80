import torch
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
K_ = 32768
for (M, N, name) in [(2048, 4096, "out_proj wgrad"), (8512, 2048, "in_proj wgrad")]:
    A = torch.randn(K_, M, device=dev, dtype=torch.bfloat16); B = torch.randn(K_, N, device=dev, dtype=torch.bfloat16)
    print(name, "single", round(timeit(lambda: A.t() @ B), 1))
    for S in (2, 4, 8):
        A3, B3 = A.view(S, K_ // S, M), B.view(S, K_ // S, N)
        print("  S", S, "bmm+sum(f32)", round(timeit(lambda: torch.bmm(A3.transpose(1, 2), B3).sum(0, dtype=torch.float32)), 1),
              " bmm+sum(bf16 out)", round(timeit(lambda: torch.bmm(A3.transpose(1, 2), B3).sum(0, dtype=torch.float32).to(torch.bfloat16)), 1))
    try:
        A3, B3 = A.view(4, K_ // 4, M), B.view(4, K_ // 4, N)
        r = torch.bmm(A3.transpose(1, 2), B3, out_dtype=torch.float32)
        print("  out_dtype fp32 bmm ok", r.dtype, round(timeit(lambda: torch.bmm(A3.transpose(1, 2), B3, out_dtype=torch.float32).sum(0)), 1))
    except Exception as e:
        print("  out_dtype not available:", str(e)[:80])

import torch
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M, N, K = 32768, 8512, 2048
for name, gen in [("zeros", lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16)), ("ones", lambda *s: torch.ones(*s, device=dev, dtype=torch.bfloat16)),
                  ("randn", lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)), ("randn*0.02", lambda *s: (torch.randn(*s, device=dev) * 0.02).bfloat16())]:
    x, w = gen(M, K), gen(N, K)
    t = timeit(lambda: torch.nn.functional.linear(x, w))
    print(f"{name:12s} in_proj fwd {t:8.1f} us  {2*M*N*K/t/1e9:7.1f} TFLOP/s")
x = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); w = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: x @ w); print(f"8192^3 randn {t:8.1f} us {2*8192**3/t/1e9:7.1f} TFLOP/s")
x = torch.zeros(8192, 8192, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: x @ x); print(f"8192^3 zeros {t:8.1f} us {2*8192**3/t/1e9:7.1f} TFLOP/s")

import torch, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tools.bench_scan import timeit
dev = torch.device("cuda:0")
for n in (256, 1024):
    x = torch.randn(n * 1024 * 1024 // 2, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    t = min(timeit(lambda: y.copy_(x), 20, 3) for _ in range(3))
    print(f"copy {n} MB: {t*1e3:7.1f} us  {2*n*1.048576/t/1e3:6.2f} TB/s (read + write)")
    t = min(timeit(lambda: x.sum(), 20, 3) for _ in range(3))
    print(f"sum  {n} MB: {t*1e3:7.1f} us  {n*1.048576/t/1e3:6.2f} TB/s (read)")
    t = min(timeit(lambda: y.fill_(1.0), 20, 3) for _ in range(3))
    print(f"fill {n} MB: {t*1e3:7.1f} us  {n*1.048576/t/1e3:6.2f} TB/s (write)")
    t = min(timeit(lambda: torch.add(x, x, out=y), 20, 3) for _ in range(3))
    print(f"add  {n} MB: {t*1e3:7.1f} us  {2*n*1.048576/t/1e3:6.2f} TB/s (1 read + 1 write)")

"""Developer tool: Mamba-1 selective_scan forward (and, with --bwd, forward + backward) at BASELINE configs[0] (B 2, L 1024, D 768,
N 16 fp32) and scaled batches; --bld: channel-last (B, L, D) views as the Mamba-1 module holds them; --batches 2,16,64.
OMK_SELSCAN_LANES=0 keeps the lanes-are-channels sweep out (chunked scan, copies for channel-last views)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.selective_scan import selective_scan_fn  # noqa: E402
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
Dm, L, N = 768, 1024, 16
BLD = "--bld" in sys.argv
BATCHES = [int(v) for v in sys.argv[sys.argv.index("--batches") + 1].split(",")] if "--batches" in sys.argv else [2, 16, 64]
for dtype in (torch.float32, torch.bfloat16):
    for Bsz in BATCHES:
        torch.manual_seed(0)
        if BLD:
            u, delta, z = (torch.randn(Bsz, L, Dm, device=dev).to(dtype).transpose(1, 2) for _ in range(3))
            Bm, Cm = (torch.randn(Bsz, L, N, device=dev).to(dtype).transpose(1, 2) for _ in range(2))
        else:
            u, delta, z = (torch.randn(Bsz, Dm, L, device=dev).to(dtype) for _ in range(3))
            Bm, Cm = torch.randn(Bsz, N, L, device=dev).to(dtype), torch.randn(Bsz, N, L, device=dev).to(dtype)
        A = -(torch.rand(Dm, N, device=dev) + 0.1)
        D, db = torch.randn(Dm, device=dev), 0.1 * torch.randn(Dm, device=dev)
        ms = min(timeit(lambda: selective_scan_fn(u, delta, A, Bm, Cm, D, z, db, True), 20, 3) for _ in range(3))
        es = 4 if dtype == torch.float32 else 2
        nb = Bsz * L * (4 * Dm * es + 2 * N * es)
        print(f"{str(dtype):15s} B={Bsz:3d}: {ms * 1e3:8.1f} us  {Bsz * L * Dm / ms / 1e3:9.1f} M-elem/s  {nb / ms / 1e6:7.1f} GB/s = {nb / ms / 1e6 / 80:5.1f} % of 8 TB/s  (layout={'bld' if BLD else 'bdl'}, LANES={os.environ.get('OMK_SELSCAN_LANES', 'auto')})", flush=True)
        if "--bwd" in sys.argv:
            leaves = [t.detach().clone().requires_grad_() for t in (u, delta, A, Bm, Cm, D, z, db)]
            g = torch.randn_like(u)

            def fb():
                for t in leaves:
                    t.grad = None
                selective_scan_fn(*leaves, True).backward(g)
            ms2 = min(timeit(fb, 10, 2) for _ in range(3))
            # backward alone = (forward + backward) - forward; algorithmic bytes of the backward: u, delta, z, dout read, du, ddelta, dz written
            nbb = Bsz * L * (7 * Dm * es + 2 * N * es + 2 * N * 4)
            print(f"{'':15s}        fwd+bwd {ms2 * 1e3:8.1f} us   bwd alone ~{(ms2 - ms) * 1e3:8.1f} us  {nbb / (ms2 - ms) / 1e6:7.1f} GB/s = {nbb / (ms2 - ms) / 1e6 / 80:5.1f} % of 8 TB/s", flush=True)

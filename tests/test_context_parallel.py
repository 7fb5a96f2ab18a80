"""Row f2 of SURVEY.md section 8: the SSD scan of ONE sequence sharded along L over two ranks (gloo, CPU, emulated kernels)
== the single-process scan of the whole sequence -- outputs, final state and every gradient (the boundary-state exchange is
differentiable: gradients flow back to the rank that produced the state)."""
import os

import pytest
import torch


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _inputs(dtype):
    g = torch.Generator().manual_seed(3)
    Bsz, L, H, P, N, G = 2, 96, 4, 8, 16, 2
    x = torch.randn(Bsz, L, H, P, generator=g).to(dtype)
    dt = (torch.randn(Bsz, L, H, generator=g) * 0.5).to(dtype)
    A = -(torch.rand(H, generator=g) * 4 + 0.5)
    Bm, Cm = torch.randn(Bsz, L, G, N, generator=g).to(dtype), torch.randn(Bsz, L, G, N, generator=g).to(dtype)
    D, dtb = torch.randn(H, generator=g), torch.randn(H, generator=g) * 0.5 - 1
    z = torch.randn(Bsz, L, H, P, generator=g).to(dtype)
    init = torch.randn(Bsz, H, P, N, generator=g)
    gy = torch.randn(Bsz, L, H, P, generator=g).to(dtype)
    gf = torch.randn(Bsz, H, P, N, generator=g)
    return x, dt, A, Bm, Cm, D, dtb, z, init, gy, gf


def _worker(rank, world, port, q):
    import sys
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from emu.loader import use_emulator
    from omnimamba_amd.context_parallel import mamba_chunk_scan_context_parallel
    torch.set_num_threads(1)
    dist.init_process_group("gloo")
    with use_emulator():
        x, dt, A, Bm, Cm, D, dtb, z, init, gy, gf = _inputs(torch.float32)
        L = x.shape[1]
        sl = slice(rank * L // world, (rank + 1) * L // world)
        sh = [t[:, sl].clone().requires_grad_() for t in (x, dt, Bm, Cm, z)]
        par = [t.clone().requires_grad_() for t in (A, D, dtb, init)]
        y, fin = mamba_chunk_scan_context_parallel(sh[0], sh[1], par[0], sh[2], sh[3], 16, D=par[1], z=sh[4], dt_bias=par[2],
                                                   initial_states=par[3], dt_softplus=True, return_final_states=True)
        # the final state is the same tensor on every rank: its upstream gradient is applied once (rank 0)
        torch.autograd.backward([y, fin], [gy[:, sl], gf if rank == 0 else torch.zeros_like(gf)])
        out = {"y": y.detach().numpy(), "fin": fin.detach().numpy()}
        for n, t in zip(("x", "dt", "B", "C", "z"), sh):
            out["g" + n] = t.grad.numpy()
        for n, t in zip(("A", "D", "dtb", "init"), par):
            gsum = t.grad.clone()
            dist.all_reduce(gsum)                      # parameter gradients: the sum over the shards (what DDP-style training needs)
            out["g" + n] = gsum.numpy()
        q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_context_parallel_scan_two_ranks_equals_one():
    import torch.multiprocessing as mp
    from emu.loader import use_emulator
    from omnimamba_amd.ssd_combined import mamba_chunk_scan_combined
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    with use_emulator():
        x, dt, A, Bm, Cm, D, dtb, z, init, gy, gf = _inputs(torch.float32)
        lv = [t.clone().requires_grad_() for t in (x, dt, A, Bm, Cm, D, dtb, z, init)]
        y, fin = mamba_chunk_scan_combined(lv[0], lv[1], lv[2], lv[3], lv[4], 16, D=lv[5], z=lv[7], dt_bias=lv[6], initial_states=lv[8],
                                           dt_softplus=True, return_final_states=True)
        torch.autograd.backward([y, fin], [gy, gf])
    L = x.shape[1]
    cat = lambda k: torch.cat([torch.from_numpy(got[r][k]) for r in range(2)], dim=1)
    assert rel(cat("y"), y.detach()) < 2e-5
    for r in range(2):
        assert rel(torch.from_numpy(got[r]["fin"]), fin.detach()) < 2e-5
    for k, t in (("gx", lv[0]), ("gdt", lv[1]), ("gB", lv[3]), ("gC", lv[4]), ("gz", lv[7])):
        assert rel(cat(k), t.grad) < 3e-4, k
    for k, t in (("gA", lv[2]), ("gD", lv[5]), ("gdtb", lv[6]), ("ginit", lv[8])):
        assert rel(torch.from_numpy(got[0][k]), t.grad) < 3e-4, k

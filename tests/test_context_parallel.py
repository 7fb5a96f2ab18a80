"""Row f2 of SURVEY.md section 8: ONE sequence sharded along L.
  * two gloo ranks (CPU, emulated kernels): the context-parallel scan == the single-process scan of the whole sequence -- outputs,
    final state and every gradient (the boundary-state exchange is one differentiable node every rank executes), also when the
    final state is not asked for (no rank may skip a collective its peers issue);
  * two gloo ranks: a whole Mamba2 block with cp_group (conv1d halo from the left neighbour + the scan exchange) == the block on the
    whole sequence, output and parameter gradients;
  * one process, emulator and MI355X (-m gpu): two shards walked one after the other in bf16 through the MFMA kernels (state-only
    pass, fold, scan from the folded state) == the oracle on the whole sequence under the tolerance rule of the unsharded scan."""
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


def _worker_nofinal(rank, world, port, q):
    import sys
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from emu.loader import use_emulator
    from omnimamba_amd.context_parallel import mamba_chunk_scan_context_parallel
    from omnimamba_amd.mamba2 import Mamba2
    torch.set_num_threads(1)
    dist.init_process_group("gloo")
    with use_emulator():
        x, dt, A, Bm, Cm, D, dtb, z, init, gy, gf = _inputs(torch.float32)
        L = x.shape[1]
        sl = slice(rank * L // world, (rank + 1) * L // world)
        sh = [t[:, sl].clone().requires_grad_() for t in (x, dt, Bm, Cm)]
        # (a) no final state, no initial state: rank 0's output does not depend on the gathered states -- it must still run the
        # exchange's backward collectives
        y = mamba_chunk_scan_context_parallel(sh[0], sh[1], A, sh[2], sh[3], 16, D=D, dt_bias=dtb, dt_softplus=True)
        y.backward(gy[:, sl])
        out = {"y": y.detach().numpy(), "gx": sh[0].grad.numpy(), "gB": sh[2].grad.numpy()}
        # (b) a Mamba2 block on the sharded sequence
        torch.manual_seed(0)
        m = Mamba2(32, d_state=16, headdim=8, ngroups=2, chunk_size=16)
        u = torch.randn(2, 96, 32, generator=torch.Generator().manual_seed(9))
        gu = torch.randn(2, 96, 32, generator=torch.Generator().manual_seed(10))
        ur = u[:, sl].clone().requires_grad_()
        yo = m(ur, cp_group=dist.group.WORLD)
        yo.backward(gu[:, sl])
        out["by"], out["bgu"] = yo.detach().numpy(), ur.grad.numpy()
        for n, p_ in m.named_parameters():
            g = p_.grad.clone()
            dist.all_reduce(g)
            out["bg." + n] = g.numpy()
        q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_context_parallel_without_final_state_and_whole_block():
    import torch.multiprocessing as mp
    from emu.loader import use_emulator
    from omnimamba_amd.mamba2 import Mamba2
    from omnimamba_amd.ssd_combined import mamba_chunk_scan_combined
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31300 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_worker_nofinal, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    cat = lambda k: torch.cat([torch.from_numpy(got[r][k]) for r in range(2)], dim=1)
    with use_emulator():
        x, dt, A, Bm, Cm, D, dtb, z, init, gy, gf = _inputs(torch.float32)
        lv = [t.clone().requires_grad_() for t in (x, dt, Bm, Cm)]
        y = mamba_chunk_scan_combined(lv[0], lv[1], A, lv[2], lv[3], 16, D=D, dt_bias=dtb, dt_softplus=True)
        y.backward(gy)
        assert rel(cat("y"), y.detach()) < 2e-5 and rel(cat("gx"), lv[0].grad) < 3e-4 and rel(cat("gB"), lv[2].grad) < 3e-4
        torch.manual_seed(0)
        m = Mamba2(32, d_state=16, headdim=8, ngroups=2, chunk_size=16)
        u = torch.randn(2, 96, 32, generator=torch.Generator().manual_seed(9)).requires_grad_()
        gu = torch.randn(2, 96, 32, generator=torch.Generator().manual_seed(10))
        yo = m(u)
        yo.backward(gu)
    assert rel(cat("by"), yo.detach()) < 3e-5
    assert rel(cat("bgu"), u.grad) < 3e-4
    for n, p_ in m.named_parameters():
        assert rel(torch.from_numpy(got[0]["bg." + n]), p_.grad) < 5e-4, n


@pytest.mark.parametrize("L,split", [(384, 192), (330, 128)])
def test_two_shards_in_sequence_bf16_mfma_vs_oracle(dev, L, split):
    """The three steps of the context-parallel scan walked in ONE process: state-only pass of shard 0 (omk_ssd_scan_fwd without
    `out`), fold, scan of shard 1 from the folded state -- bf16, headdim 64, d_state 128: the MFMA kernels -- against the oracle
    on the WHOLE sequence, with the budget of the unsharded scan (tests/tolerances.py).  Also: the state-only pass == the final
    state the full scan reports."""
    import oracle as O
    from tolerances import forward_budget
    from omnimamba_amd.context_parallel import shard_states, start_states_from_shards
    from omnimamba_amd.ssd_combined import mamba_chunk_scan_combined, ssd_scan_fwd
    g = torch.Generator().manual_seed(4)
    Bsz, H, P, N, G = 1, 2, 64, 128, 1
    x = torch.randn(Bsz, L, H, P, generator=g).bfloat16()
    dt = (torch.randn(Bsz, L, H, generator=g) * 0.5).bfloat16()
    A = -(torch.rand(H, generator=g) * 6 + 0.5)
    Bm, Cm = torch.randn(Bsz, L, G, N, generator=g).bfloat16(), torch.randn(Bsz, L, G, N, generator=g).bfloat16()
    D, dtb = torch.randn(H, generator=g), torch.randn(H, generator=g) * 0.5 - 2.5
    d = lambda t: t.to(dev)
    cuts = [slice(0, split), slice(split, L)]
    with torch.no_grad():
        S, ld = zip(*[shard_states(d(x[:, c]), d(dt[:, c]), d(A), d(Bm[:, c]), dt_bias=d(dtb), dt_softplus=True) for c in cuts])
        S_all, ld_all = torch.stack(S), torch.stack(ld)
        ys = []
        for r, c in enumerate(cuts):
            s_in = start_states_from_shards(S_all, ld_all, r)
            ys.append(mamba_chunk_scan_combined(d(x[:, c]), d(dt[:, c]), d(A), d(Bm[:, c]), d(Cm[:, c]), 256, D=d(D), dt_bias=d(dtb),
                                                initial_states=s_in, dt_softplus=True))
        final = start_states_from_shards(S_all, ld_all, 2)
        _, _, fin0 = ssd_scan_fwd(d(x[:, cuts[0]]), d(dt[:, cuts[0]]), d(A), d(Bm[:, cuts[0]]), d(Cm[:, cuts[0]]), dt_bias=d(dtb), dt_softplus=True,
                                  return_final_states=True)
    y = torch.cat(ys, 1)
    y64, f64, by, bf, up = forward_budget(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)
    q = rel(y64.float().bfloat16().float(), y64)
    assert rel(y.float().cpu(), y64) < (by ** 2 + q ** 2) ** 0.5, (rel(y.float().cpu(), y64), by, q, up)
    assert rel(final.cpu(), f64) < bf
    assert rel(S[0].cpu(), fin0.cpu()) < 1e-6          # same kernel arithmetic with and without the output phases

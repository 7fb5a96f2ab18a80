#!/usr/bin/env python
"""bench.py -- BASELINE.json configs[1]: one Mamba-2 block (OmniMamba-1.3B shape), forward + backward,
B=8 L=4096 d_model=2048 d_state=128, bf16 autocast over fp32 parameters, synthetic data, on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0.  value = selective-scan M-elements/s of the whole job =
world * B * L * (H*P = 4096 scanned channels) / step_time / 1e6, a step being fwd + bwd of the block (in_proj GEMM ->
fused conv1d + SSD scan + gated RMSNorm + out_proj node and its backward; with N > 1 the block is wrapped in DDP so
the parameter gradients are all-reduced over RCCL, overlapped with backward).  `roofline` is for the dominant
hand-written kernel (the SSD scan forward, algorithmic bytes SURVEY.md section 8d) timed with HIP events on the launch
stream inside the timed region; `cpu_baseline` is the CPU oracle (a port, not the reference: mamba_ssm is absent) on a
bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_LOCAL, SEQ, D_MODEL, D_STATE, HEADDIM = 8, 4096, 2048, 128, 64
H = 2 * D_MODEL // HEADDIM
D_SCAN = H * HEADDIM
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: 8 TB/s spec
SCAN_FWD_BYTES_PER_TOK = 2 * D_SCAN * 2 + 2 * D_STATE * 2 + H * 2      # x in, y out, B, C, dt   = 17,024 B
SCAN_BWD_BYTES_PER_TOK = 3 * D_SCAN * 2 + 4 * D_STATE * 2 + 2 * H * 2  # x, dy in, dx out, B, C, dB, dC, dt, ddt = 25,856 B


def cpu_baseline(seconds_budget=20.0):
    """CPU oracle of the SAME workload (oracle.mamba2_forward_ref: in_proj -> conv1d+SiLU -> SSD chunked scan -> gated
    RMSNorm -> out_proj, forward + autograd backward, fp32) on a bounded sample: batch 1, L grown until the budget is
    used; all host cores.  A port of the upstream semantics, not the reference's own code (mamba_ssm is absent)."""
    import oracle as O
    torch.manual_seed(0)
    n = torch.get_num_threads()
    p = O.Mamba2RefParams.random(D_MODEL, headdim=HEADDIM, d_state=D_STATE, chunk_size=256, seed=0)
    for v in p.__dict__.values():
        if torch.is_tensor(v):
            v.requires_grad_()
    L, spent, best = 512, 0.0, None
    while True:
        u = torch.randn(1, L, D_MODEL, requires_grad=True)
        t0 = time.perf_counter()
        y = O.mamba2_forward_ref(p, u)
        y.backward(torch.randn_like(y))
        dt_s = time.perf_counter() - t0
        spent += dt_s
        best = (L, dt_s)
        if spent + 2.2 * dt_s > seconds_budget or L >= SEQ:
            break
        L *= 2
    L, dt_s = best
    return {"value": round(L * D_SCAN / dt_s / 1e6, 3), "unit": "M-elements/s", "cores": n, "kind": "port",
            "sample": f"oracle.mamba2_forward_ref block fwd+bwd, B=1 L={L} d_model={D_MODEL} fp32, {dt_s:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from omnimamba_amd import _prof
    from omnimamba_amd._lib import get_lib
    from omnimamba_amd.gemm_tuning import use_tuned_gemms
    from omnimamba_amd.mamba2 import Mamba2
    assert get_lib().omk_is_emulated() == 0
    tuned = use_tuned_gemms()                  # recorded hipBLASLt / rocBLAS solutions for the block's GEMMs (no tuning here)

    torch.manual_seed(0)                       # identical random-init weights on every rank
    block = Mamba2(D_MODEL, d_state=D_STATE, headdim=HEADDIM, layer_idx=0, device=dev)
    model = block
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        # 32 MB buckets: out_proj.weight (33.5 MB, ready at the very start of backward) gets a bucket of its own and its
        # all-reduce overlaps the whole backward; with 64 MB it would wait for the small parameters whose gradients come
        # out of the scan / conv backward.  in_proj.weight (70 MB) is last either way: it overlaps the dx GEMM (linear.py).
        model = DDP(block, device_ids=[local_rank], gradient_as_bucket_view=True, bucket_cap_mb=32)
    torch.manual_seed(1234 + rank)             # per-rank synthetic batch (weak scaling: B_LOCAL per GPU)
    u = torch.randn(B_LOCAL, SEQ, D_MODEL, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(B_LOCAL, SEQ, D_MODEL, device=dev, dtype=torch.bfloat16)

    def step():
        ur = u.detach().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(ur)
        y.backward(dy)
        for p in block.parameters():
            p.grad = None

    for _ in range(args.warmup):
        step()
    _prof.ENABLED = True
    _prof.reset()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    _prof.ENABLED = False
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B_LOCAL * SEQ * D_SCAN / (elapsed / args.steps) / 1e6

    if rank == 0:
        prof = _prof.summary()
        tok = B_LOCAL * SEQ
        n_f, ms_f = prof.get("ssd_scan_fwd", (0, float("nan")))
        n_b, ms_b = prof.get("ssd_scan_bwd", (0, float("nan")))
        fwd_bytes = tok * SCAN_FWD_BYTES_PER_TOK
        ach = fwd_bytes / (ms_f * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of the same kernel at this shape (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); counters cannot be read inside this process
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "ssd_fwd_traffic.json")) as fh:
                tj = json.load(fh)
                traffic, traffic_src = int(tj["traffic_bytes_per_launch"]), tj.get("source")
        except (OSError, KeyError, ValueError):
            pass
        out = {
            "metric": "selective-scan M-elements/sec", "value": round(value, 1), "unit": "M-elements/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "single Mamba-2 block fwd+bwd (BASELINE.json configs[1])", "batch_per_gpu": B_LOCAL,
                       "global_batch": B_LOCAL * world, "seq_len": SEQ, "d_model": D_MODEL, "d_state": D_STATE,
                       "headdim": HEADDIM, "nheads": H, "params": "fp32 master, bf16 autocast",
                       "parallelism": f"dp{world}" if world > 1 else "single", "library_gemm_solutions": "recorded (TunableOp file)" if tuned else "default"},
            "roofline": {"bound": "hbm", "kernel": "omk_ssd_scan_fwd (ssd_mfma_a3_kernel<GS_Y> + ssd_dt_prep_kernel)",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": fwd_bytes, "launch_ms": round(ms_f, 4),
                         "launches_timed": n_f},
            "scan_bwd": {"launch_ms": round(ms_b, 4), "algorithmic_bytes_per_launch": tok * SCAN_BWD_BYTES_PER_TOK,
                         "achieved_GBs": round(tok * SCAN_BWD_BYTES_PER_TOK / (ms_b * 1e-3) / 1e9, 1), "launches_timed": n_b},
            "tokens_per_s": round(world * tok / (elapsed / args.steps), 1),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

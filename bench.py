#!/usr/bin/env python
"""bench.py -- BASELINE.json configs[1]: one Mamba-2 block (OmniMamba-1.3B shape), forward + backward,
B=8 L=4096 d_model=2048 d_state=128, bf16 autocast over fp32 parameters, synthetic data, on N GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE
in the environment) or started plainly, in which case it re-executes itself under torch.distributed.run on 127.0.0.1.

Prints ONE JSON line on rank 0.  value = selective-scan M-elements/s of the whole job =
world * B * L * (H*P = 4096 scanned channels) / step_time / 1e6, a step being fwd + bwd of the block (in_proj GEMM ->
fused conv1d + SSD scan + gated RMSNorm + out_proj node and its backward; with N > 1 the block is wrapped in DDP so the
parameter gradients are all-reduced over RCCL, overlapped with backward).  The timed region is EXACTLY K steps (`steps` = K);
a sustained region of the same step (>= --min-seconds, default 5 s: the driver's utilisation sampler has a 5 s period) follows and
is reported separately under `sustained`.
  roofline / roofline_bwd   the SSD scan forward / backward launches (algorithmic bytes of SURVEY.md section 8d) timed
                            with HIP events on the launch stream inside the timed region
  cpu_baseline              the CPU oracle (a port: mamba_ssm is absent) on a bounded sample of the same workload
  train_1p3b                BASELINE configs[3]: OmniMamba-1.3B stage-1 MMU step (projector + MMU LoRA train), L=2048,
                            bf16 autocast, DDP over RCCL when N > 1 -- tokens/s of the whole job
  train_1p3b_stage2         BASELINE configs[4]: stage-2 unified fine-tune step, one T2I + one MMU forward of L=8192 each,
                            every parameter trains (5.9 GB of fp32 gradients all-reduced per step when N > 1)
  selscan_cfg1              BASELINE configs[0]: Mamba-1 selective_scan at B2 L1024 D768 N16 fp32 -- HIP kernel next
                            to the CPU selective_scan_ref restatement on the host cores
  scan_target               the north-star target shape: the scan alone at L = 8192, d_model 2048, B = 8 and B = 1
  decode_1p3b               BASELINE configs[2]: 1.3B T2I greedy decode, 72-token prompt + 256 tokens: time to first token, ms/token
"""
import argparse
import json
import math
import os
import socket
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


def selscan_cfg1(dev):
    """BASELINE configs[0] (SURVEY.md section 8d 'Cfg 1 inputs'): the host restatement of selective_scan_ref, best of 3, next to the HIP selective_scan_fn on the same tensors (and at 32x the batch, where the launch is long
    enough to read a bandwidth from)."""
    import oracle as O
    from omnimamba_amd.selective_scan import selective_scan_fn
    torch.manual_seed(0)
    Bsz, Dm, L, N = 2, 768, 1024, 16
    u, delta = torch.randn(Bsz, Dm, L), torch.rand(Bsz, Dm, L) * 0.5
    A = -(torch.rand(Dm, N) + 0.1)
    Bm, Cm = torch.randn(Bsz, N, L), torch.randn(Bsz, N, L)
    D, z, db = torch.randn(Dm), torch.randn(Bsz, Dm, L), 0.1 * torch.randn(Dm)
    best = float("inf")
    for _ in range(3):                    # ~7 s per run on the GPU box's 128 host cores (a Python loop over 1024 steps)
        t0 = time.perf_counter()
        ref = O.selective_scan_ref(u, delta, A, Bm, Cm, D, z, db, True)
        best = min(best, time.perf_counter() - t0)
    out = {"shape": {"B": Bsz, "L": L, "D": Dm, "N": N, "dtype": "f32"},
           "cpu_ref": {"value": round(Bsz * L * Dm / best / 1e6, 3), "unit": "M-elements/s", "cores": torch.get_num_threads(),
                       "kind": "port", "sample": f"oracle.selective_scan_ref, best of 3, {best * 1e3:.1f} ms"}}
    def timed(fn, n, warm):
        # best of three back-to-back blocks of n calls: at B = 2 a call is ~30 us and host bound, and the host has just run 128 threads of
        # the CPU restatement -- one slow block (0.18 ms per call on one box of round 5) is the host, not the kernel
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best_ms = float("inf")
        for _ in range(3):
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best_ms = min(best_ms, e0.elapsed_time(e1) / n)
        return best_ms

    for rep in (1, 32):
        g = [t.to(dev).repeat(*([rep] + [1] * (t.dim() - 1))) if t.dim() == 3 else t.to(dev) for t in (u, delta, A, Bm, Cm, D, z, db)]
        got = selective_scan_fn(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], True)
        if rep == 1:
            err = ((got.cpu() - ref).norm() / ref.norm()).item()
            assert err < 1e-3, f"selective_scan_fn vs selective_scan_ref: rel-L2 {err:.2e}"
            out["rel_l2_vs_cpu_ref"] = err
        ms = timed(lambda: selective_scan_fn(g[0], g[1], g[2], g[3], g[4], g[5], g[6], g[7], True), 50, 5)
        nb = rep * Bsz * L * (4 * Dm * 4 + 2 * N * 4)          # u, delta, z in, out + B, C
        out[f"hip_B{rep * Bsz}"] = {"value": round(rep * Bsz * L * Dm / (ms * 1e-3) / 1e6, 1), "unit": "M-elements/s", "launch_ms": round(ms, 4),
                                    "algorithmic_bytes": nb, "achieved_GBs": round(nb / (ms * 1e-3) / 1e9, 1),
                                    "frac_of_hbm_peak": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        # forward + backward through autograd (the backward alone = the difference; its Python wrapper costs ~0.2 ms at B = 2)
        leaves = [t.detach().clone().requires_grad_() for t in g]
        go = torch.randn_like(got)

        def fb():
            for t in leaves:
                t.grad = None
            selective_scan_fn(*leaves, True).backward(go)
        out[f"hip_B{rep * Bsz}"]["fwd_bwd_ms"] = round(timed(fb, 20, 3), 4)
        if rep > 1:
            # the same tensors as channel-last (B, L, D) views -- what the Mamba-1 module holds after its in_proj, the north star's
            # "(B, L, D) laid out for coalesced HBM loads": read and written as they lie by the lanes-are-channels sweep (selscan.hip)
            cl = [t.transpose(1, 2).contiguous().transpose(1, 2) if t.dim() == 3 else t for t in g]
            got_cl = selective_scan_fn(cl[0], cl[1], cl[2], cl[3], cl[4], cl[5], cl[6], cl[7], True)
            assert got_cl.stride(1) == 1 and torch.allclose(got_cl, got, rtol=1e-4, atol=1e-4)
            ms = timed(lambda: selective_scan_fn(cl[0], cl[1], cl[2], cl[3], cl[4], cl[5], cl[6], cl[7], True), 50, 5)
            out[f"hip_B{rep * Bsz}_channel_last"] = {"value": round(rep * Bsz * L * Dm / (ms * 1e-3) / 1e6, 1), "unit": "M-elements/s", "launch_ms": round(ms, 4),
                                                     "algorithmic_bytes": nb, "achieved_GBs": round(nb / (ms * 1e-3) / 1e9, 1),
                                                     "frac_of_hbm_peak": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            # forward + backward on the same channel-last views (round 6: selscan_bwd_lanes_kernel walks them as they lie; until then the
            # backward made L-contiguous copies for the chunked scan)
            lcl = [t.detach().clone().requires_grad_() for t in cl]
            gcl = torch.randn_like(got_cl)

            def fb_cl():
                for t in lcl:
                    t.grad = None
                selective_scan_fn(*lcl, True).backward(gcl)
            out[f"hip_B{rep * Bsz}_channel_last"]["fwd_bwd_ms"] = round(timed(fb_cl, 20, 3), 4)
    return out


def train_1p3b(dev, rank, world, steps=10, warmup=2, batch=8, seqlen=2048, stage2=False, dist_on=None):
    """BASELINE configs[3]: OmniMamba-1.3B stage-1 MMU pretrain step on synthetic image features + text ids, L = 2048:
    images_feat (B, 729, 2176) -> projector, text ids of length L - 733, labels = ids; stage 'align' with only the MMU
    task configured (projector + MMU LoRA adapters train); bf16 autocast, AdamW, clip 1.0.
    stage2=True: BASELINE configs[4] -- stage 'finetune', one T2I + one MMU forward of L tokens each per step, one backward,
    every parameter of the stack trains (position tables sized >= L: the documented deviation from the reference's caps)."""
    from omnimamba_amd.omni import OmniMambaPath
    from omnimamba_amd.stack import StackConfig
    from omnimamba_amd.train import Stage2Step, TrainConfig, synthetic_batch, wrap_ddp
    dist_on = world > 1 if dist_on is None else dist_on
    torch.manual_seed(0)
    tasks = ("t2i", "mmu") if stage2 else ("mmu",)
    cfg = StackConfig.omnimamba_1_3b(t2i_task=stage2, mmu_task=True, mmu_positions=max(seqlen, 1500), t2i_positions=max(seqlen, 329))
    model = OmniMambaPath(cfg, stage="finetune" if stage2 else "align", device=dev, dtype=torch.float32)
    tc = TrainConfig()
    net = wrap_ddp(model, tc, device_ids=[dev.index]) if dist_on else None
    step = Stage2Step(model, tc, ddp_model=net)
    data = synthetic_batch(cfg, batch, seqlen, dev, torch.bfloat16, rank=rank, tasks=tasks)
    for _ in range(warmup):
        step(data)
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    # per-step HIP events on the current stream (rank 0's own steps; the headline figure is the barrier-bracketed wall time)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    step.time_backward = True
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        step(data)
        evs[i + 1].record()
    torch.cuda.synchronize()
    if dist_on:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    bwd = sorted(step.take_backward_ms())
    step.time_backward = False
    bwd_ms = bwd[len(bwd) // 2] if bwd else float("nan")
    loss = sum(float(step.last[t]) for t in tasks)
    assert math.isfinite(loss)
    out = {"tokens_per_s": round(world * len(tasks) * batch * seqlen * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 2), "steps": steps,
           "ms_per_step_min": round(per[0], 2), "ms_per_step_median": round(per[len(per) // 2], 2), "ms_per_step_max": round(per[-1], 2),
           "warmup": warmup, "loss": round(loss, 4), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
           "config": {"workload": "OmniMamba-1.3B stage-2 unified fine-tune step, T2I + MMU (BASELINE.json configs[4])" if stage2 else
                                  "OmniMamba-1.3B stage-1 MMU pretrain step (BASELINE.json configs[3])", "n_layer": cfg.n_layer,
                      "d_model": cfg.d_model, "seq_len": seqlen, "batch_per_gpu": batch, "global_batch": batch * world,
                      "trainable_params": sum(p.numel() for p in model.parameters() if p.requires_grad),
                      "params": sum(p.numel() for p in model.parameters()), "dtype": "bf16 autocast, fp32 masters",
                      "parallelism": f"dp{world}" if world > 1 else "single"}}
    out["comm_model"] = comm_model(out["config"]["trainable_params"], tc.bucket_cap_mb, dt / steps * 1e3, bwd_ms, world)
    if not dist_on and world == 1 and dev.type == "cuda":
        # what the DDP wrapper itself costs (bucket copies, hooks, the RCCL all-reduce of one rank): the SAME model and optimizer behind
        # DistributedDataParallel on an nccl group of one, step time minus the plain step time above (next #6)
        import torch.distributed as dist
        own_pg = not dist.is_initialized()
        try:
            if own_pg:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            net = wrap_ddp(model, tc, device_ids=[dev.index])
            step.net = net
            for _ in range(2):
                step(data)
            torch.cuda.synchronize()
            n2 = max(3, steps // 2)
            ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(n2 + 1)]
            ev2[0].record()
            for i in range(n2):
                step(data)
                ev2[i + 1].record()
            torch.cuda.synchronize()
            per2 = sorted(ev2[i].elapsed_time(ev2[i + 1]) for i in range(n2))
            out["ddp_world1"] = {"ms_per_step_median": round(per2[len(per2) // 2], 2), "steps": n2, "backend": "nccl (RCCL), world size 1",
                                 "grad_compression": tc.grad_compression}
            out["ddp_overhead_ms"] = round(per2[len(per2) // 2] - per[len(per) // 2], 2)
        except Exception as e:   # noqa: BLE001 -- a reported extra, never a reason to lose the bench line
            out["ddp_overhead_ms"] = None
            out["ddp_world1"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        finally:
            if own_pg and dist.is_initialized():
                dist.destroy_process_group()
    del step, net, model
    torch.cuda.empty_cache()
    return out


XGMI_LINK_GBS = 153.0     # per direction and link; 7 links per GPU, point to point (MI355X_MICROARCH.md / task brief)


def comm_model(trainable_params, bucket_cap_mb, step_ms, backward_ms, world):
    """when the N devices are not there, the modelled gradient all-reduce next to the measured backward it has
    to hide under.  fp32 gradients (4 B per trainable parameter; 2 B with TrainConfig.grad_compression = 'bf16'), DDP buckets of
    bucket_cap_mb, reduce-scatter + all-gather volume 2 (N - 1) / N x bytes per GPU.  Two bounds for the time: ONE ring (every hop
    on one 153 GB/s link) and ALL links (the N - 1 peers reached directly, each over its own link -- what a fully connected xGMI
    node allows).  The all-reduce of a bucket starts when the backward has produced it, so all but the last bucket can run under
    the backward; the last one (the first layers' gradients) has nothing left to hide behind:
    exposed = max(t_allreduce / buckets, t_allreduce - backward x (buckets - 1) / buckets).  Predicted
    efficiency = step / (step + exposed); the measured numbers at N > 1 come from the driver's SCALE run of this file."""
    out = {"measured_on_gpus": world, "step_ms": round(step_ms, 2), "backward_ms": round(backward_ms, 2), "trainable_params": trainable_params,
           "xgmi_link_GBs": XGMI_LINK_GBS, "links_per_gpu": 7, "predicted": {}}
    for comp, bpe in (("fp32", 4), ("bf16_compression", 2)):
        nbytes = trainable_params * bpe
        buckets = max(1, math.ceil(trainable_params * 4 / (bucket_cap_mb * 2 ** 20)))     # buckets are cut on the fp32 gradients
        row = {"gradient_bytes": nbytes, "buckets": buckets}
        for n in (2, 4, 8):
            vol = 2.0 * (n - 1) / n * nbytes
            t_ring = vol / (XGMI_LINK_GBS * 1e9) * 1e3
            t_all = vol / (XGMI_LINK_GBS * 1e9 * (n - 1)) * 1e3
            hide = backward_ms * (buckets - 1) / buckets
            ex_ring, ex_all = max(t_ring / buckets, t_ring - hide), max(t_all / buckets, t_all - hide)
            row[f"n{n}"] = {"allreduce_ms_one_ring": round(t_ring, 2), "allreduce_ms_all_links": round(t_all, 2),
                            "exposed_ms_one_ring": round(ex_ring, 2), "exposed_ms_all_links": round(ex_all, 2),
                            "efficiency_one_ring": round(step_ms / (step_ms + ex_ring), 4),
                            "efficiency_all_links": round(step_ms / (step_ms + ex_all), 4)}
        out["predicted"][comp] = row
    return out


def scan_target(dev):
    """The north-star target shape: the Mamba-2 chunked scan (omk_ssd_scan_fwd through mamba_chunk_scan_combined) at L = 8192,
    d_model 2048 (H 64, P 64, N 128, one group), bf16, B = 8 and B = 1, and the plain (inference-form) forward at the shape of the
    timed step (B 8, L 4096); HIP events on the launch stream, algorithmic bytes of SURVEY.md section 8d (17 024 B per token)."""
    from omnimamba_amd.ssd_combined import mamba_chunk_scan_combined, scan_options

    def time_it(run, n=40):
        with torch.no_grad():
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                run()
            e1.record()
            torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    out = {}
    for L, Bsz in ((4096, 8), (8192, 8), (8192, 1)):   # (the first: the plain forward at the shape of the timed step, whose own forward also writes window states)
        torch.manual_seed(0)
        x = torch.randn(Bsz, L, H, HEADDIM, device=dev, dtype=torch.bfloat16)
        dt = (torch.randn(Bsz, L, H, device=dev) * 0.5).bfloat16()
        A = -(torch.rand(H, device=dev) * 15 + 1)
        Bm, Cm = (torch.randn(Bsz, L, 1, D_STATE, device=dev, dtype=torch.bfloat16) for _ in range(2))
        D, dtb = torch.ones(H, device=dev), torch.randn(H, device=dev) * 0.5 - 2
        run = lambda: mamba_chunk_scan_combined(x, dt, A, Bm, Cm, 256, D=D, dt_bias=dtb, dt_softplus=True)
        ms = time_it(run)
        nb = Bsz * L * SCAN_FWD_BYTES_PER_TOK
        out[f"B{Bsz}_L{L}"] = {"launch_ms": round(ms, 4), "algorithmic_bytes": nb, "achieved_GBs": round(nb / (ms * 1e-3) / 1e9, 1),
                               "frac_of_hbm_peak": round(nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                               "M_elements_per_s": round(Bsz * L * D_SCAN / (ms * 1e-3) / 1e6, 1)}
        if Bsz == 8:
            # the same launch with the per-call parity options (OmkSsdFwd.flags): what the bare 1e-3 on slow-decay heads costs
            with scan_options(khilo=True):
                mk = time_it(run, 20)
            with scan_options(precise=True):
                mp = time_it(run, 20)
            out[f"B{Bsz}_L{L}"]["options"] = {"khilo_launch_ms": round(mk, 4), "khilo_price": round(mk / ms - 1, 3),
                                               "precise_launch_ms": round(mp, 4), "precise_price": round(mp / ms - 1, 3)}
    return out


def decode_1p3b(dev):
    """BASELINE configs[2]: OmniMamba-1.3B T2I autoregressive decode -- 72-token prompt, 256 greedy image tokens, batch 1, fp32
    weights (the reference's inference default, scripts/inference_t2i.py:21-26), hipGraph replay of the single-token step.
    time_to_first_token = prefill of the prompt + the first sampled id (one host sync); ms_per_token = the 256-token loop."""
    from omnimamba_amd.generation import decode
    from omnimamba_amd.stack import OmniMambaLM, StackConfig
    torch.manual_seed(0)
    cfg = StackConfig.omnimamba_1_3b()
    model = OmniMambaLM(cfg, device=dev, dtype=torch.float32).eval()
    P, new = 72, 256
    ids = torch.zeros(1, P, dtype=torch.long, device=dev)
    emb = torch.randn(1, P, cfg.d_model, device=dev) * 0.02 + model.backbone.pos_embed[:, :P]
    decode(ids, emb, model, P + new, top_k=1, task="t2i", cg=True)            # warm-up: captures the graph
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    decode(ids, emb, model, P + 1, top_k=1, task="t2i", cg=True)
    torch.cuda.synchronize()
    ttft = time.perf_counter() - t0
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        seq = decode(ids, emb, model, P + new, top_k=1, task="t2i", cg=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert seq.shape == (1, P + new)
    n_param = sum(p.numel() for p in model.parameters())
    ms_tok = (best - ttft) / (new - 1) * 1e3
    out = {"workload": "OmniMamba-1.3B T2I greedy decode (BASELINE.json configs[2])", "batch": 1, "prompt": P, "new_tokens": new,
           "time_to_first_token_ms": round(ttft * 1e3, 3), "ms_per_token": round(ms_tok, 4), "tokens_per_s": round(1e3 / ms_tok, 1),
           "total_ms": round(best * 1e3, 2), "weights_GBs": round(n_param * 4 / (ms_tok * 1e-3) / 1e9, 1),
           "frac_of_hbm_peak": round(n_param * 4 / (ms_tok * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "params": n_param, "dtype": "f32"}
    # BASELINE.md cfg 3 also lists batch 8 / 32 and bf16 weights (scripts/inference_t2i.py:29-46 batches prompts): ms per decode STEP,
    # tokens/s of the whole batch, bytes per step = weights + the recurrent state read and written (48 layers x (conv + ssm state))
    def more(model_, bsz, wbytes, new_=64):
        ids_ = torch.zeros(bsz, P, dtype=torch.long, device=dev)
        emb_ = (torch.randn(bsz, P, cfg.d_model, device=dev) * 0.02 + model_.backbone.pos_embed[:, :P].float()).to(next(model_.parameters()).dtype)
        model_._decoding_cache = None
        decode(ids_, emb_, model_, P + new_, top_k=1, task="t2i", cg=True)
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        decode(ids_, emb_, model_, P + 1, top_k=1, task="t2i", cg=True)
        torch.cuda.synchronize()
        tf_ = time.perf_counter() - t0_
        b_ = float("inf")
        for _ in range(2):
            t0_ = time.perf_counter()
            decode(ids_, emb_, model_, P + new_, top_k=1, task="t2i", cg=True)
            torch.cuda.synchronize()
            b_ = min(b_, time.perf_counter() - t0_)
        ms_ = (b_ - tf_) / (new_ - 1) * 1e3
        sbytes = 2 * bsz * cfg.n_layer * (64 * 64 * 128 + (2 * cfg.d_model + 2 * 128) * 4) * wbytes
        model_._decoding_cache = None
        return {"ms_per_step": round(ms_, 4), "tokens_per_s": round(bsz * 1e3 / ms_, 1),
                "GBs_weights_plus_state": round((n_param * wbytes + sbytes) / (ms_ * 1e-3) / 1e9, 1)}
    out["batches"] = {"f32_B8": more(model, 8, 4), "f32_B32": more(model, 32, 4)}
    model = model.to(torch.bfloat16)
    out["batches"]["bf16_B1"] = more(model, 1, 2, new_=128)
    out["batches"]["bf16_B8"] = more(model, 8, 2)
    # the VQ decode tail behind the 256 ids (mamba_vlm.py:104-108; VQ-16 geometry, random weights): ids -> 3 x 256 x 256 pixels
    from omnimamba_amd.vq_tail import VQDecodeTail
    tail = VQDecodeTail().to(dev).eval()
    img_ids = seq[:, P:].clamp(0, 16383)

    def t_ms(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        out["vq_tail_ms"] = {"eager_f32": round(t_ms(lambda: tail.decode_to_img(img_ids)), 3),
                             "graph_f32": round(t_ms(lambda: tail.graphed(img_ids)), 3),
                             "graph_bf16": round(t_ms(lambda: tail.graphed(img_ids, torch.bfloat16)), 3),
                             "params": sum(p.numel() for p in tail.parameters())}
        # (round 6: the channels-last variant of the library convolutions is no longer timed here -- MIOpen's bf16 NHWC kernels for the
        # decoder's 3 x 3 convolutions were 2.6 x SLOWER than its NCHW ones on this stack, 17.2 against 6.9 ms in BENCH_r05; the tail
        # keeps NCHW, VQDecodeTail.set_channels_last stays as an option for library versions where that changes)
    del model, tail
    torch.cuda.empty_cache()
    return out


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run ... bench.py <same args>`."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def main():
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")   # the VQ tail's library convolutions: no exhaustive find (5 s of naive kernels per run)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # read by HSA at initialisation: before the first torch.cuda call of this process
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--min-seconds", type=float, default=5.0, help="length of the sustained region behind the K timed steps (0 = none)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-1p3b", action="store_true")
    ap.add_argument("--no-selscan-cfg1", action="store_true")
    ap.add_argument("--no-scan-target", action="store_true")
    ap.add_argument("--no-decode", action="store_true")
    ap.add_argument("--backend", default=None, help="process-group backend (default nccl = RCCL; the CPU launch test passes gloo)")
    ap.add_argument("--dry-launch", action="store_true", help="only initialise the process group and report ranks (CPU test of the launch path)")
    ap.add_argument("--force-dist", action="store_true", help="take the multi-rank path (launcher, process group, DDP, barriers, max over ranks) even with --gpus 1: "
                                                              "runs the exact code of the N > 1 scaling runs on RCCL where only one GPU exists")
    args = ap.parse_args()

    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    # stdout carries ONE line, the JSON of rank 0.  Everything else a library writes there -- RCCL prints its version banner through C stdio as
    # soon as a process group exists (N > 1, --force-dist, the ddp_overhead_ms measurement), buffered on a pipe until exit, i.e. AFTER the
    # JSON -- goes to stderr: file descriptor 1 now points where 2 does, and the line is written to a duplicate of the original descriptor.
    sys.stdout.flush()
    real_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if args.dry_launch:
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(args.backend or "gloo")
            t = torch.tensor([float(rank)])
            dist.all_reduce(t)
            ok = t.item() == world * (world - 1) / 2
            dist.destroy_process_group()
        else:
            ok = True
        if rank == 0:
            real_out.write(json.dumps({"dry_launch": True, "n_gpus": world, "ok": bool(ok)}) + "\n")
            real_out.flush()
        return
    assert torch.cuda.is_available(), "bench.py needs the MI355X (there is no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(args.backend or "nccl", device_id=dev, rank=rank, world_size=world)

    from omnimamba_amd import _prof
    from omnimamba_amd._lib import get_lib
    from omnimamba_amd.gemm_tuning import use_tuned_gemms
    from omnimamba_amd.mamba2 import Mamba2
    assert get_lib().omk_is_emulated() == 0
    tuned = use_tuned_gemms()                  # recorded hipBLASLt / rocBLAS solutions for the block's GEMMs (no tuning here)

    torch.manual_seed(0)                       # identical random-init weights on every rank
    block = Mamba2(D_MODEL, d_state=D_STATE, headdim=HEADDIM, layer_idx=0, device=dev)
    model = block
    if dist_on:
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

    def timed(n):
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        return el

    for _ in range(args.warmup):
        step()
    # ---- the contract's timed region: EXACTLY K steps between barrier + synchronize pairs, max over ranks
    nsteps = args.steps
    _prof.ENABLED = True
    _prof.reset()
    elapsed = timed(nsteps)
    ms_per_step = elapsed / nsteps * 1e3
    value = world * B_LOCAL * SEQ * D_SCAN / (elapsed / nsteps) / 1e6
    # ---- a sustained region behind it (same step, >= --min-seconds): the driver's utilisation sampler has a 5 s period, K = 20
    # steps are 0.13 s; reported next to the headline, never instead of it.  The HIP-event timing of the scan launches runs over both.
    sustained = None
    if args.min_seconds > 0:
        n_sus = max(nsteps, math.ceil(1.05 * args.min_seconds / (elapsed / nsteps)))
        el_sus = timed(n_sus)
        sustained = {"steps": n_sus, "seconds": round(el_sus, 3), "ms_per_step": round(el_sus / n_sus * 1e3, 3),
                     "value": round(world * B_LOCAL * SEQ * D_SCAN / (el_sus / n_sus) / 1e6, 1)}
    _prof.ENABLED = False

    out = None
    if rank == 0:
        prof = _prof.summary()
        tok = B_LOCAL * SEQ
        n_f, ms_f = prof.get("ssd_scan_fwd", (0, float("nan")))
        n_b, ms_b = prof.get("ssd_scan_bwd", (0, float("nan")))
        fwd_bytes, bwd_bytes = tok * SCAN_FWD_BYTES_PER_TOK, tok * SCAN_BWD_BYTES_PER_TOK
        ach_f, ach_b = fwd_bytes / (ms_f * 1e-3) / 1e9, bwd_bytes / (ms_b * 1e-3) / 1e9
        # HBM bytes per launch from the PMC passes of the same kernels at this shape (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); counters cannot be read inside this process
        traffic = {}
        from omnimamba_amd.ssd_combined import save_window_states_enabled
        save_ws = save_window_states_enabled()
        # the kernels the timed launches really were (omk_ssd_last_kernels): a PMC file recorded for OTHER kernels is refused -- traffic and
        # mfma_busy are then null and traffic_source says why (tests/test_bench_contract.py asserts the match)
        ran = _prof.kernels()
        ids = {"fwd": ran.get("ssd_scan_fwd"), "bwd": ran.get("ssd_scan_bwd")}
        for key, fn in (("fwd", "ssd_fwd_traffic.json"), ("bwd", "ssd_bwd_traffic.json")):
            try:
                with open(os.path.join(ROOT, "profiles", fn)) as fh:
                    tj = json.load(fh)
                    want = tj["with_window_states"]["kernel_ids"] if (key == "fwd" and save_ws and "with_window_states" in tj) else tj["kernel_ids"]
                    if ids[key] is not None and want != ids[key]:
                        traffic[key] = (None, f"STALE: profiles/{fn} was recorded for [{want}], this run launched [{ids[key]}] -- re-run tools/pmc_r06.sh + tools/make_traffic_json.py")
                        continue
                    traffic[key] = (int(tj["traffic_bytes_per_launch"]), tj.get("source"))
                    traffic[key + "_mfma"] = tj.get("mfma_busy")
                    if key == "fwd" and "with_window_states" in tj:
                        traffic["fwd_ws"] = int(tj["with_window_states"]["traffic_bytes_per_launch"])
                        traffic["fwd_ws_mfma"] = tj["with_window_states"].get("mfma_busy")
            except (OSError, KeyError, ValueError):
                traffic[key] = (None, None)
        out = {
            "metric": "selective-scan M-elements/sec", "value": round(value, 1), "unit": "M-elements/s",
            "n_gpus": world, "steps": nsteps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "timed_seconds": round(elapsed, 4), "sustained": sustained,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "single Mamba-2 block fwd+bwd (BASELINE.json configs[1])", "batch_per_gpu": B_LOCAL,
                       "global_batch": B_LOCAL * world, "seq_len": SEQ, "d_model": D_MODEL, "d_state": D_STATE,
                       "headdim": HEADDIM, "nheads": H, "params": "fp32 master, bf16 autocast",
                       "parallelism": f"dp{world}" if world > 1 else "single", "process_group": (dist.get_backend() if dist_on else None), "library_gemm_solutions": "recorded (TunableOp file)" if tuned else "default"},
            "roofline": {"bound": "hbm", "kernel": "omk_ssd_scan_fwd as the training step launches it (ssd_a8_kernel<GS_Y, DUMP> + ssd_dt_prep_vec_kernel" + (
                             "; the forward also leaves its window states behind for the backward: +256 MiB of writes that are not algorithmic bytes; the plain forward of this shape is scan_target.B8_L4096)"
                             if save_ws else ")"),
                         "achieved": round(ach_f, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_f / HBM_PEAK_GBS, 4),
                         # (both instantiations -- with and without the window-state dumps -- are under the counters)
                         "traffic": None if traffic["fwd"][0] is None else (traffic.get("fwd_ws", traffic["fwd"][0] + B_LOCAL * ((SEQ + 127) // 128) * H * 16384) if save_ws else traffic["fwd"][0]),
                         "traffic_source": traffic["fwd"][1],
                         # share of the SIMD-cycles of the launch the matrix pipe is busy (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),
                         # from the same PMC passes as `traffic`: = the share of the dense bf16 MFMA peak at the clock the kernel runs at)
                         "mfma_busy": (traffic.get("fwd_ws_mfma") if save_ws else traffic.get("fwd_mfma")),
                         "kernel_ids": ids["fwd"],
                         "algorithmic_bytes_per_launch": fwd_bytes, "launch_ms": round(ms_f, 4), "launches_timed": n_f},
            "roofline_bwd": {"bound": "hbm", "kernel": "omk_ssd_scan_bwd (dt prep + " + ("" if save_ws else "state-only forward pass + ") + "dx scan with window-state dumps + ssd_cp_kernel + folds + finish" + (
                                 "; forward window states saved by the training forward)" if save_ws else ")"),
                             "achieved": round(ach_b, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach_b / HBM_PEAK_GBS, 4),
                             "traffic": traffic["bwd"][0], "traffic_source": traffic["bwd"][1], "mfma_busy": traffic.get("bwd_mfma"),
                             "kernel_ids": ids["bwd"],
                             "algorithmic_bytes_per_launch": bwd_bytes, "launch_ms": round(ms_b, 4), "launches_timed": n_b},
            "tokens_per_s": round(world * tok / (elapsed / nsteps), 1),
        }
    del model, block, u, dy
    torch.cuda.empty_cache()
    extra_s = None if (args.no_selscan_cfg1 or world > 1 or rank != 0) else selscan_cfg1(dev)
    extra_tg = None if (args.no_scan_target or world > 1 or rank != 0) else scan_target(dev)
    extra_d = None if (args.no_decode or world > 1 or rank != 0) else decode_1p3b(dev)
    extra_t = None if args.no_train_1p3b else train_1p3b(dev, rank, world, dist_on=dist_on)      # every rank takes part (DDP)
    torch.cuda.reset_peak_memory_stats()
    extra_t2 = None if args.no_train_1p3b else train_1p3b(dev, rank, world, steps=5, warmup=2, batch=2, seqlen=8192, stage2=True, dist_on=dist_on)   # two warm-up steps: the caching allocator settles in the second
    if rank == 0:
        out["train_1p3b"] = extra_t
        out["train_1p3b_stage2"] = extra_t2
        out["selscan_cfg1"] = extra_s
        out["scan_target"] = extra_tg
        out["decode_1p3b"] = extra_d
        out["cpu_baseline"] = cpu_baseline() if (not args.no_cpu_baseline and world == 1) else None
        # the JSON line is the ONLY line on the real stdout (see the top of main)
        sys.stdout.flush()
        real_out.write(json.dumps(out) + "\n")
        real_out.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

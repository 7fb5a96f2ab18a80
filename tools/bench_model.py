"""BASELINE.json configs[2..4] on the synthetic OmniMamba-1.3B stack (random init, synthetic data), 1 GPU or N GPUs via
torch.distributed.run:   python tools/bench_model.py [decode|train] [--batch B] [--seqlen L] [--steps K]
Prints one JSON line per workload (rank 0)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.generation import decode  # noqa: E402
from omnimamba_amd.omni import OmniMambaPath  # noqa: E402
from omnimamba_amd.stack import OmniMambaLM, StackConfig  # noqa: E402
from omnimamba_amd.train import Stage2Step, TrainConfig, init_distributed, synthetic_batch, wrap_ddp  # noqa: E402


def bench_decode(args, dev):
    """configs[2]: T2I autoregressive decode, 72-token prompt + 256 image tokens, greedy, hipGraph replay, fp32 weights
    (the reference inference scripts never cast the model: scripts/inference_t2i.py:21-26)."""
    torch.manual_seed(0)
    cfg = StackConfig.omnimamba_1_3b()
    wdt = torch.bfloat16 if args.weights == "bf16" else torch.float32   # the reference keeps fp32; bf16 = a model cast by the user
    model = OmniMambaLM(cfg, device=dev, dtype=wdt).eval()
    B, P, new = args.batch, 72, 256
    ids = torch.zeros(B, P, dtype=torch.long, device=dev)
    emb = torch.randn(B, P, cfg.d_model, device=dev, dtype=wdt) * 0.02 + model.backbone.pos_embed[:, :P].to(wdt)
    out = {}
    for cg in (True, False) if args.eager_too else (True,):
        decode(ids, emb, model, P + new, top_k=1, task="t2i", cg=cg)          # warm-up (captures the graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seq = decode(ids, emb, model, P + new, top_k=1, task="t2i", cg=cg)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert seq.shape == (B, P + new)
        out["graph" if cg else "eager"] = dt
    n_param = sum(p.numel() for p in model.parameters())
    ms_tok = out["graph"] / new * 1e3
    print(json.dumps({"workload": "OmniMamba-1.3B T2I decode (configs[2])", "batch": B, "prompt": P, "new_tokens": new,
                      "ms_per_token": round(ms_tok, 3), "tokens_per_s": round(B * new / out["graph"], 1),
                      "weights_GBs": round(n_param * (2 if wdt == torch.bfloat16 else 4) / (ms_tok * 1e-3) / 1e9, 1), "params": n_param,
                      "dtype": "bf16" if wdt == torch.bfloat16 else "f32",
                      "eager_ms_per_token": round(out["eager"] / new * 1e3, 3) if "eager" in out else None}), flush=True)


def bench_decode_mmu(args, dev):
    """MMU generation the way scripts/inference_mmu.py runs it: <|mmu|> <|soi|> [729 projected image positions] <|eoi|> <|sot|> + a 47-token
    question = 780 prompt positions (but 51 prompt ids), greedy, hipGraph replay of the step, fp32 weights; 128 new tokens."""
    from omnimamba_amd.omni import OmniMambaPath
    torch.manual_seed(0)
    cfg = StackConfig.omnimamba_1_3b()
    model = OmniMambaPath(cfg, stage="inference", device=dev, dtype=torch.float32)
    B, Q, new = args.batch, 47, 128
    q = torch.randint(0, 50000, (B, Q), device=dev)
    feat = torch.randn(B, 729, cfg.fused_vision_dim, device=dev)
    P = 4 + 729 + Q
    res = {}
    for n in (1, new):
        model.mmu_generate(feat, q, max_length=P + n, cg=True)                # warm-up (captures)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seq = model.mmu_generate(feat, q, max_length=P + n, cg=True)
        torch.cuda.synchronize()
        res[n] = time.perf_counter() - t0
        assert seq.shape == (B, 4 + Q + n)
    print(json.dumps({"workload": "OmniMamba-1.3B MMU generation (scripts/inference_mmu.py shape)", "batch": B, "prompt_positions": P,
                      "prompt_ids": 4 + Q, "new_tokens": new, "time_to_first_token_ms": round(res[1] * 1e3, 2),
                      "ms_per_token": round((res[new] - res[1]) / (new - 1) * 1e3, 3), "dtype": "f32"}), flush=True)


def bench_train(args, dev, rank, world):
    """configs[3]/[4]-style Stage-2 step: one T2I + one MMU forward, one backward, clip, AdamW; bf16 autocast over fp32
    master weights; DDP over RCCL when world > 1."""
    torch.manual_seed(0)
    L = args.seqlen
    tasks = tuple(args.tasks.split(","))
    cfg = StackConfig.omnimamba_1_3b(t2i_positions=max(L, 329), mmu_positions=max(L, 1500), t2i_task="t2i" in tasks, mmu_task="mmu" in tasks)
    model = OmniMambaPath(cfg, stage=args.stage, device=dev, dtype=torch.float32)
    tc = TrainConfig()
    net = wrap_ddp(model, tc, device_ids=[dev.index]) if world > 1 else None
    step = Stage2Step(model, tc, ddp_model=net)
    batch = synthetic_batch(cfg, args.batch, L, dev, torch.bfloat16, rank=rank, tasks=tasks)
    losses = []
    for _ in range(args.warmup):
        losses.append(step(batch))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(step(batch))
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = (time.perf_counter() - t0) / args.steps
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    if rank == 0:
        trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
        print(json.dumps({"workload": f"OmniMamba-1.3B step (stage {args.stage}), tasks {tasks} x L={L}", "n_gpus": world,
                          "batch_per_gpu": args.batch, "ms_per_step": round(dt * 1e3, 2),
                          "tokens_per_s": round(world * len(tasks) * args.batch * L / dt, 1), "trainable_params": trainable,
                          "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2), "dtype": "bf16 autocast",
                          "losses": [round(float(x["loss"] if isinstance(x, dict) else x), 4) for x in losses]}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["decode", "train"])
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)   # one warm-up step is not enough: optimizer state and GEMM heuristics settle in the second
    ap.add_argument("--stage", default="finetune")
    ap.add_argument("--tasks", default="t2i,mmu", help="t2i,mmu (stage 2) or mmu (stage-1 MMU pretrain, BASELINE configs[3])")
    ap.add_argument("--eager-too", action="store_true")
    ap.add_argument("--task", default="t2i", choices=["t2i", "mmu"], help="decode: T2I (configs[2]) or the MMU generation of scripts/inference_mmu.py")
    ap.add_argument("--weights", choices=["f32", "bf16"], default="f32")
    args = ap.parse_args()
    rank, local, world = init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if args.what == "decode":
        (bench_decode_mmu if args.task == "mmu" else bench_decode)(args, dev)
    else:
        bench_train(args, dev, rank, world)


if __name__ == "__main__":
    main()

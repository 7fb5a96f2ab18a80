"""Single-node data-parallel training step around the path (reference
/root/reference/trainer.py:113-127 compute_loss = t2i loss + mmu loss, two forwards / one backward;
train_stage2.py:16-44: bf16 autocast, AdamW(0.9, 0.95, wd 0), grad clip 1.0, cosine-with-min-lr).

One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on ROCm; "gloo" in the CPU tests).  DDP with
gradient_as_bucket_view and LARGE buckets: a ring on xGMI is per-link bound (7 links x ~153 GB/s), so RCCL needs big
messages to spread over all links; bucket all-reduce overlaps the (bandwidth-bound) backward.  No NCCL_ALGO=Tree
(train_stage2.py:48 -- an inter-node setting).  No per-step .item(): losses are logged from a detached device scalar.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class TrainConfig:
    lr: float = 1e-4
    min_lr_ratio: float = 0.1
    warmup_steps: int = 0
    max_steps: int = 1000
    betas: tuple = (0.9, 0.95)
    weight_decay: float = 0.0
    clip: float = 1.0
    bucket_cap_mb: int = 128
    amp_dtype: torch.dtype = torch.bfloat16


def cosine_with_min_lr(step, cfg: TrainConfig):
    if step < cfg.warmup_steps:
        return (step + 1) / max(1, cfg.warmup_steps)
    p = min(1.0, (step - cfg.warmup_steps) / max(1, cfg.max_steps - cfg.warmup_steps))
    return cfg.min_lr_ratio + (1 - cfg.min_lr_ratio) * 0.5 * (1 + math.cos(math.pi * p))


def init_distributed(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torch.distributed.run).  Returns (rank, local, world)."""
    import torch.distributed as dist
    rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def wrap_ddp(model, cfg: TrainConfig, device_ids=None):
    from torch.nn.parallel import DistributedDataParallel as DDP
    # two forwards share one backward and each activates only one task's adapters -> parameters are used by exactly one
    # of the two graphs; static_graph is not needed, but every trainable parameter must receive a gradient in the step,
    # which Stage2 (both tasks per step) guarantees.
    return DDP(model, device_ids=device_ids, gradient_as_bucket_view=True, bucket_cap_mb=cfg.bucket_cap_mb,
               broadcast_buffers=False)


def lm_loss(logits, labels):
    """Shift-by-one cross entropy (reference models/mamba_vlm.py:88-102); labels == -100 are ignored."""
    return F.cross_entropy(logits[:, :-1].reshape(-1, logits.shape[-1]).float(), labels[:, 1:].reshape(-1), ignore_index=-100)


class Stage2Step:
    """loss = t2i_loss + mmu_loss from two forwards, one backward, clip, AdamW, cosine LR (trainer.py:113-127)."""

    def __init__(self, model, cfg: TrainConfig, ddp_model=None):
        self.model, self.cfg = model, cfg
        self.net = ddp_model if ddp_model is not None else model
        params = [p for p in model.parameters() if p.requires_grad]
        # one fused kernel per parameter chunk on the GPU: the foreach form makes ~15 passes over the 1.39 B fp32 parameters
        # and their two moments (33 ms of a 480 ms step), the fused one a single read-modify-write
        fused = bool(params) and all(p.is_cuda for p in params) and os.environ.get("OMK_FUSED_ADAMW", "1") != "0"
        self.opt = torch.optim.AdamW(params, lr=cfg.lr, betas=cfg.betas, weight_decay=cfg.weight_decay, fused=fused)
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: cosine_with_min_lr(s, cfg))
        self.last = {}

    def __call__(self, batch):
        dev_type = "cuda" if next(self.model.parameters()).is_cuda else "cpu"
        self.opt.zero_grad(set_to_none=True)
        total = 0.0
        with torch.autocast(dev_type, dtype=self.cfg.amp_dtype, enabled=self.cfg.amp_dtype != torch.float32):
            for task in ("t2i", "mmu"):
                if task not in batch:
                    continue
                emb, labels = batch[task]
                out = self.net(None, emb, task=task)
                logits = out.t2i_logits if task == "t2i" else out.mmu_logits
                loss = lm_loss(logits, labels)
                self.last[task] = loss.detach()
                total = total + loss
        total.backward()
        if self.cfg.clip:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.grad is not None], self.cfg.clip)
        self.opt.step()
        self.sched.step()
        return total.detach()


def synthetic_batch(cfg, batch, seqlen, device, dtype, rank=0, tasks=("t2i", "mmu"), step=0):
    """Synthetic embeddings + labels of the reference shapes: seeds = 1234 + rank."""
    g = torch.Generator(device="cpu").manual_seed(1234 + rank + 1000003 * step)
    out = {}
    for task in tasks:
        vocab = cfg.vqvae_vocab_size if task == "t2i" else cfg.vocab_size
        emb = torch.randn(batch, seqlen, cfg.d_model, generator=g).to(device=device, dtype=dtype)
        labels = torch.randint(0, vocab, (batch, seqlen), generator=g).to(device)
        out[task] = (emb, labels)
    return out

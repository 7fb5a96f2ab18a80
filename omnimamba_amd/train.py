"""Single-node data-parallel training step around the path (reference
/root/reference/trainer.py:113-127 compute_loss = t2i loss + mmu loss, two forwards / one backward;
train_stage2.py:16-44: bf16 autocast, AdamW(0.9, 0.95, wd 0), grad clip 1.0, cosine-with-min-lr).

One process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on ROCm; "gloo" in the CPU tests).  DDP with
gradient_as_bucket_view and LARGE buckets: a ring on xGMI is per-link bound (7 links x ~153 GB/s), so RCCL needs big
messages to spread over all links; bucket all-reduce overlaps the (bandwidth-bound) backward.  No NCCL_ALGO=Tree
(train_stage2.py:48 -- an inter-node setting).  No per-step .item(): losses are logged from a detached device scalar.
"""
from __future__ import annotations

import collections
import math
import os
from dataclasses import dataclass

import torch


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
    grad_compression: str = "none"     # "bf16": the bucket all-reduce runs on bf16 copies of the fp32 gradient buckets (half the xGMI bytes)


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
    net = DDP(model, device_ids=device_ids, gradient_as_bucket_view=True, bucket_cap_mb=cfg.bucket_cap_mb,
              broadcast_buffers=False)
    if cfg.grad_compression == "bf16":
        # SURVEY.md section 2.3 C1 "optional bf16 compression hook": stage 2 all-reduces 5.9 GB of fp32 gradients per step
        # (/root/reference/train_stage2.py:37-38 trains every parameter); the hook casts each bucket to bf16, all-reduces, and casts
        # back into the fp32 bucket view -- half the bytes on every xGMI link, one bf16 rounding per gradient element and rank sum
        from torch.distributed.algorithms.ddp_comm_hooks import default_hooks
        net.register_comm_hook(state=None, hook=default_hooks.bf16_compress_hook)
    elif cfg.grad_compression != "none":
        raise ValueError(f"grad_compression: 'none' or 'bf16', not {cfg.grad_compression!r}")
    return net


class Stage2Step:
    """loss = model(inputs, 't2i') + model(inputs, 'mmu') from two forwards, one backward, clip, AdamW, cosine LR
    (trainer.py:113-127; HF Trainer defaults of train_stage2.py:16-44).  ``model`` is omnimamba_amd.omni.OmniMambaPath
    (forward(inputs, task) -> loss, like the reference's OmniMamba.forward); a batch holding only one of the two flows
    runs the stage-1 single-task step."""

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
        self.time_backward = False   # bench.py: HIP events around the backward (the window the gradient all-reduce has to hide in)
        self.backward_events = collections.deque(maxlen=256)   # (start, end) HIP event pairs of the last backwards

    def take_backward_ms(self):
        """Durations (ms) of the backwards timed since the last call (time_backward = True); the event pairs are dropped."""
        out = [a.elapsed_time(b) for a, b in self.backward_events]
        self.backward_events.clear()
        return out

    def __call__(self, batch):
        dev_type = "cuda" if next(self.model.parameters()).is_cuda else "cpu"
        self.opt.zero_grad(set_to_none=True)
        total = 0.0
        with torch.autocast(dev_type, dtype=self.cfg.amp_dtype, enabled=self.cfg.amp_dtype != torch.float32):
            for task in ("t2i", "mmu"):
                if f"{task}_flow" not in batch:
                    continue
                loss = self.net(batch, task)
                self.last[task] = loss.detach()      # logged from the device scalar: no per-step .item() (trainer.py:122-125)
                total = total + loss
        if self.time_backward and dev_type == "cuda":
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            total.backward()
            e1.record()
            self.backward_events.append((e0, e1))
        else:
            total.backward()
        if self.cfg.clip:
            torch.nn.utils.clip_grad_norm_([p for p in self.model.parameters() if p.grad is not None], self.cfg.clip)
        self.opt.step()
        self.sched.step()
        return total.detach()


def synthetic_batch(cfg, batch, seqlen, device, dtype, rank=0, tasks=("t2i", "mmu"), step=0, caption_len=73):
    """Synthetic inputs of the reference shapes, total sequence length ``seqlen`` per task:
      t2i_flow: caption ids (B, caption_len) ~ U[0, vocab), image ids (B, seqlen - caption_len) ~ U[0, vq vocab)
                (reference: 73 + 256 = 329, omnimamba.py:264);
      mmu_flow: images_feat ~ N(0, 1) (B, img_sq_len, fused_vision_dim) standing for the frozen vision towers' output,
                text ids (B, seqlen - img_sq_len - 4) ~ U[0, vocab), labels = ids (reference: 4 + 729 + 449, :190-218).
    Seeds = 1234 + rank (DistributedSampler-style per-rank shards, trainer.py:50-56,79-85)."""
    g = torch.Generator(device="cpu").manual_seed(1234 + rank + 1000003 * step)
    out = {}
    if "t2i" in tasks:
        n_img = seqlen - caption_len
        assert n_img > 0
        out["t2i_flow"] = {"caption_ids": torch.randint(0, cfg.vocab_size, (batch, caption_len), generator=g).to(device),
                           "inputs": torch.randint(0, cfg.vqvae_vocab_size, (batch, n_img), generator=g).to(device)}
    if "mmu" in tasks:
        n_txt = seqlen - cfg.img_sq_len - 4
        assert n_txt > 0
        ids = torch.randint(0, cfg.vocab_size, (batch, n_txt), generator=g).to(device)
        out["mmu_flow"] = {"images_feat": torch.randn(batch, cfg.img_sq_len, cfg.fused_vision_dim, generator=g).to(device=device, dtype=dtype),
                           "input_ids": ids, "labels": ids.clone()}
    return out


def shard_batch(batch, rank, world):
    """Rows [rank * B / world, (rank + 1) * B / world) of every tensor in a synthetic batch."""
    def cut(t):
        n = t.shape[0] // world
        return t[rank * n:(rank + 1) * n]
    return {flow: {k: (cut(v) if torch.is_tensor(v) else v) for k, v in d.items()} for flow, d in batch.items()}

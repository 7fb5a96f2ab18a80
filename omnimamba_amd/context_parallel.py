"""Context-parallel Mamba-2: ONE sequence cut along L over the ranks of a process group (SURVEY.md section 8 rows e / f2).

The reference has no sequence parallelism (its length scaling comes from the O(L) recurrence alone);
the regime it advertises (assets/teaser.png (c): 4 K - 128 K tokens) is where a single sequence stops fitting one GPU's
activation memory.  The SSD recurrence needs exactly ONE exchange per layer for that, the conv1d in front of it a 3-token halo:

  rank r holds tokens [r L/W, (r + 1) L/W) of every sequence
  0. conv1d halo: the last d_conv - 1 pre-conv xBC tokens of rank r - 1 enter rank r's conv as `initial_states`
     (3 x 4352 values per sequence at the 1.3B shape) -- all-gather of the tails, every rank picks its left neighbour's
  1. state-only pass over the shard from a ZERO start (omk_ssd_scan_fwd with `out` absent: about a third of a scan):
     S_loc = end state, fp32 (B, H, P, N); the shard's total log-decay is A * sum dt'
  2. all-gather of (S_loc, log-decay) -- 2.1 MB per sequence and layer at the 1.3B shape, over xGMI -- and the exclusive scan over
     the ranks   s_r = sum_{q < r} (prod_{q < j < r} a_j) S_q   (a_j = shard decay per head), initial_states in front of shard 0
  3. the scan proper of the shard FROM s_r (`initial_states` of the same MFMA kernel): y of the shard with ONE output rounding,
     exactly what the unsharded scan computes -- no correction pass, no (B, L, H, P) fp32 temporary -- and
     S_final = a_r s_r + S_loc on the last rank.

Everything is differentiable: the exchange is ONE autograd node (`_CpExchange`) whose backward all-reduces the gradients of the
gathered states -- it sits in every rank's graph (the scan proper always takes `initial_states = s_r`, zeros on rank 0), so no
rank can skip a collective its peers issue.  One process per GPU, backend "nccl" (= RCCL); the CPU test runs two gloo ranks on the
emulated kernels, the -m gpu test walks two shards on one MI355X in bf16 through the MFMA kernels and compares with the oracle on
the whole sequence.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .ssd_combined import mamba_chunk_scan_combined, ssd_final_state


def _dt_eff(dt, dt_bias, dt_softplus, dt_limit):
    v = dt.float()
    if dt_bias is not None:
        v = v + dt_bias.float()
    if dt_softplus:
        v = F.softplus(v)
    if dt_limit != (0.0, float("inf")):
        v = v.clamp(dt_limit[0], dt_limit[1])
    return v


def start_states_from_shards(S_all, logdec_all, rank, initial_states=None):
    """Exclusive scan over the ranks: S_all (W, B, H, P, N) end states from zero starts, logdec_all (W, B, H) natural-log
    decay of every shard -> state at the START of shard `rank` (initial_states enters in front of shard 0)."""
    s = torch.zeros_like(S_all[0]) if initial_states is None else initial_states.float()
    for q in range(rank):
        s = torch.exp(logdec_all[q])[..., None, None] * s + S_all[q]
    return s


class _LocalGroup:
    """The exchange of a 'group' whose shards are walked one after the other in ONE process (tests, single-GPU long sequences):
    all_gather = the list the caller filled."""

    def __init__(self, world):
        self.world, self.rank = world, 0


class _CpExchange(torch.autograd.Function):
    """(S_loc, logdec_loc, initial_states) of this rank -> (state at the start of this rank's shard, state behind the whole
    sequence).  Forward: two all-gathers.  Backward: the local gradients of the gathered tensors (autograd of
    start_states_from_shards on this rank's two outputs) summed over the ranks -- one all-reduce each, issued by EVERY rank."""

    @staticmethod
    def forward(ctx, S_loc, ld_loc, init, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        Ss = [torch.empty_like(S_loc) for _ in range(world)]
        ls = [torch.empty_like(ld_loc) for _ in range(world)]
        dist.all_gather(Ss, S_loc.contiguous(), group=group)
        dist.all_gather(ls, ld_loc.contiguous(), group=group)
        S_all, ld_all = torch.stack(Ss, 0), torch.stack(ls, 0)
        ctx.save_for_backward(S_all, ld_all, init)
        ctx.group, ctx.rank, ctx.world = group, rank, world
        return start_states_from_shards(S_all, ld_all, rank, init), start_states_from_shards(S_all, ld_all, world, init)

    @staticmethod
    def backward(ctx, ds_in, dfinal):
        S_all, ld_all, init = ctx.saved_tensors
        with torch.enable_grad():
            Sa, la = S_all.detach().requires_grad_(), ld_all.detach().requires_grad_()
            ii = None if init is None else init.detach().float().requires_grad_()
            outs = (start_states_from_shards(Sa, la, ctx.rank, ii), start_states_from_shards(Sa, la, ctx.world, ii))
            leaves = [Sa, la] + ([ii] if ii is not None else [])
            pairs = [(o, g.float()) for o, g in zip(outs, (ds_in, dfinal)) if o.requires_grad]   # rank 0 without initial_states: s_in is a constant
            grads = (torch.autograd.grad([o for o, _ in pairs], leaves, [g for _, g in pairs], allow_unused=True) if pairs
                     else [None] * len(leaves))
        gS = torch.zeros_like(S_all) if grads[0] is None else grads[0]
        gl = torch.zeros_like(ld_all) if grads[1] is None else grads[1]
        dist.all_reduce(gS, op=dist.ReduceOp.SUM, group=ctx.group)
        dist.all_reduce(gl, op=dist.ReduceOp.SUM, group=ctx.group)
        # initial_states is a replicated input: every rank returns ITS contribution (the caller sums parameter-like gradients over
        # the ranks, as for A / D / dt_bias)
        gi = None if ii is None else (torch.zeros_like(ii) if grads[2] is None else grads[2]).to(init.dtype)
        return gS[ctx.rank], gl[ctx.rank], gi, None


class _HaloExchange(torch.autograd.Function):
    """tail (B, C, W - 1) of every rank's pre-conv input -> the left neighbour's tail (zeros on rank 0).  Backward: the gradient of
    rank r + 1's halo returns to rank r's tail.  All-gather both ways, so every rank issues the same collectives."""

    @staticmethod
    def forward(ctx, tail, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        ts = [torch.empty_like(tail) for _ in range(world)]
        dist.all_gather(ts, tail.contiguous(), group=group)
        ctx.group, ctx.rank, ctx.world = group, rank, world
        return ts[rank - 1] if rank > 0 else torch.zeros_like(tail)

    @staticmethod
    def backward(ctx, g):
        gs = [torch.empty_like(g) for _ in range(ctx.world)]
        dist.all_gather(gs, g.contiguous(), group=ctx.group)
        return (gs[ctx.rank + 1] if ctx.rank + 1 < ctx.world else torch.zeros_like(g)), None


def shard_states(x, dt, A, B, dt_bias=None, dt_softplus=False, dt_limit=(0.0, float("inf"))):
    """Step 1 for one shard: (end state from a zero start (B, H, P, N) fp32, natural-log decay of the whole shard (B, H))."""
    S_loc = ssd_final_state(x, dt, A, B, dt_bias=dt_bias, dt_softplus=dt_softplus, dt_limit=dt_limit)
    ld = (_dt_eff(dt, dt_bias, dt_softplus, dt_limit) * A.float()).sum(1)
    return S_loc, ld


def mamba_chunk_scan_context_parallel(x, dt, A, B, C, chunk_size, D=None, z=None, dt_bias=None, initial_states=None,
                                      dt_softplus=False, dt_limit=(0.0, float("inf")), return_final_states=False, group=None):
    """mamba_chunk_scan_combined for a sequence sharded along L over `group` (this rank's shard: x (B, L/W, H, P), ...).
    initial_states: state in front of the WHOLE sequence (every rank passes the same tensor or None).
    Returns this shard's output [, the state after the WHOLE sequence (identical on every rank)]."""
    if group is None:
        group = dist.group.WORLD
    S_loc, ld = shard_states(x, dt, A, B, dt_bias, dt_softplus, dt_limit)
    s_in, final = _CpExchange.apply(S_loc, ld, initial_states, group)
    y = mamba_chunk_scan_combined(x, dt, A, B, C, chunk_size, D=D, z=z, dt_bias=dt_bias, initial_states=s_in, dt_softplus=dt_softplus,
                                  dt_limit=dt_limit)
    return (y, final) if return_final_states else y


def conv1d_halo(xBC_t, width, group=None):
    """xBC_t: this shard's pre-conv input (B, C, L_loc) -> the `initial_states` (B, C, width - 1) of its causal conv1d: the last
    width - 1 tokens of the left neighbour's shard (zeros on rank 0)."""
    if group is None:
        group = dist.group.WORLD
    n = width - 1
    tail = xBC_t[..., -n:] if xBC_t.shape[-1] >= n else F.pad(xBC_t, (n - xBC_t.shape[-1], 0))
    return _HaloExchange.apply(tail.contiguous(), group)

"""Context-parallel SSD scan: ONE sequence cut along L over the ranks of a process group (SURVEY.md section 8 rows e / f2).

The reference has no sequence parallelism (its length scaling comes from the O(L) recurrence alone);
the regime it advertises (assets/teaser.png (c): 4 K - 128 K tokens) is where a single sequence stops fitting one GPU's
activation memory.  The SSD recurrence needs exactly ONE exchange for that:

  rank r holds tokens [r L/W, (r + 1) L/W) of every sequence
  1. local scan from a ZERO start (omk_ssd_scan_fwd, the same MFMA kernel): y_loc, S_loc = end state, fp32 (B, H, P, N)
  2. all-gather of (S_loc, log-decay of the whole shard) -- 2.1 MB per sequence and layer at the 1.3B shape, over xGMI --
     and an exclusive scan over the ranks:  s_r = sum_{q < r} (prod_{q < j < r} a_j) S_q      (a_j = shard decay per head)
  3. correction  y_t += C_t . (exp(cs_t) s_r)   with cs_t the inclusive log-decay prefix inside the shard: a (tokens x N) x
     (N x P) GEMM per (batch, head) on the library (hipBLASLt), and  S_final = a_r s_r + S_loc.

No second pass over the shard, no collective inside the scan kernel.  Everything around the scan is differentiable
torch code and the all-gather has a backward (sum of the gradient slices), so the same function trains: gradients of the
boundary states flow back to the ranks that produced them.  One process per GPU, backend "nccl" (= RCCL); the CPU test
runs two gloo ranks on the emulated kernels and compares with the single-process scan of the whole sequence.
"""
from __future__ import annotations

import torch
import torch.distributed as dist
import torch.nn.functional as F

from .ssd_combined import mamba_chunk_scan_combined


class _AllGatherCat(torch.autograd.Function):
    """all_gather along a new leading dim with a backward that works on every backend (all_reduce of the stacked gradient)."""

    @staticmethod
    def forward(ctx, t, group):
        ctx.group = group
        world = dist.get_world_size(group)
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t.contiguous(), group=group)
        return torch.stack(out, 0)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=ctx.group)
        return g[dist.get_rank(ctx.group)], None


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


def mamba_chunk_scan_context_parallel(x, dt, A, B, C, chunk_size, D=None, z=None, dt_bias=None, initial_states=None,
                                      dt_softplus=False, dt_limit=(0.0, float("inf")), return_final_states=False, group=None):
    """mamba_chunk_scan_combined for a sequence sharded along L over `group` (this rank's shard: x (B, L/W, H, P), ...).
    initial_states: state in front of the WHOLE sequence (every rank passes the same tensor or None).
    Returns this shard's output [, the state after the WHOLE sequence (identical on every rank)]."""
    if group is None:
        group = dist.group.WORLD
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H, G = x.shape[2], B.shape[2]
    y_loc, S_loc = mamba_chunk_scan_combined(x, dt, A, B, C, chunk_size, D=D, z=None, dt_bias=dt_bias, dt_softplus=dt_softplus,
                                             dt_limit=dt_limit, return_final_states=True)
    cs = torch.cumsum(_dt_eff(dt, dt_bias, dt_softplus, dt_limit) * A.float(), dim=1)        # (B, L_loc, H) natural log
    S_all = _AllGatherCat.apply(S_loc.float(), group)
    ld_all = _AllGatherCat.apply(cs[:, -1], group)
    s_in = start_states_from_shards(S_all, ld_all, rank, initial_states)
    # correction: y[b, t, h, p] += exp(cs[b, t, h]) * sum_n C[b, t, g(h), n] s_in[b, h, p, n]
    # one GEMM per (batch, group) over ALL heads of the group -- (L_loc x N) x (N x (H / G) P) -- with the decay applied to the
    # product (C is shared by the heads of a group: expanding it per head first would be a (B, L, H, N) fp32 temporary, 1 GB at the
    # 1.3B shape with L_loc = 4096)
    Pd = x.shape[3]
    sg = s_in.view(s_in.shape[0], G, H // G, Pd, s_in.shape[-1])                              # (B, G, H / G, P, N)
    corr = torch.einsum("blgn,bgkpn->blgkp", C.float(), sg).reshape(x.shape[0], x.shape[1], H, Pd)
    y = torch.addcmul(y_loc.float(), corr, torch.exp(cs)[..., None])
    if z is not None:
        y = y * F.silu(z.float())
    y = y.to(x.dtype)
    if not return_final_states:
        return y
    final = start_states_from_shards(S_all, ld_all, world, initial_states)                    # state behind the last shard
    return y, final

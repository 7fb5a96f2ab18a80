"""Base projection + task LoRA as ONE library GEMM: the rank joins the contraction dimension.

    y = x W^T + s h B^T = [x | s h | 0] [W | B | 0]^T            h = A(dropout(x)), s = lora_alpha / r

The reference adds the LoRA branch to the base output afterwards (models/stage2/lora.py:263-279): at training token counts
that is a second full read + write of the (tokens, 8512) in_proj output -- 157 us per layer and task on the MI355X even as one
streaming kernel (omk_lora_add), 4 % of the 1.3B training step.  With the contraction dimension padded from 2048 to 2048 + 64
the library GEMM takes the same time (`tools/probe_gemm_kext.py`: 468 us both; K = 2048 + 8 alone is 9 % slower, and an
extended INPUT gradient (N = 2056 / 2112) is 50 % slower -- so only the forward uses the extension) and the add disappears.

The extended weight [W | B | 0] lives in a per-module bf16 buffer: the W part is refreshed when the master weight changes
(never, while the base weight is frozen -- which also removes the per-call fp32 -> bf16 cast of the 70 MB master), the B part on
every call (out_features x r elements).  Backward: dx = dy W (the plain K = 2048 GEMM on a strided view of the buffer),
dh = s dy B and dB = s dy^T h as skinny GEMMs, dW -- when the base weight trains -- by `linear._WGradFn` behind this node.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import lora_add as LA
from .linear import weight_grad

PAD = 64          # columns appended to the contraction dimension (rank <= PAD); a multiple of the GEMM's K tile
MIN_TOKENS = 512  # below this the add is not worth a second copy of x


def applies(x2d: torch.Tensor, weight: torch.Tensor, rank: int, adt: torch.dtype) -> bool:
    return (adt in (torch.bfloat16, torch.float16) and x2d.dim() == 2 and x2d.shape[0] >= MIN_TOKENS
            and rank <= PAD and weight.shape[1] % 8 == 0)


def extended_weight(module, weight: torch.Tensor, lora_b: torch.Tensor, adt: torch.dtype) -> torch.Tensor:
    """[W | B | 0] as (out, in + PAD) in `adt`, cached on `module` (not a registered buffer: it never enters a state_dict)."""
    out_f, in_f = weight.shape
    try:
        key = (weight.data_ptr(), weight._version, weight.device, adt)
    except RuntimeError:   # inference tensors have no version counter: rebuild every call
        key = None
    buf = getattr(module, "_omk_we", None)
    with torch.no_grad():
        if buf is None or buf.shape != (out_f, in_f + PAD) or buf.device != weight.device or buf.dtype != adt:
            buf = torch.zeros(out_f, in_f + PAD, dtype=adt, device=weight.device)
            module._omk_we, module._omk_we_key = buf, None
        if key is None or module._omk_we_key != key or os.environ.get("OMK_CAST_CACHE", "1") == "0":
            buf[:, :in_f].copy_(weight)
            module._omk_we_key = key
        buf[:, in_f:in_f + lora_b.shape[1]].copy_(lora_b)
    return buf


class _LoraExtFn(torch.autograd.Function):
    """The whole task-LoRA projection as one node: dropout -> A -> extended GEMM; backward with the A branch folded into dx."""

    @staticmethod
    def forward(ctx, x2d, we, lora_a, lora_b, scale, in_f, p_drop):
        adt = we.dtype
        r = lora_a.shape[0]
        mask = None
        xd = x2d
        if p_drop > 0.0:
            xd, mask = torch.ops.aten.native_dropout(x2d, p_drop, True)
        h2d = xd @ lora_a.to(adt).t()
        xe = torch.empty(x2d.shape[0], in_f + PAD, dtype=adt, device=x2d.device)
        xe[:, :in_f].copy_(x2d)
        torch.mul(h2d, scale, out=xe[:, in_f:in_f + r])
        xe[:, in_f + r:].zero_()
        ctx.save_for_backward(xd, h2d, mask, lora_a, lora_b)
        # the buffer is NOT a saved tensor: its B columns are rewritten by the next call (the other task's forward runs before
        # this call's backward), which is harmless -- backward only reads the W columns -- but would trip the version check
        ctx.we = we
        ctx.scale, ctx.in_f, ctx.p_drop = float(scale), in_f, float(p_drop)
        return F.linear(xe, we)

    @staticmethod
    def backward(ctx, dy):
        xd, h2d, mask, lora_a, lora_b = ctx.saved_tensors
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        adt = dyc.dtype
        dx = da = db = None
        if os.environ.get("OMK_LORA_UP_FUSED", "1") != "0" and LA.up_bwd_applies(dyc, h2d, lora_b):
            # dh = dy B and dB = dy^T h from ONE pass over dy (omk_lora_up_bwd): as two library GEMMs each streams the 279 MB
            dh32, db32 = LA.lora_up_bwd(dyc, h2d, lora_b)
            dh = (dh32 * ctx.scale).to(adt)                           # (tokens, r)
            if ctx.needs_input_grad[3]:
                db = (db32 * ctx.scale).to(lora_b.dtype)
        else:
            dh = (dyc @ lora_b.to(adt)) * ctx.scale
            if ctx.needs_input_grad[3]:
                db = weight_grad(dyc, h2d.contiguous(), torch.float32).mul_(ctx.scale).to(lora_b.dtype)
        if ctx.needs_input_grad[2]:
            da = weight_grad(dh.contiguous(), xd if xd.is_contiguous() else xd.contiguous(), torch.float32).to(lora_a.dtype)
        if ctx.needs_input_grad[0]:
            dx = dyc @ ctx.we[:, :ctx.in_f]                           # the plain input gradient: rows of the buffer, stride in + PAD
            # + dropout'(dh A): one streaming pass over dx (omk_lora_add with the dropout mask) instead of a K = 8 GEMM into a
            # second (tokens, in) tensor, the mask-and-scale pass and the add (126 -> 41 us per layer and task at 16 k tokens)
            keep = 1.0 / (1.0 - ctx.p_drop) if mask is not None else 1.0
            a_t = lora_a.to(adt).t().contiguous()                     # (in, r): the kernel's "B" operand
            if LA.applies(dx, dh, a_t):
                LA.lora_add_(dx, dh.contiguous(), a_t, keep, mask)
            else:
                upd = dh @ lora_a.to(adt)
                dx += upd * (mask.to(adt) * keep) if mask is not None else upd
        return dx, None, da, db, None, None, None


def lora_ext_linear(module, x2d, weight, lora_a, lora_b, scale, adt, p_drop):
    """x2d (tokens, in) -> (tokens, out) in `adt` = x W^T + scale * dropout_p(x) A^T B^T; gradients to x2d, lora_a, lora_b (the
    base weight's gradient, if it trains, belongs to the caller: `linear._WGradFn`)."""
    we = extended_weight(module, weight, lora_b, adt)
    return _LoraExtFn.apply(x2d.to(adt), we, lora_a, lora_b, scale, weight.shape[1], p_drop)

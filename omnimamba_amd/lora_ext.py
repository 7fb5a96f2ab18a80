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

import torch
import torch.nn.functional as F

from .linear import weight_grad

PAD = 64          # columns appended to the contraction dimension (rank <= PAD); a multiple of the GEMM's K tile
MIN_TOKENS = 512  # below this the add is not worth a second copy of x


def applies(x2d: torch.Tensor, weight: torch.Tensor, rank: int, adt: torch.dtype) -> bool:
    return (adt in (torch.bfloat16, torch.float16) and x2d.dim() == 2 and x2d.shape[0] >= MIN_TOKENS
            and rank <= PAD and weight.shape[1] % 8 == 0)


def extended_weight(module, weight: torch.Tensor, lora_b: torch.Tensor, adt: torch.dtype) -> torch.Tensor:
    """[W | B | 0] as (out, in + PAD) in `adt`, cached on `module` (not a registered buffer: it never enters a state_dict)."""
    out_f, in_f = weight.shape
    key = (weight.data_ptr(), weight._version, weight.device, adt)
    buf = getattr(module, "_omk_we", None)
    with torch.no_grad():
        if buf is None or buf.shape != (out_f, in_f + PAD) or buf.device != weight.device or buf.dtype != adt:
            buf = torch.zeros(out_f, in_f + PAD, dtype=adt, device=weight.device)
            module._omk_we, module._omk_we_key = buf, None
        if module._omk_we_key != key:
            buf[:, :in_f].copy_(weight)
            module._omk_we_key = key
        buf[:, in_f:in_f + lora_b.shape[1]].copy_(lora_b)
    return buf


class _LoraExtFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, h2d, we, lora_b, scale, in_f):
        r = h2d.shape[1]
        xe = torch.empty(x2d.shape[0], in_f + PAD, dtype=we.dtype, device=x2d.device)
        xe[:, :in_f].copy_(x2d)
        torch.mul(h2d, scale, out=xe[:, in_f:in_f + r])
        xe[:, in_f + r:].zero_()
        ctx.save_for_backward(h2d, lora_b)
        # the buffer is NOT a saved tensor: its B columns are rewritten by the next call (the other task's forward runs before
        # this call's backward), which is harmless -- backward only reads the W columns -- but would trip the version check
        ctx.we = we
        ctx.scale, ctx.in_f = float(scale), in_f
        return F.linear(xe, we)

    @staticmethod
    def backward(ctx, dy):
        h2d, lora_b = ctx.saved_tensors
        dx = dh = db = None
        if ctx.needs_input_grad[0]:
            dx = dy @ ctx.we[:, :ctx.in_f]                  # the plain input gradient: rows of the buffer, stride in + PAD
        if ctx.needs_input_grad[1]:
            dh = (dy @ lora_b.to(dy.dtype)) * ctx.scale
        if ctx.needs_input_grad[3]:
            dyc = dy if dy.is_contiguous() else dy.contiguous()
            db = weight_grad(dyc, h2d.to(dy.dtype).contiguous(), torch.float32).mul_(ctx.scale).to(lora_b.dtype)
        return dx, dh, None, db, None, None


def lora_ext_linear(module, x2d, h2d, weight, lora_b, scale, adt):
    """x2d (tokens, in), h2d (tokens, r) -> (tokens, out) in `adt`; gradients to x2d, h2d and lora_b (the base weight's gradient, if
    it trains, belongs to the caller: `linear._WGradFn`)."""
    we = extended_weight(module, weight, lora_b, adt)
    return _LoraExtFn.apply(x2d.to(adt), h2d.to(adt), we, lora_b, scale, weight.shape[1])

"""Chunked fused linear + cross-entropy for the two heads.

The reference materialises the logits of the whole batch -- `lm_head(hidden)` at models/stage2/mixer_seq_simple.py:519-521,
then `CrossEntropyLoss` over the shifted, flattened logits (models/mamba_vlm.py:88-102, models/omnimamba.py:276-279,305-306).
At vocab 50 288 x L = 8192 x batch 8 that is a 13 GB fp32 tensor plus its gradient: the largest HBM consumer of BASELINE
configs 4 / 5.  Here the tokens are walked in blocks: a block's logits come out of the library GEMM (hipBLASLt) in the
activation dtype, `omk_cross_entropy` (csrc/ce.hip) reads them once for the loss and writes the gradient
(softmax - onehot) / n_valid over them, and the two gradient GEMMs (dW += dlogits^T h, dh = dlogits W) run on that block
straight away.  Only one block of logits ever exists; backward just scales the stored gradients by the upstream scalar.

Same arithmetic as the reference under bf16 autocast: logits rounded to the activation dtype by the GEMM, softmax
statistics in fp32, mean over the labels that are not -100.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn.functional as F

from . import _capi as K
from ._lib import get_lib, require_device
from .linear import weight_grad

IGNORE_ID = -100


def _block_tokens(vocab: int, elem: int) -> int:
    """Tokens per block: about 512 MB of logits (OMK_CE_BLOCK_MB overrides), a multiple of 256."""
    mb = int(os.environ.get("OMK_CE_BLOCK_MB", "512"))
    t = max(256, (mb << 20) // (vocab * elem))
    return (t // 256) * 256


def cross_entropy_inplace(logits2d: torch.Tensor, labels1d: torch.Tensor, grad_scale: torch.Tensor | None, write_grad=True):
    """losses (T) f32; logits2d is overwritten with its gradient times grad_scale[0] when write_grad."""
    lib = get_lib()
    require_device(lib, logits2d)
    losses = torch.empty(logits2d.shape[0], dtype=torch.float32, device=logits2d.device)
    p = K.CrossEntropy(logits=K.T(logits2d), labels=labels1d.data_ptr(), losses=K.T(losses),
                       grad_scale=None if grad_scale is None else grad_scale.data_ptr(), ignore_index=IGNORE_ID, write_grad=int(write_grad))
    K.run(lib, "omk_cross_entropy", p, logits2d)
    return losses


class _FusedLinearCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden2d, weight, labels1d):
        dev = hidden2d.device
        adt = hidden2d.dtype
        if hidden2d.is_cuda and torch.is_autocast_enabled():
            adt = torch.get_autocast_dtype("cuda")
        h = hidden2d.to(adt)
        h = h if h.is_contiguous() else h.contiguous()
        w = weight.to(adt)
        if labels1d.dtype != torch.int64:
            raise TypeError(f"fused_linear_cross_entropy: labels must be int64 (the kernel reads 8-byte ids), got {labels1d.dtype}")
        labels1d = labels1d.contiguous()
        T, V = h.shape[0], w.shape[0]
        # the mean runs over exactly the rows the kernel scores: ids outside [0, V) (which torch's cross_entropy rejects with a device
        # assert) get neither loss nor gradient there, so they do not enter the denominator either.  No counted row at all: 0, where
        # torch returns NaN -- a batch of nothing but ignored labels does not poison an accumulated training loss.
        counted = ((labels1d >= 0) & (labels1d < w.shape[0])).sum().to(torch.float32)      # (IGNORE_ID = -100 lies outside)
        inv = (1.0 / counted.clamp_min(1.0)).reshape(1)              # device scalar: no host sync
        need_dh, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dh = torch.empty_like(h) if need_dh else None
        dw = torch.zeros(weight.shape, dtype=torch.float32, device=dev) if need_dw else None
        total = torch.zeros((), dtype=torch.float32, device=dev)
        step = _block_tokens(V, h.element_size())
        for t0 in range(0, T, step):
            hb = h[t0:t0 + step]
            logits = F.linear(hb, w)                                   # (block, V), activation dtype: the only logits alive
            losses = cross_entropy_inplace(logits, labels1d[t0:t0 + step], inv, write_grad=need_dh or need_dw)
            total += losses.sum()
            if need_dh:
                torch.mm(logits, w, out=dh[t0:t0 + step])
            if need_dw:
                dw += weight_grad(logits, hb, torch.float32)
        ctx.save_for_backward(dh, dw)
        ctx.in_dtype, ctx.w_dtype = hidden2d.dtype, weight.dtype
        return total * inv[0]

    @staticmethod
    def backward(ctx, g):
        dh, dw = ctx.saved_tensors
        return (None if dh is None else (dh * g.to(dh.dtype)).to(ctx.in_dtype),
                None if dw is None else (dw * g).to(ctx.w_dtype), None)


def fused_linear_cross_entropy(hidden2d, weight, labels1d):
    """mean over labels != -100 of CE(hidden2d @ weight^T, labels1d) without materialising the (tokens, vocab) logits.
    Edge behaviour that differs from torch.nn.CrossEntropyLoss: when EVERY label is -100 the result is 0 (torch: NaN); labels
    outside [0, V) other than -100 contribute zero loss and gradient and are NOT counted in the mean (torch raises) -- callers
    validate their label range (the reference only ever produces ids < V and -100, omnimamba.py:190-218)."""
    return _FusedLinearCE.apply(hidden2d, weight, labels1d)


def applies(hidden2d, weight) -> bool:
    try:
        lib = get_lib()
    except RuntimeError:
        return False
    on_lib_device = hidden2d.is_cuda != bool(lib.omk_is_emulated())
    elem = 2 if (hidden2d.is_cuda and torch.is_autocast_enabled()) else hidden2d.element_size()
    return on_lib_device and os.environ.get("OMK_FUSED_CE", "1") != "0" and (weight.shape[0] * elem) % 16 == 0

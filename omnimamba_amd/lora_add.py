"""result + scaling * h @ lora_B^T as one streaming pass over `result`, in place (omk_lora_add, csrc/lora_add.hip).

The reference adds its task LoRA as `result += lora_B(lora_A(dropout(x))) * scaling` (models/stage2/lora.py:263-279).  At
training token counts the rank-8 update is a K = 8 GEMM: a full read + write of the (tokens, out_features) result, and as
`torch.addmm` additionally a copy of it.  Here it is one kernel that reads the result once and writes it once.  Backward:
two skinny library GEMMs (dh = scaling * dy @ B, dB = scaling * dy^T @ h); the gradient of `result` passes through.
"""
from __future__ import annotations

import torch

from . import _capi as K
from ._lib import get_lib, require_device


def applies(result2d: torch.Tensor, h2d: torch.Tensor, lora_b: torch.Tensor) -> bool:
    vec = 4 if result2d.dtype == torch.float32 else 8
    try:
        on_lib_device = result2d.is_cuda != bool(get_lib().omk_is_emulated())   # HIP build: GPU tensors; emulator (tests): CPU
    except RuntimeError:
        return False
    return (on_lib_device and result2d.dim() == 2 and result2d.stride(1) == 1 and h2d.stride(1) == 1 and h2d.dtype == result2d.dtype
            and h2d.shape[1] in (8, 16) and result2d.shape[1] % vec == 0 and lora_b.stride(1) == 1
            and result2d.data_ptr() % 16 == 0 and h2d.data_ptr() % 16 == 0
            and (result2d.stride(0) * result2d.element_size()) % 16 == 0 and (h2d.stride(0) * h2d.element_size()) % 16 == 0)


class _LoraAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, result2d, h2d, lora_b, scale):
        lib = get_lib()
        require_device(lib, result2d, h2d, lora_b)
        # in place on `result2d`, declared to autograd (mark_dirty bumps the version counter): a producer or hook that
        # saved the base projection's output for its own backward now raises instead of silently reading the sum
        p = K.LoraAdd(out=K.T(result2d), h=K.T(h2d), lora_b=K.T(lora_b), scale=float(scale))
        K.run(lib, "omk_lora_add", p, result2d)
        ctx.mark_dirty(result2d)
        ctx.save_for_backward(h2d, lora_b)
        ctx.scale = float(scale)
        return result2d

    @staticmethod
    def backward(ctx, dy):
        h2d, lora_b = ctx.saved_tensors
        dh = db = None
        if ctx.needs_input_grad[1]:
            dh = (dy @ lora_b.to(dy.dtype)) * ctx.scale
        if ctx.needs_input_grad[2]:
            db = (dy.t() @ h2d.to(dy.dtype)).to(lora_b.dtype) * ctx.scale
        return dy, dh, db, None


def lora_add(result2d, h2d, lora_b, scale):
    """result2d (tokens, out) + scale * h2d (tokens, r) @ lora_b (out, r)^T; overwrites result2d's storage."""
    return _LoraAdd.apply(result2d, h2d, lora_b, scale)

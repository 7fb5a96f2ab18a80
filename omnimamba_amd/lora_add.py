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


# autograd nodes that may have produced the base projection's output WITHOUT saving that output for their own backward
# (omnimamba_amd.linear's two nodes, the library linear / matmul nodes, pure view nodes).  Anything else -- an activation
# checkpoint wrapper, a node with saved-tensor hooks on its result -- makes TaskLoRALinear take the out-of-place addmm.
_SAFE_PRODUCERS = {"_XGradFnBackward", "_WGradFnBackward", "MmBackward0", "AddmmBackward0", "LinearBackward0", "BmmBackward0",
                   "ViewBackward0", "UnsafeViewBackward0", "ReshapeAliasBackward0", "AliasBackward0", "ToCopyBackward0"}


def producer_is_safe(result: torch.Tensor) -> bool:
    """True when overwriting `result` cannot invalidate what its producer saved (see _SAFE_PRODUCERS)."""
    fn = result.grad_fn
    return fn is None or type(fn).__name__ in _SAFE_PRODUCERS


class _LoraAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, result2d, h2d, lora_b, scale):
        lib = get_lib()
        require_device(lib, result2d, h2d, lora_b)
        # in place on the storage of `result2d`: nothing keeps the base projection's output for backward (its gradient only
        # needs dy, x and W), so overwriting it is safe; autograd sees an ordinary out-of-place node returning an alias.
        # (ctx.mark_dirty cannot be used: the base projection's output is a view made inside a custom Function --
        # linear._WGradFn -- and autograd forbids in-place on those.)  `producer_is_safe` guards the assumption.
        out = result2d.detach()
        p = K.LoraAdd(out=K.T(out), h=K.T(h2d), lora_b=K.T(lora_b), scale=float(scale))
        K.run(lib, "omk_lora_add", p, out)
        ctx.save_for_backward(h2d, lora_b)
        ctx.scale = float(scale)
        return out

    @staticmethod
    def backward(ctx, dy):
        h2d, lora_b = ctx.saved_tensors
        dh = db = None
        if ctx.needs_input_grad[1]:
            dh = (dy @ lora_b.to(dy.dtype)) * ctx.scale
        if ctx.needs_input_grad[2]:
            db = (dy.t() @ h2d.to(dy.dtype)).to(lora_b.dtype) * ctx.scale
        return dy, dh, db, None


def lora_add_(out2d, h2d, lora_b, scale, mask=None):
    """The kernel itself, no autograd: out2d += scale * h2d @ lora_b^T in place; with `mask` (bool / uint8, out2d's shape) only
    where the mask is set.  `applies(out2d, h2d, lora_b)` must hold."""
    lib = get_lib()
    require_device(lib, out2d, h2d, lora_b, mask)
    p = K.LoraAdd(out=K.T(out2d), h=K.T(h2d), lora_b=K.T(lora_b), mask=K.T(mask), scale=float(scale))
    K.run(lib, "omk_lora_add", p, out2d)
    return out2d


def lora_add(result2d, h2d, lora_b, scale):
    """result2d (tokens, out) + scale * h2d (tokens, r) @ lora_b (out, r)^T; overwrites result2d's storage."""
    return _LoraAdd.apply(result2d, h2d, lora_b, scale)


def up_bwd_applies(dy2d: torch.Tensor, h2d: torch.Tensor, lora_b: torch.Tensor) -> bool:
    try:
        on_lib_device = dy2d.is_cuda != bool(get_lib().omk_is_emulated())
    except RuntimeError:
        return False
    return (on_lib_device and dy2d.dim() == 2 and dy2d.dtype == torch.bfloat16 and h2d.dtype == torch.bfloat16 and h2d.shape[1] == 8
            and dy2d.shape[1] % 8 == 0 and dy2d.stride(1) == 1 and h2d.stride(1) == 1 and lora_b.stride(1) == 1
            and dy2d.data_ptr() % 16 == 0 and h2d.data_ptr() % 16 == 0 and (dy2d.stride(0) * 2) % 16 == 0 and (h2d.stride(0) * 2) % 16 == 0)


def lora_up_bwd(dy2d, h2d, lora_b):
    """(dy2d @ lora_b, dy2d^T @ h2d) as fp32 (tokens, r) and (out, r) from ONE pass over dy2d (omk_lora_up_bwd leaves per-column-
    block / per-token-chunk partial sums, added up here: deterministic).  No autograd."""
    import ctypes as C
    lib = get_lib()
    require_device(lib, dy2d, h2d, lora_b)
    T, N, r = dy2d.shape[0], dy2d.shape[1], h2d.shape[1]
    parts = (C.c_int32 * 2)()
    K.check(lib, lib.omk_lora_up_bwd_parts(T, N, parts), "omk_lora_up_bwd_parts")
    nb, nc = int(parts[0]), int(parts[1])
    buf = torch.empty((nb * T + nc * N) * r, dtype=torch.float32, device=dy2d.device)
    dh_p, db_p = buf[:nb * T * r].view(nb, T, r), buf[nb * T * r:].view(nc, N, r)
    K.run(lib, "omk_lora_up_bwd", K.LoraUpBwd(dy=K.T(dy2d), lora_b=K.T(lora_b), h=K.T(h2d), dh=K.T(dh_p), dlora_b=K.T(db_p)), dy2d)
    return dh_p.sum(0), db_p.sum(0)

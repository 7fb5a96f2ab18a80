"""Gated RMSNorm on the MI355X (``Mamba2.norm(y, z)``).

Mirrors ``mamba_ssm.ops.triton.layernorm_gated.{rmsnorm_fn, RMSNorm}``; reached from the reference through
Mamba2.forward / Mamba2.step (/root/reference/models/stage2/block.py:117).  Kernels: omk_norm_gated_fwd / _bwd
(omnimamba_amd/csrc/norms.hip).  norm_before_gate=False (the reference's configuration): rmsnorm(x * silu(z)) * w.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _capi as K
from ._lib import get_lib, require_device


def _rows(t, cols):
    t2 = t.reshape(-1, cols)
    return t2 if t2.stride(-1) == 1 else t2.contiguous()


class NormGatedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, z=None, eps=1e-6, group_size=None, norm_before_gate=True):
        lib = get_lib()
        require_device(lib, x, weight, bias, z)
        shape = x.shape
        cols = shape[-1]
        x2 = _rows(x, cols)
        z2 = None if z is None else _rows(z, cols)
        if z2 is not None and z2.dtype != x2.dtype:
            z2 = z2.to(x2.dtype)
        y = torch.empty_like(x2)
        gs = cols if group_size is None else group_size
        if x2.shape[0] > 0:
            p = K.NormGatedFwd(x=K.T(x2), z=K.T(z2), weight=K.T(weight), bias=K.T(bias), y=K.T(y), rstd=K.T(None),
                               group_size=gs, eps=eps, norm_before_gate=int(norm_before_gate))
            K.run(lib, "omk_norm_gated_fwd", p, x2)
        ctx.save_for_backward(x2, z2, weight, bias)
        ctx.shape, ctx.eps, ctx.gs, ctx.nbg = shape, eps, gs, norm_before_gate
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy):
        lib = get_lib()
        x2, z2, weight, bias = ctx.saved_tensors
        cols = ctx.shape[-1]
        dy2 = _rows(dy, cols)
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        dx = torch.empty_like(x2)
        dz = None if z2 is None else torch.empty_like(z2)
        dw = torch.zeros(cols, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[1] else None   # frozen weight: no reduction
        if x2.shape[0] > 0:
            p = K.NormGatedBwd(dy=K.T(dy2), x=K.T(x2), z=K.T(z2), weight=K.T(weight), dx=K.T(dx), dz=K.T(dz),
                               dweight=K.T(dw), group_size=ctx.gs, eps=ctx.eps, norm_before_gate=int(ctx.nbg))
            ws = K.workspace(lib, "omk_norm_gated_bwd_workspace_bytes", p, dy2)  # noqa: F841
            K.run(lib, "omk_norm_gated_bwd", p, dy2)
        db = None
        if bias is not None and ctx.needs_input_grad[2]:
            g = dy2.float()
            if z2 is not None and ctx.nbg:
                g = g * torch.nn.functional.silu(z2.float())
            db = g.sum(0).to(bias.dtype)
        return (dx.reshape(ctx.shape), None if dw is None else dw.to(weight.dtype), db, None if dz is None else dz.reshape(ctx.shape),
                None, None, None)


def rmsnorm_fn(x, weight, bias, z=None, eps=1e-6, group_size=None, norm_before_gate=True):
    return NormGatedFn.apply(x, weight, bias, z, eps, group_size, norm_before_gate)


class RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-5, group_size=None, norm_before_gate=True, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.empty(hidden_size, **factory_kwargs))
        self.register_parameter("bias", None)
        self.group_size = group_size
        self.norm_before_gate = norm_before_gate
        nn.init.ones_(self.weight)

    def forward(self, x, z=None):
        """If z is not None: norm(x) * silu(z) if norm_before_gate, else norm(x * silu(z))."""
        return rmsnorm_fn(x, self.weight, self.bias, z=z, eps=self.eps, group_size=self.group_size,
                          norm_before_gate=self.norm_before_gate)

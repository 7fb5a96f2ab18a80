"""Fused residual-add + RMSNorm / LayerNorm on the MI355X.

Mirrors ``mamba_ssm.ops.triton.layer_norm`` as the reference uses it
(/root/reference/models/stage2/block.py:10,86-95 ; models/stage2/mixer_seq_simple.py:30,341-343,428-437):
``layer_norm_fn`` / ``rms_norm_fn`` / ``RMSNorm``.  The arithmetic is the HIP kernel pair
omk_add_norm_fwd / omk_add_norm_bwd (omnimamba_amd/csrc/norms.hip); there is no PyTorch fallback.
"""
from __future__ import annotations

import torch
from torch import nn

from . import _capi as K
from ._lib import get_lib, require_device


def _rows(t, cols):
    t2 = t.reshape(-1, cols)
    return t2 if t2.stride(-1) == 1 else t2.contiguous()


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                is_rms_norm=False):
        lib = get_lib()
        require_device(lib, x, weight, bias, residual)
        shape = x.shape
        cols = shape[-1]
        x2 = _rows(x, cols)
        res2 = None if residual is None else _rows(residual, cols)
        if res2 is not None and res2.shape != x2.shape:
            raise RuntimeError("layer_norm_fn: residual shape mismatch")
        res_dtype = residual.dtype if residual is not None else (torch.float32 if residual_in_fp32 else None)
        need_res_out = residual is not None or (res_dtype is not None and res_dtype != x.dtype)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        res_out = torch.empty(rows, cols, dtype=res_dtype, device=x.device) if need_res_out else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        mean = None if is_rms_norm else torch.empty(rows, dtype=torch.float32, device=x.device)
        if rows > 0:
            p = K.AddNormFwd(x=K.T(x2), residual=K.T(res2), weight=K.T(weight), bias=K.T(bias), y=K.T(y),
                             residual_out=K.T(res_out), rstd=K.T(rstd), mean=K.T(mean), eps=eps,
                             is_rms_norm=int(is_rms_norm))
            K.run(lib, "omk_add_norm_fwd", p, x2)
        ctx.save_for_backward(res_out if res_out is not None else x2, weight, bias, mean, rstd)
        ctx.shape, ctx.eps, ctx.is_rms_norm = shape, eps, is_rms_norm
        ctx.has_residual, ctx.prenorm, ctx.x_dtype = residual is not None, prenorm, x.dtype
        y = y.reshape(shape)
        if not prenorm:
            return y
        return y, (res_out.reshape(shape) if res_out is not None else x)

    @staticmethod
    def backward(ctx, dy, *args):
        lib = get_lib()
        xsum, weight, bias, mean, rstd = ctx.saved_tensors
        cols = ctx.shape[-1]
        dy2 = _rows(dy, cols)
        if dy2.dtype != ctx.x_dtype:
            dy2 = dy2.to(ctx.x_dtype)
        dres = None
        if ctx.prenorm and args[0] is not None:
            dres = _rows(args[0], cols)
            if dres.dtype != xsum.dtype:
                dres = dres.to(xsum.dtype)
        rows = dy2.shape[0]
        dx = torch.empty(rows, cols, dtype=ctx.x_dtype, device=dy.device)
        dres_in = None
        if ctx.has_residual and xsum.dtype != ctx.x_dtype:
            dres_in = torch.empty(rows, cols, dtype=xsum.dtype, device=dy.device)
        # frozen norm weights (every RMSNorm in the 'align' stage): the kernel then writes no partial sums and the reduction is not launched
        dw = torch.zeros(cols, dtype=torch.float32, device=dy.device) if ctx.needs_input_grad[1] else None
        db = None if (bias is None or not ctx.needs_input_grad[2]) else torch.zeros(cols, dtype=torch.float32, device=dy.device)
        if rows > 0:
            p = K.AddNormBwd(dy=K.T(dy2), dresidual_out=K.T(dres), xsum=K.T(xsum), weight=K.T(weight), rstd=K.T(rstd),
                             mean=K.T(mean), dx=K.T(dx), dresidual_in=K.T(dres_in), dweight=K.T(dw), dbias=K.T(db),
                             is_rms_norm=int(ctx.is_rms_norm), has_bias=int(bias is not None))
            ws = K.workspace(lib, "omk_add_norm_bwd_workspace_bytes", p, dy2)  # noqa: F841 (kept alive until launch)
            K.run(lib, "omk_add_norm_bwd", p, dy2)
        dx = dx.reshape(ctx.shape)
        if ctx.has_residual:
            dres_ret = dres_in.reshape(ctx.shape) if dres_in is not None else dx
        else:
            dres_ret = None
        return (dx, None if dw is None else dw.to(weight.dtype), None if db is None else db.to(bias.dtype), dres_ret, None, None, None, None)


def layer_norm_fn(x, weight, bias, residual=None, x1=None, weight1=None, bias1=None, eps=1e-6, dropout_p=0.0,
                  rowscale=None, prenorm=False, residual_in_fp32=False, is_rms_norm=False,
                  return_dropout_mask=False):
    """Same signature as upstream ``layer_norm_fn``; returns ``y`` or ``(y, residual_out)`` when ``prenorm``."""
    if x1 is not None or weight1 is not None or bias1 is not None or rowscale is not None or return_dropout_mask:
        raise NotImplementedError("layer_norm_fn: x1/weight1/bias1/rowscale/return_dropout_mask are not on the "
                                  "OmniMamba path (block.py:86-95 never passes them)")
    if dropout_p != 0.0:
        raise NotImplementedError("layer_norm_fn: dropout_p > 0 is not on the OmniMamba path")
    return LayerNormFn.apply(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)


def rms_norm_fn(x, weight, bias, residual=None, x1=None, weight1=None, bias1=None, eps=1e-6, dropout_p=0.0,
                rowscale=None, prenorm=False, residual_in_fp32=False, return_dropout_mask=False):
    return layer_norm_fn(x, weight, bias, residual, x1, weight1, bias1, eps, dropout_p, rowscale, prenorm,
                         residual_in_fp32, True, return_dropout_mask)


class RMSNorm(nn.Module):
    """Upstream ``mamba_ssm.ops.triton.layer_norm.RMSNorm``: ``.weight``, ``.bias = None``, ``.eps``
    (read at block.py:88-93)."""

    def __init__(self, hidden_size, eps=1e-5, dropout_p=0.0, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.eps = eps
        if dropout_p > 0.0:
            raise NotImplementedError("RMSNorm dropout is not on the OmniMamba path")
        self.drop = None
        self.weight = nn.Parameter(torch.empty(hidden_size, **factory_kwargs))
        self.register_parameter("bias", None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.ones_(self.weight)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps, prenorm=prenorm,
                           residual_in_fp32=residual_in_fp32)

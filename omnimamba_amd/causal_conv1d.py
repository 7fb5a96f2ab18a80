"""Causal depthwise conv1d (+bias, +SiLU) on the MI355X.

Mirrors ``causal_conv1d.causal_conv1d_fn`` / ``causal_conv1d_update`` (causal-conv1d==1.4.0, pinned at
/root/reference/requirements.txt:12) as Mamba2.forward / Mamba2.step use them
(/root/reference/models/stage2/mixer_seq_simple.py:17,200-205 -> block.py:117).  Kernels:
omk_causal_conv1d_{fwd,bwd,update} (omnimamba_amd/csrc/conv1d.hip).  No PyTorch fallback.
"""
from __future__ import annotations

import torch

from . import _capi as K
from ._lib import get_lib, require_device


def _act_flag(activation):
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu, or swish")
    return int(activation is not None)


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias=None, initial_states=None, return_final_states=False, final_states_out=None,
                activation=None):
        lib = get_lib()
        require_device(lib, x, weight, bias, initial_states, final_states_out)
        if x.dim() != 3:
            raise ValueError("causal_conv1d_fn: x must be (batch, dim, seqlen)")
        if x.stride(2) != 1 and x.stride(1) != 1:
            x = x.contiguous()
        B, Cc, L = x.shape
        W = weight.shape[1]
        if initial_states is not None and initial_states.dtype != x.dtype:
            initial_states = initial_states.to(x.dtype)
        out = torch.empty_like(x)  # keeps x's (channel-last or channel-first) strides
        fin = None
        if return_final_states:
            if final_states_out is not None:
                fin = final_states_out
            else:  # upstream allocates channel-last: (B, W-1, C).transpose(1, 2)
                fin = torch.empty(B, W - 1, Cc, dtype=x.dtype, device=x.device).transpose(1, 2)
        if x.numel() > 0:
            p = K.Conv1dFwd(x=K.T(x), weight=K.T(weight), bias=K.T(bias), initial_states=K.T(initial_states), out=K.T(out),
                            final_states=K.T(fin), silu=_act_flag(activation))
            K.run(lib, "omk_causal_conv1d_fwd", p, x)
        ctx.save_for_backward(x, weight, bias, initial_states)
        ctx.silu = _act_flag(activation)
        ctx.return_final_states = return_final_states
        return (out, fin) if return_final_states else out

    @staticmethod
    def backward(ctx, dout, *args):
        lib = get_lib()
        x, weight, bias, initial_states = ctx.saved_tensors
        if ctx.return_final_states and args and args[0] is not None and bool((args[0] != 0).any()):
            raise NotImplementedError("gradient through final_states is not supported")
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        if dout.stride() != x.stride():
            d2 = torch.empty_like(x)
            d2.copy_(dout)
            dout = d2
        dx = torch.empty_like(x)
        dw = torch.zeros(weight.shape, dtype=torch.float32, device=x.device)
        db = None if bias is None else torch.zeros(bias.shape, dtype=torch.float32, device=x.device)
        dinit = None
        if initial_states is not None and ctx.needs_input_grad[3]:
            dinit = torch.empty_like(initial_states)
        if x.numel() > 0:
            p = K.Conv1dBwd(x=K.T(x), weight=K.T(weight), bias=K.T(bias), initial_states=K.T(initial_states), dout=K.T(dout),
                            dx=K.T(dx), dweight=K.T(dw), dbias=K.T(db), dinitial_states=K.T(dinit), silu=ctx.silu)
            ws = K.workspace(lib, "omk_causal_conv1d_bwd_workspace_bytes", p, x)   # partial dw / db rows: no atomics, the same sums on every run
            K.run(lib, "omk_causal_conv1d_bwd", p, x)
        return dx, dw.to(weight.dtype), None if bias is None else db.to(bias.dtype), dinit, None, None, None


def causal_conv1d_fn(x, weight, bias=None, seq_idx=None, initial_states=None, return_final_states=False,
                     final_states_out=None, activation=None):
    """x: (batch, dim, seqlen); weight: (dim, width); bias: (dim,); initial_states: (batch, dim, width-1).
    Returns out (batch, dim, seqlen) [, final_states (batch, dim, width-1)]."""
    if seq_idx is not None:
        raise NotImplementedError("seq_idx never reaches the mixer in OmniMamba (mixer_seq_simple.py:375,408-420)")
    return CausalConv1dFn.apply(x, weight, bias, initial_states, return_final_states, final_states_out, activation)


def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None, cache_seqlens=None,
                         conv_state_indices=None):
    """x: (batch, dim) or (batch, dim, seqlen); conv_state: (batch, dim, state_len >= width-1), updated in place."""
    if cache_seqlens is not None or conv_state_indices is not None:
        raise NotImplementedError("cache_seqlens / conv_state_indices are not on the OmniMamba path")
    lib = get_lib()
    require_device(lib, x, conv_state, weight, bias)
    squeeze = x.dim() == 2
    x3 = x.unsqueeze(-1) if squeeze else x
    out = torch.empty_like(x3)
    if x3.numel() > 0:
        p = K.Conv1dUpdate(x=K.T(x3), conv_state=K.T(conv_state), weight=K.T(weight), bias=K.T(bias), out=K.T(out),
                           silu=_act_flag(activation))
        K.run(lib, "omk_causal_conv1d_update", p, x3)
    return out.squeeze(-1) if squeeze else out

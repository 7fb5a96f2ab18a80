"""Single-token SSM state update on the MI355X (decode step of Mamba2.step).

Mirrors ``mamba_ssm.ops.triton.selective_state_update.selective_state_update``; reference reach:
/root/reference/models/stage2/generation.py:195-211,412-424 -> MixerModel.forward -> Block -> Mamba2.step.
Kernel: omk_selective_state_update (omnimamba_amd/csrc/state_update.hip), in place on ``state``, graph-capturable.
"""
from __future__ import annotations

import torch

from . import _capi as K
from ._lib import get_lib, require_device


def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False,
                           state_batch_indices=None):
    """state: (batch, dim, dstate) or (batch, nheads, dim, dstate), updated IN PLACE.
    x, dt, z: (batch, dim) or (batch, nheads, dim); A: (dim, dstate) or (nheads, dim, dstate);
    B, C: (batch, dstate) or (batch, ngroups, dstate); D, dt_bias: (dim) or (nheads, dim).  Returns out like x."""
    if state_batch_indices is not None:
        raise NotImplementedError("state_batch_indices is not on the OmniMamba path")
    lib = get_lib()
    require_device(lib, state, x, dt, A, B, C, D, z, dt_bias)
    has_heads = state.dim() > 3
    if not has_heads:
        state_v, x_v, dt_v, A_v = state.unsqueeze(1), x.unsqueeze(1), dt.unsqueeze(1), A.unsqueeze(0)
        B_v, C_v = (B.unsqueeze(1) if B.dim() == 2 else B), (C.unsqueeze(1) if C.dim() == 2 else C)
        D_v = None if D is None else D.unsqueeze(0)
        z_v = None if z is None else z.unsqueeze(1)
        tb_v = None if dt_bias is None else dt_bias.unsqueeze(0)
    else:
        state_v, x_v, dt_v, A_v, B_v, C_v, D_v, z_v, tb_v = state, x, dt, A, B, C, D, z, dt_bias
    if B_v.dtype != x_v.dtype:
        B_v = B_v.to(x_v.dtype)
    if C_v.dtype != x_v.dtype:
        C_v = C_v.to(x_v.dtype)
    if z_v is not None and z_v.dtype != x_v.dtype:
        z_v = z_v.to(x_v.dtype)
    out = torch.empty_like(x_v)
    if x_v.numel() > 0:
        p = K.StateUpdate(state=K.T(state_v), x=K.T(x_v), dt=K.T(dt_v), A=K.T(A_v), Bm=K.T(B_v), Cm=K.T(C_v), D=K.T(D_v),
                          z=K.T(z_v), dt_bias=K.T(tb_v), out=K.T(out), dt_softplus=int(dt_softplus))
        K.run(lib, "omk_selective_state_update", p, x_v)
    return out if has_heads else out.squeeze(1)

"""Mamba-1 selective scan on the MI355X (``selective_scan_fn`` signature; BASELINE.json configs[0]).

Mirrors ``mamba_ssm.ops.selective_scan_interface.selective_scan_fn`` (importable mixer alternative at
/root/reference/models/stage2/mixer_seq_simple.py:16,197-201).  Kernel: omk_selective_scan_fwd
(omnimamba_amd/csrc/selscan.hip), coalesced for both (B, D, L) and channel-last (B, L, D) storage.
"""
from __future__ import annotations

import torch

from . import _capi as K
from ._lib import get_lib, require_device


class SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        lib = get_lib()
        require_device(lib, u, delta, A, B, C, D, z, delta_bias)
        if A.is_complex():
            raise NotImplementedError("complex A is not supported")
        if delta.dtype != u.dtype:
            delta = delta.to(u.dtype)
        if z is not None and z.dtype != u.dtype:
            z = z.to(u.dtype)
        B4 = B.unsqueeze(1) if B.dim() == 3 else B
        C4 = C.unsqueeze(1) if C.dim() == 3 else C
        out = torch.empty_like(u)
        Bsz, Dm, L = u.shape
        last = torch.empty(Bsz, Dm, A.shape[1], dtype=torch.float32, device=u.device) if return_last_state else None
        if u.numel() > 0:
            p = K.SelScanFwd(u=K.T(u), delta=K.T(delta), A=K.T(A.float() if A.dtype != torch.float32 else A), Bm=K.T(B4),
                             Cm=K.T(C4), D=K.T(D), z=K.T(z), delta_bias=K.T(delta_bias), out=K.T(out),
                             last_state=K.T(last), delta_softplus=int(delta_softplus))
            K.run(lib, "omk_selective_scan_fwd", p, u)
        ctx.return_last_state = return_last_state
        return (out, last) if return_last_state else out

    @staticmethod
    def backward(ctx, *grads):
        raise NotImplementedError("selective_scan_fn backward (Mamba-1) is not implemented: OmniMamba's shipped "
                                  "configs only build Mamba2 mixers (models/stage2/config_mamba.py:16)")


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """u, delta, z: (batch, dim, L); A: (dim, dstate); B, C: (dim, dstate) | (batch, dstate, L) |
    (batch, ngroups, dstate, L); D, delta_bias: (dim).  out: (batch, dim, L) [, last_state (batch, dim, dstate) fp32]."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)

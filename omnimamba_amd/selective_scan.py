"""Mamba-1 selective scan on the MI355X (``selective_scan_fn`` signature; BASELINE.json configs[0]).

Mirrors ``mamba_ssm.ops.selective_scan_interface.selective_scan_fn`` (importable mixer alternative at
/root/reference/models/stage2/mixer_seq_simple.py:16,197-201).  Kernels: omk_selective_scan_fwd / _bwd
(omnimamba_amd/csrc/selscan.hip), coalesced for both (B, D, L) and channel-last (B, L, D) storage.
"""
from __future__ import annotations

import ctypes as _C
import os

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
        Bsz, Dm, L = u.shape
        out = None
        if u.stride(1) == 1 and Dm > 1 and u.numel() > 0:
            # channel-last views of a (B, L, D) projection (what the Mamba-1 module holds).  With enough sequences the lanes-are-channels
            # sweep reads them as they lie (omk_selective_scan_fwd_form == 2, selscan.hip); the output then is channel-last as well
            o = torch.empty(Bsz, L, Dm, dtype=u.dtype, device=u.device).transpose(1, 2)
            probe = K.SelScanFwd(u=K.T(u), delta=K.T(delta), A=K.T(A), Bm=K.T(B4), Cm=K.T(C4), D=K.T(D), z=K.T(z),
                                 delta_bias=K.T(delta_bias), out=K.T(o), last_state=K.T(None), pass_states=K.T(None),
                                 delta_softplus=int(delta_softplus))
            if lib.omk_selective_scan_fwd_form(_C.byref(probe)) == 2:
                out = o
        took_lanes = out is not None
        if out is None and L >= 64:
            # the chunked associative scan wants L-contiguous rows (lanes = time): one copy pass here costs far less than the per-channel
            # sequential kernel
            u, delta = (t if t.stride(-1) == 1 else t.contiguous() for t in (u, delta))
            z = z if z is None or z.stride(-1) == 1 else z.contiguous()
            B4 = B4 if B4.dim() != 4 or B4.stride(-1) == 1 else B4.contiguous()
            C4 = C4 if C4.dim() != 4 or C4.stride(-1) == 1 else C4.contiguous()
        if out is None:
            out = torch.empty_like(u)
        last = torch.empty(Bsz, Dm, A.shape[1], dtype=torch.float32, device=u.device) if return_last_state else None
        Af = A.float() if A.dtype != torch.float32 else A     # bound to a local: the kernel reads it after this line
        # a backward will follow and the chunked form applies: keep the state in front of every 512-token pass (B D L / 512 N floats) --
        # the backward then needs no second forward pass
        ps = None
        lanes_fwd = took_lanes and B4.dim() == 4 and C4.dim() == 4 and A.shape[1] <= 16
        if lanes_fwd and any(ctx.needs_input_grad) and u.numel() > 0 and not os.environ.get("OMK_SELSCAN_NO_PASS_STATES"):
            # the lanes-are-channels forward took channel-last views: so will the backward, from the state in front of every 16-token tile
            # (B, L / 16, N, D floats, channels innermost) -- without the tensor the backward runs the forward sweep once more to get them
            ps = torch.empty(Bsz, (L + 15) // 16, A.shape[1], Dm, dtype=torch.float32, device=u.device)
        if (ps is None and not lanes_fwd and L >= 64 and any(ctx.needs_input_grad) and B4.dim() == 4 and C4.dim() == 4 and B4.dtype == u.dtype and C4.dtype == u.dtype and
                (Dm // B4.shape[1]) % 8 == 0 and (out.stride(1) == 1 or all(t is None or t.stride(-1) == 1 for t in (u, delta, z, B4, C4))) and
                not os.environ.get("OMK_SELSCAN_SEQ") and not os.environ.get("OMK_SELSCAN_NO_PASS_STATES")):
            # = the conditions of the chunked backward (selscan.hip); without the tensor the backward runs a state-only forward pass first
            ps = torch.empty(Bsz, Dm, (L + 511) // 512, A.shape[1], dtype=torch.float32, device=u.device)
        if u.numel() > 0:
            p = K.SelScanFwd(u=K.T(u), delta=K.T(delta), A=K.T(Af), Bm=K.T(B4),
                             Cm=K.T(C4), D=K.T(D), z=K.T(z), delta_bias=K.T(delta_bias), out=K.T(out),
                             last_state=K.T(last), pass_states=K.T(ps), delta_softplus=int(delta_softplus))
            K.run(lib, "omk_selective_scan_fwd", p, u)
        ctx.return_last_state = return_last_state
        ctx.delta_softplus = bool(delta_softplus)
        ctx.b3, ctx.c3 = B.dim() == 3, C.dim() == 3
        ctx.save_for_backward(u, delta, A, B4, C4, D, z, delta_bias, ps)
        if return_last_state:
            ctx.mark_non_differentiable(last)
        return (out, last) if return_last_state else out

    @staticmethod
    def backward(ctx, dout, *unused):
        """omk_selective_scan_bwd: L-contiguous inputs with input-dependent B / C run the chunked associative scan in both
        directions (any d_state <= 64); other layouts the per-channel sequential kernel (d_state <= 16)."""
        lib = get_lib()
        u, delta, A, B4, C4, D, z, delta_bias, ps = ctx.saved_tensors
        dout = dout.to(u.dtype)
        lanes = False
        tile_ps = ps is not None and ps.shape[-1] == u.shape[1] and ps.shape[1] == (u.shape[2] + 15) // 16
        if u.stride(1) == 1 and u.shape[1] > 1 and u.numel() > 0 and (ps is None or tile_ps) and os.environ.get("OMK_SELSCAN_BWD_LANES", "1") != "0":   # (=0: developer A/B against the copies + chunked scan)
            # channel-last views (what the Mamba-1 module holds): with enough sequences the lanes-are-channels reverse sweep reads and
            # writes them as they lie (omk_selective_scan_bwd_form == 2, selscan.hip: selscan_bwd_lanes_kernel) -- no L-contiguous copies
            Bsz, Dm, L = u.shape
            cl = lambda t: t if t.stride(1) == 1 else t.transpose(1, 2).contiguous().transpose(1, 2)
            delta_c, dout_c, z_c = cl(delta), cl(dout), None if z is None else cl(z)
            du_c = torch.empty(Bsz, L, Dm, dtype=u.dtype, device=u.device).transpose(1, 2)
            dd_c = torch.empty(Bsz, L, Dm, dtype=u.dtype, device=u.device).transpose(1, 2)
            dz_c = None if z is None else torch.empty(Bsz, L, Dm, dtype=u.dtype, device=u.device).transpose(1, 2)
            probe = K.SelScanBwd(u=K.T(u), delta=K.T(delta_c), A=K.T(A.float() if A.dtype != torch.float32 else A), Bm=K.T(B4), Cm=K.T(C4),
                                 D=K.T(D), z=K.T(z_c), delta_bias=K.T(delta_bias), dout=K.T(dout_c), du=K.T(du_c), ddelta=K.T(dd_c),
                                 dA=K.T(None), dB=K.T(None), dC=K.T(None), dD=K.T(None), dz=K.T(dz_c), ddelta_bias=K.T(None),
                                 pass_states=K.T(ps), delta_softplus=int(ctx.delta_softplus))
            if lib.omk_selective_scan_bwd_form(_C.byref(probe)) == 2:
                lanes, delta, dout, z = True, delta_c, dout_c, z_c
        if not lanes and ps is not None and tile_ps:
            ps = None          # (tile states are of no use to the other forms)
        if not lanes and u.shape[-1] >= 64:
            # (the forward may have read channel-last views as they lay: the chunked backward wants rows along L)
            u, delta, dout = (t if t.stride(-1) == 1 else t.contiguous() for t in (u, delta, dout))
            z = z if z is None or z.stride(-1) == 1 else z.contiguous()
            B4 = B4 if B4.dim() != 4 or B4.stride(-1) == 1 else B4.contiguous()
            C4 = C4 if C4.dim() != 4 or C4.stride(-1) == 1 else C4.contiguous()
        Af = A.float() if A.dtype != torch.float32 else A
        if lanes:
            du, ddelta, dz = du_c, dd_c, dz_c
        else:
            du, ddelta = torch.empty_like(u), torch.empty_like(delta)
            dz = None if z is None else torch.empty_like(z)
        # the fp32 accumulators (the kernel adds into them) as slices of ONE zeroed buffer: one fill launch instead of five
        shapes = [A.shape, B4.shape, C4.shape, None if D is None else D.shape, None if delta_bias is None else delta_bias.shape]
        sizes = [0 if sh is None else int(torch.Size(sh).numel()) for sh in shapes]
        sizes = [(n + 3) // 4 * 4 for n in sizes]                 # 16-byte aligned slices
        acc = torch.zeros(sum(sizes), dtype=torch.float32, device=u.device)
        parts, o = [], 0
        for sh, n in zip(shapes, sizes):
            parts.append(None if sh is None else acc[o:o + int(torch.Size(sh).numel())].view(sh))
            o += n
        dA, dB, dC, dD, ddb = parts
        if u.numel() > 0:
            p = K.SelScanBwd(u=K.T(u), delta=K.T(delta), A=K.T(Af), Bm=K.T(B4), Cm=K.T(C4), D=K.T(D), z=K.T(z),
                             delta_bias=K.T(delta_bias), dout=K.T(dout), du=K.T(du), ddelta=K.T(ddelta), dA=K.T(dA), dB=K.T(dB),
                             dC=K.T(dC), dD=K.T(dD), dz=K.T(dz), ddelta_bias=K.T(ddb), pass_states=K.T(ps),
                             delta_softplus=int(ctx.delta_softplus))
            ws = K.workspace(lib, "omk_selective_scan_bwd_workspace_bytes", p, u)  # noqa: F841
            K.run(lib, "omk_selective_scan_bwd", p, u)
        dB = (dB.squeeze(1) if ctx.b3 else dB).to(B4.dtype)
        dC = (dC.squeeze(1) if ctx.c3 else dC).to(C4.dtype)
        return (du, ddelta, dA.to(A.dtype), dB, dC, None if D is None else dD.to(D.dtype), dz,
                None if delta_bias is None else ddb.to(delta_bias.dtype), None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """u, delta, z: (batch, dim, L); A: (dim, dstate); B, C: (dim, dstate) | (batch, dstate, L) |
    (batch, ngroups, dstate, L); D, delta_bias: (dim).  out: (batch, dim, L) [, last_state (batch, dim, dstate) fp32]."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False, return_last_state=False):
    """The pure-PyTorch form of the same signature (``mamba_ssm.ops.selective_scan_interface.selective_scan_ref``, the
    function BASELINE.json configs[0] times on the host): the recurrence of SURVEY.md Appendix A.5 written with torch ops,
    any device, fp32 arithmetic.  It is an API of its own, not a fallback of ``selective_scan_fn``."""
    dtype_in = u.dtype
    u, delta = u.float(), delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias[..., None].float()
    if delta_softplus:
        delta = torch.nn.functional.softplus(delta)
    batch, dim, dstate = u.shape[0], A.shape[0], A.shape[1]
    is_variable_B, is_variable_C = B.dim() >= 3, C.dim() >= 3
    B, C = B.float(), C.float()
    if is_variable_B and B.dim() == 3:
        B = B.unsqueeze(1)
    if is_variable_C and C.dim() == 3:
        C = C.unsqueeze(1)
    x = A.new_zeros((batch, dim, dstate), dtype=torch.float32)
    deltaA = torch.exp(torch.einsum("bdl,dn->bdln", delta, A.float()))
    if not is_variable_B:
        deltaB_u = torch.einsum("bdl,dn,bdl->bdln", delta, B, u)
    else:
        Bx = B.repeat_interleave(dim // B.shape[1], dim=1)                      # (batch, dim, dstate, L)
        deltaB_u = torch.einsum("bdl,bdnl,bdl->bdln", delta, Bx, u)
    if is_variable_C:
        Cx = C.repeat_interleave(dim // C.shape[1], dim=1)
    ys = []
    for i in range(u.shape[2]):
        x = deltaA[:, :, i] * x + deltaB_u[:, :, i]
        ys.append(torch.einsum("bdn,dn->bd", x, C) if not is_variable_C else torch.einsum("bdn,bdn->bd", x, Cx[:, :, :, i]))
    y = torch.stack(ys, dim=2)
    out = y if D is None else y + u * D.float()[..., None]
    if z is not None:
        out = out * torch.nn.functional.silu(z.float())
    out = out.to(dtype_in)
    return out if not return_last_state else (out, x)


def mamba_inner_fn(xz, conv1d_weight, conv1d_bias, x_proj_weight, delta_proj_weight, out_proj_weight, out_proj_bias, A, B=None,
                   C=None, D=None, delta_bias=None, B_proj_bias=None, C_proj_bias=None, delta_softplus=True):
    """The Mamba-1 mixer after in_proj (``mamba_ssm.ops.selective_scan_interface.mamba_inner_fn``): causal conv1d + SiLU on
    x, x_proj -> (dt, B, C), dt_proj, selective scan gated by z, out_proj.  xz: (batch, 2 * d_inner, L).  Upstream fuses these
    into one autograd node to save memory; here they are this package's own differentiable ops in sequence (omk_causal_conv1d,
    library GEMMs, omk_selective_scan) -- same arithmetic, same argument list."""
    import torch.nn.functional as F
    from .causal_conv1d import causal_conv1d_fn
    L = xz.shape[-1]
    d_inner, dt_rank = delta_proj_weight.shape
    d_state = A.shape[-1]
    x, z = xz.chunk(2, dim=1)
    w = conv1d_weight.squeeze(1) if conv1d_weight.dim() == 3 else conv1d_weight
    x = causal_conv1d_fn(x, w, conv1d_bias, activation="silu")
    x_dbl = F.linear(x.transpose(1, 2).reshape(-1, d_inner), x_proj_weight.to(x.dtype))                 # (batch * L, dt_rank + 2 d_state)
    # delta, B, C stay channel-last views of token-major GEMM outputs (like x and z): no copies here -- selective_scan_fn reads them
    # as they lie when the batch fills the chip (lanes = channels) and makes its own L-contiguous copies otherwise
    delta = F.linear(x_dbl[:, :dt_rank], delta_proj_weight.to(x.dtype)).view(-1, L, d_inner).transpose(1, 2)   # (batch, d_inner, L)
    if B is None:
        B = x_dbl[:, dt_rank:dt_rank + d_state]
        if B_proj_bias is not None:
            B = B + B_proj_bias.to(B.dtype)
        B = B.reshape(-1, L, d_state).transpose(1, 2)                                                        # (batch, d_state, L)
    if C is None:
        C = x_dbl[:, -d_state:]
        if C_proj_bias is not None:
            C = C + C_proj_bias.to(C.dtype)
        C = C.reshape(-1, L, d_state).transpose(1, 2)
    y = selective_scan_fn(x, delta, A, B, C, D, z=z, delta_bias=delta_bias, delta_softplus=delta_softplus)
    return F.linear(y.transpose(1, 2), out_proj_weight.to(y.dtype), None if out_proj_bias is None else out_proj_bias.to(y.dtype))

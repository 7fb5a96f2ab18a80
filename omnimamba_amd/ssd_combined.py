"""Mamba-2 SSD scan on the MI355X.

Mirrors ``mamba_ssm.ops.triton.ssd_combined.{mamba_chunk_scan_combined, mamba_split_conv1d_scan_combined}``
(the names HF wraps at transformers/models/mamba2/modeling_mamba2.py:166-187,253-268 and that the reference reaches
through Mamba2.forward, /root/reference/models/stage2/block.py:117).  Arithmetic: omk_ssd_scan_fwd / _bwd
(omnimamba_amd/csrc/ssd.hip, ssd_mfma.hip), omk_causal_conv1d_*, omk_norm_gated_*; the projection GEMMs stay on
hipBLASLt through torch.  No PyTorch fallback for the scan.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn.functional as F

from . import _capi as K
from . import _prof
from ._lib import get_lib, require_device
from .causal_conv1d import causal_conv1d_fn
from .layernorm_gated import rmsnorm_fn
from .linear import frozen_cast, weight_grad

_INF = float("inf")


def _last_contig(t):
    return t if t is None or t.stride(-1) == 1 else t.contiguous()


# ---- per-call options of the scan (OmkSsdFwd.flags / OmkSsdBwd.flags, include/omk.h).  The library reads no environment variable that
# changes the scan's numerics; a caller picks them per call -- `flags=` on the raw functions, or this context on everything below it
# (the autograd nodes record the flags of their forward for their backward).
import contextlib
import threading

_scan_opts = threading.local()


def current_scan_flags() -> int:
    return getattr(_scan_opts, "flags", 0)


@contextlib.contextmanager
def scan_options(precise=False, khilo=False, every_chunk=False, no_split=False, column_slice=False, sequential_bwd=False):
    """precise: hi + lo copy of the carried state and of the state-update operand -- the bare 1e-3 of the fp32 recurrence on every
    head (price: profiles/r06_precise.txt); khilo: only the state-update operand; every_chunk / column_slice / sequential_bwd /
    no_split: the older kernels and arithmetic orders the tests compare against."""
    f = ((K.SSD_PRECISE if precise else 0) | (K.SSD_KHILO if khilo else 0) | (K.SSD_EVERY_CHUNK if every_chunk else 0)
         | (K.SSD_NO_SPLIT if no_split else 0) | (K.SSD_COLUMN_SLICE if column_slice else 0)
         | (K.SSD_SEQUENTIAL_BWD if sequential_bwd else 0))
    prev = current_scan_flags()
    _scan_opts.flags = prev | f
    try:
        yield
    finally:
        _scan_opts.flags = prev


def save_window_states_enabled() -> bool:
    """OMK_SSD_SAVE_WINDOW_STATES=0: a training forward does not keep its window states (16 KB per head and 128 tokens); the
    backward then recomputes them with a state pass over x, as upstream's backward recomputes its chunk states."""
    return os.environ.get("OMK_SSD_SAVE_WINDOW_STATES", "1") != "0"


def ssd_scan_fwd(x, dt, A, B, C, D=None, z=None, dt_bias=None, initial_states=None, dt_softplus=False,
                 dt_limit=(0.0, _INF), return_final_states=False, want_out_x=False, chunk_size=256,
                 force_generic=False, save_window_states=False, flags=None):
    """Raw (non-autograd) forward: returns (out, out_x | None, final_states | None), with save_window_states=True a fourth
    element: the opaque window-state tensor omk_ssd_scan_bwd takes (None when this forward cannot produce it)."""
    lib = get_lib()
    require_device(lib, x, dt, A, B, C, D, z, dt_bias, initial_states)
    x, B, C, z = _last_contig(x), _last_contig(B), _last_contig(C), _last_contig(z)
    if B.dtype != x.dtype:
        B = B.to(x.dtype)
    if C.dtype != x.dtype:
        C = C.to(x.dtype)
    if z is not None and z.dtype != x.dtype:
        z = z.to(x.dtype)
    A = A.float().contiguous()
    Bsz, L, H, P = x.shape
    N = B.shape[-1]
    out = torch.empty(Bsz, L, H, P, dtype=x.dtype, device=x.device)
    out_x = torch.empty_like(out) if (want_out_x and z is not None) else None
    fin = torch.empty(Bsz, H, P, N, dtype=torch.float32, device=x.device) if return_final_states else None
    wstates = None
    if x.numel() > 0:
        p = K.SsdFwd(x=K.T(x), dt=K.T(dt), A=K.T(A), Bm=K.T(B), Cm=K.T(C), D=K.T(D), z=K.T(z), dt_bias=K.T(dt_bias),
                     initial_states=K.T(initial_states), out=K.T(out), out_x=K.T(out_x), final_states=K.T(fin),
                     dt_min=float(dt_limit[0]), dt_max=float(dt_limit[1]), dt_softplus=int(dt_softplus),
                     chunk_size=int(chunk_size), force_generic=int(force_generic),
                     flags=int(current_scan_flags() if flags is None else flags) & ~K.SSD_SEQUENTIAL_BWD)
        ws = K.workspace(lib, "omk_ssd_scan_fwd_workspace_bytes", p, x)  # noqa: F841
        if save_window_states:
            nbytes = lib.omk_ssd_scan_fwd_window_states_bytes(K.C.byref(p))
            if nbytes:
                wstates = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=x.device)
                p.window_states = K.T(wstates)
        with _prof.range_("ssd_scan_fwd"):
            K.run(lib, "omk_ssd_scan_fwd", p, x)
        _prof.note_kernels("ssd_scan_fwd", lib)
    elif fin is not None:
        fin.zero_() if initial_states is None else fin.copy_(initial_states)
    if save_window_states:
        return out, out_x, fin, wstates
    return out, out_x, fin


def fused_conv_scan_enabled() -> bool:
    """OMK_K2_FUSED=0: forward-only passes run conv kernel + scan separately (the fused form: conv1d + SiLU of the x channels inside the
    scan's staging, OmkSsdFwd.conv_weight)."""
    return os.environ.get("OMK_K2_FUSED", "1") != "0"


def ssd_scan_fwd_fused_conv(xBC, dt, A, conv_weight, conv_bias, nheads, headdim, ngroups, d_state, D=None, dt_bias=None,
                            dt_softplus=True, dt_limit=(0.0, _INF), return_final_states=False, conv_state_out=None, chunk_size=256):
    """K2 fusion of the forward-only path (reference reach: models/stage2/generation.py:195-211 prefill,
    scripts/inference_mmu.py:137-147): xBC (batch, seqlen, d_ssm + 2 G N) is the PRE-conv slice of zxbcdt.  The 2 G N B / C channels go
    through a small conv launch into a dense buffer; the d_ssm x channels are convolved INSIDE the scan while it stages them
    (OmkSsdFwd.conv_weight) and never visit a conv output buffer.  conv_state_out (batch, d_ssm + 2 G N, state_len): filled with the
    last pre-conv inputs like causal_conv1d_fn(..., final_states_out=) does.  Returns (y (batch, seqlen, nheads, headdim), final_states |
    None), or None when the kernel does not take this call (the caller then runs the separate ops) -- bit-identical results either way."""
    lib = get_lib()
    Bsz, L, Ct = xBC.shape
    H, P, G, N = nheads, headdim, ngroups, d_state
    d_ssm = H * P
    if (xBC.dtype != torch.bfloat16 or P != 64 or N != 128 or xBC.stride(-1) != 1 or Ct != d_ssm + 2 * G * N or conv_weight.shape[1] > 4
            or not fused_conv_scan_enabled() or L == 0 or Bsz == 0):
        return None
    # The conv work rides on the scan's workgroups (one per head pair): it pays when they fill the chip -- 256 workgroups at batch 8 of the
    # 1.3B block: - 38 us per block -- and loses when the scan runs on a fraction of the CUs while the stand-alone conv kernel would use all of
    # them (batch 4: + 7 us, batch 1: + 5 us; profiles/r06_k2_fusion.txt).  OMK_K2_MIN_WGS: developer override of the threshold.
    if Bsz * (H // 2) < int(os.environ.get("OMK_K2_MIN_WGS", "256")):
        return None
    require_device(lib, xBC, dt, A, conv_weight, conv_bias, D, dt_bias)
    A = A.float().contiguous()
    x_pre = xBC[..., :d_ssm].unflatten(-1, (H, P))
    bc = torch.empty(Bsz, L, 2 * G * N, dtype=xBC.dtype, device=xBC.device)
    out = torch.empty(Bsz, L, H, P, dtype=xBC.dtype, device=xBC.device)
    fin = torch.empty(Bsz, H, P, N, dtype=torch.float32, device=xBC.device) if return_final_states else None
    wx, wbc = conv_weight[:d_ssm], conv_weight[d_ssm:]
    bx, bbc = (None, None) if conv_bias is None else (conv_bias[:d_ssm].contiguous(), conv_bias[d_ssm:])
    Bm, Cm = bc[..., :G * N].unflatten(-1, (G, N)), bc[..., G * N:].unflatten(-1, (G, N))
    p = K.SsdFwd(x=K.T(x_pre), dt=K.T(dt), A=K.T(A), Bm=K.T(Bm), Cm=K.T(Cm), D=K.T(D), z=K.T(None), dt_bias=K.T(dt_bias),
                 initial_states=K.T(None), out=K.T(out), out_x=K.T(None), final_states=K.T(fin), dt_min=float(dt_limit[0]),
                 dt_max=float(dt_limit[1]), dt_softplus=int(dt_softplus), chunk_size=int(chunk_size), force_generic=0,
                 flags=int(current_scan_flags()) & (K.SSD_KHILO | K.SSD_EVERY_CHUNK), conv_weight=K.T(wx), conv_bias=K.T(bx))
    ws = K.workspace(lib, "omk_ssd_scan_fwd_workspace_bytes", p, xBC)  # noqa: F841
    # the B / C conv first (the scan reads its output); its epilogue fills the B / C rows of conv_state_out
    pc = K.Conv1dFwd(x=K.T(xBC[..., d_ssm:].transpose(1, 2)), weight=K.T(wbc), bias=K.T(bbc), initial_states=K.T(None),
                     out=K.T(bc.transpose(1, 2)), final_states=K.T(None if conv_state_out is None else conv_state_out[:, d_ssm:]), silu=1)
    fn = lib.omk_ssd_scan_fwd
    import ctypes as C_
    if lib.omk_is_emulated():
        K.run(lib, "omk_causal_conv1d_fwd", pc, xBC)
        rc = fn(C_.byref(p), None)
    else:
        with torch.cuda.device(xBC.device):
            K.run(lib, "omk_causal_conv1d_fwd", pc, xBC)
            with _prof.range_("ssd_scan_fwd_fused_conv"):
                rc = fn(C_.byref(p), K.stream_of(lib, xBC))
    if rc == -4:      # OMK_EUNSUPPORTED: heads that do not pair up, a sequence the scan splits, strides outside the MFMA kernel
        return None
    K.check(lib, rc, "omk_ssd_scan_fwd (fused conv)")
    if conv_state_out is not None:      # the x rows of the conv state: the last state_len pre-conv inputs (left zero padded)
        sl = conv_state_out.shape[-1]
        cs = conv_state_out[:, :d_ssm]
        if L >= sl:
            cs.copy_(xBC[:, L - sl:, :d_ssm].transpose(1, 2))
        else:
            cs[..., :sl - L].zero_()
            cs[..., sl - L:].copy_(xBC[:, :, :d_ssm].transpose(1, 2))
    return out, fin


def ssd_scan_bwd(dout, x, dt, A, B, C, D=None, dt_bias=None, initial_states=None, dfinal_states=None,
                 dt_softplus=False, dt_limit=(0.0, _INF), chunk_size=256, need_dinit=False, force_generic=False, y=None,
                 dx_out=None, dB_out=None, dC_out=None, window_states=None, flags=None):
    """Raw backward: returns dict(dx, ddt, dA, dB, dC, dD, ddt_bias, dinitial_states).  `y` = the forward's pre-gate
    output (D*x included); with it the MFMA path applies."""
    lib = get_lib()
    x, B, C, dout, y = _last_contig(x), _last_contig(B), _last_contig(C), _last_contig(dout), _last_contig(y)
    if y is not None and y.dtype != x.dtype:
        y = y.to(x.dtype)
    if B.dtype != x.dtype:
        B = B.to(x.dtype)
    if C.dtype != x.dtype:
        C = C.to(x.dtype)
    if dout.dtype != x.dtype:
        dout = dout.to(x.dtype)
    A = A.float().contiguous()
    Bsz, L, H, P = x.shape
    G, N = B.shape[2], B.shape[3]
    dev = x.device
    dx = dx_out if dx_out is not None else torch.empty(Bsz, L, H, P, dtype=x.dtype, device=dev)
    # (B, L, H) view of a (B, H, L) buffer: the finishing pass walks tokens of one head with adjacent lanes, so its stores are
    # full rows (the (B, L, H)-contiguous form made every 4-byte store its own 64-byte write: 73 MB for 4 MB of data)
    ddt = torch.empty(Bsz, H, L, dtype=torch.float32, device=dev).transpose(1, 2)
    dA = torch.empty(H, dtype=torch.float32, device=dev)
    dB = dB_out if dB_out is not None else torch.empty(Bsz, L, G, N, dtype=x.dtype, device=dev)
    dC = dC_out if dC_out is not None else torch.empty(Bsz, L, G, N, dtype=x.dtype, device=dev)
    dD = None if D is None else torch.empty(D.shape, dtype=torch.float32, device=dev)
    ddtb = None if dt_bias is None else torch.empty(H, dtype=torch.float32, device=dev)
    dinit = torch.empty(Bsz, H, P, N, dtype=torch.float32, device=dev) if need_dinit else None
    if dfinal_states is not None:
        dfinal_states = dfinal_states.float()
    if x.numel() > 0:
        p = K.SsdBwd(x=K.T(x), dt=K.T(dt), A=K.T(A), Bm=K.T(B), Cm=K.T(C), D=K.T(D), dt_bias=K.T(dt_bias),
                     initial_states=K.T(initial_states), y=K.T(y), dout=K.T(dout), dfinal_states=K.T(dfinal_states), dx=K.T(dx),
                     ddt=K.T(ddt), dA=K.T(dA), dB=K.T(dB), dC=K.T(dC), dD=K.T(dD), ddt_bias=K.T(ddtb),
                     dinitial_states=K.T(dinit), window_states=K.T(window_states), dt_min=float(dt_limit[0]), dt_max=float(dt_limit[1]),
                     dt_softplus=int(dt_softplus), chunk_size=int(chunk_size), force_generic=int(force_generic),
                     flags=int(current_scan_flags() if flags is None else flags) & (K.SSD_EVERY_CHUNK | K.SSD_NO_SPLIT | K.SSD_COLUMN_SLICE | K.SSD_SEQUENTIAL_BWD))
        ws = K.workspace(lib, "omk_ssd_scan_bwd_workspace_bytes", p, x)  # noqa: F841
        with _prof.range_("ssd_scan_bwd"):
            K.run(lib, "omk_ssd_scan_bwd", p, x)
        _prof.note_kernels("ssd_scan_bwd", lib)
    else:
        for t in (dA, dD, ddtb):
            if t is not None:
                t.zero_()
        if dinit is not None:
            dinit.zero_() if dfinal_states is None else dinit.copy_(dfinal_states)
    return dict(dx=dx, ddt=ddt, dA=dA, dB=dB, dC=dC, dD=dD, ddt_bias=ddtb, dinitial_states=dinit)


def ssd_final_state_raw(x, dt, A, B, dt_bias=None, initial_states=None, dt_softplus=False, dt_limit=(0.0, _INF)):
    """State behind the sequence, fp32 (batch, nheads, headdim, dstate), without the scan's output: the state-only pass of
    omk_ssd_scan_fwd (`out` absent; bf16, headdim 64, d_state 128) -- about a third of a scan -- or, for other shapes, the scan
    itself with its output dropped."""
    lib = get_lib()
    require_device(lib, x, dt, A, B, dt_bias, initial_states)
    Bsz, L, H, P = x.shape
    N = B.shape[-1]
    if x.numel() == 0:
        fin = torch.zeros(Bsz, H, P, N, dtype=torch.float32, device=x.device)
        return fin if initial_states is None else fin.copy_(initial_states)
    if x.dtype == torch.bfloat16 and P == 64 and N == 128 and os.environ.get("OMK_SSD_STATE_ONLY", "1") != "0":
        x, B = _last_contig(x), _last_contig(B)
        B = B if B.dtype == x.dtype else B.to(x.dtype)
        fin = torch.empty(Bsz, H, P, N, dtype=torch.float32, device=x.device)
        p = K.SsdFwd(x=K.T(x), dt=K.T(dt), A=K.T(A.float().contiguous()), Bm=K.T(B), Cm=K.T(B), dt_bias=K.T(dt_bias),
                     initial_states=K.T(initial_states), final_states=K.T(fin), dt_min=float(dt_limit[0]), dt_max=float(dt_limit[1]),
                     dt_softplus=int(dt_softplus), chunk_size=256, flags=int(current_scan_flags()) & ~K.SSD_SEQUENTIAL_BWD)
        ws = K.workspace(lib, "omk_ssd_scan_fwd_workspace_bytes", p, x)  # noqa: F841
        fn = getattr(lib, "omk_ssd_scan_fwd")
        import ctypes as C
        if lib.omk_is_emulated():
            rc = fn(C.byref(p), None)
        else:
            with torch.cuda.device(x.device):
                rc = fn(C.byref(p), K.stream_of(lib, x))
        if rc == 0:
            return fin
        if rc != -4:       # OMK_EUNSUPPORTED (strides / alignment outside the MFMA kernel): fall through to the scan
            K.check(lib, rc, "omk_ssd_scan_fwd (state only)")
    return ssd_scan_fwd(x, dt, A, B, B, dt_bias=dt_bias, initial_states=initial_states, dt_softplus=dt_softplus, dt_limit=dt_limit,
                        return_final_states=True)[2]


class SsdFinalStateFn(torch.autograd.Function):
    """final_states of the scan as a differentiable function of (x, dt, A, B, dt_bias, initial_states) -- the first dispatch of a
    context-parallel shard.  Backward = the scan's backward with a zero output gradient and dfinal_states."""

    @staticmethod
    def forward(ctx, x, dt, A, B, dt_bias, initial_states, dt_softplus, dt_limit):
        fin = ssd_final_state_raw(x, dt, A, B, dt_bias, initial_states, dt_softplus, dt_limit)
        ctx.save_for_backward(x, dt, A, B, dt_bias, initial_states)
        ctx.cfg = (dt_softplus, dt_limit)
        return fin

    @staticmethod
    def backward(ctx, dfin):
        x, dt, A, B, dt_bias, initial_states = ctx.saved_tensors
        dt_softplus, dt_limit = ctx.cfg
        g = ssd_scan_bwd(torch.zeros_like(x), x, dt, A, B, torch.zeros_like(B), dt_bias=dt_bias, initial_states=initial_states,
                         dfinal_states=dfin, dt_softplus=dt_softplus, dt_limit=dt_limit, need_dinit=initial_states is not None)
        dinit = g["dinitial_states"]
        return (g["dx"], g["ddt"].to(dt.dtype), g["dA"].to(A.dtype), g["dB"].to(B.dtype),
                None if dt_bias is None else g["ddt_bias"].to(dt_bias.dtype),
                None if dinit is None else dinit.to(initial_states.dtype), None, None)


def ssd_final_state(x, dt, A, B, dt_bias=None, initial_states=None, dt_softplus=False, dt_limit=(0.0, _INF)):
    return SsdFinalStateFn.apply(x, dt, A, B, dt_bias, initial_states, dt_softplus, dt_limit)


def _silu_grad(z):
    s = torch.sigmoid(z)
    return s * (1 + z * (1 - s))


class MambaChunkScanCombinedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dt, A, B, C, chunk_size, D=None, z=None, dt_bias=None, initial_states=None, seq_idx=None,
                cu_seqlens=None, dt_softplus=False, dt_limit=(0.0, _INF), return_final_states=False,
                return_varlen_states=False):
        if seq_idx is not None or cu_seqlens is not None or return_varlen_states:
            raise NotImplementedError("seq_idx / cu_seqlens / varlen states never reach the mixer in OmniMamba "
                                      "(models/stage2/mixer_seq_simple.py:375,408-420)")
        keep_ws = any(ctx.needs_input_grad) and save_window_states_enabled()
        r = ssd_scan_fwd(x, dt, A, B, C, D=D, z=z, dt_bias=dt_bias, initial_states=initial_states, dt_softplus=dt_softplus,
                         dt_limit=dt_limit, return_final_states=return_final_states, want_out_x=True, chunk_size=chunk_size,
                         save_window_states=keep_ws)
        (out, out_x, fin), wst = r[:3], (r[3] if keep_ws else None)
        ctx.scan_flags = current_scan_flags()
        ctx.save_for_backward(x, dt, A, B, C, D, z, dt_bias, initial_states, out_x if z is not None else out, wst)
        ctx.dt_softplus, ctx.dt_limit, ctx.chunk_size = dt_softplus, dt_limit, chunk_size
        ctx.return_final_states = return_final_states
        return (out, fin) if return_final_states else out

    @staticmethod
    def backward(ctx, dout, *args):
        x, dt, A, B, C, D, z, dt_bias, initial_states, out_x, wst = ctx.saved_tensors
        dfinal = args[0] if ctx.return_final_states and args else None
        dz = None
        if z is not None:
            zf = z.float()
            dz = (dout.float() * out_x.float() * _silu_grad(zf)).to(z.dtype)
            dout = (dout.float() * F.silu(zf)).to(x.dtype)
        g = ssd_scan_bwd(dout, x, dt, A, B, C, D=D, dt_bias=dt_bias, initial_states=initial_states,
                         dfinal_states=dfinal, dt_softplus=ctx.dt_softplus, dt_limit=ctx.dt_limit,
                         chunk_size=ctx.chunk_size, need_dinit=initial_states is not None, y=out_x, window_states=wst,
                         flags=ctx.scan_flags)
        dinit = g["dinitial_states"]
        return (g["dx"], g["ddt"].to(dt.dtype), g["dA"].to(A.dtype), g["dB"].to(B.dtype), g["dC"].to(C.dtype), None,
                None if D is None else g["dD"].to(D.dtype), dz,
                None if dt_bias is None else g["ddt_bias"].to(dt_bias.dtype),
                None if dinit is None else dinit.to(initial_states.dtype), None, None, None, None, None, None)


def mamba_chunk_scan_combined(x, dt, A, B, C, chunk_size, D=None, z=None, dt_bias=None, initial_states=None,
                              seq_idx=None, cu_seqlens=None, dt_softplus=False, dt_limit=(0.0, _INF),
                              return_final_states=False, return_varlen_states=False):
    """x: (batch, seqlen, nheads, headdim); dt: (batch, seqlen, nheads); A: (nheads); B, C: (batch, seqlen, ngroups,
    dstate); D: (nheads, headdim) or (nheads,); z: like x; dt_bias: (nheads,); initial_states: (batch, nheads,
    headdim, dstate).  Returns out like x [, final_states (batch, nheads, headdim, dstate) fp32]."""
    return MambaChunkScanCombinedFn.apply(x, dt, A, B, C, chunk_size, D, z, dt_bias, initial_states, seq_idx,
                                          cu_seqlens, dt_softplus, dt_limit, return_final_states,
                                          return_varlen_states)


class MambaSplitConv1dScanCombinedFn(torch.autograd.Function):
    """conv1d+SiLU -> SSD scan -> gated RMSNorm -> out_proj as ONE autograd node (upstream K2; SURVEY.md section 8
    row a9).  Saves zxbcdt, the pre-norm y and the parameters.  Upstream recomputes the conv output and the norm output
    in backward to save memory; with 288 GB of HBM the default here keeps them (0.55 GB per 32k-token block call, 0.33 ms
    of recompute saved per block backward).  OMK_RECOMPUTE=1 restores upstream's memory behaviour."""

    @staticmethod
    def forward(ctx, zxbcdt, conv1d_weight, conv1d_bias, dt_bias, A, D, chunk_size, initial_states=None, seq_idx=None,
                dt_limit=(0.0, _INF), return_final_states=False, activation="silu", rmsnorm_weight=None,
                rmsnorm_eps=1e-6, outproj_weight=None, outproj_bias=None, headdim=None, ngroups=1,
                norm_before_gate=True, conv_state_out=None):
        if seq_idx is not None:
            raise NotImplementedError("seq_idx never reaches the mixer in OmniMamba")
        if activation not in ("silu", "swish"):
            raise NotImplementedError("activation must be silu/swish")
        if D.dim() == 1:
            assert headdim is not None
            H = D.shape[0]
        else:
            H, headdim = D.shape
        Bsz, L, _ = zxbcdt.shape
        P, G = headdim, ngroups
        d_ssm = H * P
        N = (conv1d_weight.shape[0] - d_ssm) // (2 * G)
        if zxbcdt.shape[-1] != 2 * d_ssm + 2 * G * N + H:
            raise NotImplementedError("d_mlp > 0 (d_ssm < d_inner) is handled by the un-fused path of Mamba2.forward")
        if zxbcdt.stride(-1) != 1:
            zxbcdt = zxbcdt.contiguous()
        z, xBC, dt = torch.split(zxbcdt, [d_ssm, d_ssm + 2 * G * N, H], dim=-1)
        use_norm = rmsnorm_weight is not None
        fused = None
        if not any(ctx.needs_input_grad) and initial_states is None and use_norm and D.dim() == 1:
            # forward-only (prefill / inference): K2 fusion -- the x channels are convolved inside the scan's staging
            fused = ssd_scan_fwd_fused_conv(xBC, dt, A, conv1d_weight, conv1d_bias, H, P, G, N, D=D, dt_bias=dt_bias, dt_softplus=True,
                                            dt_limit=dt_limit, return_final_states=return_final_states, conv_state_out=conv_state_out,
                                            chunk_size=chunk_size)
        if fused is not None:
            y, fin = fused
            out_n = rmsnorm_fn(y.reshape(Bsz, L, d_ssm), rmsnorm_weight, None, z=z, eps=rmsnorm_eps, group_size=d_ssm // G,
                               norm_before_gate=norm_before_gate)
            if outproj_weight is not None:
                w = outproj_weight if outproj_weight.dtype == out_n.dtype else frozen_cast(outproj_weight, out_n.dtype)
                out = F.linear(out_n, w, None if outproj_bias is None else outproj_bias.to(out_n.dtype))
            else:
                out = out_n
            return (out, fin) if return_final_states else out
        if conv_state_out is not None:
            # prefill: the conv kernel's epilogue leaves the last `state_len` pre-conv inputs in the
            # cache's conv_state (left zero padded when L < state_len) -- no separate pad / copy pass
            xBC_c = causal_conv1d_fn(xBC.transpose(1, 2), conv1d_weight, conv1d_bias, return_final_states=True,
                                     final_states_out=conv_state_out, activation=activation)[0].transpose(1, 2)
        else:
            xBC_c = causal_conv1d_fn(xBC.transpose(1, 2), conv1d_weight, conv1d_bias, activation=activation).transpose(1, 2)
        x, Bm, Cm = torch.split(xBC_c, [d_ssm, G * N, G * N], dim=-1)
        zz = z.reshape(Bsz, L, H, P) if z.is_contiguous() else z.unflatten(-1, (H, P))
        keep_ws = any(ctx.needs_input_grad) and save_window_states_enabled()
        r = ssd_scan_fwd(x.unflatten(-1, (H, P)), dt, A, Bm.unflatten(-1, (G, N)), Cm.unflatten(-1, (G, N)), D=D,
                         z=None if use_norm else zz, dt_bias=dt_bias, initial_states=initial_states,
                         dt_softplus=True, dt_limit=dt_limit, return_final_states=return_final_states,
                         want_out_x=True, chunk_size=chunk_size, save_window_states=keep_ws)
        (y, y_x, fin), wst = r[:3], (r[3] if keep_ws else None)
        y_pre = y if (use_norm or y_x is None) else y_x        # pre-gate / pre-norm scan output (D*x included)
        if use_norm:
            out_n = rmsnorm_fn(y.reshape(Bsz, L, d_ssm), rmsnorm_weight, None, z=z, eps=rmsnorm_eps,
                               group_size=d_ssm // G, norm_before_gate=norm_before_gate)
        else:
            out_n = y.reshape(Bsz, L, d_ssm)
        if outproj_weight is not None:
            w = outproj_weight if outproj_weight.dtype == out_n.dtype else frozen_cast(outproj_weight, out_n.dtype)   # cast once per weight version
            out = F.linear(out_n, w, None if outproj_bias is None else outproj_bias.to(out_n.dtype))
            ctx.w_cast = w          # the 16-bit copy the input gradient reuses (saved_tensors hands back new objects: no cache hit there)
        else:
            out = out_n
        keep = os.environ.get("OMK_RECOMPUTE", "0") != "1"
        # the norm output is only the operand of out_proj's WEIGHT gradient: a frozen out_proj ('align' stage) neither keeps it
        # (134 MB per 16 k tokens and layer) nor forms that gradient (a 0.27 TFLOP GEMM per layer: 7 % of the stage-1 step)
        need_wo = outproj_weight is not None and ctx.needs_input_grad[14]
        ctx.save_for_backward(zxbcdt, conv1d_weight, conv1d_bias, dt_bias, A, D, y_pre, rmsnorm_weight, outproj_weight,
                              outproj_bias, initial_states, xBC_c if keep else None,
                              out_n if (keep and use_norm and need_wo) else None, wst)
        ctx.cfg = (H, P, G, N, chunk_size, dt_limit, activation, rmsnorm_eps, norm_before_gate, return_final_states)
        ctx.scan_flags = current_scan_flags()
        return (out, fin) if return_final_states else out

    @staticmethod
    def backward(ctx, dout, *args):
        (zxbcdt, conv_w, conv_b, dt_bias, A, D, y_pre, norm_w, outproj_w, outproj_b, initial_states, xBC_saved,
         on_saved, wst) = ctx.saved_tensors
        H, P, G, N, chunk_size, dt_limit, activation, eps, nbg, ret_fin = ctx.cfg
        dfinal = args[0] if ret_fin and args else None
        Bsz, L, _ = zxbcdt.shape
        d_ssm = H * P
        dev, adt = zxbcdt.device, zxbcdt.dtype
        use_norm = norm_w is not None
        z, xBC, dt = torch.split(zxbcdt, [d_ssm, d_ssm + 2 * G * N, H], dim=-1)
        dzxbcdt = torch.empty(zxbcdt.shape, dtype=adt, device=dev)   # always batch-major dense (empty_like would keep a permuted layout)
        dz, dxBC, ddt_v = torch.split(dzxbcdt, [d_ssm, d_ssm + 2 * G * N, H], dim=-1)
        # ---- out_proj: its wgrad needs the norm output (kept, or recomputed when OMK_RECOMPUTE=1)
        dout = dout.to(adt)
        d_outproj_w = d_outproj_b = None
        y2 = y_pre.reshape(Bsz, L, d_ssm)
        lib = get_lib()
        need_wo = outproj_w is not None and ctx.needs_input_grad[14]
        on = None
        if need_wo:
            if not use_norm:
                on = (y2.float() * F.silu(z.float())).to(adt)
            elif on_saved is not None:
                on = on_saved
            else:
                with torch.no_grad():
                    on = rmsnorm_fn(y2, norm_w, None, z=z, eps=eps, group_size=d_ssm // G, norm_before_gate=nbg)
        if outproj_w is not None:
            w_c = getattr(ctx, "w_cast", None)
            d_outn = dout @ (w_c if (w_c is not None and w_c.dtype == adt) else outproj_w.to(adt))
            if need_wo:
                do2, on2 = dout.reshape(-1, dout.shape[-1]), on.reshape(-1, d_ssm)
                d_outproj_w = weight_grad(do2 if do2.is_contiguous() else do2.contiguous(), on2 if on2.is_contiguous() else on2.contiguous(),
                                          outproj_w.dtype)
            if outproj_b is not None and ctx.needs_input_grad[15]:
                d_outproj_b = dout.reshape(-1, dout.shape[-1]).sum(0).to(outproj_b.dtype)
        else:
            d_outn = dout
        # ---- gated norm (or plain gate) backward: dz lands in the z slice of dzxbcdt, no copy
        d_norm_w = None
        if use_norm:
            rows = Bsz * L
            y_r = y2.reshape(rows, d_ssm)
            z_r = z.as_strided((rows, d_ssm), (zxbcdt.stride(1), 1), z.storage_offset()) if zxbcdt.stride(0) == L * zxbcdt.stride(1) else z.reshape(rows, d_ssm)
            dz_r = dz.as_strided((rows, d_ssm), (dzxbcdt.stride(1), 1), dz.storage_offset())
            g_r = d_outn.reshape(rows, d_ssm)
            if g_r.dtype != adt:
                g_r = g_r.to(adt)
            if g_r.stride(-1) != 1:
                g_r = g_r.contiguous()
            dy = torch.empty(Bsz, L, d_ssm, dtype=adt, device=dev)
            gw = torch.zeros(d_ssm, dtype=torch.float32, device=dev) if ctx.needs_input_grad[12] else None   # frozen norm weight: no reduction
            pn = K.NormGatedBwd(dy=K.T(g_r), x=K.T(y_r), z=K.T(z_r), weight=K.T(norm_w), dx=K.T(dy.reshape(rows, d_ssm)),
                                dz=K.T(dz_r), dweight=K.T(gw), group_size=d_ssm // G, eps=eps, norm_before_gate=int(nbg))
            wsn = K.workspace(lib, "omk_norm_gated_bwd_workspace_bytes", pn, g_r)  # noqa: F841
            K.run(lib, "omk_norm_gated_bwd", pn, g_r)
            d_norm_w = None if gw is None else gw.to(norm_w.dtype)
        else:
            zf = z.float()
            dz.copy_((d_outn.float() * y2.float() * _silu_grad(zf)).to(adt))
            dy = (d_outn.float() * F.silu(zf)).to(adt)
        # ---- conv output (kept or recomputed); the SSD backward writes straight into the dxBC_conv buffer
        if xBC_saved is not None:
            xBC_c = xBC_saved
        else:
            with torch.no_grad():
                xBC_c = causal_conv1d_fn(xBC.transpose(1, 2), conv_w, conv_b, activation=activation).transpose(1, 2)
        x, Bm, Cm = torch.split(xBC_c, [d_ssm, G * N, G * N], dim=-1)
        dxBC_c = torch.empty_like(xBC_c)
        dx_v, dB_v, dC_v = torch.split(dxBC_c, [d_ssm, G * N, G * N], dim=-1)
        g = ssd_scan_bwd(dy.reshape(Bsz, L, H, P), x.unflatten(-1, (H, P)), dt, A, Bm.unflatten(-1, (G, N)),
                         Cm.unflatten(-1, (G, N)), D=D, dt_bias=dt_bias, initial_states=initial_states,
                         dfinal_states=dfinal, dt_softplus=True, dt_limit=dt_limit, chunk_size=chunk_size,
                         need_dinit=initial_states is not None, y=y_pre, dx_out=dx_v.unflatten(-1, (H, P)),
                         dB_out=dB_v.unflatten(-1, (G, N)), dC_out=dC_v.unflatten(-1, (G, N)), window_states=wst,
                         flags=ctx.scan_flags)
        ddt_v.copy_(g["ddt"])
        # ---- conv backward: dx lands in the xBC slice of dzxbcdt
        dw = torch.zeros(conv_w.shape, dtype=torch.float32, device=dev)
        db = None if conv_b is None else torch.zeros(conv_b.shape, dtype=torch.float32, device=dev)
        p = K.Conv1dBwd(x=K.T(xBC.transpose(1, 2)), weight=K.T(conv_w), bias=K.T(conv_b), initial_states=K.T(None),
                        dout=K.T(dxBC_c.transpose(1, 2)), dx=K.T(dxBC.transpose(1, 2)), dweight=K.T(dw), dbias=K.T(db),
                        dinitial_states=K.T(None), silu=1)
        cws = K.workspace(lib, "omk_causal_conv1d_bwd_workspace_bytes", p, zxbcdt)   # partial dw / db rows instead of atomics
        K.run(lib, "omk_causal_conv1d_bwd", p, zxbcdt)
        dinit = g["dinitial_states"]
        return (dzxbcdt, dw.to(conv_w.dtype), None if conv_b is None else db.to(conv_b.dtype),
                g["ddt_bias"].to(dt_bias.dtype), g["dA"].to(A.dtype), g["dD"].to(D.dtype), None,
                None if dinit is None else dinit.to(initial_states.dtype), None, None, None, None, d_norm_w, None,
                d_outproj_w, d_outproj_b, None, None, None, None)


def mamba_split_conv1d_scan_combined(zxbcdt, conv1d_weight, conv1d_bias, dt_bias, A, D, chunk_size, initial_states=None,
                                     seq_idx=None, dt_limit=(0.0, _INF), return_final_states=False, activation="silu",
                                     rmsnorm_weight=None, rmsnorm_eps=1e-6, outproj_weight=None, outproj_bias=None,
                                     headdim=None, ngroups=1, norm_before_gate=True, conv_state_out=None):
    """zxbcdt: (batch, seqlen, 2 * dim + 2 * ngroups * dstate + nheads); conv1d_weight: (dim + 2 * ngroups * dstate,
    width); dt_bias, A: (nheads); D: (nheads, headdim) or (nheads,).  Returns out (batch, seqlen, d_model | dim)
    [, final_states (batch, nheads, headdim, dstate)].  conv_state_out (extension for the prefill of a cached decode): a
    (batch, dim + 2 * ngroups * dstate, state_len >= width - 1) buffer the conv kernel fills with the last pre-conv inputs."""
    return MambaSplitConv1dScanCombinedFn.apply(zxbcdt, conv1d_weight, conv1d_bias, dt_bias, A, D, chunk_size,
                                                initial_states, seq_idx, dt_limit, return_final_states, activation,
                                                rmsnorm_weight, rmsnorm_eps, outproj_weight, outproj_bias, headdim,
                                                ngroups, norm_before_gate, conv_state_out)

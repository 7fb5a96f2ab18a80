"""CPU restatement (plain PyTorch) of the operators on the OmniMamba Mamba-2 hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function computes in fp32 (or the
dtype given by ``compute_dtype``, fp64 for golden generation) from the recurrences themselves;
nothing here is tuned.  ``[UPSTREAM]`` marks the absent mamba_ssm==2.2.2 / causal-conv1d==1.4.0
entry point a function answers to; the citation after it is the reference call site that reaches it.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F


def softplus_ref(x: torch.Tensor) -> torch.Tensor:
    """log(1+exp(x)) with the PyTorch threshold (x > 20 -> x)."""
    return torch.where(x > 20.0, x, torch.log1p(torch.exp(torch.clamp(x, max=20.0))))


def _silu(x):
    return x * torch.sigmoid(x)


# --------------------------------------------------------------------------------------
# Mamba-1 selective scan   [UPSTREAM] mamba_ssm.ops.selective_scan_interface.selective_scan_ref
# reached from models/stage2/mixer_seq_simple.py:16,197-201 (ssm_cfg.layer == "Mamba1");
# BASELINE.json configs[0] names this signature.
# --------------------------------------------------------------------------------------
def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                       return_last_state=False, compute_dtype=torch.float32):
    """u, delta, z: (B, D, L); A: (D, N); B, C: (D, N) | (B, N, L) | (B, G, N, L); D, delta_bias: (D).

    x_t = exp(delta_t A) x_{t-1} + delta_t B_t u_t ;  y_t = <C_t, x_t> ;  out = (y + D u) * silu(z).
    Streams over L keeping only the (B, D, N) state (the upstream ref materialises (B, D, L, N)).
    """
    dtype_in = u.dtype
    cd = compute_dtype
    u_ = u.to(cd)
    delta_ = delta.to(cd)
    if delta_bias is not None:
        delta_ = delta_ + delta_bias.to(cd)[None, :, None]
    if delta_softplus:
        delta_ = softplus_ref(delta_)
    Bsz, Dm, L = u_.shape
    N = A.shape[1]
    A_ = A.to(cd)

    def expand_bc(M):
        if M.dim() == 2:  # (D, N) constant over time
            return None
        M = M.to(cd)
        if M.dim() == 3:  # (B, N, L) -> one group
            M = M[:, None]
        G = M.shape[1]
        assert Dm % G == 0
        return M.repeat_interleave(Dm // G, dim=1)  # (B, D, N, L)

    Bv, Cv = expand_bc(B), expand_bc(C)
    x = torch.zeros(Bsz, Dm, N, dtype=cd)
    ys = []
    for t in range(L):
        dA = torch.exp(delta_[:, :, t, None] * A_[None])                      # (B, D, N)
        Bt = B.to(cd)[None] if Bv is None else Bv[..., t]
        Ct = C.to(cd)[None] if Cv is None else Cv[..., t]
        x = dA * x + (delta_[:, :, t] * u_[:, :, t])[..., None] * Bt
        ys.append((x * Ct).sum(-1))
    y = torch.stack(ys, dim=2)
    out = y if D is None else y + u_ * D.to(cd)[None, :, None]
    if z is not None:
        out = out * _silu(z.to(cd))
    out = out.to(dtype_in)
    return (out, x) if return_last_state else out


# --------------------------------------------------------------------------------------
# causal depthwise conv1d  [UPSTREAM] causal_conv1d.causal_conv1d_fn / causal_conv1d_update
# reached from Mamba2.forward / Mamba2.step (mixer_seq_simple.py:17,200-205 -> block.py:117)
# --------------------------------------------------------------------------------------
def causal_conv1d_ref(x, weight, bias=None, initial_states=None, return_final_states=False,
                      activation=None, compute_dtype=torch.float32):
    """x: (B, C, L); weight: (C, W); bias: (C); initial_states: (B, C, W-1).

    out[b,c,l] = act(bias[c] + sum_k w[c,k] * xpad[b,c,l+k]),  xpad = [initial_states | x]
    (zeros when no initial state).  final_states = last W-1 columns of xpad.
    """
    assert activation in (None, "silu", "swish")
    dtype_in = x.dtype
    Bsz, Cc, L = x.shape
    W = weight.shape[1]
    xf = x.to(compute_dtype)
    if initial_states is None:
        pad = torch.zeros(Bsz, Cc, W - 1, dtype=compute_dtype)
    else:
        pad = initial_states.to(compute_dtype)
    xp = torch.cat([pad, xf], dim=-1)                                          # (B, C, L+W-1)
    out = torch.zeros(Bsz, Cc, L, dtype=compute_dtype)
    wf = weight.to(compute_dtype)
    for k in range(W):
        out = out + wf[None, :, k, None] * xp[:, :, k:k + L]
    if bias is not None:
        out = out + bias.to(compute_dtype)[None, :, None]
    if activation is not None:
        out = _silu(out)
    out = out.to(dtype_in)
    if return_final_states:
        return out, xp[:, :, L:].to(dtype_in)                                   # last W-1 columns
    return out


def causal_conv1d_update_ref(x, conv_state, weight, bias=None, activation=None,
                             compute_dtype=torch.float32):
    """x: (B, C) or (B, C, T); conv_state: (B, C, S) with S >= W-1, updated IN PLACE.

    Per new column: state <- [state[:,:,1:], x_t]; out_t = act(bias + sum_k w[:,k] * window_k) where the
    window is the last W entries of [old state | x_0..x_t].
    """
    assert activation in (None, "silu", "swish")
    squeeze = x.dim() == 2
    if squeeze:
        x = x[..., None]
    dtype_in = x.dtype
    W = weight.shape[1]
    S = conv_state.shape[-1]
    assert S >= W - 1
    full = torch.cat([conv_state.to(compute_dtype), x.to(compute_dtype)], dim=-1)   # (B, C, S+T)
    T = x.shape[-1]
    wf = weight.to(compute_dtype)
    out = torch.zeros(x.shape, dtype=compute_dtype)
    for t in range(T):
        end = S + t + 1
        win = full[:, :, end - W:end]
        out[:, :, t] = (win * wf[None]).sum(-1)
    if bias is not None:
        out = out + bias.to(compute_dtype)[None, :, None]
    if activation is not None:
        out = _silu(out)
    conv_state.copy_(full[:, :, -S:].to(conv_state.dtype))
    out = out.to(dtype_in)
    return out[..., 0] if squeeze else out


# --------------------------------------------------------------------------------------
# Mamba-2 SSD scan  [UPSTREAM] mamba_ssm.ops.triton.ssd_combined.mamba_chunk_scan_combined
# reached from Mamba2.forward (block.py:117); SURVEY.md Appendix A.1-3/4
# --------------------------------------------------------------------------------------
def _prep_dt(dt, dt_bias, dt_softplus, dt_limit, cd):
    dt_ = dt.to(cd)
    if dt_bias is not None:
        dt_ = dt_ + dt_bias.to(cd)
    if dt_softplus:
        dt_ = softplus_ref(dt_)
    lo, hi = dt_limit
    if lo != 0.0 or hi != float("inf"):
        dt_ = torch.clamp(dt_, min=lo, max=hi)
    return dt_


def ssd_ref_sequential(x, dt, A, B, C, D=None, z=None, dt_bias=None, initial_states=None,
                       dt_softplus=False, dt_limit=(0.0, float("inf")), return_final_states=False,
                       compute_dtype=torch.float32):
    """Naive recurrence.  x: (B, L, H, P); dt: (B, L, H); A: (H); B, C: (B, L, G, N);
    D: (H) or (H, P); z: (B, L, H, P); dt_bias: (H); initial_states: (B, H, P, N).

    s_t = exp(dt'_t A_h) s_{t-1} + dt'_t x_t (x) B_t ;  y_t = s_t . C_t + D x_t ;  y *= silu(z).
    """
    cd = compute_dtype
    Bsz, L, H, P = x.shape
    G, N = B.shape[2], B.shape[3]
    assert H % G == 0
    dt_ = _prep_dt(dt, dt_bias, dt_softplus, dt_limit, cd)
    xf = x.to(cd)
    Bf = B.to(cd).repeat_interleave(H // G, dim=2)                             # (B, L, H, N)
    Cf = C.to(cd).repeat_interleave(H // G, dim=2)
    Af = A.to(cd)
    s = torch.zeros(Bsz, H, P, N, dtype=cd) if initial_states is None else initial_states.to(cd).clone()
    ys = []
    for t in range(L):
        dA = torch.exp(dt_[:, t] * Af[None])                                   # (B, H)
        s = s * dA[..., None, None] + (dt_[:, t, :, None] * xf[:, t])[..., None] * Bf[:, t, :, None, :]
        ys.append((s * Cf[:, t, :, None, :]).sum(-1))                          # (B, H, P)
    y = torch.stack(ys, dim=1)
    if D is not None:
        Df = D.to(cd)
        y = y + xf * (Df[None, None, :, None] if Df.dim() == 1 else Df[None, None])
    if z is not None:
        y = y * _silu(z.to(cd))
    y = y.to(x.dtype)
    return (y, s) if return_final_states else y


def ssd_ref_chunked(x, dt, A, B, C, chunk_size, D=None, z=None, dt_bias=None, initial_states=None,
                    dt_softplus=False, dt_limit=(0.0, float("inf")), return_final_states=False,
                    compute_dtype=torch.float32, emulate_upstream_rounding=False, round_output=True):
    """Chunked (SSD) form of the same recurrence -- the fast CPU baseline.

    Per chunk (local l, a_l = dt'_l A, cs_l = sum_{j<=l} a_j):
      y_l   = exp(cs_l) C_l.s_in + sum_{s<=l} (C_l.B_s) exp(cs_l - cs_s) dt'_s x_s + D x_l
      s_out = exp(cs_Q-1) s_in + sum_l exp(cs_Q-1 - cs_l) dt'_l x_l (x) B_l
    Tail positions (>= L) contribute dt' = 0.

    emulate_upstream_rounding: with 16-bit activations the published mamba_ssm==2.2.2 pipeline
    ([UPSTREAM-RECALLED], absent from /root/reference) rounds three MFMA / tl.dot operands to the activation dtype:
      (i)   the decayed, dt-scaled, masked C.B^T tile ("cb.to(x dtype)" in _chunk_scan_fwd) before it multiplies x,
      (ii)  the scaled B rows exp(cs_end - cs_l) dt'_l B_l ("b.to(x dtype)" in _chunk_state_fwd) before the chunk-state product,
      (iii) the chunk-START states handed to the scan (_state_passing_fwd(..., out_dtype=C.dtype)) before C . s_in;
    the carried state itself, all accumulations and the decay factors stay fp32.  The flag reproduces exactly these three
    roundings (activation dtype = x.dtype) on top of `compute_dtype` arithmetic; False = no intermediate rounding.
    round_output=False returns y in compute_dtype (the arithmetic result before the final cast to x.dtype).
    """
    cd = compute_dtype
    adt = x.dtype
    rnd = (lambda t: t.to(adt).to(cd)) if (emulate_upstream_rounding and adt in (torch.bfloat16, torch.float16)) else (lambda t: t)
    Bsz, L, H, P = x.shape
    G, N = B.shape[2], B.shape[3]
    Q = chunk_size
    nC = (L + Q - 1) // Q
    pad = nC * Q - L
    dt_ = _prep_dt(dt, dt_bias, dt_softplus, dt_limit, cd)

    def padL(t):
        return F.pad(t, (0, 0) * (t.dim() - 2) + (0, pad)) if pad else t

    xf = padL(x.to(cd)).reshape(Bsz, nC, Q, H, P)
    dtf = padL(dt_).reshape(Bsz, nC, Q, H)
    Bf = padL(B.to(cd)).reshape(Bsz, nC, Q, G, N)
    Cf = padL(C.to(cd)).reshape(Bsz, nC, Q, G, N)
    a = dtf * A.to(cd)[None, None, None]                                       # (B, nC, Q, H)
    cs = torch.cumsum(a, dim=2)
    rep = H // G
    # intra-chunk
    CB = torch.einsum("bclgn,bcsgn->bcgls", Cf, Bf).repeat_interleave(rep, dim=2)      # (B,nC,H,Q,Q)
    seg = cs.permute(0, 1, 3, 2)[..., :, None] - cs.permute(0, 1, 3, 2)[..., None, :]   # cs_l - cs_s
    mask = torch.tril(torch.ones(Q, Q, dtype=torch.bool))
    decay = torch.where(mask, torch.exp(torch.where(mask, seg, torch.zeros_like(seg))), torch.zeros_like(seg))
    M = rnd(CB * decay * dtf.permute(0, 1, 3, 2)[..., None, :])                       # (i)
    y = torch.einsum("bchls,bcshp->bclhp", M, xf)
    # chunk states
    w = torch.exp(cs[:, :, -1:, :] - cs) * dtf                                 # (B,nC,Q,H)
    Bh = Bf.repeat_interleave(rep, dim=3)
    Ch = Cf.repeat_interleave(rep, dim=3)
    S_loc = torch.einsum("bclhp,bclhn->bchpn", xf, rnd(w[..., None] * Bh))            # (ii)
    s = torch.zeros(Bsz, H, P, N, dtype=cd) if initial_states is None else initial_states.to(cd).clone()
    for c in range(nC):
        y[:, c] = y[:, c] + torch.exp(cs[:, c])[..., None] * torch.einsum("blhn,bhpn->blhp", Ch[:, c], rnd(s))   # (iii)
        s = torch.exp(cs[:, c, -1])[..., None, None] * s + S_loc[:, c]
    y = y.reshape(Bsz, nC * Q, H, P)[:, :L]
    xo = x.to(cd)
    if D is not None:
        Df = D.to(cd)
        y = y + xo * (Df[None, None, :, None] if Df.dim() == 1 else Df[None, None])
    if z is not None:
        y = y * _silu(z.to(cd))
    if round_output:
        y = y.to(x.dtype)
    return (y, s) if return_final_states else y


# --------------------------------------------------------------------------------------
# single-token state update  [UPSTREAM] mamba_ssm.ops.triton.selective_state_update
# reached from Mamba2.step <- models/stage2/generation.py:195-211,412-424
# --------------------------------------------------------------------------------------
def selective_state_update_ref(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False,
                               compute_dtype=torch.float32):
    """state: (B, H, P, N) [or (B, dim, N)] updated IN PLACE; x, dt, z: (B, H, P); A: (H, P, N);
    B, C: (B, G, N); D, dt_bias: (H, P).   s <- s exp(dt' A) + dt' x (x) B ;  y = s.C + D x ; y *= silu(z)."""
    cd = compute_dtype
    has_heads = state.dim() > 3
    if not has_heads:
        state_v, x, dt, A = state[:, None], x[:, None], dt[:, None], A[None]
        B, C = B[:, None], C[:, None]
        D = None if D is None else D[None]
        z = None if z is None else z[:, None]
        dt_bias = None if dt_bias is None else dt_bias[None]
    else:
        state_v = state
    Bsz, H, P, N = state_v.shape
    G = B.shape[1]
    dt_ = dt.to(cd)
    if dt_bias is not None:
        dt_ = dt_ + dt_bias.to(cd)
    if dt_softplus:
        dt_ = softplus_ref(dt_)
    dA = torch.exp(dt_[..., None] * A.to(cd))                                  # (B,H,P,N)
    Bh = B.to(cd).repeat_interleave(H // G, dim=1)                             # (B,H,N)
    Ch = C.to(cd).repeat_interleave(H // G, dim=1)
    new = state_v.to(cd) * dA + (dt_ * x.to(cd))[..., None] * Bh[:, :, None, :]
    state_v.copy_(new.to(state_v.dtype))
    y = (new * Ch[:, :, None, :]).sum(-1)
    if D is not None:
        y = y + x.to(cd) * D.to(cd)
    if z is not None:
        y = y * _silu(z.to(cd))
    y = y.to(x.dtype)
    return y if has_heads else y[:, 0]


# --------------------------------------------------------------------------------------
# norms  [UPSTREAM] mamba_ssm.ops.triton.layernorm_gated.rmsnorm_fn (Mamba2.norm) and
# mamba_ssm.ops.triton.layer_norm.layer_norm_fn (models/stage2/block.py:10,86-95;
# mixer_seq_simple.py:30,428-437)
# --------------------------------------------------------------------------------------
def rmsnorm_gated_ref(x, weight, bias=None, z=None, eps=1e-6, group_size=None, norm_before_gate=True,
                      is_rms_norm=True, compute_dtype=torch.float32):
    """norm_before_gate=False (the reference's setting): out = norm(x * silu(z)) * w (+b), per group of
    ``group_size`` lanes; True: out = norm(x) * w * silu(z)."""
    cd = compute_dtype
    dtype_in = x.dtype
    Nn = x.shape[-1]
    gs = Nn if group_size is None else group_size
    xf = x.to(cd)
    zf = None if z is None else z.to(cd)
    if zf is not None and not norm_before_gate:
        xf = xf * _silu(zf)
    xg = xf.reshape(*xf.shape[:-1], Nn // gs, gs)
    if is_rms_norm:
        rstd = torch.rsqrt(xg.pow(2).mean(-1, keepdim=True) + eps)
        xn = xg * rstd
    else:
        mu = xg.mean(-1, keepdim=True)
        rstd = torch.rsqrt((xg - mu).pow(2).mean(-1, keepdim=True) + eps)
        xn = (xg - mu) * rstd
    out = xn.reshape(xf.shape) * weight.to(cd)
    if bias is not None:
        out = out + bias.to(cd)
    if zf is not None and norm_before_gate:
        out = out * _silu(zf)
    return out.to(dtype_in)


def add_norm_ref(x, weight, bias=None, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                 is_rms_norm=False, compute_dtype=torch.float32):
    """Fused residual-add + (RMS|Layer)Norm.  r = x + residual (fp32); residual_out = r in fp32 when
    residual_in_fp32 or the incoming residual is fp32, else in x.dtype; y = norm(r) * w (+b) in x.dtype."""
    cd = compute_dtype
    r = x.to(cd)
    if residual is not None:
        r = r + residual.to(cd)
    if residual_in_fp32 or (residual is not None and residual.dtype == torch.float32):
        res_dtype = torch.float32
    else:
        res_dtype = x.dtype if residual is None else residual.dtype
    res_out = r.to(res_dtype)
    # the normalised value is computed from the (possibly rounded) stored residual, like the fused kernel's
    # single pass over r held in fp32 registers: upstream normalises the fp32 sum, so do we.
    if is_rms_norm:
        y = r * torch.rsqrt(r.pow(2).mean(-1, keepdim=True) + eps)
    else:
        mu = r.mean(-1, keepdim=True)
        y = (r - mu) * torch.rsqrt((r - mu).pow(2).mean(-1, keepdim=True) + eps)
    y = y * weight.to(cd)
    if bias is not None:
        y = y + bias.to(cd)
    y = y.to(x.dtype)
    return (y, res_out) if prenorm else y


# --------------------------------------------------------------------------------------
# fused block op  [UPSTREAM] mamba_ssm.ops.triton.ssd_combined.mamba_split_conv1d_scan_combined
# (training path of Mamba2.forward)
# --------------------------------------------------------------------------------------
def mamba_split_conv1d_scan_combined_ref(zxbcdt, conv1d_weight, conv1d_bias, dt_bias, A, D, chunk_size,
                                         initial_states=None, dt_limit=(0.0, float("inf")),
                                         return_final_states=False, activation="silu",
                                         rmsnorm_weight=None, rmsnorm_eps=1e-6, outproj_weight=None,
                                         outproj_bias=None, headdim=None, ngroups=1, norm_before_gate=True,
                                         compute_dtype=torch.float32, round_intermediates=False):
    """zxbcdt: (B, L, 2*d_ssm + 2*G*N + H) split as [z | xBC | dt].  ``round_intermediates`` emulates the
    upstream rounding points (conv output and pre-norm y stored in the activation dtype)."""
    cd = compute_dtype
    Bsz, L, _ = zxbcdt.shape
    if D.dim() == 1:
        assert headdim is not None
        H = D.shape[0]
    else:
        H, headdim = D.shape
    P = headdim
    d_ssm = H * P
    N = (conv1d_weight.shape[0] - d_ssm) // (2 * ngroups)
    assert zxbcdt.shape[-1] == 2 * d_ssm + 2 * ngroups * N + H
    act_dtype = zxbcdt.dtype
    z, xBC, dt = torch.split(zxbcdt, [d_ssm, d_ssm + 2 * ngroups * N, H], dim=-1)
    xBC_c = causal_conv1d_ref(xBC.transpose(1, 2).to(cd), conv1d_weight, conv1d_bias,
                              activation=activation, compute_dtype=cd).transpose(1, 2)
    if round_intermediates:
        xBC_c = xBC_c.to(act_dtype).to(cd)
    x, Bm, Cm = torch.split(xBC_c, [d_ssm, ngroups * N, ngroups * N], dim=-1)
    x = x.reshape(Bsz, L, H, P)
    Bm = Bm.reshape(Bsz, L, ngroups, N)
    Cm = Cm.reshape(Bsz, L, ngroups, N)
    zr = z.reshape(Bsz, L, H, P).to(cd)
    res = ssd_ref_chunked(x, dt.to(cd), A, Bm, Cm, chunk_size, D=D, z=zr if rmsnorm_weight is None else None,
                          dt_bias=dt_bias, initial_states=initial_states, dt_softplus=True, dt_limit=dt_limit,
                          return_final_states=return_final_states, compute_dtype=cd)
    y, final = (res if return_final_states else (res, None))
    y = y.reshape(Bsz, L, d_ssm)
    if round_intermediates:
        y = y.to(act_dtype).to(cd)
    if rmsnorm_weight is not None:
        y = rmsnorm_gated_ref(y, rmsnorm_weight, None, z=z.to(cd), eps=rmsnorm_eps,
                              group_size=d_ssm // ngroups, norm_before_gate=norm_before_gate, compute_dtype=cd)
    if round_intermediates:
        y = y.to(act_dtype).to(cd)
    if outproj_weight is not None:
        y = F.linear(y, outproj_weight.to(cd), None if outproj_bias is None else outproj_bias.to(cd))
    y = y.to(act_dtype)
    return (y, final) if return_final_states else y


# --------------------------------------------------------------------------------------
# whole Mamba-2 mixer  [UPSTREAM] mamba_ssm.modules.mamba2.Mamba2 (forward / step), constructed at
# models/stage2/mixer_seq_simple.py:200-205 with all-default hyper-parameters, called at block.py:117
# --------------------------------------------------------------------------------------
@dataclass
class Mamba2RefParams:
    in_proj_weight: torch.Tensor      # (d_in_proj, d_model)
    conv_weight: torch.Tensor         # (conv_dim, W)
    conv_bias: Optional[torch.Tensor]
    dt_bias: torch.Tensor             # (H)
    A_log: torch.Tensor               # (H)
    D: torch.Tensor                   # (H)
    norm_weight: torch.Tensor         # (d_ssm)
    out_proj_weight: torch.Tensor     # (d_model, d_inner)
    headdim: int = 64
    d_state: int = 128
    ngroups: int = 1
    chunk_size: int = 256
    norm_eps: float = 1e-5
    norm_before_gate: bool = False

    @staticmethod
    def random(d_model, headdim=64, d_state=128, ngroups=1, d_conv=4, expand=2, chunk_size=256, seed=0,
               dtype=torch.float32):
        """Same init distributions as the upstream module's defaults (A~U(1,16), dt log-uniform[1e-3,0.1])."""
        g = torch.Generator().manual_seed(seed)
        d_inner = expand * d_model
        H = d_inner // headdim
        conv_dim = d_inner + 2 * ngroups * d_state
        d_in_proj = 2 * d_inner + 2 * ngroups * d_state + H
        k_in = 1.0 / math.sqrt(d_model)
        k_out = 1.0 / math.sqrt(d_inner)
        k_conv = 1.0 / math.sqrt(d_conv)
        u = lambda *s: torch.rand(*s, generator=g)
        dt = torch.exp(u(H) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3)).clamp(min=1e-4)
        return Mamba2RefParams(
            in_proj_weight=((u(d_in_proj, d_model) * 2 - 1) * k_in).to(dtype),
            conv_weight=((u(conv_dim, d_conv) * 2 - 1) * k_conv).to(dtype),
            conv_bias=((u(conv_dim) * 2 - 1) * k_conv).to(dtype),
            dt_bias=(dt + torch.log(-torch.expm1(-dt))).to(dtype),
            A_log=torch.log(1.0 + 15.0 * u(H)).to(dtype),
            D=torch.ones(H, dtype=dtype),
            norm_weight=torch.ones(d_inner, dtype=dtype),
            out_proj_weight=((u(d_model, d_inner) * 2 - 1) * k_out).to(dtype),
            headdim=headdim, d_state=d_state, ngroups=ngroups, chunk_size=chunk_size)


def mamba2_forward_ref(p: Mamba2RefParams, u, conv_state=None, ssm_state=None, compute_dtype=torch.float32):
    """u: (B, L, d_model) -> (B, L, d_model).  When the two state tensors are given they are fully
    overwritten (prefill-with-cache semantics): conv_state (B, conv_dim, W) gets the
    last W pre-conv xBC columns (left zero padded), ssm_state (B, H, P, N) the final SSM state."""
    cd = compute_dtype
    Bsz, L, _ = u.shape
    H = p.A_log.shape[0]
    P, N, G = p.headdim, p.d_state, p.ngroups
    d_ssm = H * P
    zxbcdt = F.linear(u.to(cd), p.in_proj_weight.to(cd))
    z, xBC, dt = torch.split(zxbcdt, [d_ssm, d_ssm + 2 * G * N, H], dim=-1)
    W = p.conv_weight.shape[1]
    if conv_state is not None:
        xt = xBC.transpose(1, 2)
        conv_state.copy_(F.pad(xt, (W - L, 0)).to(conv_state.dtype) if L < W else xt[:, :, -W:].to(conv_state.dtype))
    xc = causal_conv1d_ref(xBC.transpose(1, 2), p.conv_weight, p.conv_bias, activation="silu",
                           compute_dtype=cd).transpose(1, 2)
    x, Bm, Cm = torch.split(xc, [d_ssm, G * N, G * N], dim=-1)
    A = -torch.exp(p.A_log.to(cd))
    y, final = ssd_ref_chunked(x.reshape(Bsz, L, H, P), dt, A, Bm.reshape(Bsz, L, G, N), Cm.reshape(Bsz, L, G, N),
                               p.chunk_size, D=p.D, dt_bias=p.dt_bias, dt_softplus=True,
                               return_final_states=True, compute_dtype=cd)
    if ssm_state is not None:
        ssm_state.copy_(final.to(ssm_state.dtype))
    y = rmsnorm_gated_ref(y.reshape(Bsz, L, d_ssm), p.norm_weight, None, z=z, eps=p.norm_eps,
                          group_size=d_ssm // G, norm_before_gate=p.norm_before_gate, compute_dtype=cd)
    return F.linear(y, p.out_proj_weight.to(cd)).to(u.dtype)


def mamba2_step_ref(p: Mamba2RefParams, u, conv_state, ssm_state, compute_dtype=torch.float32):
    """One decode step.  u: (B, 1, d_model); both states updated in place."""
    cd = compute_dtype
    Bsz = u.shape[0]
    H = p.A_log.shape[0]
    P, N, G = p.headdim, p.d_state, p.ngroups
    d_ssm = H * P
    zxbcdt = F.linear(u[:, 0].to(cd), p.in_proj_weight.to(cd))
    z, xBC, dt = torch.split(zxbcdt, [d_ssm, d_ssm + 2 * G * N, H], dim=-1)
    xc = causal_conv1d_update_ref(xBC, conv_state, p.conv_weight, p.conv_bias, activation="silu", compute_dtype=cd)
    x, Bm, Cm = torch.split(xc, [d_ssm, G * N, G * N], dim=-1)
    A = -torch.exp(p.A_log.to(cd))
    y = selective_state_update_ref(
        ssm_state, x.reshape(Bsz, H, P), dt[:, :, None].expand(Bsz, H, P), A[:, None, None].expand(H, P, N),
        Bm.reshape(Bsz, G, N), Cm.reshape(Bsz, G, N), D=p.D[:, None].expand(H, P),
        dt_bias=p.dt_bias[:, None].expand(H, P), dt_softplus=True, compute_dtype=cd)
    y = rmsnorm_gated_ref(y.reshape(Bsz, d_ssm), p.norm_weight, None, z=z, eps=p.norm_eps,
                          group_size=d_ssm // G, norm_before_gate=p.norm_before_gate, compute_dtype=cd)
    return F.linear(y, p.out_proj_weight.to(cd)).to(u.dtype)[:, None]

"""Dense projections of the block (in_proj, out_proj): plain library GEMMs through torch, with one MI355X-specific
choice -- how the WEIGHT gradient is formed.

dW = dy^T x contracts over the tokens (K = batch * seqlen = 32 768 at the benchmark shape) into a small (out, in) matrix.
For in_proj that is 8512 x 2048 = 34 x 8 = 272 output tiles of 256 x 256 on a 256-CU device: one full round plus a
16-tile tail, i.e. half the chip idles for half the GEMM (measured 1.45-1.5 ms = 0.76 PFLOP/s with every hipBLASLt /
rocBLAS solution TunableOp could find).  Splitting the token dimension into S slices and running ONE batched GEMM
(S x tiles work items, fp32 partial outputs) followed by an fp32 sum fills the rounds: 1.15 ms at S = 4.  The fp32
partials make the result more accurate than the single bf16-output GEMM it replaces (the weights are fp32 masters).

`linear(x, weight, bias)` is `F.linear` with that backward; it is what `Mamba2` uses for a plain `nn.Linear` in_proj and
what `TaskLoRALinear` uses for its base weight.  Anything unusual (no GPU, fp32 autocast-off training, tiny token counts)
takes the ordinary matmul.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

_TILE = 256


def _n_cu(device) -> int:
    try:
        return torch.cuda.get_device_properties(device).multi_processor_count
    except Exception:
        return 256


def split_factor(m: int, n: int, k: int, device) -> int:
    """Slices of the contraction dimension for a (m x n) output: the smallest S in {2, 4, 8} whose S * tiles work items
    fill at least 80 % of the CU rounds, when the unsplit GEMM fills less than 75 %."""
    if os.environ.get("OMK_WGRAD_SPLITK", "1") == "0":
        return 1
    cu = _n_cu(device)
    tiles = -(-m // _TILE) * -(-n // _TILE)

    def eff(s):
        items = s * tiles
        return items / (-(-items // cu) * cu)

    if eff(1) >= 0.75:
        return 1
    best = 1
    # a handful of tiles (the rank-8 LoRA factors: 8 tiles for A) may be cut much finer: slices down to 512 tokens
    cands, kmin = ((2, 4, 8, 16, 32), 512) if tiles <= 16 else ((2, 4, 8), 2048)
    for s in cands:
        if k % s or k // s < kmin:
            break
        best = s
        if eff(s) >= 0.8:
            break
    return best


def weight_grad(dy2d: torch.Tensor, x2d: torch.Tensor, out_dtype: torch.dtype) -> torch.Tensor:
    """dW (out, in) = dy2d^T (tokens, out) @ x2d (tokens, in), accumulated in fp32."""
    k, m = dy2d.shape
    n = x2d.shape[1]
    if (dy2d.is_cuda and dy2d.dtype in (torch.bfloat16, torch.float16) and dy2d.is_contiguous() and x2d.is_contiguous()):
        s = split_factor(m, n, k, dy2d.device)
        if s > 1:
            part = torch.bmm(dy2d.view(s, k // s, m).transpose(1, 2), x2d.view(s, k // s, n), out_dtype=torch.float32)
            return part.sum(0).to(out_dtype)
    return (dy2d.t() @ x2d).to(out_dtype)


class _XGradFn(torch.autograd.Function):
    """y = F.linear(x, W, b); its backward only forms dx."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        if torch.is_autocast_enabled():
            adt = torch.get_autocast_dtype("cuda")
            x, weight = x.to(adt), weight.to(adt)
            bias = None if bias is None else bias.to(adt)
        ctx.save_for_backward(weight)
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        (weight,) = ctx.saved_tensors
        return (dy @ weight.to(dy.dtype)) if ctx.needs_input_grad[0] else None, None, None


class _WGradFn(torch.autograd.Function):
    """Identity on y that owns the weight / bias gradients.  It sits BEHIND the GEMM node in the graph, so backward reaches
    it first: dW is formed, the weight's AccumulateGrad (highest priority in the autograd engine) runs, and under DDP the
    bucket's all-reduce is in flight while the dx GEMM of `_XGradFn` computes.  For the first layer of a stack -- or the
    single block of bench.py -- this is the only overlap the last (largest) weight gradient can get."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, y, x, weight, bias):
        ctx.wdtype, ctx.bdtype = weight.dtype, None if bias is None else bias.dtype
        if torch.is_autocast_enabled():
            x = x.to(torch.get_autocast_dtype("cuda"))
        ctx.save_for_backward(x)
        return y.view_as(y)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dw = db = None
        dy2 = dy.reshape(-1, dy.shape[-1])
        if ctx.needs_input_grad[2]:
            dy2c = dy2 if dy2.is_contiguous() else dy2.contiguous()
            x2 = x.reshape(-1, x.shape[-1])
            dw = weight_grad(dy2c, x2 if x2.is_contiguous() else x2.contiguous(), ctx.wdtype)
        if ctx.bdtype is not None and ctx.needs_input_grad[3]:
            db = dy2.sum(0).to(ctx.bdtype)
        return dy, None, dw, db


def frozen_cast(weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """`weight.to(dtype)` kept on the parameter object while its version stands: under autocast an fp32 master was cast again on
    every call -- 20 us and 105 MB of traffic per in_proj call.  A frozen weight (every base projection in the 'align' stage) is
    cast once; a training one once per optimizer step instead of once per task forward (the version counter moves with the step).
    The key is (storage address, autograd version): every in-place update torch itself makes (optimizers, `copy_`,
    `load_state_dict`) moves the version.  In-place writes through `weight.data` do NOT -- code that updates parameters that way
    must run with OMK_CAST_CACHE=0 (which also makes `lora_ext` rebuild its extended weight on every call)."""
    if os.environ.get("OMK_CAST_CACHE", "1") == "0":
        return weight.detach().to(dtype)
    c = getattr(weight, "_omk_cast", None)
    try:
        key = (weight.data_ptr(), weight._version)
    except RuntimeError:   # inference tensors (a model built under torch.inference_mode) carry no version counter: no cache
        return weight.detach().to(dtype)
    if c is not None and c[0] == key and c[1].dtype == dtype and c[1].device == weight.device:
        return c[1]
    t = weight.detach().to(dtype)
    weight._omk_cast = (key, t)
    return t


def linear(x, weight, bias=None):
    """F.linear whose weight gradient is formed with `weight_grad` (same forward arithmetic), as two autograd nodes so that
    the weight gradient is ready -- and its all-reduce started -- before the input gradient is computed."""
    if not x.is_cuda:
        return F.linear(x, weight, bias)
    # the GEMM node sees detached parameters: an edge from it to the weight would make the weight's AccumulateGrad wait
    # for the dx GEMM as well
    w_in = weight.detach()
    if torch.is_autocast_enabled() and weight.dtype != torch.get_autocast_dtype("cuda") and weight.numel() >= (1 << 20):
        w_in = frozen_cast(weight, torch.get_autocast_dtype("cuda"))
    y = _XGradFn.apply(x, w_in, None if bias is None else bias.detach())
    if torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad)):
        y = _WGradFn.apply(y, x.detach(), weight, bias)
    return y

"""Decode-step projections fused with the normalisation in front of them (omk_norm_linear, csrc/norm_linear.hip).

At one token per sequence the reference runs `layer_norm_fn` (block.py:86-95), the task LoRA `Linear` (lora.py:185-279:
base GEMV, A GEMV, B GEMV, scale, add) and later `RMSNormGated` + `out_proj` (upstream Mamba2.step) as separate launches
of a few microseconds each; here each group is ONE kernel that streams the weight matrix once.  Inference only (no
autograd); one to eight sequences per call.  Callers fall back to the unfused ops when `applies()` says no.
"""
from __future__ import annotations

import torch

from . import _capi as K
from ._lib import get_lib, require_device

MAX_BATCH = 8     # sequences per call; larger decode batches take the separate ops


def _uniform(weight, *others) -> bool:
    """The templated kernels' dtype rule: fp32 or bf16 weights, every other tensor of the same dtype."""
    return weight.dtype in (torch.float32, torch.bfloat16) and all(t is None or t.dtype == weight.dtype for t in others)


def applies(x: torch.Tensor, weight: torch.Tensor, norm_weight=None, *same_dtype) -> bool:
    """Fused path preconditions (shape / dtype / no autograd).  One sequence: any supported dtype mix.  Two to eight
    sequences: the uniform-dtype kernel only -- pass the norm weight and every tensor that must share the weight's dtype
    (gate, LoRA factors, bias)."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad):
        return False
    if x.dim() != 2 or x.shape[0] > MAX_BATCH or x.shape[0] == 0:
        return False
    vec = 4 if weight.dtype == torch.float32 else 8
    ok = (weight.dim() == 2 and weight.stride(1) == 1 and weight.shape[1] % 1024 == 0 and weight.shape[1] <= 8192
          and weight.stride(0) % vec == 0 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)
    if x.shape[0] == 1:
        return ok and x.shape[1] * 4 <= 96 * 1024
    nb = 2 if x.shape[0] <= 2 else (4 if x.shape[0] <= 4 else 8)
    return (ok and norm_weight is not None and _uniform(weight, x, norm_weight, *same_dtype) and weight.shape[1] in (1024, 2048, 4096)
            and nb * weight.shape[1] * weight.element_size() <= 144 * 1024 and weight.shape[0] <= 64 * 1024)


def conv_tail_applies(x, weight, norm_weight, conv_state, conv_weight, conv_bias, lora_a=None, bias=None, residual=None) -> bool:
    """Whether `norm_linear(..., conv_state=...)` is served (the uniform-dtype kernel: fp32 or bf16 everywhere)."""
    dt = weight.dtype
    same = lambda t: t is None or t.dtype == dt
    W, S = conv_weight.shape[-1], conv_state.shape[-1]
    return (dt in (torch.float32, torch.bfloat16) and x.dtype == dt and norm_weight is not None and same(norm_weight) and same(lora_a)
            and same(bias) and same(conv_bias) and conv_state.dtype == dt and conv_weight.dtype == dt
            and (residual is None or residual.dtype in (torch.float32, dt)) and weight.shape[1] in (1024, 2048, 4096)
            and 2 <= W <= 4 and W - 1 <= S <= 4 and (lora_a is None or lora_a.shape[0] <= 8))


def norm_linear(x, weight, bias=None, *, norm_weight=None, eps=1e-5, residual=None, residual_out_dtype=None, z=None,
                group_size=None, norm_before_gate=False, lora_a=None, lora_b=None, lora_scale=0.0, out_dtype=None,
                conv_state=None, conv_weight=None, conv_bias=None, conv_offset=0, conv_silu=True):
    """out = norm(x [+ residual] | gated by z) @ weight^T [+ bias] [+ lora_scale * (n @ lora_a^T) @ lora_b^T].
    x: (B, in).  Returns out, or (out, residual_out) when `residual_out_dtype` is given (residual_out = x + residual).
    conv_state (B, C, S) + conv_weight (C, W): output columns [conv_offset, conv_offset + C) additionally go through
    causal_conv1d_update (+ SiLU): out holds the convolved values and conv_state is rolled in place."""
    lib = get_lib()
    require_device(lib, x, weight, bias, norm_weight, residual, z, lora_a, lora_b, conv_state, conv_weight, conv_bias)
    if x.stride(-1) != 1:
        x = x.contiguous()
    if z is not None and (z.dtype != x.dtype or z.stride(-1) != 1):
        z = z.to(x.dtype).contiguous()
    B = x.shape[0]
    out = torch.empty(B, weight.shape[0], dtype=out_dtype or x.dtype, device=x.device)
    ro = None if residual_out_dtype is None else torch.empty(B, x.shape[1], dtype=residual_out_dtype, device=x.device)
    p = K.NormLinear(x=K.T(x), residual=K.T(residual), z=K.T(z), norm_weight=K.T(norm_weight), weight=K.T(weight), bias=K.T(bias),
                     lora_a=K.T(lora_a), lora_b=K.T(lora_b), residual_out=K.T(ro), out=K.T(out),
                     conv_state=K.T(conv_state), conv_weight=K.T(conv_weight), conv_bias=K.T(conv_bias),
                     group_size=0 if group_size is None else int(group_size), conv_offset=int(conv_offset), eps=float(eps),
                     lora_scale=float(lora_scale), norm_before_gate=int(bool(norm_before_gate)), conv_silu=int(bool(conv_silu)))
    K.run(lib, "omk_norm_linear", p, x)
    return out if ro is None else (out, ro)

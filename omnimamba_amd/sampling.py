"""On-device token sampling: omk_sample behind the reference's `sample` signature
(/root/reference/models/stage2/generation.py:87-121).  One launch, no host scalar after it: usable inside a captured decode step."""
from __future__ import annotations

from typing import Optional

import torch

from . import _capi as K
from ._lib import get_lib

MAX_TOP_K = 64


def applies(logits: torch.Tensor, top_k: int, min_p: float = 0.0, top_p: float = 0.0) -> bool:
    """The HIP sampler covers the reference's top_k == 1 short cut, its top_k > 0 branch up to 64 candidates, and its whole-vocabulary
    branch (top_k == 0): the plain multinomial (t2i_generate's default arguments), the top-p cut over all tokens and the min_p filter."""
    try:
        lib = get_lib()
    except RuntimeError:
        return False
    on_lib_device = logits.is_cuda != bool(lib.omk_is_emulated())
    full = top_k == 0 and 0.0 <= min_p < 1.0 and top_p <= 1.0
    return on_lib_device and (1 <= top_k <= MAX_TOP_K or full) and logits.dim() == 2 and logits.stride(1) == 1 and logits.dtype in (torch.float32, torch.bfloat16, torch.float16)


def sample_device(logits: torch.Tensor, top_k: int = 1, top_p: float = 0.0, temperature: float = 1.0, seed: int = 0,
                  step_counter: Optional[torch.Tensor] = None, offset: int = 0, out: Optional[torch.Tensor] = None, min_p: float = 0.0) -> torch.Tensor:
    """(batch, vocab) -> (batch,) int64 ids.  seed / step_counter / offset select the Philox stream: the same triple gives the same
    ids (row b uses its own stream); step_counter is a device int64 scalar tensor read by the kernel (advance it yourself)."""
    lib = get_lib()
    if out is None:
        out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    if step_counter is not None:
        assert step_counter.dtype == torch.int64 and step_counter.device == logits.device
    p = K.Sample(logits=K.T(logits), top_k=int(top_k), top_p=float(top_p), temperature=float(temperature), min_p=float(min_p) if top_k == 0 else 0.0, seed=int(seed) & (2 ** 64 - 1),
                 offset=int(offset), step_counter=None if step_counter is None else step_counter.data_ptr())
    p.out_ids.data = out.data_ptr()
    p.out_ids.ndim = 1
    p.out_ids.shape[0] = out.shape[0]
    p.out_ids.stride[0] = out.stride(0)
    K.run(lib, "omk_sample", p, logits)
    return out

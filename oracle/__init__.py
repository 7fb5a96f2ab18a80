"""CPU oracle for the OmniMamba Mamba-2 hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch (fp32/fp64, CPU) restatement of the arithmetic that the
reference delegates to the absent third-party packages ``mamba_ssm==2.2.2`` and
``causal-conv1d==1.4.0`` (/root/reference/requirements.txt:12-13).  Reference call sites
the restatement answers to: models/stage2/mixer_seq_simple.py:15-20,30,200-205 and
models/stage2/block.py:10,86-95,117.

PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden vectors or
fixtures for this path, and the upstream packages cannot be built or imported here
(nvcc + Triton, not vendored, no network).  The oracle is therefore pinned by
  (1) first-principles self-consistency (naive recurrence == chunked form == step form),
  (2) the independent pure-PyTorch restatement of the same upstream ops that ships in the
      installed ``transformers`` (models/mamba2/modeling_mamba2.py, models/mamba/modeling_mamba.py),
  (3) outputs of the reference's own importable Python (models/stage2/lora.py, block.py)
      captured by tests/golden/make_golden.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package, and only as the checker.  The product path (``omnimamba_amd``) never
imports it and fails loudly when the HIP library is missing.
"""
from .ops import (  # noqa: F401
    softplus_ref,
    selective_scan_ref,
    causal_conv1d_ref,
    causal_conv1d_update_ref,
    ssd_ref_sequential,
    ssd_ref_chunked,
    selective_state_update_ref,
    rmsnorm_gated_ref,
    add_norm_ref,
    mamba_split_conv1d_scan_combined_ref,
    Mamba2RefParams,
    mamba2_forward_ref,
    mamba2_step_ref,
)

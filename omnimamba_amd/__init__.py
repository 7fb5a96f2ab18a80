"""omnimamba_amd -- MI355X (gfx950) native kernels behind OmniMamba's Mamba-2 selective-scan hot path.

Host side mirrors the operator API the reference imports from mamba_ssm / causal_conv1d
(/root/reference/models/stage2/mixer_seq_simple.py:15-20,30 ; block.py:10); all arithmetic happens in
hand-written HIP kernels behind the C ABI declared in include/omk.h.
"""
__version__ = "0.1.0"

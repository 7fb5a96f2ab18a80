"""Drop-in check against the reference's OWN Python (only where /root/reference exists, i.e. the build container; skipped on
the GPU box): its models/stage2/block.py::Block imports `mamba_ssm.ops.triton.layer_norm` -- resolved by this repo's facade --
and must run unchanged on our Mamba2 / RMSNorm / layer_norm_fn and agree with our own ResidualBlock."""
import importlib.util
import os
import sys

import pytest
import torch

REF = "/root/reference/models/stage2/block.py"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference sources are only present in the build container")
def test_reference_block_runs_on_our_operators():
    from functools import partial
    from emu.loader import use_emulator
    spec = importlib.util.spec_from_file_location("ref_block", REF)
    ref_block = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_block)                       # executes `from mamba_ssm.ops.triton.layer_norm import ...`
    from mamba_ssm.modules.mamba2 import Mamba2
    from mamba_ssm.ops.triton.layer_norm import RMSNorm
    from omnimamba_amd.stack import ResidualBlock, StackConfig
    with use_emulator():
        torch.manual_seed(0)
        mixer_cls = partial(Mamba2, layer_idx=0, d_state=16, headdim=8, chunk_size=16)
        blk = ref_block.Block(32, mixer_cls, torch.nn.Identity, norm_cls=partial(RMSNorm, eps=1e-5), fused_add_norm=True,
                              residual_in_fp32=True)
        ours = ResidualBlock(32, 0, StackConfig(d_model=32, n_layer=1, ssm_cfg=dict(d_state=16, headdim=8, chunk_size=16)))
        ours.load_state_dict(blk.state_dict())
        h, res = torch.randn(2, 9, 32), torch.randn(2, 9, 32)
        y1, r1 = blk(h, res)
        y2, r2 = ours(h, res)
        assert r1.dtype == torch.float32 and torch.equal(r1, r2) and torch.allclose(y1, y2, atol=1e-6)
        with torch.autocast("cpu", dtype=torch.bfloat16):   # training runs under bf16 AMP (train_stage2.py:21,37)
            y1, r1 = blk(h.bfloat16(), None)                # first block: residual=None, bf16 activations, fp32 residual out
        assert r1.dtype == torch.float32 and y1.dtype == torch.bfloat16


@pytest.mark.skipif(not os.path.exists(REF), reason="reference sources are only present in the build container")
def test_reference_block_with_a_gated_mlp():
    """d_intermediate > 0 (no shipped config, but create_block builds it: mixer_seq_simple.py:168-230): the reference's Block with this
    repo's GatedMLP as mlp_cls -- second fused add + norm, then the MLP -- against the same composition written out."""
    from functools import partial
    import torch.nn.functional as F
    from emu.loader import use_emulator
    spec = importlib.util.spec_from_file_location("ref_block2", REF)
    ref_block = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_block)
    from mamba_ssm.modules.mamba2 import Mamba2
    from mamba_ssm.modules.mlp import GatedMLP
    from mamba_ssm.ops.triton.layer_norm import RMSNorm
    import oracle as O
    with use_emulator():
        torch.manual_seed(1)
        mlp = GatedMLP(32, hidden_features=40, out_features=32, multiple_of=16)
        assert mlp.fc1.weight.shape == (96, 32) and mlp.fc2.weight.shape == (32, 48)      # 40 -> 48 = next multiple of 16
        blk = ref_block.Block(32, partial(Mamba2, layer_idx=0, d_state=16, headdim=8, chunk_size=16),
                              partial(GatedMLP, hidden_features=40, out_features=32, multiple_of=16), norm_cls=partial(RMSNorm, eps=1e-5),
                              fused_add_norm=True, residual_in_fp32=True)
        h, res = torch.randn(2, 9, 32), torch.randn(2, 9, 32)
        y, r = blk(h, res)
        # written out: mixer half, then residual + norm2 + MLP
        hn, r1 = O.add_norm_ref(h, blk.norm.weight.detach(), None, residual=res, eps=1e-5, prenorm=True, residual_in_fp32=True, is_rms_norm=True)
        m = blk.mixer(hn)
        hn2, r2 = O.add_norm_ref(m, blk.norm2.weight.detach(), None, residual=r1, eps=1e-5, prenorm=True, residual_in_fp32=True, is_rms_norm=True)
        a, g = F.linear(hn2, blk.mlp.fc1.weight).chunk(2, dim=-1)
        want = F.linear(a * F.silu(g), blk.mlp.fc2.weight)
        assert torch.allclose(r, r2, atol=1e-6) and torch.allclose(y, want, atol=1e-5)

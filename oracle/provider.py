"""TEST INFRASTRUCTURE ONLY: an in-memory ``mamba_ssm`` / ``causal_conv1d`` provider backed by the CPU oracle.

The reference's own Python (/root/reference/models/stage2/{block,lora,mixer_seq_simple,generation}.py) imports seven
names from the absent third-party packages (mixer_seq_simple.py:15-20,30; block.py:10).  ``install()`` registers modules
of those names in ``sys.modules`` whose arithmetic is ``oracle.ops`` (plain PyTorch, fp32), so that the reference's
classes can be instantiated and RUN in the build container by tests/golden/make_golden.py WITHOUT touching any kernel
under test.  The numbers it produces are committed as fixtures; the HIP path must reproduce them.

Never imported by ``omnimamba_amd`` (the product has its own facade packages ``mamba_ssm/`` and ``causal_conv1d/`` at
the repo root); ``install()`` shadows that facade only inside the generating process.
"""
from __future__ import annotations

import math
import sys
import types
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops as O


class RMSNorm(nn.Module):
    """[UPSTREAM] mamba_ssm.ops.triton.layer_norm.RMSNorm: weight only, eps, bias=None."""

    def __init__(self, hidden_size, eps=1e-5, dropout_p=0.0, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.drop = None
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return O.add_norm_ref(x, self.weight, None, residual=residual, eps=self.eps, prenorm=prenorm,
                              residual_in_fp32=residual_in_fp32, is_rms_norm=True)


def layer_norm_fn(x, weight, bias, residual=None, x1=None, weight1=None, bias1=None, eps=1e-6, dropout_p=0.0,
                  rowscale=None, prenorm=False, residual_in_fp32=False, is_rms_norm=False, return_dropout_mask=False):
    assert x1 is None and weight1 is None and rowscale is None and dropout_p == 0.0
    return O.add_norm_ref(x, weight, bias, residual=residual, eps=eps, prenorm=prenorm, residual_in_fp32=residual_in_fp32,
                          is_rms_norm=is_rms_norm)


def rms_norm_fn(x, weight, bias, residual=None, x1=None, weight1=None, bias1=None, eps=1e-6, dropout_p=0.0, rowscale=None,
                prenorm=False, residual_in_fp32=False, return_dropout_mask=False):
    return layer_norm_fn(x, weight, bias, residual=residual, eps=eps, prenorm=prenorm, residual_in_fp32=residual_in_fp32,
                         is_rms_norm=True)


class _GatedNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-5, group_size=None, norm_before_gate=True, device=None, dtype=None):
        super().__init__()
        self.eps, self.group_size, self.norm_before_gate = eps, group_size, norm_before_gate
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)


class Mamba2(nn.Module):
    """[UPSTREAM] mamba_ssm.modules.mamba2.Mamba2 restated on the oracle ops: same
    constructor arguments, parameter names and methods; forward / prefill-with-cache / step."""

    def __init__(self, d_model, d_state=128, d_conv=4, conv_init=None, expand=2, headdim=64, d_ssm=None, ngroups=1,
                 A_init_range=(1, 16), D_has_hdim=False, rmsnorm=True, norm_before_gate=False, dt_min=0.001, dt_max=0.1,
                 dt_init_floor=1e-4, dt_limit=(0.0, float("inf")), bias=False, conv_bias=True, chunk_size=256,
                 use_mem_eff_path=True, layer_idx=None, process_group=None, sequence_parallel=True, device=None, dtype=None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        assert process_group is None and d_ssm is None and not D_has_hdim and rmsnorm
        self.d_model, self.d_state, self.d_conv, self.expand, self.headdim = d_model, d_state, d_conv, expand, headdim
        self.d_inner = expand * d_model
        self.d_ssm = self.d_inner
        self.ngroups, self.nheads = ngroups, self.d_inner // headdim
        self.norm_before_gate, self.dt_limit, self.chunk_size, self.layer_idx = norm_before_gate, dt_limit, chunk_size, layer_idx
        self.activation = "silu"
        d_in_proj = 2 * self.d_inner + 2 * ngroups * d_state + self.nheads
        self.in_proj = nn.Linear(d_model, d_in_proj, bias=bias, **fk)
        conv_dim = self.d_ssm + 2 * ngroups * d_state
        self.conv1d = nn.Conv1d(conv_dim, conv_dim, bias=conv_bias, kernel_size=d_conv, groups=conv_dim, padding=d_conv - 1, **fk)
        dt = torch.exp(torch.rand(self.nheads, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
        self.dt_bias = nn.Parameter(dt + torch.log(-torch.expm1(-dt)))
        self.dt_bias._no_weight_decay = True
        A = torch.empty(self.nheads, dtype=torch.float32, device=device).uniform_(*A_init_range)
        self.A_log = nn.Parameter(torch.log(A).to(dtype=dtype))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.nheads, device=device))
        self.D._no_weight_decay = True
        self.norm = _GatedNorm(self.d_ssm, eps=1e-5, norm_before_gate=norm_before_gate, group_size=self.d_ssm // ngroups, **fk)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)

    def _split(self, zxbcdt):
        return torch.split(zxbcdt, [self.d_ssm, self.d_ssm + 2 * self.ngroups * self.d_state, self.nheads], dim=-1)

    def _gnorm(self, y, z):
        return O.rmsnorm_gated_ref(y, self.norm.weight, None, z=z, eps=self.norm.eps, group_size=self.norm.group_size,
                                   norm_before_gate=self.norm.norm_before_gate)

    def forward(self, u, seqlen=None, seq_idx=None, cu_seqlens=None, inference_params=None):
        assert seqlen is None and seq_idx is None and cu_seqlens is None
        Bsz, L, _ = u.shape
        conv_state = ssm_state = None
        if inference_params is not None:
            conv_state, ssm_state = self._get_states_from_cache(inference_params, Bsz)
            if inference_params.seqlen_offset > 0:
                out, _, _ = self.step(u, conv_state, ssm_state)
                return out
        H, P, N, G = self.nheads, self.headdim, self.d_state, self.ngroups
        zxbcdt = self.in_proj(u)                                     # the reference's LoRA Linear once swapped in
        z, xBC, dt = self._split(zxbcdt)
        W = self.d_conv
        if conv_state is not None:
            xt = xBC.transpose(1, 2)
            conv_state.copy_(F.pad(xt, (W - L, 0)) if L < W else xt[:, :, -W:])
        xc = O.causal_conv1d_ref(xBC.transpose(1, 2), self.conv1d.weight.squeeze(1), self.conv1d.bias,
                                 activation="silu").transpose(1, 2)
        x, Bm, Cm = torch.split(xc, [self.d_ssm, G * N, G * N], dim=-1)
        A = -torch.exp(self.A_log.float())
        y, final = O.ssd_ref_chunked(x.reshape(Bsz, L, H, P), dt, A, Bm.reshape(Bsz, L, G, N), Cm.reshape(Bsz, L, G, N),
                                     self.chunk_size, D=self.D, dt_bias=self.dt_bias, dt_softplus=True, dt_limit=self.dt_limit,
                                     return_final_states=True)
        if ssm_state is not None:
            ssm_state.copy_(final)
        return self.out_proj(self._gnorm(y.reshape(Bsz, L, self.d_ssm), z).to(u.dtype))

    def step(self, hidden_states, conv_state, ssm_state):
        assert hidden_states.shape[1] == 1
        Bsz = hidden_states.shape[0]
        H, P, N, G = self.nheads, self.headdim, self.d_state, self.ngroups
        z, xBC, dt = self._split(self.in_proj(hidden_states.squeeze(1)))
        xc = O.causal_conv1d_update_ref(xBC, conv_state, self.conv1d.weight.squeeze(1), self.conv1d.bias, activation="silu")
        x, Bm, Cm = torch.split(xc, [self.d_ssm, G * N, G * N], dim=-1)
        A = -torch.exp(self.A_log.float())
        y = O.selective_state_update_ref(ssm_state, x.reshape(Bsz, H, P), dt[:, :, None].expand(Bsz, H, P),
                                         A[:, None, None].expand(H, P, N), Bm.reshape(Bsz, G, N), Cm.reshape(Bsz, G, N),
                                         D=self.D[:, None].expand(H, P), dt_bias=self.dt_bias[:, None].expand(H, P), dt_softplus=True)
        out = self.out_proj(self._gnorm(y.reshape(Bsz, self.d_ssm), z).to(hidden_states.dtype))
        return out.unsqueeze(1), conv_state, ssm_state

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        dev = self.out_proj.weight.device
        conv_state = torch.zeros(batch_size, self.conv1d.weight.shape[0], self.d_conv, device=dev,
                                 dtype=self.conv1d.weight.dtype if dtype is None else dtype)
        ssm_state = torch.zeros(batch_size, self.nheads, self.headdim, self.d_state, device=dev,
                                dtype=self.in_proj.weight.dtype if dtype is None else dtype)
        return conv_state, ssm_state

    def _get_states_from_cache(self, inference_params, batch_size, initialize_states=False):
        if self.layer_idx not in inference_params.key_value_memory_dict:
            inference_params.key_value_memory_dict[self.layer_idx] = self.allocate_inference_cache(batch_size, 0)
        conv_state, ssm_state = inference_params.key_value_memory_dict[self.layer_idx]
        if initialize_states:
            conv_state.zero_()
            ssm_state.zero_()
        return conv_state, ssm_state


class _NotBuilt(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("never constructed by the shipped OmniMamba configs (config_mamba.py:7,16-17)")


@dataclass
class MambaConfig:
    d_model: int = 2560
    d_intermediate: int = 0
    n_layer: int = 64
    vocab_size: int = 50277
    ssm_cfg: dict = field(default_factory=dict)
    attn_layer_idx: list = field(default_factory=list)
    attn_cfg: dict = field(default_factory=dict)
    rms_norm: bool = True
    residual_in_fp32: bool = True
    fused_add_norm: bool = True
    pad_vocab_size_multiple: int = 8
    tie_embeddings: bool = True


def install():
    """Register the provider under the import paths the reference uses.  Returns the names it replaced."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__path__ = []
        sys.modules[name] = m
        return m

    def _nohub(*a, **k):
        raise RuntimeError("no network / hub access in the build container")

    mod("mamba_ssm")
    mod("mamba_ssm.models")
    mod("mamba_ssm.models.config_mamba", MambaConfig=MambaConfig)
    mod("mamba_ssm.modules")
    mod("mamba_ssm.modules.mamba2", Mamba2=Mamba2)
    mod("mamba_ssm.modules.mamba_simple", Mamba=_NotBuilt)
    mod("mamba_ssm.modules.mha", MHA=_NotBuilt)
    mod("mamba_ssm.modules.mlp", GatedMLP=_NotBuilt)
    mod("mamba_ssm.utils")
    mod("mamba_ssm.utils.hf", load_config_hf=_nohub, load_state_dict_hf=_nohub)
    mod("mamba_ssm.ops")
    mod("mamba_ssm.ops.triton")
    mod("mamba_ssm.ops.triton.layer_norm", RMSNorm=RMSNorm, layer_norm_fn=layer_norm_fn, rms_norm_fn=rms_norm_fn)

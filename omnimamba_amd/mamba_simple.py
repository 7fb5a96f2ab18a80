"""Mamba-1 mixer on the MI355X: same constructor, parameters, state-dict keys and methods as
``mamba_ssm.modules.mamba_simple.Mamba`` (mamba_ssm==2.2.2), which the reference imports and builds when
``ssm_cfg.layer == 'Mamba1'`` (/root/reference/models/stage2/mixer_seq_simple.py:16,196-205; not selected by the shipped
configs, config_mamba.py:16, but its scan is the op BASELINE.json configs[0] names).

  forward(h)                 no cache : in_proj -> mamba_inner_fn (conv1d + SiLU, x_proj, dt_proj, selective scan, out_proj)
  forward(h, ip, offset 0)   prefill  : the same ops unfused; conv_state / ssm_state fully overwritten
  step(h, conv, ssm)         decode   : causal_conv1d_update + selective_state_update (un-headed form), in place
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .causal_conv1d import causal_conv1d_fn, causal_conv1d_update
from .selective_scan import mamba_inner_fn, selective_scan_fn
from .selective_state_update import selective_state_update


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1, dt_init="random",
                 dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True, layer_idx=None, device=None,
                 dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path, self.layer_idx = use_fast_path, layer_idx
        self.in_proj = nn.Linear(d_model, self.d_inner * 2, bias=bias, **fk)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv, groups=self.d_inner,
                                padding=d_conv - 1, **fk)
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + d_state * 2, bias=False, **fk)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)
        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(self.d_inner, **fk) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
        with torch.no_grad():
            self.dt_proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))     # softplus^-1
        self.dt_proj.bias._no_reinit = True                                 # the reference's _init_weights keeps it (mixer_seq_simple.py:239-241)
        A = torch.arange(1, d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1).contiguous()
        self.A_log = nn.Parameter(torch.log(A))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **fk)

    def forward(self, hidden_states, inference_params=None):
        """hidden_states: (batch, seqlen, d_model) -> same shape."""
        batch, seqlen, _ = hidden_states.shape
        conv_state, ssm_state = None, None
        if inference_params is not None:
            conv_state, ssm_state = self._get_states_from_cache(inference_params, batch)
            if inference_params.seqlen_offset > 0:
                out, _, _ = self.step(hidden_states, conv_state, ssm_state)
                return out
        xz = self.in_proj(hidden_states).transpose(1, 2)                    # (batch, 2 d_inner, seqlen), seqlen-strided view
        A = -torch.exp(self.A_log.float())
        if self.use_fast_path and inference_params is None:
            return mamba_inner_fn(xz, self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight,
                                  self.out_proj.weight, self.out_proj.bias, A, None, None, self.D.float(),
                                  delta_bias=self.dt_proj.bias.float(), delta_softplus=True)
        x, z = xz.chunk(2, dim=1)
        if conv_state is not None:
            conv_state.copy_(F.pad(x, (self.d_conv - x.shape[-1], 0)))      # last d_conv inputs, left zero padded
        x = causal_conv1d_fn(x, self.conv1d.weight.squeeze(1), self.conv1d.bias, activation=self.activation)
        x_dbl = self.x_proj(x.transpose(1, 2).reshape(-1, self.d_inner))
        dt, B, C = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        # channel-last like x and z (views of token-major GEMM outputs, no copies): with enough sequences the scan reads them as they
        # lie (selscan.hip, lanes = channels); otherwise selective_scan_fn makes the L-contiguous copies the chunked scan wants
        dt = F.linear(dt, self.dt_proj.weight).view(batch, seqlen, self.d_inner).transpose(1, 2)
        B = B.reshape(batch, seqlen, self.d_state).transpose(1, 2)
        C = C.reshape(batch, seqlen, self.d_state).transpose(1, 2)
        y = selective_scan_fn(x, dt, A, B, C, self.D.float(), z=z, delta_bias=self.dt_proj.bias.float(), delta_softplus=True,
                              return_last_state=ssm_state is not None)
        if ssm_state is not None:
            y, last_state = y
            ssm_state.copy_(last_state)
        return self.out_proj(y.transpose(1, 2))

    def step(self, hidden_states, conv_state, ssm_state):
        assert hidden_states.shape[1] == 1, "Only support decoding with 1 token at a time for now"
        xz = self.in_proj(hidden_states.squeeze(1))
        x, z = xz.chunk(2, dim=-1)
        x = causal_conv1d_update(x, conv_state, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.activation)
        x_db = self.x_proj(x)
        dt, B, C = torch.split(x_db, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = F.linear(dt, self.dt_proj.weight)                              # the bias goes in with the softplus below
        A = -torch.exp(self.A_log.float())
        y = selective_state_update(ssm_state, x, dt, A, B, C, self.D, z=z, dt_bias=self.dt_proj.bias, dt_softplus=True)
        return self.out_proj(y).unsqueeze(1), conv_state, ssm_state

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        device = self.out_proj.weight.device
        conv_state = torch.zeros(batch_size, self.d_model * self.expand, self.d_conv, device=device,
                                 dtype=self.conv1d.weight.dtype if dtype is None else dtype)
        ssm_state = torch.zeros(batch_size, self.d_model * self.expand, self.d_state, device=device,
                                dtype=self.dt_proj.weight.dtype if dtype is None else dtype)
        return conv_state, ssm_state

    def _get_states_from_cache(self, inference_params, batch_size, initialize_states=False):
        assert self.layer_idx is not None
        if self.layer_idx not in inference_params.key_value_memory_dict:
            inference_params.key_value_memory_dict[self.layer_idx] = self.allocate_inference_cache(batch_size, 0)
        conv_state, ssm_state = inference_params.key_value_memory_dict[self.layer_idx]
        if initialize_states:
            conv_state.zero_()
            ssm_state.zero_()
        return conv_state, ssm_state

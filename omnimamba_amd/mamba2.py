"""Mamba2 mixer on the MI355X: same constructor, parameters, state-dict keys and methods as
``mamba_ssm.modules.mamba2.Mamba2`` (mamba_ssm==2.2.2), the only mixer the reference instantiates
(/root/reference/models/stage2/mixer_seq_simple.py:17,196-205 with ssm_cfg={'layer': 'Mamba2'},
models/stage2/config_mamba.py:16; called at models/stage2/block.py:117,149-150).

  forward(u)            training / no cache : in_proj (hipBLASLt) -> fused conv1d+SSD+gated-norm+out_proj node
  forward(u, ip, off=0) prefill with cache  : conv_state / ssm_state are fully overwritten
  step(u, conv, ssm)    decode              : causal_conv1d_update + selective_state_update, in place
``in_proj`` stays an nn.Linear attribute invoked through __call__ because the reference swaps it for its task-switched
LoRA Linear (models/stage2/lora.py:90-106) and sets ``.task_types`` on it (mixer_seq_simple.py:368-371).
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .causal_conv1d import causal_conv1d_fn, causal_conv1d_update
from .layernorm_gated import RMSNorm as RMSNormGated
from . import norm_linear as NL
from .linear import linear
from .selective_state_update import selective_state_update
from .ssd_combined import mamba_chunk_scan_combined, mamba_split_conv1d_scan_combined


class Mamba2(nn.Module):
    def __init__(self, d_model, d_state=128, d_conv=4, conv_init=None, expand=2, headdim=64, d_ssm=None, ngroups=1,
                 A_init_range=(1, 16), D_has_hdim=False, rmsnorm=True, norm_before_gate=False, dt_min=0.001,
                 dt_max=0.1, dt_init_floor=1e-4, dt_limit=(0.0, float("inf")), bias=False, conv_bias=True,
                 chunk_size=256, use_mem_eff_path=True, layer_idx=None, process_group=None, sequence_parallel=True,
                 device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        if process_group is not None:
            raise NotImplementedError("tensor/sequence parallel Mamba2 is not on the OmniMamba path "
                                      "(the reference never passes process_group)")
        self.d_model = d_model
        self.d_state = d_state
        self.d_conv = d_conv
        self.conv_init = conv_init
        self.expand = expand
        self.process_group = None
        self.sequence_parallel = sequence_parallel
        self.world_size = 1
        self.local_rank = 0
        self.d_inner = self.expand * self.d_model
        self.headdim = headdim
        self.d_ssm = self.d_inner if d_ssm is None else d_ssm
        assert ngroups >= 1 and self.d_ssm % self.headdim == 0
        self.ngroups = ngroups
        self.nheads = self.d_ssm // self.headdim
        self.D_has_hdim = D_has_hdim
        self.rmsnorm = rmsnorm
        self.norm_before_gate = norm_before_gate
        self.dt_limit = dt_limit
        self.activation = "silu"
        self.chunk_size = chunk_size
        self.use_mem_eff_path = use_mem_eff_path
        self.layer_idx = layer_idx

        # order: [z, x, B, C, dt]
        d_in_proj = 2 * self.d_inner + 2 * self.ngroups * self.d_state + self.nheads
        self.in_proj = nn.Linear(self.d_model, d_in_proj, bias=bias, **factory_kwargs)
        conv_dim = self.d_ssm + 2 * self.ngroups * self.d_state
        self.conv1d = nn.Conv1d(in_channels=conv_dim, out_channels=conv_dim, bias=conv_bias, kernel_size=d_conv,
                                groups=conv_dim, padding=d_conv - 1, **factory_kwargs)
        if self.conv_init is not None:
            nn.init.uniform_(self.conv1d.weight, -self.conv_init, self.conv_init)
        self.act = nn.SiLU()

        # dt bias: softplus^-1 of a log-uniform dt in [dt_min, dt_max]
        dt = torch.exp(torch.rand(self.nheads, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min))
        dt = torch.clamp(dt, min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        self.dt_bias = nn.Parameter(inv_dt)
        self.dt_bias._no_weight_decay = True

        assert A_init_range[0] > 0 and A_init_range[1] >= A_init_range[0]
        A = torch.empty(self.nheads, dtype=torch.float32, device=device).uniform_(*A_init_range)
        self.A_log = nn.Parameter(torch.log(A).to(dtype=dtype))
        self.A_log._no_weight_decay = True

        self.D = nn.Parameter(torch.ones(self.d_ssm if self.D_has_hdim else self.nheads, device=device))
        self.D._no_weight_decay = True

        if self.rmsnorm:
            self.norm = RMSNormGated(self.d_ssm, eps=1e-5, norm_before_gate=self.norm_before_gate,
                                     group_size=self.d_ssm // ngroups, **factory_kwargs)
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **factory_kwargs)

    def _D(self):
        return self.D.view(self.nheads, self.headdim) if self.D_has_hdim else self.D

    def forward(self, u, seqlen=None, seq_idx=None, cu_seqlens=None, inference_params=None, cp_group=None):
        """u: (batch, seqlen, hidden_dim) -> same shape.  cp_group (extension): a torch.distributed group
        over which the SEQUENCE is cut -- u is this rank's shard (batch, seqlen / world, hidden_dim); the conv1d takes a 3-token halo
        from the left neighbour and the scan one boundary-state exchange (omnimamba_amd/context_parallel.py)."""
        if seq_idx is not None or cu_seqlens is not None:
            raise NotImplementedError("seq_idx / cu_seqlens never reach the mixer in OmniMamba")
        if cp_group is not None:
            if inference_params is not None or seqlen is not None:
                raise NotImplementedError("context-parallel forward: training / prefill-free path only")
            return self._forward_context_parallel(u, cp_group)
        seqlen_og = seqlen
        if seqlen is None:
            batch, seqlen, _ = u.shape
        else:
            batch_seqlen, _ = u.shape
            batch = batch_seqlen // seqlen

        conv_state, ssm_state = None, None
        if inference_params is not None:
            conv_state, ssm_state = self._get_states_from_cache(inference_params, batch)
            if inference_params.seqlen_offset > 0:
                out, _, _ = self.step(u, conv_state, ssm_state)
                return out
            if not torch.is_grad_enabled():
                self._A_inference()   # prefill: bring the step's persistent -exp(A_log) buffer up to date (graph replays read it)

        # a plain nn.Linear in_proj goes through omnimamba_amd.linear (same forward GEMM, token-split weight gradient);
        # any replacement module (the reference's LoRA wrapper) is simply called
        zxbcdt = linear(u, self.in_proj.weight, self.in_proj.bias) if type(self.in_proj) is nn.Linear else self.in_proj(u)
        if seqlen_og is not None:
            zxbcdt = zxbcdt.view(batch, seqlen, -1)
        # (inference: the persistent buffer the decode step reads -- two elementwise launches per layer less in a prefill)
        A = self._A_inference() if (inference_params is not None and not torch.is_grad_enabled()) else -torch.exp(self.A_log.float())
        dt_limit_kwargs = {} if self.dt_limit == (0.0, float("inf")) else dict(dt_limit=self.dt_limit)
        d_mlp = (zxbcdt.shape[-1] - 2 * self.d_ssm - 2 * self.ngroups * self.d_state - self.nheads) // 2
        # prefill of a cached decode takes the same fused node: conv1d + SiLU with the conv_state fill in
        # its epilogue -> scan with the final state -> gated norm -> out_proj, instead of the five separate ops of upstream's branch
        fused_prefill = (inference_params is not None and conv_state is not None and not torch.is_grad_enabled()
                         and conv_state.dtype == zxbcdt.dtype and os.environ.get("OMK_FUSED_PREFILL", "1") != "0")
        if self.use_mem_eff_path and d_mlp == 0 and fused_prefill:
            out, last_state = mamba_split_conv1d_scan_combined(
                zxbcdt, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.dt_bias, A, D=self._D(),
                chunk_size=self.chunk_size, activation=self.activation, return_final_states=True,
                rmsnorm_weight=self.norm.weight if self.rmsnorm else None,
                rmsnorm_eps=self.norm.eps if self.rmsnorm else 1e-6, outproj_weight=self.out_proj.weight,
                outproj_bias=self.out_proj.bias, headdim=None if self.D_has_hdim else self.headdim,
                ngroups=self.ngroups, norm_before_gate=self.norm_before_gate, conv_state_out=conv_state, **dt_limit_kwargs)
            ssm_state.copy_(last_state)
            if seqlen_og is not None:
                out = out.reshape(batch * seqlen, -1)
            return out
        if self.use_mem_eff_path and inference_params is None and d_mlp == 0:
            out = mamba_split_conv1d_scan_combined(
                zxbcdt, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.dt_bias, A, D=self._D(),
                chunk_size=self.chunk_size, seq_idx=seq_idx, activation=self.activation,
                rmsnorm_weight=self.norm.weight if self.rmsnorm else None,
                rmsnorm_eps=self.norm.eps if self.rmsnorm else 1e-6, outproj_weight=self.out_proj.weight,
                outproj_bias=self.out_proj.bias, headdim=None if self.D_has_hdim else self.headdim,
                ngroups=self.ngroups, norm_before_gate=self.norm_before_gate, **dt_limit_kwargs)
            if seqlen_og is not None:
                out = out.reshape(batch * seqlen, -1)
            return out

        z0, x0, z, xBC, dt = torch.split(
            zxbcdt, [d_mlp, d_mlp, self.d_ssm, self.d_ssm + 2 * self.ngroups * self.d_state, self.nheads], dim=-1)
        if conv_state is not None:
            # conv_state <- last d_conv columns of the pre-conv xBC (left zero padded): fully overwritten
            xBC_t = xBC.transpose(1, 2)
            conv_state.copy_(F.pad(xBC_t, (self.d_conv - xBC_t.shape[-1], 0)))
        xBC = causal_conv1d_fn(xBC.transpose(1, 2), self.conv1d.weight.squeeze(1), self.conv1d.bias,
                               activation=self.activation).transpose(1, 2)
        x, B, C = torch.split(xBC, [self.d_ssm, self.ngroups * self.d_state, self.ngroups * self.d_state], dim=-1)
        y = mamba_chunk_scan_combined(
            x.unflatten(-1, (self.nheads, self.headdim)), dt, A, B.unflatten(-1, (self.ngroups, self.d_state)),
            C.unflatten(-1, (self.ngroups, self.d_state)), chunk_size=self.chunk_size, D=self._D(),
            z=z.unflatten(-1, (self.nheads, self.headdim)) if not self.rmsnorm else None, dt_bias=self.dt_bias,
            dt_softplus=True, return_final_states=ssm_state is not None, **dt_limit_kwargs)
        if ssm_state is not None:
            y, last_state = y
            ssm_state.copy_(last_state)
        y = y.flatten(-2)
        if self.rmsnorm:
            y = self.norm(y, z)
        if d_mlp > 0:
            y = torch.cat([F.silu(z0) * x0, y], dim=-1)
        if seqlen_og is not None:
            y = y.reshape(batch * seqlen, -1)
        return self.out_proj(y)

    def _forward_context_parallel(self, u, group):
        from . import context_parallel as CP
        zxbcdt = linear(u, self.in_proj.weight, self.in_proj.bias) if type(self.in_proj) is nn.Linear else self.in_proj(u)
        A = -torch.exp(self.A_log.float())
        d_mlp = (zxbcdt.shape[-1] - 2 * self.d_ssm - 2 * self.ngroups * self.d_state - self.nheads) // 2
        if d_mlp != 0:
            raise NotImplementedError("context-parallel forward: d_ssm < d_inner is not on the OmniMamba path")
        z, xBC, dt = torch.split(zxbcdt, [self.d_ssm, self.d_ssm + 2 * self.ngroups * self.d_state, self.nheads], dim=-1)
        xBC_t = xBC.transpose(1, 2)
        halo = CP.conv1d_halo(xBC_t, self.d_conv, group)
        xBC = causal_conv1d_fn(xBC_t, self.conv1d.weight.squeeze(1), self.conv1d.bias, initial_states=halo,
                               activation=self.activation).transpose(1, 2)
        x, B, C = torch.split(xBC, [self.d_ssm, self.ngroups * self.d_state, self.ngroups * self.d_state], dim=-1)
        y = CP.mamba_chunk_scan_context_parallel(
            x.unflatten(-1, (self.nheads, self.headdim)), dt, A, B.unflatten(-1, (self.ngroups, self.d_state)),
            C.unflatten(-1, (self.ngroups, self.d_state)), self.chunk_size, D=self._D(),
            z=z.unflatten(-1, (self.nheads, self.headdim)) if not self.rmsnorm else None, dt_bias=self.dt_bias,
            dt_softplus=True, dt_limit=self.dt_limit, group=group)
        y = y.flatten(-2)
        if self.rmsnorm:
            y = self.norm(y, z)
        return self.out_proj(y)

    def _A_inference(self):
        """-exp(A_log) for the decode step.  ONE persistent buffer per module, refreshed IN PLACE when A_log's version
        moved: a captured hipGraph of the step bakes the buffer's address, so an in-place refresh (every prefill passes
        through here, see forward) keeps replays valid after optimizer steps / load_state_dict into the same model --
        the reference recomputes A inside the graph (generation.py:372-434 captures the whole step).  While a capture is
        in progress nothing is allocated or refreshed: a missing or stale buffer is replaced by an in-graph recompute."""
        if torch.is_grad_enabled() and self.A_log.requires_grad:
            return -torch.exp(self.A_log.float())
        buf, ver = getattr(self, "_A_buf", None), getattr(self, "_A_ver", None)
        fresh = buf is not None and ver == self.A_log._version and buf.device == self.A_log.device
        if fresh:
            return buf
        if self.A_log.is_cuda and torch.cuda.is_current_stream_capturing():
            return -torch.exp(self.A_log.float())
        with torch.no_grad():
            if buf is None or buf.device != self.A_log.device or buf.shape != self.A_log.shape:
                with torch.inference_mode(False):   # a normal tensor: later refreshes may come from outside inference_mode
                    self._A_buf = buf = torch.empty(self.A_log.shape, dtype=torch.float32, device=self.A_log.device)
            buf.copy_(-torch.exp(self.A_log.float()))
        self._A_ver = self.A_log._version
        return buf

    def step(self, hidden_states, conv_state, ssm_state):
        """hidden_states: (batch, 1, d_model); both states updated in place. -> (out (batch, 1, d_model), conv, ssm)"""
        assert hidden_states.shape[1] == 1, "Only support decoding with 1 token at a time for now"
        zxbcdt = self.in_proj(hidden_states.squeeze(1))
        return self.step_from_zxbcdt(zxbcdt, conv_state, ssm_state).unsqueeze(1), conv_state, ssm_state

    def step_from_zxbcdt(self, zxbcdt, conv_state, ssm_state, conv_done=False):
        """The decode step after in_proj: zxbcdt (batch, d_in_proj) -> out (batch, d_model).  Split out so that callers
        which fuse the block's pre-norm into the in_proj GEMV (stack.ResidualBlock) can enter here; conv_done: the xBC
        columns already went through the convolution update (norm_linear's conv tail)."""
        d_mlp = (zxbcdt.shape[-1] - 2 * self.d_ssm - 2 * self.ngroups * self.d_state - self.nheads) // 2
        z0, x0, z, xBC, dt = torch.split(
            zxbcdt, [d_mlp, d_mlp, self.d_ssm, self.d_ssm + 2 * self.ngroups * self.d_state, self.nheads], dim=-1)
        if not conv_done:
            xBC = causal_conv1d_update(xBC, conv_state, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.activation)
        x, B, C = torch.split(xBC, [self.d_ssm, self.ngroups * self.d_state, self.ngroups * self.d_state], dim=-1)
        A = self._A_inference()
        H, P, N = self.nheads, self.headdim, self.d_state
        batch = x.shape[0]
        # stride-0 expansions: the kernel takes the tied-head fast path (A/dt/dt_bias constant over (p, n))
        A_e = A[:, None, None].expand(H, P, N)
        dt_e = dt[:, :, None].expand(batch, H, P)
        dt_bias_e = self.dt_bias[:, None].expand(H, P)
        D_e = self.D.view(H, P) if self.D_has_hdim else self.D[:, None].expand(H, P)
        y = selective_state_update(ssm_state, x.view(batch, H, P), dt_e, A_e, B.view(batch, self.ngroups, N),
                                   C.view(batch, self.ngroups, N), D_e, z=z.view(batch, H, P) if not self.rmsnorm else None,
                                   dt_bias=dt_bias_e, dt_softplus=True)
        y = y.reshape(batch, H * P)
        if (self.rmsnorm and d_mlp == 0 and type(self.out_proj) is nn.Linear and self.norm.bias is None
                and NL.applies(y, self.out_proj.weight, self.norm.weight, z, self.out_proj.bias)):
            # gated RMSNorm + out_proj in one launch (weights streamed once)
            return NL.norm_linear(y, self.out_proj.weight, self.out_proj.bias, norm_weight=self.norm.weight, eps=self.norm.eps,
                                  z=z, group_size=self.norm.group_size, norm_before_gate=self.norm.norm_before_gate)
        if self.rmsnorm:
            y = self.norm(y, z)
        if d_mlp > 0:
            y = torch.cat([F.silu(z0) * x0, y], dim=-1)
        return self.out_proj(y)

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        device = self.out_proj.weight.device
        conv_dtype = self.conv1d.weight.dtype if dtype is None else dtype
        # channel-last storage (batch, conv_dim, d_conv) like upstream: adjacent lanes = adjacent channels
        conv_state = torch.zeros(batch_size, self.d_conv, self.conv1d.weight.shape[0], device=device,
                                 dtype=conv_dtype).transpose(1, 2)
        ssm_dtype = self.in_proj.weight.dtype if dtype is None else dtype
        ssm_state = torch.zeros(batch_size, self.nheads, self.headdim, self.d_state, device=device, dtype=ssm_dtype)
        return conv_state, ssm_state

    def _get_states_from_cache(self, inference_params, batch_size, initialize_states=False):
        assert self.layer_idx is not None
        if self.layer_idx not in inference_params.key_value_memory_dict:
            conv_state, ssm_state = self.allocate_inference_cache(batch_size, 0)
            inference_params.key_value_memory_dict[self.layer_idx] = (conv_state, ssm_state)
        else:
            conv_state, ssm_state = inference_params.key_value_memory_dict[self.layer_idx]
            if initialize_states:
                conv_state.zero_()
                ssm_state.zero_()
        return conv_state, ssm_state

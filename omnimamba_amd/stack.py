"""The sequence model around the hot path, restated for synthetic benchmarks and tests (SURVEY.md section 8 rows
a1, a3; reference: /root/reference/models/stage2/{block.py, lora.py, mixer_seq_simple.py}, models/mamba_vlm.py:115-116).

Not a port of OmniMamba: only what drives the Mamba-2 path -- token / image-token embeddings, N x [fused add+RMSNorm ->
Mamba2 (in_proj wrapped by the task-switched LoRA)], final fused norm, the two tied heads.  Vision towers, VQ-VAE,
tokenizer and captions are inputs of the path and are replaced by synthetic tensors.
State-dict keys match the reference's MambaLMHeadModel KEY FOR KEY (`backbone.{embedding, img_embeddings.word_embeddings,
img_embeddings.project_in.projector.{0,2,4}, pos_embed, caption_embed.cap_proj.{fc1,fc2}, mmu_pos_embed, layers.{i}.norm,
layers.{i}.mixer.*, norm_f}`, `lm_head`, `img_head`; mixer_seq_simple.py:296-304,353,484-502) so that
`OmniMamba-1.3b.pth` / `mamba2-1.3b` checkpoints load with strict=True (section 8f-4; tests/test_reference_fixtures.py
loads a state dict written by the reference's own classes).  `omnimamba_amd.omni.OmniMambaPath` adds the projector and the
`llm_backbone.mamba.` prefix of the top-level checkpoint (models/omnimamba.py:88-103).
"""
from __future__ import annotations

import os
import math
from collections import namedtuple
from dataclasses import dataclass, field

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lora_add as LA
from . import lora_ext as LE
from . import norm_linear as NL
from .generation import GenerationMixin
from .layer_norm import RMSNorm, layer_norm_fn
from .linear import _WGradFn, linear
from .mamba2 import Mamba2

CausalLMOutput = namedtuple("CausalLMOutput", ["t2i_logits", "mmu_logits"])


@dataclass
class StackConfig:
    d_model: int = 2048
    n_layer: int = 48
    vocab_size: int = 50277
    pad_vocab_size_multiple: int = 16
    vqvae_vocab_size: int = 16384
    num_tokens: int = 256                  # image tokens of a T2I sample
    t2i_positions: int = 256 + 73          # reference: num_tokens + 73 (mixer_seq_simple.py:298-299)
    mmu_positions: int = 1500              # reference cap (mixer_seq_simple.py:302-303); raise for L > 1500 (documented deviation)
    ssm_cfg: dict = field(default_factory=dict)
    norm_epsilon: float = 1e-5
    residual_in_fp32: bool = True
    lora_r: int = 8
    lora_alpha: int = 32
    lora_dropout: float = 0.05
    t2i_task: bool = True
    mmu_task: bool = True
    token_drop: float = 0.0                # GPT2Embeddings token dropout (config_mamba.py:34)
    img_sq_len: int = 729                  # SigLIP + DINOv2 patch tokens (mixer_seq_simple.py:305)
    fused_vision_dim: int = 2176           # DINOv2-L 1024 + SigLIP-so400m 1152 (dinosiglip_vit.py:37-160)

    @staticmethod
    def omnimamba_1_3b(**kw):
        return StackConfig(d_model=2048, n_layer=48, **kw)

    @property
    def padded_vocab(self):
        m = self.pad_vocab_size_multiple
        return self.vocab_size + (-self.vocab_size) % m


class FusedMLPProjector(nn.Module):
    """Linear(in, 4 in) -> GELU -> Linear(4 in, out) -> GELU -> Linear(out, out), all with bias
    (reference models/cobra/nn_utils.py:38-54; parameter names `projector.{0,2,4}.{weight,bias}`)."""

    def __init__(self, fused_vision_dim, llm_dim, device=None, dtype=None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        self.initial_projection_dim = fused_vision_dim * 4
        self.projector = nn.Sequential(nn.Linear(fused_vision_dim, self.initial_projection_dim, bias=True, **fk), nn.GELU(),
                                       nn.Linear(self.initial_projection_dim, llm_dim, bias=True, **fk), nn.GELU(),
                                       nn.Linear(llm_dim, llm_dim, bias=True, **fk))

    def forward(self, fused_img_patches):
        return self.projector(fused_img_patches)


class ImageTokenEmbeddings(nn.Module):
    """VQ image-token embedding = table lookup + FusedMLPProjector(d, d) (reference GPT2Embeddings built with
    max_position_embeddings = -1 and word_embed_proj_dim = d_model, mixer_seq_simple.py:36-89,297)."""

    def __init__(self, d_model, vocab_size, token_drop=0.0, device=None, dtype=None):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab_size, d_model, device=device, dtype=dtype)
        self.project_in = FusedMLPProjector(d_model, d_model, device=device, dtype=dtype)
        self.token_dropout = nn.Dropout(token_drop)

    # ---- decode: the module is a function of the token id alone, so one pass over the whole codebook (16 384 rows) replaces the
    # three GEMVs over 151 MB of weights that every generated token would otherwise stream (1.3B size).  Built outside any graph
    # capture by `prepare_decode()` (generation.decode calls it), refreshed when a parameter's version changes.
    def _table_key(self):
        return tuple((p.data_ptr(), p._version, p.dtype, p.device) for p in self.parameters())

    @torch.no_grad()
    def prepare_decode(self):
        if os.environ.get("OMK_IMG_EMBED_TABLE", "1") == "0":
            self._omk_tab = self._omk_tab_key = None
            return
        key = self._table_key()
        if getattr(self, "_omk_tab_key", None) != key:
            ids = torch.arange(self.word_embeddings.num_embeddings, device=self.word_embeddings.weight.device)
            new = self.project_in(self.word_embeddings(ids))
            tab = getattr(self, "_omk_tab", None)
            # refresh IN PLACE: a captured decode graph (kept in model._decoding_cache across calls) has this tensor's address baked
            # into its F.embedding node; only a shape / dtype / device change reallocates (and such a change invalidates the graph too)
            if tab is not None and tab.shape == new.shape and tab.dtype == new.dtype and tab.device == new.device:
                tab.copy_(new)
            else:
                self._omk_tab = new
            self._omk_tab_key = key

    def forward(self, input_ids, position_ids=None):
        tab = getattr(self, "_omk_tab", None)
        if (tab is not None and input_ids.shape[-1] == 1 and not self.training and not torch.is_grad_enabled()
                and self._omk_tab_key == self._table_key()):
            return F.embedding(input_ids, tab)
        return self.project_in(self.token_dropout(self.word_embeddings(input_ids)))


class CaptionEmbedder(nn.Module):
    """fc1 -> GELU(tanh) -> fc2 without biases on the caption embeddings (mixer_seq_simple.py:124-164)."""

    class _MLP(nn.Module):
        def __init__(self, d, device=None, dtype=None):
            super().__init__()
            self.fc1 = nn.Linear(d, d, bias=False, device=device, dtype=dtype)
            self.act = nn.GELU(approximate="tanh")
            self.fc2 = nn.Linear(d, d, bias=False, device=device, dtype=dtype)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    def __init__(self, in_channels, hidden_size, device=None, dtype=None):
        super().__init__()
        assert in_channels == hidden_size
        self.cap_proj = CaptionEmbedder._MLP(hidden_size, device=device, dtype=dtype)

    def forward(self, caption, train=False, force_drop_ids=None):
        return self.cap_proj(caption)


class TaskLoRALinear(nn.Linear):
    """Task-switched LoRA on a dense layer (reference models/stage2/lora.py:185-279 with lora_nums = 1):
    y = x W^T (+b) + scaling * B_task(A_task(dropout(x))),  task in {'t2i', 'mmu'} chosen by ``self.task_types``.
    Parameter names match the reference (`{task}_lora_A0.weight`, `{task}_lora_B0.weight`)."""

    def __init__(self, in_features, out_features, r=8, lora_alpha=32, lora_dropout=0.05, bias=False, device=None, dtype=None):
        super().__init__(in_features, out_features, bias=bias, device=device, dtype=dtype)
        self.r, self.lora_alpha = r, lora_alpha
        self.scaling = lora_alpha / r
        self.lora_dropout = nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else nn.Identity()
        self.task_types = "t2i"
        self.disable_adapters = False
        for task in ("mmu", "t2i"):
            setattr(self, f"{task}_lora_A0", nn.Linear(in_features, r, bias=False, device=device, dtype=dtype))
            setattr(self, f"{task}_lora_B0", nn.Linear(r, out_features, bias=False, device=device, dtype=dtype))
        self.weight.requires_grad = False
        for task in ("mmu", "t2i"):
            nn.init.kaiming_uniform_(getattr(self, f"{task}_lora_A0").weight, a=math.sqrt(5))
            nn.init.zeros_(getattr(self, f"{task}_lora_B0").weight)

    def forward(self, x):
        if self.disable_adapters or self.task_types not in ("t2i", "mmu"):
            return linear(x, self.weight, self.bias)
        A = getattr(self, f"{self.task_types}_lora_A0")
        B = getattr(self, f"{self.task_types}_lora_B0")
        # ---- many tokens, 16-bit GEMMs: the rank joins the contraction dimension, base + LoRA are ONE GEMM and one autograd node
        adt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        x2 = x.reshape(-1, x.shape[-1])
        if self.bias is None and os.environ.get("OMK_LORA_EXT", "1") != "0" and LE.applies(x2, self.weight, self.r, adt):
            # the dropout module's own mode decides (stage 'align' leaves in_proj in eval() and re-enables train() on the
            # modules named *lora*, reference omnimamba.py:137-151; lora.py:270-274 calls self.lora_dropout(x) unconditionally)
            p_drop = self.lora_dropout.p if (isinstance(self.lora_dropout, nn.Dropout) and self.lora_dropout.training) else 0.0
            y = LE.lora_ext_linear(self, x2, self.weight.detach(), A.weight, B.weight, self.scaling, adt, p_drop)
            if torch.is_grad_enabled() and self.weight.requires_grad:
                y = _WGradFn.apply(y, x2.detach(), self.weight, None)   # the base weight trains ('finetune'): its gradient node
            return y.view(*x.shape[:-1], y.shape[-1])
        h = linear(self.lora_dropout(x), A.weight) if os.environ.get("OMK_LORA_A_PLAIN") != "1" else A(self.lora_dropout(x))   # token-split dA
        result = linear(x, self.weight, self.bias)   # F.linear; token-split weight gradient when the base weight trains
        # result + scaling * B(h) as ONE GEMM with a beta = 1 epilogue: the separate scale and add passes over the
        # (tokens, 8512) tensor cost two extra HBM round trips per call (8.7 % of the 1.3B training step)
        out_f = result.shape[-1]
        # (an in-place addmm_ on the base GEMM's output would save the 279 MB copy the out-of-place form starts with, but
        # the library then picks a much slower GEMM: 461 -> 509 ms per 1.3B training step)
        r2, h2 = result.reshape(-1, out_f), h.reshape(-1, h.shape[-1])
        if os.environ.get("OMK_LORA_ADDMM") != "1" and r2.data_ptr() == result.data_ptr() and LA.applies(r2, h2, B.weight) and LA.producer_is_safe(result):
            # one streaming pass over the result (read once, write once) instead of the library's copy + K = 8 GEMM
            return LA.lora_add(r2, h2, B.weight, self.scaling).view(result.shape)
        fused = torch.addmm(r2, h2, B.weight.t().to(h.dtype), alpha=self.scaling)
        return fused.view(result.shape)


class ResidualBlock(nn.Module):
    """Add -> RMSNorm -> Mixer with the residual stream kept in fp32 (reference block.py:71-117, fused_add_norm path)."""

    def __init__(self, d_model, layer_idx, cfg: StackConfig, device=None, dtype=None):
        super().__init__()
        self.residual_in_fp32 = cfg.residual_in_fp32
        self.norm = RMSNorm(d_model, eps=cfg.norm_epsilon, device=device, dtype=dtype)
        self.mixer = Mamba2(d_model, layer_idx=layer_idx, device=device, dtype=dtype, **cfg.ssm_cfg)
        self.layer_idx = layer_idx

    def forward(self, hidden_states, residual=None, inference_params=None):
        if inference_params is not None and inference_params.seqlen_offset > 0 and hidden_states.shape[1] == 1:
            fused = self._decode_step_fused(hidden_states, residual, inference_params)
            if fused is not None:
                return fused
        hidden_states, residual = layer_norm_fn(hidden_states, self.norm.weight, self.norm.bias, residual=residual,
                                                prenorm=True, residual_in_fp32=self.residual_in_fp32, eps=self.norm.eps,
                                                is_rms_norm=True)
        return self.mixer(hidden_states, inference_params=inference_params), residual

    def _decode_step_fused(self, hidden_states, residual, inference_params):
        """One-token step with add + RMSNorm + in_proj (+ the task's LoRA) as a single launch (omk_norm_linear), then the
        rest of Mamba2.step.  Returns None when the fused kernel does not apply (training, large batch, odd shapes)."""
        ip = self.mixer.in_proj
        x2 = hidden_states.squeeze(1)
        if self.norm.bias is not None or not isinstance(ip, nn.Linear):
            return None
        lora = {}
        if isinstance(ip, TaskLoRALinear):
            if not ip.disable_adapters and ip.task_types in ("t2i", "mmu"):
                if ip.training and not isinstance(ip.lora_dropout, nn.Identity):
                    return None                       # dropout on the LoRA input: only the unfused path implements it
                lora = dict(lora_a=getattr(ip, f"{ip.task_types}_lora_A0").weight, lora_b=getattr(ip, f"{ip.task_types}_lora_B0").weight,
                            lora_scale=ip.scaling)
        elif type(ip) is not nn.Linear:
            return None
        if lora and (lora["lora_a"].shape[0] > 8 and x2.shape[0] > 1):
            return None
        if not NL.applies(x2, ip.weight, self.norm.weight, lora.get("lora_a"), lora.get("lora_b"), ip.bias):
            return None
        if x2.shape[0] > 1 and residual is not None and residual.dtype not in (torch.float32, x2.dtype):
            return None
        ro_dtype = torch.float32 if (self.residual_in_fp32 or (residual is not None and residual.dtype == torch.float32)) else x2.dtype
        conv_state, ssm_state = self.mixer._get_states_from_cache(inference_params, x2.shape[0])
        res2 = None if residual is None else residual.squeeze(1)
        m, conv = self.mixer, {}
        cw = m.conv1d.weight.squeeze(1)
        d_mlp = (ip.weight.shape[0] - 2 * m.d_ssm - 2 * m.ngroups * m.d_state - m.nheads) // 2
        if (m.activation in ("silu", "swish") and os.environ.get("OMK_DECODE_CONV_SEPARATE") != "1"
                and NL.conv_tail_applies(x2, ip.weight, self.norm.weight, conv_state, cw, m.conv1d.bias, lora.get("lora_a"), ip.bias, res2)):
            # the convolution of the new xBC inputs rides on the in_proj launch (one launch less per layer-step)
            conv = dict(conv_state=conv_state, conv_weight=cw, conv_bias=m.conv1d.bias, conv_offset=2 * d_mlp + m.d_ssm)
        zxbcdt, new_res = NL.norm_linear(x2, ip.weight, ip.bias, norm_weight=self.norm.weight, eps=self.norm.eps,
                                         residual=res2, residual_out_dtype=ro_dtype, **lora, **conv)
        out = self.mixer.step_from_zxbcdt(zxbcdt, conv_state, ssm_state, conv_done=bool(conv))
        return out.unsqueeze(1), new_res.unsqueeze(1)

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)


class MixerStack(nn.Module):
    """The reference's MixerModel (mixer_seq_simple.py:265-440) without the dead adaLN branches."""

    def __init__(self, cfg: StackConfig, device=None, dtype=None):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        self.cfg = cfg
        d = cfg.d_model
        self.t2i_task, self.mmu_task = cfg.t2i_task, cfg.mmu_task
        self.img_sq_len = cfg.img_sq_len
        if cfg.t2i_task:
            self.img_embeddings = ImageTokenEmbeddings(d, cfg.vqvae_vocab_size, token_drop=cfg.token_drop, **fk)
            self.pos_embed = nn.Parameter(nn.init.trunc_normal_(torch.zeros(1, cfg.t2i_positions, d, **fk), 0.0, 0.02))
            self.caption_embed = CaptionEmbedder(d, d, **fk)
        if cfg.mmu_task:
            self.mmu_pos_embed = nn.Parameter(nn.init.trunc_normal_(torch.zeros(1, cfg.mmu_positions, d, **fk), 0.0, 0.02))
        self.embedding = nn.Embedding(cfg.padded_vocab, d, **fk)
        self.layers = nn.ModuleList([ResidualBlock(d, i, cfg, **fk) for i in range(cfg.n_layer)])
        self.norm_f = RMSNorm(d, eps=cfg.norm_epsilon, **fk)
        # reference _init_weights (mixer_seq_simple.py:233-262): Linear biases zero, embeddings N(0, 0.02),
        # out_proj.weight and fc2.weight kaiming-uniform / sqrt(n_layer)
        for m in self.modules():
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=0.02)
        for n, p in self.named_parameters():
            if n.endswith("out_proj.weight") or n.endswith("fc2.weight"):
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(cfg.n_layer)
        # reference _find_and_replace (lora.py:78-112): every mixer.in_proj becomes the task-switched LoRA Linear that
        # shares the base weight tensor
        for blk in self.layers:
            old = blk.mixer.in_proj
            new = TaskLoRALinear(old.in_features, old.out_features, r=cfg.lora_r, lora_alpha=cfg.lora_alpha,
                                 lora_dropout=cfg.lora_dropout, bias=old.bias is not None, device=old.weight.device,
                                 dtype=old.weight.dtype)
            new.weight = old.weight
            new.weight.requires_grad = False
            blk.mixer.in_proj = new

    def set_lora_mode(self, task):
        for blk in self.layers:
            blk.mixer.in_proj.task_types = task

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return {i: blk.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs) for i, blk in enumerate(self.layers)}

    def forward(self, input_ids, input_embeddings, position_ids=None, task="t2i", inference_params=None):
        """Training / prefill: ``input_embeddings`` (B, L, d) given -- for 'mmu' the position table is added here when
        ``position_ids`` is None, for 't2i' the CALLER has added ``pos_embed`` (reference omnimamba.py:264,319-320).
        Decode: ``input_ids`` (B, 1) + ``position_ids`` (reference MixerModel.forward, mixer_seq_simple.py:375-440)."""
        self.set_lora_mode(task)
        if input_embeddings is not None:
            h = input_embeddings
            if task == "mmu" and position_ids is None:
                h = h + self.mmu_pos_embed[:, : h.shape[1]]
        else:
            if task == "t2i":
                h = self.img_embeddings(input_ids)
                pe = self.pos_embed.expand(h.shape[0], -1, -1)
            else:
                h = self.embedding(input_ids)
                pe = self.mmu_pos_embed.expand(h.shape[0], -1, -1)
            h = h + pe.gather(1, position_ids.unsqueeze(-1).expand(-1, -1, pe.size(-1)))
        residual = None
        for blk in self.layers:
            h, residual = blk(h, residual, inference_params=inference_params)
        return layer_norm_fn(h, self.norm_f.weight, self.norm_f.bias, eps=self.norm_f.eps, residual=residual, prenorm=False,
                             residual_in_fp32=self.cfg.residual_in_fp32, is_rms_norm=True)


class OmniMambaLM(nn.Module, GenerationMixin):
    """backbone + tied heads (reference MambaLMHeadModel, mixer_seq_simple.py:443-524: an nn.Module with the GenerationMixin)."""

    def __init__(self, cfg: StackConfig, device=None, dtype=None):
        super().__init__()
        self.cfg = cfg
        self.backbone = MixerStack(cfg, device=device, dtype=dtype)
        if cfg.t2i_task:
            self.img_head = nn.Linear(cfg.d_model, cfg.vqvae_vocab_size, bias=False, device=device, dtype=dtype)
        self.lm_head = nn.Linear(cfg.d_model, cfg.padded_vocab, bias=False, device=device, dtype=dtype)
        self.tie_weights()
        self._decoding_cache = None

    def tie_weights(self):
        """mixer_seq_simple.py:498-502: both heads share their embedding tables."""
        if self.cfg.t2i_task:
            self.img_head.weight = self.backbone.img_embeddings.word_embeddings.weight
        self.lm_head.weight = self.backbone.embedding.weight

    def get_input_embeddings(self):
        return self.backbone.embedding

    def get_output_embeddings(self):
        return self.lm_head

    def set_input_embeddings(self, value):
        self.backbone.embedding = value

    def resize_token_embeddings(self, new_num_tokens=None, pad_to_multiple_of=None):
        """mixer_seq_simple.py:562-676, reached from OmniMamba.__init__ (omnimamba.py:103: `len(tokenizer)` padded to
        `pad_vocab_size_multiple`): a new table of the padded size with the old rows copied and the added rows N(0, 0.02), heads tied again.
        Same size -> the old table, untouched (the case of the shipped checkpoints: 50 287 tokens padded to 50 288)."""
        old = self.get_input_embeddings()
        if new_num_tokens is None and pad_to_multiple_of is None:
            return old
        n_new = old.weight.shape[0] if new_num_tokens is None else int(new_num_tokens)
        if pad_to_multiple_of is not None:
            if not isinstance(pad_to_multiple_of, int):
                raise ValueError(f"pad_to_multiple_of must be an integer, got {pad_to_multiple_of!r}")
            n_new = -(-n_new // pad_to_multiple_of) * pad_to_multiple_of
        n_old, d = old.weight.shape
        if n_new != n_old:
            new = nn.Embedding(n_new, d, device=old.weight.device, dtype=old.weight.dtype)
            with torch.no_grad():
                nn.init.normal_(new.weight, std=0.02)
                new.weight[: min(n_old, n_new)] = old.weight[: min(n_old, n_new)]
            new.weight.requires_grad_(old.weight.requires_grad)
            self.set_input_embeddings(new)
            self.lm_head = nn.Linear(d, n_new, bias=False, device=old.weight.device, dtype=old.weight.dtype)
            self.cfg.vocab_size, self._decoding_cache = n_new, None
            if n_new % self.cfg.pad_vocab_size_multiple:
                self.cfg.pad_vocab_size_multiple = 1          # (resized without padding: the table IS the vocabulary)
        self.tie_weights()
        return self.get_input_embeddings()

    # ---- a directory with config.json + pytorch_model.bin (mixer_seq_simple.py:526-549; no hub access here: local paths only)
    def save_pretrained(self, save_directory):
        import json
        from dataclasses import asdict
        os.makedirs(save_directory, exist_ok=True)
        torch.save(self.state_dict(), os.path.join(save_directory, "pytorch_model.bin"))
        with open(os.path.join(save_directory, "config.json"), "w") as fh:
            json.dump(asdict(self.cfg), fh, indent=4)

    @classmethod
    def from_pretrained(cls, pretrained_model_name, device=None, dtype=None, strict=True, **kwargs):
        import json
        if not os.path.isdir(pretrained_model_name):
            raise FileNotFoundError(f"{pretrained_model_name}: a local directory with config.json and pytorch_model.bin is expected (no hub access)")
        with open(os.path.join(pretrained_model_name, "config.json")) as fh:
            raw = json.load(fh)
        known = {f.name for f in StackConfig.__dataclass_fields__.values()}
        model = cls(StackConfig(**{k: v for k, v in raw.items() if k in known}), device=device, dtype=dtype, **kwargs)
        sd = torch.load(os.path.join(pretrained_model_name, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        model.load_state_dict(sd, strict=strict)
        return model

    def prepare_decode(self, task="t2i"):
        """Called by generation.decode before the first step (and before any graph capture): per-token-id table of the image-token
        embedding MLP for the T2I loop (its inputs are sampled VQ ids)."""
        if task == "t2i" and self.cfg.t2i_task:
            self.backbone.img_embeddings.prepare_decode()
        # the step graphs read every mixer's persistent -exp(A_log) buffer; with the PREFILL captured too (generation.PrefillGraph) no
        # eager forward refreshes it any more after an in-place weight update -- do it here, outside any capture
        if not torch.is_grad_enabled():
            for blk in self.backbone.layers:
                if hasattr(blk.mixer, "_A_inference"):
                    blk.mixer._A_inference()

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.backbone.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)

    def forward(self, input_ids, input_embeddings, position_ids=None, cond=None, task=None, inference_params=None,
                num_last_tokens=0):
        h = self.backbone(input_ids, input_embeddings, position_ids, task, inference_params=inference_params)
        if num_last_tokens > 0:
            h = h[:, -num_last_tokens:]
        return CausalLMOutput(t2i_logits=self.img_head(h) if task == "t2i" else None,
                              mmu_logits=self.lm_head(h) if task == "mmu" else None)

    def set_stage(self, stage: str):
        """Which parameters train (reference omnimamba.py:119-188).  'align': the stack is frozen except -- with the T2I
        task -- img_embeddings, embedding, pos_embed, caption_embed, img_head, and the LoRA adapters; 'finetune':
        everything (mamba.requires_grad_(True) overrides the LoRA base-weight freeze, Appendix C of SURVEY.md).
        Deviation (SURVEY.md hard parts): adapters of a task the model is not configured for stay frozen, because DDP with
        find_unused_parameters=False (train_stage2.py:38) cannot carry parameters that never receive a gradient."""
        if stage == "finetune":
            for p in self.parameters():
                p.requires_grad_(True)
        elif stage == "align":
            tasks = [t for t, on in (("t2i", self.cfg.t2i_task), ("mmu", self.cfg.mmu_task)) if on]
            for n, p in self.named_parameters():
                on = any(f"{t}_lora_" in n for t in tasks)
                if self.cfg.t2i_task and (n.startswith("backbone.img_embeddings.") or n == "backbone.embedding.weight"
                                          or n == "backbone.pos_embed" or n.startswith("backbone.caption_embed.")
                                          or n == "img_head.weight" or n == "lm_head.weight"):
                    on = True
                p.requires_grad_(on)
        else:
            raise ValueError(stage)
        for t, on in (("t2i", self.cfg.t2i_task), ("mmu", self.cfg.mmu_task)):
            if not on:
                for n, p in self.named_parameters():
                    if f"{t}_lora_" in n:
                        p.requires_grad_(False)

"""The VQ decode tail of T2I generation: sampled image-token ids -> pixels (last item).

Reference: ``MambaVLM.decode_to_img`` (/root/reference/models/mamba_vlm.py:104-108) calls ``vqvae.decode_code(index, shape=[B, 8, 16, 16])``
= ``quantize.get_codebook_entry`` -> ``post_quant_conv`` -> ``decoder`` of the LlamaGen VQ-16 tokenizer
(/root/reference/llamagen_tokenizer/tokenizer_image/vq_model.py:48-55, 128-194, 261-277, 279-377).  This module is that tail only
-- no encoder, no quantiser losses -- with parameter names equal to the reference's ``VQModel.state_dict()`` (``quantize.embedding.weight``,
``post_quant_conv.*``, ``decoder.*``), so ``vq_ds16_t2i.pt`` loads key for key (``load_reference_state_dict`` drops ``encoder.*`` /
``quant_conv.*`` / ``quantize.codebook_used``).

It is a convolutional network behind the hot path, not a scan: the kernels are the library's (MIOpen convolutions, the fused attention of
``scaled_dot_product_attention``).  What is decided here is how it runs on the MI355X, by measurement (`bench.py` -> `decode_1p3b.vq_tail_ms`,
batch 1, 256 ids -> 3 x 256 x 256, 42.6 M parameters): eager fp32 6.4 ms, one hipGraph replay (``graphed``) 6.25 ms, replay under bf16 autocast
5.8 ms, bf16 + channels-last 6.6 ms -- the convolutions at 128^2 / 256^2 pixels are the time, not the 60 launches, and NHWC buys nothing
from the library here, so NCHW is the default and channels-last an option (``set_channels_last``).  Against the 440 ms of the 256-token
decode loop in front of it the tail is 1.4 % of a generated image.  The l2-normalised codebook is cached per weight version (the
reference normalises all 16 384 rows on every call).
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


def _gn(c: int) -> nn.GroupNorm:
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class _Res(nn.Module):
    """GroupNorm -> swish -> 3x3 conv, twice, plus the (1x1-projected when the width changes) input (vq_model.py:279-314)."""

    def __init__(self, cin: int, cout: int, p_drop: float = 0.0):
        super().__init__()
        self.norm1, self.conv1 = _gn(cin), nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2, self.conv2 = _gn(cout), nn.Conv2d(cout, cout, 3, padding=1)
        self.dropout = nn.Dropout(p_drop)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(self.dropout(F.silu(self.norm2(h))))
        return h + (self.nin_shortcut(x) if hasattr(self, "nin_shortcut") else x)


class _Attn(nn.Module):
    """Single-head self-attention over the h * w positions, scale c ** -0.5, 1x1-conv projections (vq_model.py:317-351)."""

    def __init__(self, c: int):
        super().__init__()
        self.norm = _gn(c)
        self.q, self.k, self.v, self.proj_out = (nn.Conv2d(c, c, 1) for _ in range(4))

    def forward(self, x):
        b, c, hh, ww = x.shape
        n = self.norm(x)
        # (b, 1, hw, c): one head of width c; the fused attention kernel applies the c ** -0.5 scale and the softmax over keys
        q, k, v = (f(n).flatten(2).transpose(1, 2).unsqueeze(1) for f in (self.q, self.k, self.v))
        o = F.scaled_dot_product_attention(q, k, v)
        o = o.squeeze(1).transpose(1, 2).reshape(b, c, hh, ww)
        return x + self.proj_out(o)


class _Up(nn.Module):
    def __init__(self, c: int):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Decoder(nn.Module):
    """vq_model.py:128-194: conv_in, mid = (res, attn, res), levels from the coarsest up (res x (n + 1), attention on the coarsest
    level only, nearest x2 + conv between levels), norm_out -> swish -> conv_out."""

    def __init__(self, z_channels: int, ch: int, ch_mult: Sequence[int], num_res_blocks: int, out_channels: int, p_drop: float):
        super().__init__()
        top = len(ch_mult) - 1
        c = ch * ch_mult[top]
        self.conv_in = nn.Conv2d(z_channels, c, 3, padding=1)
        self.mid = nn.ModuleList([_Res(c, c, p_drop), _Attn(c), _Res(c, c, p_drop)])
        self.conv_blocks = nn.ModuleList()
        for lvl in range(top, -1, -1):
            blk = nn.Module()
            blk.res, blk.attn = nn.ModuleList(), nn.ModuleList()
            for _ in range(num_res_blocks + 1):
                blk.res.append(_Res(c, ch * ch_mult[lvl], p_drop))
                c = ch * ch_mult[lvl]
                if lvl == top:
                    blk.attn.append(_Attn(c))
            if lvl != 0:
                blk.upsample = _Up(c)
            self.conv_blocks.append(blk)
        self.norm_out = _gn(c)
        self.conv_out = nn.Conv2d(c, out_channels, 3, padding=1)

    def forward(self, z):
        h = self.conv_in(z)
        for m in self.mid:
            h = m(h)
        for blk in self.conv_blocks:
            for i, r in enumerate(blk.res):
                h = r(h)
                if len(blk.attn):
                    h = blk.attn[i](h)
            if hasattr(blk, "upsample"):
                h = blk.upsample(h)
        return self.conv_out(F.silu(self.norm_out(h)))


class VQDecodeTail(nn.Module):
    """``decode_code`` of the reference's VQ-16 tokenizer (defaults = ``VQ_16()``: 16 384 codes of width 8, z_channels 256, ch 128,
    multipliers (1, 1, 2, 2, 4): 16 x 16 codes -> 256 x 256 pixels)."""

    def __init__(self, codebook_size: int = 16384, codebook_embed_dim: int = 8, codebook_l2_norm: bool = True, z_channels: int = 256,
                 ch: int = 128, ch_mult: Sequence[int] = (1, 1, 2, 2, 4), num_res_blocks: int = 2, out_channels: int = 3, dropout_p: float = 0.0):
        super().__init__()
        self.l2_norm = codebook_l2_norm
        self.quantize = nn.Module()
        self.quantize.embedding = nn.Embedding(codebook_size, codebook_embed_dim)
        with torch.no_grad():   # the reference's initialisation (vq_model.py:208-211)
            self.quantize.embedding.weight.uniform_(-1.0 / codebook_size, 1.0 / codebook_size)
            if codebook_l2_norm:
                self.quantize.embedding.weight.copy_(F.normalize(self.quantize.embedding.weight, p=2, dim=-1))
        self.post_quant_conv = nn.Conv2d(codebook_embed_dim, z_channels, 1)
        self.decoder = _Decoder(z_channels, ch, ch_mult, num_res_blocks, out_channels, dropout_p)
        self.channels_last = False   # set_channels_last(): NHWC weights and activations (measured: no gain on this network, see the header)
        self._cb = None        # (key, normalised codebook)
        self._graph = None     # (key, graph, static ids, static image)

    # ---- codebook lookup (vq_model.py:261-277)
    def _codebook(self) -> torch.Tensor:
        w = self.quantize.embedding.weight
        if not self.l2_norm:
            return w
        try:
            key = (w.data_ptr(), w._version, w.dtype, w.device)
        except RuntimeError:   # inference tensors carry no version counter
            return F.normalize(w, p=2, dim=-1)
        if self._cb is None or self._cb[0] != key:
            self._cb = (key, F.normalize(w.detach(), p=2, dim=-1))
        return self._cb[1]

    def get_codebook_entry(self, indices, shape=None, channel_first=True):
        zq = self._codebook()[indices]
        if shape is None:
            return zq
        if channel_first:   # shape = (batch, channel, height, width): rows arrive position-major
            return zq.reshape(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2)   # = channels-last storage of (b, c, h, w)
        return zq.view(shape)

    def decode(self, quant, dtype: Optional[torch.dtype] = None):
        """post_quant_conv -> decoder (vq_model.py:52-55).  ``dtype`` (bf16 / fp16): autocast the convolutions on the GPU."""
        if quant.dim() == 4:
            quant = quant.contiguous(memory_format=torch.channels_last if self.channels_last else torch.contiguous_format)
        if dtype is not None and quant.is_cuda:
            with torch.autocast("cuda", dtype=dtype):
                return self.decoder(self.post_quant_conv(quant))
        return self.decoder(self.post_quant_conv(quant))

    def decode_code(self, code_b, shape=None, channel_first=True, dtype: Optional[torch.dtype] = None):
        return self.decode(self.get_codebook_entry(code_b, shape, channel_first), dtype)

    @torch.no_grad()
    def decode_to_img(self, index, dtype: Optional[torch.dtype] = None):
        """mamba_vlm.py:104-108: ids (B, 256) -> images (B, 3, 256, 256) for the VQ-16 geometry."""
        side = int(round(index.shape[-1] ** 0.5))
        e = self.quantize.embedding.embedding_dim
        return self.decode_code(index.reshape(-1), shape=[index.shape[0], e, side, side], dtype=dtype)

    def set_channels_last(self, on: bool = True):
        self.channels_last = bool(on)
        self.to(memory_format=torch.channels_last if on else torch.contiguous_format)
        self._graph = None
        return self

    # ---- the whole tail as one graph replay (fixed batch / token count / dtype; frozen weights)
    # (inference mode, not just no_grad: when the process's first capture -- the decode loop's -- ran under it, the generator's graph
    # state tensors are inference tensors, and a later capture outside inference mode cannot update them)
    @torch.inference_mode()
    def graphed(self, index, dtype: Optional[torch.dtype] = None):
        if not index.is_cuda:
            return self.decode_to_img(index, dtype)
        key = (tuple(index.shape), index.dtype, index.device, dtype, self.channels_last)
        if self._graph is None or self._graph[0] != key:
            ids = index.clone()
            side = torch.cuda.Stream(device=index.device)
            side.wait_stream(torch.cuda.current_stream(index.device))
            with torch.cuda.stream(side):
                for _ in range(2):   # library workspaces / algorithm choice happen outside the capture
                    self.decode_to_img(ids, dtype)
            torch.cuda.current_stream(index.device).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                img = self.decode_to_img(ids, dtype)
            self._graph = (key, g, ids, img)
        _, g, ids, img = self._graph
        ids.copy_(index)
        g.replay()
        return img

    # ---- checkpoint of the reference tokenizer (vq_ds16_t2i.pt: {"model": VQModel.state_dict()})
    def load_reference_state_dict(self, state_dict, strict: bool = True):
        sd = state_dict.get("model", state_dict) if isinstance(state_dict, dict) else state_dict
        drop = ("encoder.", "quant_conv.", "quantize.codebook_used")
        return self.load_state_dict({k: v for k, v in sd.items() if not k.startswith(drop)}, strict=strict)

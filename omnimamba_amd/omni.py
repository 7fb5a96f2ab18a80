"""The callers either side of the path: sequence construction and losses of the top-level model, restated for the
synthetic BASELINE configs 4 / 5 and for checkpoint compatibility (reference
/root/reference/models/omnimamba.py:49-337, models/mamba_vlm.py:88-108).

Only what drives the Mamba-2 path is here: projector MLP, embedding of the two task sequences, position tables, shifted
cross-entropy, greedy T2I generation.  The frozen vision towers and the VQ-VAE are inputs / outputs of the path and are
replaced by synthetic tensors: ``images_feat`` (B, 729, 2176) stands for ``vision_backbone(pixel_values)`` and the 256
sampled codes are returned instead of ``vqvae.decode_code`` unless ``attach_vq_tail()`` put the decode half of the tokenizer in place
(omnimamba_amd/vq_tail.py; the vision towers and the VQ encoder stay out: SURVEY.md section 2.1 rows 13, 18).

State-dict keys: ``llm_backbone.mamba.*`` and ``projector.projector.{0,2,4}.*`` exactly as ``OmniMamba.state_dict()``
writes them; ``load_reference_state_dict`` drops the ``vision_backbone.`` / ``llm_backbone.vqvae.`` entries of a full
checkpoint (omnimamba.py:88-103) and loads everything else strictly.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import fused_ce
from .generation import decode
from .stack import FusedMLPProjector, OmniMambaLM, StackConfig

IGNORE_ID = -100     # UniversalPrompting(ignore_id=-100), mamba_vlm.py:33-38
# ids the gpt-neox-20b tokenizer (50277 entries) hands out when [PAD] and the nine special tokens are appended in the
# order of mamba_vlm.py:33-37 / prompting_utils.py:24-27 -- they land inside the 50288 padded rows
SPECIAL_IDS = {"[PAD]": 50277, "<|soi|>": 50278, "<|eoi|>": 50279, "<|sot|>": 50280, "<|eot|>": 50281, "<|t2i|>": 50282,
               "<|mmu|>": 50283, "<|soc|>": 50284, "<|eoc|>": 50285, "<|lvg|>": 50286}


class _LLMBackbone(nn.Module):
    """Holder that gives the stack the reference's attribute path ``llm_backbone.mamba`` (models/mamba_vlm.py:14-24)."""

    def __init__(self, cfg: StackConfig, device=None, dtype=None):
        super().__init__()
        self.mamba = OmniMambaLM(cfg, device=device, dtype=dtype)
        self.d_model, self.num_tokens = cfg.d_model, cfg.num_tokens

    def embed_input_ids(self, input_ids):
        return self.mamba.get_input_embeddings()(input_ids)

    def forward(self, x, c, cond=None, task="t2i"):
        """``MambaVLM.forward`` (models/mamba_vlm.py:88-102) as the reference's ``OmniMamba.forward`` calls it (omnimamba.py:275,305):
        embeddings x, labels c -> (shifted logits flattened to (tokens, vocab), shifted labels flattened).  The materialised-logits
        form; ``OmniMambaPath.forward`` takes the fused linear + cross-entropy instead and never builds this tensor."""
        if cond is not None:
            raise NotImplementedError("cond is always None in OmniMamba (omnimamba.py:275,305)")
        out = self.mamba(input_ids=None, input_embeddings=x, cond=None, task=task)
        logits = out.t2i_logits if task == "t2i" else out.mmu_logits
        return logits[..., :-1, :].reshape(-1, logits.shape[-1]), c[..., 1:].reshape(-1)

    @torch.no_grad()
    def decode_to_img(self, index):
        """models/mamba_vlm.py:104-108 (needs ``OmniMambaPath.attach_vq_tail()``: the decode half of the VQ tokenizer)."""
        tail = getattr(self, "vqvae", None)
        if tail is None:
            raise RuntimeError("decode_to_img needs OmniMambaPath.attach_vq_tail() first")
        return tail.decode_to_img(index)


def shifted_ce(hidden, head_weight, labels, loss_impl=None):
    """mamba_vlm.py:88-102 + omnimamba.py:276-279,305-306: logits[..., :-1, :] against labels[..., 1:], mean over the
    labels that are not -100.  ``loss_impl(hidden2d, weight, labels1d)`` may replace the materialised-logits form."""
    h = hidden[:, :-1].reshape(-1, hidden.shape[-1])
    t = labels[:, 1:].reshape(-1)
    if loss_impl is not None:
        return loss_impl(h, head_weight, t)
    if fused_ce.applies(h, head_weight):
        # token blocks: GEMM -> omk_cross_entropy (loss + gradient over the block's logits) -> gradient GEMMs; the (tokens, vocab)
        # fp32 logits of the reference (13 GB at vocab 50 288, L 8192, batch 8) never exist
        return fused_ce.fused_linear_cross_entropy(h, head_weight, t)
    return F.cross_entropy(F.linear(h, head_weight.to(h.dtype)).float(), t, ignore_index=IGNORE_ID)


class OmniMambaPath(nn.Module):
    def __init__(self, cfg: StackConfig, stage: str = "finetune", special_ids=None, device=None, dtype=None, loss_impl=None):
        super().__init__()
        self.cfg = cfg
        self.llm_backbone = _LLMBackbone(cfg, device=device, dtype=dtype)
        if cfg.mmu_task:
            self.projector = FusedMLPProjector(cfg.fused_vision_dim, cfg.d_model, device=device, dtype=dtype)
        self.special_ids = dict(SPECIAL_IDS if special_ids is None else special_ids)
        self.loss_impl = loss_impl
        self.set_stage(stage)

    # ---- which parameters train (omnimamba.py:119-188)
    def set_stage(self, stage):
        if stage == "inference":
            self.eval()
            self.requires_grad_(False)
            return
        self.train()
        self.llm_backbone.mamba.set_stage(stage)
        if self.cfg.mmu_task:
            self.projector.requires_grad_(True)       # trained in both 'align' and 'finetune'
        if stage == "align":
            self.llm_backbone.mamba.eval()            # llm_backbone.eval(): dropouts off ...
            for n, m in self.llm_backbone.mamba.named_modules():
                if "lora" in n.lower() or (self.cfg.t2i_task and ("img_embeddings" in n or "caption_embed" in n or n == "img_head")):
                    m.train()                         # ... except the modules the stage trains (:137-151)

    @property
    def backbone(self):
        return self.llm_backbone.mamba.backbone

    def load_reference_state_dict(self, state_dict, strict=True):
        vq = {k[len("llm_backbone.vqvae."):]: v for k, v in state_dict.items() if k.startswith("llm_backbone.vqvae.")}
        sd = {k: v for k, v in state_dict.items() if not (k.startswith("vision_backbone.") or k.startswith("llm_backbone.vqvae."))}
        tail = getattr(self.llm_backbone, "vqvae", None)
        if tail is not None:
            if vq:
                tail.load_reference_state_dict(vq, strict=strict)
            sd.update({"llm_backbone.vqvae." + k: v for k, v in tail.state_dict().items()})
        return self.load_state_dict(sd, strict=strict)

    def attach_vq_tail(self, tail=None):
        """The decode half of the frozen VQ-16 tokenizer where the reference keeps it (``llm_backbone.vqvae``, mamba_vlm.py:19,55-69):
        ``t2i_generate(..., decode_images=True)`` then ends in pixels like omnimamba.py:334-336.  Optional -- without it the path returns
        the sampled ids and its state dict has the reference's MambaLMHeadModel keys only."""
        from .vq_tail import VQDecodeTail
        p = next(self.llm_backbone.mamba.parameters())
        tail = VQDecodeTail() if tail is None else tail
        self.llm_backbone.vqvae = tail.to(p.device).eval().requires_grad_(False)
        return self.llm_backbone.vqvae

    # ---- sequence construction (omnimamba.py:190-218,253-307)
    def _sp(self, name, like):
        return torch.full((like.shape[0], 1), self.special_ids[name], dtype=torch.long, device=like.device)

    def t2i_sequence(self, image_ids, caption_ids):
        bb = self.backbone
        img = bb.img_embeddings(image_ids)
        txt = bb.caption_embed(self.llm_backbone.embed_input_ids(caption_ids), train=True)
        emb = torch.cat((txt[:, :-1], img, txt[:, -1:]), dim=1)
        ign = lambda n: torch.full((image_ids.shape[0], n), IGNORE_ID, dtype=torch.long, device=image_ids.device)
        labels = torch.cat([ign(caption_ids.shape[1] - 1), image_ids, ign(1)], dim=1)
        return emb + bb.pos_embed[:, : emb.shape[1]], labels

    def mmu_sequence(self, images_feat, input_ids, labels, multimodal_indices=None):
        """images_feat None = a text-only batch: zero image embeddings of img_sq_len positions (omnimamba.py:222-250).
        multimodal_indices (the reference's batch key, omnimamba.py:281-301): the rows that carry an image; the others get the zero
        embeddings of the text-only form.  (The reference moves the image rows to the front of the batch; the loss is a mean over
        tokens, so the rows stay where they are here.)"""
        ids = torch.cat([self._sp("<|mmu|>", input_ids), self._sp("<|soi|>", input_ids), self._sp("<|eoi|>", input_ids),
                         self._sp("<|sot|>", input_ids), input_ids], dim=1)
        txt = self.llm_backbone.embed_input_ids(ids)
        n_rows = ids.shape[0]
        if images_feat is not None and multimodal_indices is not None and len(multimodal_indices) == 0:
            images_feat = None
        if images_feat is None:
            img = torch.zeros(n_rows, self.backbone.img_sq_len, txt.shape[-1], device=txt.device, dtype=txt.dtype)
        elif multimodal_indices is not None and len(multimodal_indices) < n_rows:
            idx = torch.as_tensor(multimodal_indices, device=txt.device, dtype=torch.long)
            some = self.projector(images_feat[idx] if images_feat.shape[0] == n_rows else images_feat)
            img = torch.zeros(n_rows, some.shape[1], txt.shape[-1], device=txt.device, dtype=some.dtype).index_copy(0, idx, some)
        else:
            img = self.projector(images_feat)
        emb = torch.cat((txt[:, :2], img.to(txt.dtype), txt[:, 2:]), dim=1)
        ign = lambda n: torch.full((ids.shape[0], n), IGNORE_ID, dtype=torch.long, device=ids.device)
        return emb, torch.cat([ign(2), ign(img.shape[1]), ign(2), labels.to(ids.device)], dim=1)

    def forward(self, inputs, task="t2i"):
        """inputs: {'t2i_flow': {'inputs': image ids (B, n_img), 'caption_ids': (B, Lc)},
                    'mmu_flow': {'images_feat': (B, 729, 2176) | None, 'input_ids': (B, T), 'labels': (B, T),
                                 'multimodal_indices': optional rows that carry an image}} -> loss."""
        lm = self.llm_backbone.mamba
        if task == "t2i":
            f = inputs["t2i_flow"]
            emb, labels = self.t2i_sequence(f["inputs"], f["caption_ids"])
            hidden = lm.backbone(None, emb, None, "t2i")
            return shifted_ce(hidden, lm.img_head.weight, labels, self.loss_impl)
        f = inputs["mmu_flow"]
        emb, labels = self.mmu_sequence(f.get("images_feat"), f["input_ids"], f["labels"], f.get("multimodal_indices"))
        hidden = lm.backbone(None, emb, None, "mmu")
        return shifted_ce(hidden, lm.lm_head.weight, labels, self.loss_impl)

    @staticmethod
    def codebook_entries(codebook, indices, shape, l2_norm=True):
        """The entry of the VQ tail: `quantize.get_codebook_entry` (llamagen_tokenizer/tokenizer_image/vq_model.py:261-277,
        reached from mamba_vlm.py:104-108 with shape [B, 8, 16, 16]): l2-normalised codebook rows of the sampled ids as a
        channel-first latent.  (The whole tail incl. the convolutional decoder: omnimamba_amd/vq_tail.py, attach_vq_tail().)"""
        emb = F.normalize(codebook, p=2, dim=-1) if l2_norm else codebook
        zq = emb[indices.reshape(-1)]
        return zq.reshape(shape[0], shape[2], shape[3], shape[1]).permute(0, 3, 1, 2).contiguous()

    # ---- MMU generation (scripts/inference_mmu.py:55-95: the prompt the script assembles by hand, then mamba.generate)
    @torch.no_grad()
    def mmu_generate(self, images_feat, input_ids, max_length=2048, eos_token_id=None, temperature=1.0, top_k=1, top_p=0.0, cg=True):
        """<|mmu|> <|soi|> [projector(images_feat)] <|eoi|> <|sot|> question ids -> the reference's greedy continuation.  Returns the
        token matrix the script decodes: the 4 + len(question) prompt ids followed by the generated ids (the image positions carry no
        ids, as in the script)."""
        ids = torch.cat([self._sp("<|mmu|>", input_ids), self._sp("<|soi|>", input_ids), self._sp("<|eoi|>", input_ids),
                         self._sp("<|sot|>", input_ids), input_ids], dim=1)
        txt = self.llm_backbone.embed_input_ids(ids)
        img = self.projector(images_feat).to(txt.dtype)
        emb = torch.cat((txt[:, :2], img, txt[:, 2:]), dim=1)
        return self.llm_backbone.mamba.generate(input_ids=ids, input_embeddings=emb, cond=None, eos_token_id=eos_token_id, max_length=max_length,
                                                temperature=temperature, top_p=top_p, top_k=top_k, cg=cg, task="mmu")

    # ---- T2I generation (omnimamba.py:311-337 minus the VQ decoder network)
    @torch.no_grad()
    def t2i_generate(self, text_ids, temperature=1.0, top_k=0, top_p=1.0, fast=True, decode_images=False, image_dtype=None):
        bb = self.backbone
        emb = bb.caption_embed(self.llm_backbone.embed_input_ids(text_ids), train=False)
        emb = emb + bb.pos_embed[:, : emb.shape[1]]
        max_length = self.llm_backbone.num_tokens + emb.shape[1]
        # graph: the sampling runs inside the captured step (generation.GreedyLoopGraph): argmax for top_k = 1 (same ids as the host
        # loop), omk_sample (top-k / temperature / top-p / Philox draw in one launch) for 1 < top_k <= 64 and for the whole vocabulary
        # (top_k = 0 without a top-p cut: this function's own defaults)
        x = decode(text_ids, emb, self.llm_backbone.mamba, max_length, top_k=top_k, top_p=top_p, temperature=temperature,
                   cg=fast, task="t2i", device_loop=fast and (1 <= top_k <= 64 or (top_k == 0 and (top_p <= 0.0 or top_p >= 1.0))))
        self.llm_backbone.mamba._decoding_cache = None
        tokens = x[: text_ids.shape[0], emb.shape[1]:]
        if not decode_images:
            return tokens
        tail = getattr(self.llm_backbone, "vqvae", None)
        if tail is None:
            raise RuntimeError("t2i_generate(decode_images=True) needs attach_vq_tail() first")
        return tail.graphed(tokens, image_dtype) if fast else tail.decode_to_img(tokens, image_dtype)   # mamba_vlm.py:104-108

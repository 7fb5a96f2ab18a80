"""Autoregressive decode around Mamba2.step (reference
/root/reference/models/stage2/generation.py:19-36 InferenceParams, :87-121 sample, :125-266 decode,
:308-434 graph cache).  Integer state -- seqlen_offset, position_ids, lengths_per_sample, the stop rule, greedy argmax
-- is reproduced exactly; the 1-token step is captured once in a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed,
which the omk_* kernels allow because they never allocate, synchronise or read host scalars.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Optional

import torch

from . import sampling as SMP


@dataclass
class InferenceParams:
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: Optional[torch.Tensor] = None

    def reset(self, max_seqlen, max_batch_size):
        self.max_seqlen = max_seqlen
        self.max_batch_size = max_batch_size
        self.seqlen_offset = 0
        if self.lengths_per_sample is not None:
            self.lengths_per_sample.zero_()


def _top_p_filter_(logits, top_p):
    if top_p <= 0.0 or top_p >= 1.0:
        return
    sorted_logits, sorted_idx = torch.sort(logits, descending=False)
    remove = sorted_logits.softmax(dim=-1).cumsum(dim=-1) <= (1 - top_p)
    logits.masked_fill_(remove.scatter(1, sorted_idx, remove), float("-inf"))


MAX_PREFILL_GRAPHS = 4


def _prefill_graph_ok(seqlen):
    """Short prompts are launch bound: capture them.  Long ones (MMU prompts) run eager -- one graph per length would pin memory."""
    import os
    return os.environ.get("OMK_PREFILL_GRAPH", "1") != "0" and seqlen <= 512


def _stream_base():
    """Philox stream position of the next sampling call: 62 random bits from torch's default generator.  The reference draws with
    torch.multinomial, which advances the global generator -- successive calls are independent, and `torch.manual_seed(s)` followed by
    the same calls reproduces them.  Drawing the position from that generator gives the device sampler both properties (a fixed
    position per call made every decode of the same prompt return the same ids; a module-level counter was not reset by re-seeding)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))


def modify_logit_for_repetition_penalty(logits, prev_output_tokens, repetition_penalty=1.0):
    """Repetition penalty of arXiv:1909.05858 as the reference applies it (generation.py:73-85): the logits (batch, vocab) of every id in
    ``prev_output_tokens`` (batch, n) are multiplied by the penalty when negative and divided by it otherwise; in place, returns logits."""
    if repetition_penalty == 1.0:
        return logits
    score = torch.gather(logits, 1, prev_output_tokens)
    score = torch.where(score < 0, score * repetition_penalty, score / repetition_penalty)
    logits.scatter_(1, prev_output_tokens, score)
    return logits


def sample(logits, top_k=1, top_p=0.0, min_p=0.0, temperature=1.0):
    """(batch, vocab) -> (batch,) token ids.  top_k == 1 is greedy argmax (what the inference scripts use).  With 1 < top_k <= 64 on
    the MI355X the draw is one omk_sample launch (top-k select, temperature, top-p, Philox inverse-CDF draw: csrc/sample.hip) seeded
    from torch's generator -- same distribution as the reference's topk + softmax + multinomial chain, no host round trip."""
    if top_k == 1:
        return logits.argmax(dim=-1)
    if top_p > 0.0:
        assert top_p <= 1.0, "top-p should be in (0, 1]."
    kk = min(top_k, logits.size(-1)) if top_k > 0 else 0
    if logits.is_cuda and SMP.applies(logits, kk, min_p=min_p if top_k <= 0 else 0.0, top_p=top_p):
        # 1 < top_k <= 64, or the whole-vocabulary branch (t2i_generate's default arguments: top_k 0, top_p 1.0; its top-p cut; its min_p filter)
        return SMP.sample_device(logits, top_k=kk, top_p=top_p, temperature=temperature, seed=torch.initial_seed(), offset=_stream_base(),
                                 min_p=min_p if top_k <= 0 else 0.0)
    if top_k > 0:
        top_k = min(top_k, logits.size(-1))
        vals, idx = torch.topk(logits, top_k, dim=-1)
        if temperature != 1.0:
            vals = vals / temperature
        _top_p_filter_(vals, top_p)
        pick = torch.multinomial(torch.softmax(vals, dim=-1), num_samples=1).squeeze(-1)
        return idx[torch.arange(idx.shape[0], device=idx.device), pick]
    if min_p > 0.0:
        # reference generation.py:107-113: the threshold max_prob * min_p (a PROBABILITY at temperature 1) is compared with
        # the raw LOGITS, the filter is skipped unless 0 < threshold < 1 everywhere, and temperature is applied afterwards
        work = logits.clone()
        thr = torch.softmax(work, dim=-1).max(dim=-1, keepdim=True)[0] * min_p
        if not ((thr <= 0.0).any() or (thr >= 1.0).any()):
            work.masked_fill_(work < thr, float("-inf"))
        if temperature != 1.0:
            work /= temperature
        return torch.multinomial(torch.softmax(work, dim=-1), num_samples=1).squeeze(-1)
    work = logits / temperature if temperature != 1.0 else logits.clone()
    _top_p_filter_(work, top_p)
    return torch.multinomial(torch.softmax(work, dim=-1), num_samples=1).squeeze(-1)


class StepGraph:
    """The captured 1-token model step: static input buffers refreshed by copy_, states updated in place."""

    def __init__(self, model, inference_params, batch_size, max_seqlen, task, n_warmups=2, mempool=None):
        dev = next(iter(model.parameters())).device
        self.ip = inference_params
        self.input_ids = torch.zeros(batch_size, 1, dtype=torch.long, device=dev)
        self.position_ids = torch.zeros(batch_size, 1, dtype=torch.long, device=dev)
        off = inference_params.seqlen_offset
        # warm-ups run the STEP branch (offset > 0); they scribble on the caches, which prefill fully overwrites later
        inference_params.seqlen_offset = max_seqlen - 1
        inference_params.lengths_per_sample[:] = inference_params.seqlen_offset

        def fwd():
            out = model(self.input_ids, None, position_ids=self.position_ids, task=task, inference_params=inference_params,
                        num_last_tokens=1)
            return (out.t2i_logits if task == "t2i" else out.mmu_logits).squeeze(1)

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(n_warmups):
                fwd()
            s.synchronize()
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                torch.distributed.barrier()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=mempool):
            self.logits = fwd()
        inference_params.seqlen_offset = off

    def run(self, new_ids, new_pos, seqlen):
        self.ip.lengths_per_sample[:] = seqlen
        self.input_ids.copy_(new_ids)
        self.position_ids.copy_(new_pos)
        self.graph.replay()
        return self.logits.clone()


class PrefillGraph:
    """The captured PREFILL of a fixed (batch, prompt length): one replay instead of ~20 eager launches per layer (the T2I prompt is
    always the same length -- caption of 73 ids, omnimamba.py:264 -- and at 72 tokens the prefill is launch bound: 16 ms eager for
    the 1.3B stack).  Static embedding buffer refreshed by copy_, the caches filled in place by the fused prefill node."""

    def __init__(self, model, inference_params, batch_size, seqlen, d_model, task, dtype, n_warmups=2, mempool=None):
        dev = next(iter(model.parameters())).device
        self.ip = inference_params
        self.emb = torch.zeros(batch_size, seqlen, d_model, dtype=dtype, device=dev)
        off = inference_params.seqlen_offset
        inference_params.seqlen_offset = 0

        def fwd():
            out = model(None, self.emb, position_ids=None, task=task, inference_params=inference_params, num_last_tokens=1)
            return (out.t2i_logits if task == "t2i" else out.mmu_logits).squeeze(1)

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(n_warmups):
                fwd()
            s.synchronize()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=mempool):
            self.logits = fwd()
        inference_params.seqlen_offset = off

    def run(self, embeddings):
        self.emb.copy_(embeddings)
        self.graph.replay()
        return self.logits.clone()


class GreedyLoopGraph:
    """Decoding with the SAMPLING ON THE DEVICE (greedy, or top-k / top-p / temperature through omk_sample): one captured graph holds the 1-token
    model step, the argmax, the write of the new id into the step's own input buffer and into the output slot, and the
    position / length counters.  The host only replays it -- no per-token copy_, clone, argmax launch or torch.cat from
    Python (the reference's loop, generation.py:239-257, does all of those per token).  Same ids as the host loop; used
    when top_k == 1 and there is no EOS test (the T2I path: exactly num_tokens codes, omnimamba.py:321)."""

    def __init__(self, model, inference_params, batch_size, max_seqlen, n_steps, task, n_warmups=2, top_k=1, top_p=0.0, temperature=1.0,
                 seed=0, min_p=0.0):
        dev = next(iter(model.parameters())).device
        self.ip = inference_params
        self.draw = torch.zeros((), dtype=torch.int64, device=dev)      # Philox stream position, advanced inside the graph
        self.input_ids = torch.zeros(batch_size, 1, dtype=torch.long, device=dev)
        self.position_ids = torch.zeros(batch_size, 1, dtype=torch.long, device=dev)
        self.slot = torch.zeros(batch_size, 1, dtype=torch.long, device=dev)
        self.tokens = torch.zeros(batch_size, max(n_steps, 1), dtype=torch.long, device=dev)
        off = inference_params.seqlen_offset
        inference_params.seqlen_offset = max_seqlen - 1        # warm-ups run the STEP branch; prefill overwrites the caches later
        inference_params.lengths_per_sample[:] = inference_params.seqlen_offset

        def step():
            out = model(self.input_ids, None, position_ids=self.position_ids, task=task, inference_params=inference_params,
                        num_last_tokens=1)
            lg = (out.t2i_logits if task == "t2i" else out.mmu_logits).squeeze(1)
            if top_k == 1:
                nxt = lg.argmax(dim=-1, keepdim=True)
            else:   # top-k / temperature / top-p / draw as ONE launch that reads its stream position from device memory
                nxt = SMP.sample_device(lg, top_k=top_k, top_p=top_p, temperature=temperature, seed=seed, step_counter=self.draw, min_p=min_p).unsqueeze(1)
                self.draw.add_(1)
            self.tokens.scatter_(1, self.slot.clamp(max=self.tokens.shape[1] - 1), nxt)
            self.input_ids.copy_(nxt)
            self.position_ids.add_(1)
            self.slot.add_(1)
            inference_params.lengths_per_sample.add_(1)

        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(n_warmups):
                self.position_ids.zero_()
                step()
            s.synchronize()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        self.position_ids.zero_()
        with torch.cuda.graph(self.graph):
            step()
        inference_params.seqlen_offset = off

    def run(self, first_ids, first_pos, n_steps, draw0=1):
        self.input_ids.copy_(first_ids)
        self.position_ids.fill_(first_pos)
        self.slot.zero_()
        self.draw.fill_(draw0)
        self.ip.lengths_per_sample[:] = first_pos
        for _ in range(n_steps):
            self.graph.replay()
        return self.tokens[:, :n_steps].clone()


@dataclass
class DecodeOutput:
    """What ``generate(return_dict_in_generate=True)`` hands back (the reference returns transformers' Greedy / Sample
    DecoderOnlyOutput, generation.py:254-255: the two fields its callers read)."""
    sequences: torch.Tensor
    scores: Optional[tuple] = None


class GenerationMixin:
    """``model.generate(...)`` as the reference's scripts call it on ``llm_backbone.mamba`` (models/stage2/generation.py:269-293;
    scripts/inference_mmu.py:84-94, omnimamba.py:322-330): a thin front of ``decode`` -- same argument names and defaults, the
    token matrix by default, ``DecodeOutput`` with ``return_dict_in_generate=True`` (scores only with ``output_scores=True``)."""

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError

    def generate(self, input_ids, input_embeddings, max_length, top_k=1, top_p=0.0, min_p=0.0, temperature=1.0, eos_token_id=None,
                 return_dict_in_generate=False, output_scores=False, cond=None, **kwargs):
        if cond is not None:
            raise NotImplementedError("cond is never passed by OmniMamba (always None: omnimamba.py:322-330, inference_mmu.py:84-94)")
        sc = [] if output_scores else None
        seqs = decode(input_ids, input_embeddings, self, max_length, top_k=top_k, top_p=top_p, min_p=min_p, temperature=temperature,
                      eos_token_id=eos_token_id, scores=sc, **kwargs)
        if not return_dict_in_generate:
            return seqs
        return DecodeOutput(sequences=seqs, scores=None if sc is None else tuple(sc))


@torch.inference_mode()
def decode(input_ids, input_embeddings, model, max_length, top_k=1, top_p=0.0, min_p=0.0, temperature=1.0,
           eos_token_id=None, teacher_outputs=None, vocab_size=None, cg=False, task="t2i", trace=None, device_loop=False, scores=None,
           repetition_penalty=1.0, enable_timing=False, streamer=None):
    """Prefill with ``input_embeddings`` (batch, prompt positions, d), then sample until ``seqlen_offset >= max_length - 1`` (or
    EOS).  Returns the token matrix (batch, input_ids.shape[1] + n_sampled): the given prompt ids followed by the sampled ids.
    ``trace`` (optional list) receives (seqlen_offset, position_id) per model call -- the integer state checked bit-exact;
    ``scores`` (optional list) the logits every sampled token was drawn from (the reference's ``output.scores``)."""
    if repetition_penalty != 1.0:
        # never passed by OmniMamba's scripts; taken through the host loop with the reference's own arithmetic (generation.py:73-85,246-252:
        # gather / scale / scatter on a clone of the logits).  The reference's branch ALSO appends every sampled id twice to the matrix it
        # returns (`sequences_cat` is extended inside the branch and again behind it): reproduced, so that a caller sees the same tensor.
        device_loop = False
    if streamer is not None:
        streamer.put(input_ids.cpu())
        device_loop = False                       # a streamer wants every id as it is sampled: host loop
    t_start = time.perf_counter() if enable_timing else None
    batch_size, seqlen_og = input_ids.shape
    # the prompt the model SEES is the embedding sequence: the reference advances seqlen_offset by its length (generation.py:236-245,
    # `sequences = [input_embeddings]`), which is longer than input_ids when image embeddings were spliced in (scripts/inference_mmu.py:
    # 4 + question ids against 4 + 729 + question positions); input_ids only lead the returned matrix
    seqlen_pr = input_embeddings.shape[1]
    dev = input_embeddings.device
    graph = None
    if hasattr(model, "prepare_decode"):
        model.prepare_decode(task)      # per-token-id tables of the embedding MLPs, built outside any graph capture
    if (device_loop and cg and (1 <= top_k <= SMP.MAX_TOP_K or (top_k == 0 and 0.0 <= min_p < 1.0)) and eos_token_id is None and teacher_outputs is None
            and vocab_size is None and trace is None and scores is None):
        return _decode_device_loop(input_ids, input_embeddings, model, max_length, task, top_k, top_p, temperature, min_p=min_p if top_k == 0 else 0.0)
    if cg:
        # the reference's cache rule (generation.py:308-369): one set of state tensors and one graph memory pool, thrown away
        # only when the device / dtype changes or a LARGER batch or max_seqlen arrives; captured steps are kept per
        # (batch, decoding length 1) -- here per task as well, because the captured step bakes the task's LoRA and head.
        # (Deviation: a SMALLER batch re-allocates too; the reference would run it on the larger state tensors.)
        cache = getattr(model, "_decoding_cache", None)
        p0 = next(iter(model.parameters()))
        if (cache is None or cache.get("kind") != "host" or (cache["device"], cache["dtype"]) != (p0.device, p0.dtype)
                or batch_size != cache["max_batch_size"] or max_length > cache["max_seqlen"]):
            ip = InferenceParams(max_seqlen=max_length, max_batch_size=batch_size, seqlen_offset=seqlen_pr,
                                 key_value_memory_dict=model.allocate_inference_cache(batch_size, max_length, p0.dtype),
                                 lengths_per_sample=torch.full((batch_size,), seqlen_pr, dtype=torch.int32, device=dev))
            cache = {"kind": "host", "device": p0.device, "dtype": p0.dtype, "max_batch_size": batch_size, "max_seqlen": max_length,
                     "ip": ip, "mempool": torch.cuda.graphs.graph_pool_handle(), "graphs": {}}
            model._decoding_cache = cache
        if (batch_size, 1, task) not in cache["graphs"]:
            cache["graphs"][batch_size, 1, task] = StepGraph(model, cache["ip"], batch_size, cache["max_seqlen"], task, mempool=cache["mempool"])
        inference_params, graph = cache["ip"], cache["graphs"][batch_size, 1, task]
        inference_params.reset(max_length, batch_size)
        pkey = ("prefill", batch_size, seqlen_pr, task, input_embeddings.dtype)
        if _prefill_graph_ok(seqlen_pr) and pkey not in cache["graphs"]:
            # at most MAX_PREFILL_GRAPHS captured prompt lengths per model (the least recently USED goes first: a hit below moves its
            # key to the end): text prompts of every length would otherwise pin a graph + static buffers each
            pkeys = [k for k in cache["graphs"] if k[0] == "prefill"]
            if len(pkeys) >= MAX_PREFILL_GRAPHS:
                del cache["graphs"][pkeys[0]]
            cache["graphs"][pkey] = PrefillGraph(model, cache["ip"], batch_size, seqlen_pr, input_embeddings.shape[-1], task,
                                                 input_embeddings.dtype, mempool=cache["mempool"])
            inference_params.reset(max_length, batch_size)
        prefill_graph = cache["graphs"].get(pkey)
        if prefill_graph is not None:
            cache["graphs"][pkey] = cache["graphs"].pop(pkey)     # most recently used last
    else:
        inference_params = InferenceParams(max_seqlen=max_length, max_batch_size=batch_size)
        prefill_graph = None

    def logits_of(out):
        lg = (out.t2i_logits if task == "t2i" else out.mmu_logits).squeeze(1)
        return lg[..., :vocab_size] if vocab_size is not None else lg

    # size of the task's position table (the reference's models cap it: 256 + 73 for T2I, 1500 for MMU, mixer_seq_simple.py:298-303).
    # A step beyond it is an out-of-range gather on the device in the reference (a device-side assert); here the host loop knows the
    # position and says so before anything is launched.
    cfg_ = getattr(model, "cfg", None)
    n_pos = None if cfg_ is None else getattr(cfg_, "t2i_positions" if task == "t2i" else "mmu_positions", None)

    def get_logits(tokens, embeddings):
        decoding = inference_params.seqlen_offset > 0
        if decoding and n_pos is not None and inference_params.seqlen_offset >= n_pos:
            raise IndexError(f"decode: position {inference_params.seqlen_offset} is outside the {task} position table of {n_pos} rows "
                             "(StackConfig.{t2i,mmu}_positions; the reference's table has the same size)")
        pos = torch.full((batch_size, 1), inference_params.seqlen_offset, dtype=torch.long, device=dev) if decoding else None
        if trace is not None:
            trace.append((inference_params.seqlen_offset, None if pos is None else int(pos[0, 0])))
        if graph is not None and decoding:
            lg = graph.run(tokens, pos, inference_params.seqlen_offset)
            return lg[..., :vocab_size] if vocab_size is not None else lg
        if prefill_graph is not None and not decoding:
            lg = prefill_graph.run(embeddings)
            return lg[..., :vocab_size] if vocab_size is not None else lg
        return logits_of(model(tokens, embeddings, position_ids=pos, task=task, inference_params=inference_params,
                               num_last_tokens=1))

    def should_stop(cur):
        if inference_params.seqlen_offset == 0:
            return False
        if eos_token_id is not None and bool((cur == eos_token_id).all()):
            return True
        return inference_params.seqlen_offset >= max_length - 1

    seqs = input_ids
    last, first, n_in = None, True, seqlen_pr
    while not should_stop(last):
        lg = get_logits(None, input_embeddings) if first else get_logits(last, None)
        inference_params.seqlen_offset += n_in
        first, n_in = False, 1
        if scores is not None:
            scores.append(lg.clone() if graph is not None else lg)     # (a captured step returns its static output buffer)
        if repetition_penalty != 1.0:
            lg = modify_logit_for_repetition_penalty(lg.clone(), seqs, repetition_penalty)
        if teacher_outputs is not None and teacher_outputs.shape[1] > inference_params.seqlen_offset:
            tok = teacher_outputs[:, inference_params.seqlen_offset]
        else:
            tok = sample(lg, top_k=top_k, top_p=top_p, min_p=min_p, temperature=temperature)
        last = tok.unsqueeze(1)
        if repetition_penalty != 1.0:
            seqs = torch.cat([seqs, last], dim=1)      # (the reference's double append, see above)
        seqs = torch.cat([seqs, last], dim=1)
        if streamer is not None:
            streamer.put(last.cpu())
    if streamer is not None:
        streamer.end()
    if enable_timing:
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        print(f"Prompt processing + decoding time: {(time.perf_counter() - t_start) * 1e3:.0f}ms")     # (generation.py:262-264)
    return seqs


def _decode_device_loop(input_ids, input_embeddings, model, max_length, task, top_k=1, top_p=0.0, temperature=1.0, min_p=0.0):
    """Prefill (eager, fills the caches), then max_length - P - 1 replays of GreedyLoopGraph."""
    batch_size, seqlen_pr = input_ids.shape[0], input_embeddings.shape[1]   # (the embedding sequence is the prompt: see decode)
    dev = input_embeddings.device
    n_steps = max_length - 1 - seqlen_pr
    cfg_ = getattr(model, "cfg", None)
    n_pos = None if cfg_ is None else getattr(cfg_, "t2i_positions" if task == "t2i" else "mmu_positions", None)
    if n_pos is not None and max_length - 1 > n_pos:   # the loop runs to its end on the device: every position must be in the table
        raise IndexError(f"decode: max_length {max_length} runs past the {task} position table of {n_pos} rows")
    cache = getattr(model, "_decoding_cache", None)
    seed = torch.initial_seed()
    key = ("device_loop", batch_size, max_length, n_steps, task, top_k, top_p, temperature, min_p, seed if top_k != 1 else 0)
    if cache is None or cache.get("kind") != "device_loop" or cache.get("key") != key:
        dtype = next(iter(model.parameters())).dtype
        ip = InferenceParams(max_seqlen=max_length, max_batch_size=batch_size, seqlen_offset=seqlen_pr,
                             key_value_memory_dict=model.allocate_inference_cache(batch_size, max_length, dtype),
                             lengths_per_sample=torch.full((batch_size,), seqlen_pr, dtype=torch.int32, device=dev))
        cache = {"kind": "device_loop", "key": key, "ip": ip,
                 "graph": GreedyLoopGraph(model, ip, batch_size, max_length, n_steps, task, top_k=top_k, top_p=top_p, temperature=temperature, seed=seed, min_p=min_p)}
        if _prefill_graph_ok(seqlen_pr):
            ip.reset(max_length, batch_size)
            cache["prefill"] = PrefillGraph(model, ip, batch_size, seqlen_pr, input_embeddings.shape[-1], task, input_embeddings.dtype)
        model._decoding_cache = cache
    ip, graph = cache["ip"], cache["graph"]
    ip.reset(max_length, batch_size)
    pg = cache.get("prefill")
    if pg is not None and pg.emb.shape == input_embeddings.shape and pg.emb.dtype == input_embeddings.dtype:
        lg0 = pg.run(input_embeddings)
    else:
        out = model(None, input_embeddings, position_ids=None, task=task, inference_params=ip, num_last_tokens=1)
        lg0 = (out.t2i_logits if task == "t2i" else out.mmu_logits).squeeze(1)
    base = 0 if top_k == 1 else _stream_base()   # this call's stretch of the Philox stream: base, base + 1, ... (one position per token)
    first = (lg0.argmax(dim=-1, keepdim=True) if top_k == 1 else
             SMP.sample_device(lg0, top_k=top_k, top_p=top_p, temperature=temperature, seed=seed, offset=base, min_p=min_p).unsqueeze(1))
    ip.seqlen_offset = seqlen_pr
    seqs = torch.cat([input_ids, first], dim=1)
    if n_steps > 0:
        seqs = torch.cat([seqs, graph.run(first, seqlen_pr, n_steps, draw0=base + 1)], dim=1)
        ip.seqlen_offset = seqlen_pr + n_steps
    return seqs

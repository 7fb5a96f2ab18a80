"""Rows a1/a3/a12/a14 of SURVEY.md section 8 on a toy model, through the emulated kernels (CPU) and on the MI355X (-m gpu):
stack forward == oracle composition, LoRA == the reference's own lora.Linear (golden), greedy decode == argmax of the
full forward with a bit-exact integer trace, 2-rank DDP (gloo) grads == single-process grads."""
import os

import numpy as np
import pytest
import torch

import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30)).item()


def tiny_cfg():
    from omnimamba_amd.stack import StackConfig
    return StackConfig(d_model=32, n_layer=2, vocab_size=50, pad_vocab_size_multiple=16, vqvae_vocab_size=40, num_tokens=8,
                       t2i_positions=24, mmu_positions=40, ssm_cfg=dict(d_state=16, headdim=8, chunk_size=16), lora_dropout=0.0,
                       img_sq_len=5, fused_vision_dim=12)


TINY_SPECIAL = {"<|soi|>": 51, "<|eoi|>": 52, "<|sot|>": 53, "<|mmu|>": 56}   # inside the 64 padded rows of the toy vocabulary


def tiny_path(stage="finetune"):
    from omnimamba_amd.omni import OmniMambaPath
    return OmniMambaPath(tiny_cfg(), stage=stage, special_ids=TINY_SPECIAL)


def test_lora_matches_reference_golden():
    """TaskLoRALinear vs outputs captured from the reference's own models/stage2/lora.py (tests/golden/make_golden.py)."""
    from omnimamba_amd.stack import TaskLoRALinear
    g = np.load(os.path.join(GOLD, "lora_reference.npz"))
    m = TaskLoRALinear(12, 20, r=8, lora_alpha=32, lora_dropout=0.05).eval()
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")}
    assert sorted(sd) == sorted(m.state_dict().keys())
    m.load_state_dict(sd)
    x = torch.from_numpy(g["x"])
    for task in ("t2i", "mmu"):
        m.task_types = task
        assert torch.allclose(m(x), torch.from_numpy(g[f"y_{task}"]), atol=1e-6)


def test_stack_forward_matches_oracle(dev):
    from omnimamba_amd.stack import OmniMambaLM
    torch.manual_seed(0)
    cfg = tiny_cfg()
    model = OmniMambaLM(cfg).to(dev).eval()
    emb = torch.randn(2, 11, 32)
    out = model(None, emb.to(dev), task="mmu").mmu_logits
    # oracle composition
    h = emb + model.backbone.mmu_pos_embed[:, :11].detach().cpu()
    res = None
    for blk in model.backbone.layers:
        y, res = O.add_norm_ref(h, blk.norm.weight.detach().cpu(), None, residual=res, eps=1e-5, prenorm=True, residual_in_fp32=True, is_rms_norm=True)
        mx = blk.mixer
        p = O.Mamba2RefParams(in_proj_weight=mx.in_proj.weight.detach().cpu(), conv_weight=mx.conv1d.weight.detach().cpu().squeeze(1),
                              conv_bias=mx.conv1d.bias.detach().cpu(), dt_bias=mx.dt_bias.detach().cpu(), A_log=mx.A_log.detach().cpu(),
                              D=mx.D.detach().cpu(), norm_weight=mx.norm.weight.detach().cpu(), out_proj_weight=mx.out_proj.weight.detach().cpu(),
                              headdim=8, d_state=16, chunk_size=16)
        h = O.mamba2_forward_ref(p, y)      # LoRA B is zero-initialised: adapters contribute nothing at init
    hf = O.add_norm_ref(h, model.backbone.norm_f.weight.detach().cpu(), None, residual=res, eps=1e-5, prenorm=False, residual_in_fp32=True, is_rms_norm=True)
    ref = hf @ model.lm_head.weight.detach().cpu().t()
    assert rel(out, ref) < 1e-4


def test_greedy_decode_trace_and_tokens(dev):
    """decode(): prefill + steps must give exactly the argmax tokens of the full (no-cache) forward, with the reference's
    integer trace: offsets 0, P, P+1, ... ; position_ids = offset ; exactly max_length - P sampled tokens."""
    from omnimamba_amd.generation import decode
    from omnimamba_amd.stack import OmniMambaLM
    torch.manual_seed(1)
    cfg = tiny_cfg()
    model = OmniMambaLM(cfg).to(dev).eval()
    with torch.no_grad():
        model.backbone.img_embeddings.word_embeddings.weight.mul_(30.0)   # make logits well separated so argmax is robust
    B, Pn, max_len = 2, 5, 12
    prompt_ids = torch.zeros(B, Pn, dtype=torch.long, device=dev)
    prompt_emb = torch.randn(B, Pn, 32).to(dev)
    trace = []
    seqs = decode(prompt_ids, prompt_emb, model, max_len, top_k=1, task="t2i", trace=trace)
    assert seqs.shape == (B, max_len)
    assert [t[0] for t in trace] == [0] + list(range(Pn, max_len - 1)) and [t[1] for t in trace] == [None] + list(range(Pn, max_len - 1))
    # teacher-forced full forward over the generated ids reproduces every greedy choice
    toks = seqs[:, Pn:]
    with torch.no_grad():
        # decode steps add pos_embed[position] to the image-token embeddings; the prompt embeddings arrive with theirs
        # already added by the caller (reference omnimamba.py:319-320)
        emb_gen = model.backbone.img_embeddings(toks[:, :-1]) + model.backbone.pos_embed[:, Pn:max_len - 1]
        full = model(None, torch.cat([prompt_emb, emb_gen], 1), task="t2i").t2i_logits
    assert torch.equal(full[:, Pn - 1:].argmax(-1).cpu(), toks.cpu())


def test_decode_past_the_position_table_raises_on_the_host(dev):
    """The reference's position tables are finite (256 + 73 rows for T2I, 1500 for MMU, mixer_seq_simple.py:298-303) and a decode
    step beyond them is an out-of-range gather on the device there.  Here the host loop raises IndexError before the step is launched
    (found on the MI355X: a 1500-token fp32 MMU prompt + 2 new tokens aborted the queue with a hardware exception)."""
    from omnimamba_amd.generation import decode
    from omnimamba_amd.stack import OmniMambaLM
    torch.manual_seed(2)
    cfg = tiny_cfg()
    model = OmniMambaLM(cfg).to(dev).eval()
    Pn = cfg.mmu_positions - 1
    ids = torch.zeros(1, Pn, dtype=torch.long, device=dev)
    emb = torch.randn(1, Pn, cfg.d_model).to(dev)
    out = decode(ids, emb, model, Pn + 2, top_k=1, task="mmu")            # positions Pn (= table size - 1): the last legal step
    assert out.shape == (1, Pn + 2)
    with pytest.raises(IndexError):
        decode(ids, emb, model, Pn + 3, top_k=1, task="mmu")              # one step further


@pytest.mark.parametrize("task,Bsz", [("t2i", 1), ("mmu", 1), ("t2i", 3)])
def test_fused_decode_step_equals_unfused(dev, task, Bsz, monkeypatch):
    """The one-launch add + norm + in_proj + LoRA and gated-norm + out_proj of the decode step (omk_norm_linear) against the
    same step through the separate ops, on a stack wide enough for the fused kernel to apply (d_model 1024)."""
    from omnimamba_amd import norm_linear as NL
    from omnimamba_amd.generation import InferenceParams
    from omnimamba_amd.stack import OmniMambaLM, StackConfig
    cfg = StackConfig(d_model=1024, n_layer=1, vocab_size=50, pad_vocab_size_multiple=16, vqvae_vocab_size=40, num_tokens=8,
                      t2i_positions=24, mmu_positions=40, ssm_cfg=dict(d_state=16, headdim=64, chunk_size=16), lora_dropout=0.05)
    torch.manual_seed(0)
    model = OmniMambaLM(cfg).to(dev).eval()
    with torch.no_grad():
        for blk in model.backbone.layers:       # non-trivial adapters
            for t in ("t2i", "mmu"):
                getattr(blk.mixer.in_proj, f"{t}_lora_B0").weight.normal_(std=0.05)
    emb = torch.randn(Bsz, 6, cfg.d_model).to(dev)
    calls = {"n": 0}
    real = NL.norm_linear

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)

    outs = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setattr(NL, "applies", lambda *a, **k: False)
        monkeypatch.setattr(NL, "norm_linear", counting)
        ip = InferenceParams(max_seqlen=32, max_batch_size=Bsz)
        with torch.no_grad():
            model(None, emb, task=task, inference_params=ip, num_last_tokens=1)
            ip.seqlen_offset = 6
            ids = torch.full((Bsz, 1), 3).to(dev)
            pos = torch.full((Bsz, 1), 6, dtype=torch.long).to(dev)
            logits = []
            for step in range(2):
                o = model(ids, None, position_ids=pos + step, task=task, inference_params=ip, num_last_tokens=1)
                ip.seqlen_offset += 1
                logits.append(o.t2i_logits if task == "t2i" else o.mmu_logits)
        states = [ip.key_value_memory_dict[i][1].clone() for i in range(cfg.n_layer)]
        outs.append((torch.cat(logits, 1), states))
    assert calls["n"] == 2 * cfg.n_layer * 2          # pre-norm + in_proj and gated norm + out_proj, every layer, every step
    assert rel(outs[0][0], outs[1][0]) < 2e-5
    for a, b in zip(outs[0][1], outs[1][1]):
        assert rel(a, b) < 2e-5


@pytest.mark.gpu
def test_decode_hipgraph_equals_eager():
    from omnimamba_amd.generation import decode
    from omnimamba_amd.stack import OmniMambaLM
    dev = torch.device("cuda:0")
    torch.manual_seed(2)
    model = OmniMambaLM(tiny_cfg()).to(dev).eval()
    with torch.no_grad():
        model.backbone.img_embeddings.word_embeddings.weight.mul_(30.0)
    ids, emb = torch.zeros(2, 5, dtype=torch.long, device=dev), torch.randn(2, 5, 32, device=dev)
    a = decode(ids, emb, model, 14, top_k=1, task="t2i", cg=False)
    b = decode(ids, emb, model, 14, top_k=1, task="t2i", cg=True)
    c = decode(ids, emb, model, 14, top_k=1, task="t2i", cg=True)    # replay of the cached graph, states reset by prefill
    assert torch.equal(a, b) and torch.equal(a, c)


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from emu.loader import use_emulator
    from omnimamba_amd.train import Stage2Step, TrainConfig, init_distributed, shard_batch, synthetic_batch, wrap_ddp
    torch.set_num_threads(1)
    with use_emulator():
        init_distributed("gloo")
        torch.manual_seed(0)
        cfg = tiny_cfg()
        model = tiny_path("finetune")
        tc = TrainConfig(amp_dtype=torch.float32, clip=0.0, lr=0.0)
        step = Stage2Step(model, tc, ddp_model=wrap_ddp(model, tc))
        full = synthetic_batch(cfg, 2 * world, 14, "cpu", torch.float32, rank=0, caption_len=4)
        step(shard_batch(full, rank, world))
        grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        if rank == 0:
            q.put({k: v.numpy() for k, v in grads.items()})
        dist.barrier()
        dist.destroy_process_group()


def test_ddp_gloo_two_ranks_equals_single_process():
    """Row a14 / section 8e: gradient of the mean loss over the global batch, 2 ranks x 2 samples (gloo, CPU, emulated
    kernels) == 1 process x 4 samples."""
    import torch.multiprocessing as mp
    from emu.loader import use_emulator
    from omnimamba_amd.train import Stage2Step, TrainConfig, synthetic_batch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    with use_emulator():
        torch.manual_seed(0)
        cfg = tiny_cfg()
        model = tiny_path("finetune")
        tc = TrainConfig(amp_dtype=torch.float32, clip=0.0, lr=0.0)
        Stage2Step(model, tc)(synthetic_batch(cfg, 4, 14, "cpu", torch.float32, rank=0, caption_len=4))
        ref = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert sorted(ref) == sorted(got)
    assert sorted(ref) == sorted(n for n, p in model.named_parameters() if p.requires_grad)   # nothing trainable is left unused
    for n in ref:
        assert rel(torch.from_numpy(got[n]), ref[n]) < 1e-4, n


def test_image_token_embedding_table_for_decode():
    """The image-token embedding is a function of the token id: at decode (one id per sequence, eval, no grad) a per-id table built
    by prepare_decode() replaces the MLP; it equals the MLP, is ignored once a parameter changed, and never enters a state dict."""
    from omnimamba_amd.stack import ImageTokenEmbeddings
    torch.manual_seed(0)
    m = ImageTokenEmbeddings(32, 50).eval()
    ids = torch.randint(0, 50, (3, 1))
    with torch.no_grad():
        ref = m(ids)
        m.prepare_decode()
        assert m._omk_tab.shape == (50, 32)
        got = m(ids)
        assert torch.allclose(got, ref, atol=1e-6) and got.shape == ref.shape
        long_ids = torch.randint(0, 50, (2, 5))
        assert torch.allclose(m(long_ids), m.project_in(m.word_embeddings(long_ids)))       # prefill: the MLP
        m.project_in.projector[0].weight.mul_(2.0)                                          # stale table: the MLP again
        assert torch.allclose(m(ids), m.project_in(m.word_embeddings(ids)))
        m.prepare_decode()
        assert torch.allclose(m(ids), m.project_in(m.word_embeddings(ids)), atol=1e-6)
    assert all("_omk" not in k for k in m.state_dict())
    m.train()
    assert m(ids).requires_grad                                                             # training: never the table


def test_mmu_generate_assembles_the_scripts_prompt(dev):
    """scripts/inference_mmu.py:55-95: <|mmu|> <|soi|> [image embeddings] <|eoi|> <|sot|> question -> mamba.generate(task='mmu'); the ids
    are the ones decode() gives for the hand-assembled embeddings, the prompt ids lead the returned matrix."""
    from omnimamba_amd.generation import decode
    torch.manual_seed(4)
    model = tiny_path("inference").to(dev)
    q = torch.randint(0, 50, (2, 5), device=dev)
    feat = torch.randn(2, 5, 12, device=dev)
    got = model.mmu_generate(feat, q, max_length=24, cg=False)
    ids = torch.cat([torch.full((2, 1), TINY_SPECIAL[k], device=dev) for k in ("<|mmu|>", "<|soi|>", "<|eoi|>", "<|sot|>")] + [q], dim=1)
    txt = model.llm_backbone.embed_input_ids(ids)
    emb = torch.cat((txt[:, :2], model.projector(feat), txt[:, 2:]), dim=1)
    trace = []
    want = decode(ids, emb, model.llm_backbone.mamba, 24, top_k=1, task="mmu", cg=False, trace=trace)
    assert torch.equal(got[:, :9], ids) and torch.equal(got, want)
    # the model saw 4 + 5 image + 5 question = 14 prompt positions: offsets / position ids continue from 14 (generation.py:236-245 of the
    # reference advances by the EMBEDDING length), and 24 - 1 - 14 + 1 = 10 ids are sampled behind the 9 prompt ids
    assert trace == [(0, None)] + [(o, o) for o in range(14, 23)] and got.shape[1] == 9 + 10


def test_llm_backbone_forward_is_the_references_shifted_logits(dev):
    """mamba_vlm.py:88-102: (embeddings, labels) -> (logits[..., :-1, :] flattened, labels[..., 1:] flattened); its cross-entropy is the
    loss OmniMambaPath.forward forms without materialising the logits."""
    torch.manual_seed(6)
    model = tiny_path("inference").to(dev)
    ids, cap = torch.randint(0, 40, (2, 8), device=dev), torch.randint(0, 50, (2, 6), device=dev)
    emb, labels = model.t2i_sequence(ids, cap)
    with torch.no_grad():
        logits, target = model.llm_backbone(emb, labels, cond=None, task="t2i")
        full = model.llm_backbone.mamba(None, emb, task="t2i").t2i_logits
    assert logits.shape == (2 * (emb.shape[1] - 1), full.shape[-1]) and target.shape == (2 * (emb.shape[1] - 1),)
    assert torch.equal(logits, full[:, :-1].reshape(-1, full.shape[-1])) and torch.equal(target, labels[:, 1:].reshape(-1))
    with pytest.raises(RuntimeError):
        model.llm_backbone.decode_to_img(ids)


def test_mixed_mmu_batch_text_only_rows_get_zero_image_embeddings(dev):
    """omnimamba.py:281-301: `multimodal_indices` names the rows with an image; the loss of the mixed batch is the token mean over the
    image rows run with their features and the text-only rows run with zero image embeddings."""
    torch.manual_seed(8)
    model = tiny_path("inference").to(dev)
    ids, labels = torch.randint(0, 50, (3, 6), device=dev), torch.randint(0, 50, (3, 6), device=dev)
    feat = torch.randn(3, 5, 12, device=dev)
    with torch.no_grad():
        mixed = model({"mmu_flow": {"images_feat": feat, "input_ids": ids, "labels": labels, "multimodal_indices": torch.tensor([0, 2])}}, task="mmu")
        emb, lab = model.mmu_sequence(feat, ids, labels, torch.tensor([0, 2]))
        e_img, _ = model.mmu_sequence(feat[[0, 2]], ids[[0, 2]], labels[[0, 2]])
        e_txt, _ = model.mmu_sequence(None, ids[[1]], labels[[1]])
        none = model({"mmu_flow": {"images_feat": feat, "input_ids": ids, "labels": labels, "multimodal_indices": torch.tensor([], dtype=torch.long)}}, task="mmu")
        text = model({"mmu_flow": {"images_feat": None, "input_ids": ids, "labels": labels}}, task="mmu")
    assert torch.allclose(emb[[0, 2]], e_img) and torch.allclose(emb[[1]], e_txt) and float(emb[1, 2:7].abs().max()) == 0.0
    assert torch.isfinite(mixed) and rel(none, text) < 1e-6


def test_resize_token_embeddings_and_pretrained_directory(tmp_path):
    """mixer_seq_simple.py:526-676: save_pretrained / from_pretrained through a directory; resize_token_embeddings pads, keeps the old
    rows, re-ties the head, and is a no-op at the same padded size (the shipped checkpoints' case, omnimamba.py:103)."""
    from omnimamba_amd.stack import OmniMambaLM
    torch.manual_seed(9)
    m = OmniMambaLM(tiny_cfg())
    assert m.get_output_embeddings() is m.lm_head and m.lm_head.weight is m.get_input_embeddings().weight
    old = m.get_input_embeddings()
    w0 = old.weight.detach().clone()
    assert m.resize_token_embeddings(60, pad_to_multiple_of=16) is old            # 60 -> 64 = the table's size: untouched
    new = m.resize_token_embeddings(70, pad_to_multiple_of=16)
    assert new.weight.shape == (80, 32) and torch.equal(new.weight[:64], w0) and m.lm_head.weight is new.weight
    assert m.cfg.vocab_size == 80 and float(new.weight[64:].std()) < 0.05
    m.save_pretrained(tmp_path / "ck")
    m2 = OmniMambaLM.from_pretrained(str(tmp_path / "ck"))
    assert list(m2.state_dict().keys()) == list(m.state_dict().keys())
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    with pytest.raises(FileNotFoundError):
        OmniMambaLM.from_pretrained("state-spaces/mamba2-1.3b")

"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only (it imports the reference's
own Python from /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

  lora_reference.npz   inputs / state dict / outputs of the reference's models/stage2/lora.py::Linear (both task modes,
                       eval mode) -- pins omnimamba_amd.stack.TaskLoRALinear
  oracle_ops.npz       seeded input/output vectors of the CPU oracle for every op on the path (fp32) -- pins the oracle
                       against drift and is what the HIP kernels are compared with in tests/test_golden.py
  reference_model.npz  the reference's UNMODIFIED models/stage2/{block.py, lora.py, mixer_seq_simple.py, generation.py}
                       instantiated and run here on the oracle-backed `mamba_ssm` provider (oracle/provider.py -- plain
                       PyTorch, no kernel under test): state dict of a 2-layer MambaLMHeadModel, Block.forward
                       (hidden, residual), logits of both tasks, and the greedy generate() id sequence with the
                       (seqlen_offset, position_ids) trace of every model call (SURVEY.md section 8c ii-iv)
  vq_tail_reference.npz  the reference's UNMODIFIED llamagen_tokenizer/tokenizer_image/vq_model.py classes (VectorQuantizer, Decoder at
                       a small width, a 1x1 post_quant_conv) composed the way VQModel.decode_code composes them: state dict, sampled
                       ids, the codebook entries and the decoded image -- pins omnimamba_amd.vq_tail.VQDecodeTail
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402


def lora_reference():
    path = "/root/reference/models/stage2/lora.py"
    spec = importlib.util.spec_from_file_location("ref_lora", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_lora"] = mod
    # lora.py imports its config module relatively; load it under the expected package name
    cfg_spec = importlib.util.spec_from_file_location("ref_lora_config", "/root/reference/models/stage2/lora_config.py")
    try:
        spec.loader.exec_module(mod)
    except ImportError:
        import types
        pkg = types.ModuleType("models"); pkg.__path__ = ["/root/reference/models"]
        sub = types.ModuleType("models.stage2"); sub.__path__ = ["/root/reference/models/stage2"]
        sys.modules.update({"models": pkg, "models.stage2": sub})
        spec = importlib.util.spec_from_file_location("models.stage2.lora", path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules["models.stage2.lora"] = mod
        spec.loader.exec_module(mod)
    torch.manual_seed(0)
    m = mod.Linear(12, 20, r=8, lora_alpha=32, lora_nums=1, lora_dropout=0.05, bias=False)
    m.eval()   # the reference's eval() override returns None
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn_like(p) * 0.1)     # B is zero-initialised: make the adapters visible
    x = torch.randn(3, 5, 12)
    out = {"x": x.numpy()}
    for task in ("t2i", "mmu"):
        m.task_types = task
        out[f"y_{task}"] = m(x).detach().numpy()
    for k, v in m.state_dict().items():
        out["sd." + k] = v.numpy()
    np.savez(os.path.join(HERE, "lora_reference.npz"), **out)


def oracle_ops():
    g = torch.Generator().manual_seed(20260928)
    r = lambda *s: torch.randn(*s, generator=g)
    out = {}
    # SSD: non-multiple L, groups, D, z, dt_bias, initial state, final state
    Bsz, L, H, P, N, G = 1, 75, 4, 8, 16, 2
    x, dt, A = r(Bsz, L, H, P), r(Bsz, L, H) * 0.5, -(torch.rand(H, generator=g) * 4 + 0.5)
    Bm, Cm, D, z, dtb, init = r(Bsz, L, G, N), r(Bsz, L, G, N), r(H), r(Bsz, L, H, P), r(H) * 0.5 - 1, r(Bsz, H, P, N)
    y, fin = O.ssd_ref_sequential(x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dtb, initial_states=init, dt_softplus=True, return_final_states=True)
    for k, v in dict(x=x, dt=dt, A=A, B=Bm, C=Cm, D=D, z=z, dt_bias=dtb, init=init, y=y, fin=fin).items():
        out["ssd." + k] = v.numpy()
    # conv1d + update
    xc, w, b = r(2, 12, 19), r(12, 4), r(12)
    out.update({"conv.x": xc.numpy(), "conv.w": w.numpy(), "conv.b": b.numpy(),
                "conv.y": O.causal_conv1d_ref(xc, w, b, activation="silu").numpy()})
    # state update continuing from the SSD final state (prefill <-> decode consistency)
    st = fin.clone()
    xs, dts, Bs, Cs = r(Bsz, H, P), r(Bsz, H), r(Bsz, G, N), r(Bsz, G, N)
    ys = O.selective_state_update_ref(st, xs, dts[..., None].expand(Bsz, H, P), A[:, None, None].expand(H, P, N), Bs, Cs,
                                      D=D[:, None].expand(H, P), dt_bias=dtb[:, None].expand(H, P), dt_softplus=True)
    out.update({"su.x": xs.numpy(), "su.dt": dts.numpy(), "su.B": Bs.numpy(), "su.C": Cs.numpy(), "su.y": ys.numpy(), "su.state": st.numpy()})
    # norms
    xn, zn, wn, rn = r(5, 64), r(5, 64), r(64), r(5, 64)
    out.update({"norm.x": xn.numpy(), "norm.z": zn.numpy(), "norm.w": wn.numpy(), "norm.res": rn.numpy(),
                "norm.gated": O.rmsnorm_gated_ref(xn, wn, None, zn, eps=1e-5, group_size=32, norm_before_gate=False).numpy(),
                "norm.add": O.add_norm_ref(xn, wn, None, residual=rn, eps=1e-5, prenorm=False, is_rms_norm=True).numpy()})
    # Mamba-1 selective scan, grouped B/C
    u, dl, A1 = r(2, 6, 33), torch.rand(2, 6, 33, generator=g) * 0.5, -(torch.rand(6, 4, generator=g) + 0.1)
    Bg, Cg, D1, z1, db = r(2, 2, 4, 33), r(2, 2, 4, 33), r(6), r(2, 6, 33), r(6) * 0.1
    o1, last = O.selective_scan_ref(u, dl, A1, Bg, Cg, D1, z1, db, True, True)
    out.update({"ss.u": u.numpy(), "ss.delta": dl.numpy(), "ss.A": A1.numpy(), "ss.B": Bg.numpy(), "ss.C": Cg.numpy(), "ss.D": D1.numpy(),
                "ss.z": z1.numpy(), "ss.db": db.numpy(), "ss.out": o1.numpy(), "ss.last": last.numpy()})
    np.savez_compressed(os.path.join(HERE, "oracle_ops.npz"), **out)


def reference_model():
    """captures from the reference's own classes.  `transformers` 5.x dropped the two
    output classes generation.py:16 imports (the reference pins 4.46.1): they are injected as plain containers."""
    import oracle.provider as P
    import transformers.generation as TG
    sys.path.insert(0, "/root/reference")
    P.install()

    class _Out:
        def __init__(self, sequences=None, scores=None):
            self.sequences, self.scores = sequences, scores
    for n in ("GreedySearchDecoderOnlyOutput", "SampleDecoderOnlyOutput"):
        if not hasattr(TG, n):
            setattr(TG, n, _Out)
    from models.stage2.config_mamba import MambaConfig
    from models.stage2.mixer_seq_simple import MambaLMHeadModel

    torch.manual_seed(20260928)
    cfg = MambaConfig(d_model=32, n_layer=2, vqvae_vocab_size=40, num_tokens=8, vocab_size=50,
                      ssm_cfg={"layer": "Mamba2", "d_state": 16, "headdim": 8, "chunk_size": 16}, t2i_task=True, mmu_task=True)
    model = MambaLMHeadModel(cfg)
    model.eval()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn_like(p) * 0.05)          # zero-initialised upstream: make both adapters visible
        model.backbone.img_embeddings.word_embeddings.weight.mul_(30.0)   # well separated logits: robust argmax
        model.backbone.embedding.weight.mul_(6.0)
    out = {"sd." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    out["cfg"] = np.array([32, 2, 40, 8, 50, 16, 8, 16])   # d_model, n_layer, vq vocab, num_tokens, vocab, d_state, headdim, chunk

    # (ii) Block.forward: first block (residual None) and second block (fp32 residual in)
    h = torch.randn(2, 11, 32)
    with torch.no_grad():
        model.backbone.set_lora_mode("mmu")
        h1, r1 = model.backbone.layers[0](h, None, None, task="mmu")
        h2, r2 = model.backbone.layers[1](h1, r1, None, task="mmu")
    out.update({"block.h": h.numpy(), "block.h1": h1.numpy(), "block.r1": r1.numpy(), "block.h2": h2.numpy(), "block.r2": r2.numpy()})

    # (iv) MambaLMHeadModel logits, both tasks (layer order, LoRA switch, final norm, tied heads)
    emb = torch.randn(2, 19, 32)
    with torch.no_grad():
        out["fwd.emb"] = emb.numpy()
        out["fwd.t2i_logits"] = model(None, emb, task="t2i").t2i_logits.numpy()
        out["fwd.mmu_logits"] = model(None, emb, task="mmu").mmu_logits.numpy()

    # (iii) greedy generate(): ids + the integer trace of every model call
    # (last case, drawn after the others so that their numbers do not move: FEWER prompt ids than prompt positions, the shape of
    # scripts/inference_mmu.py -- 4 + question ids against 4 + image + question embeddings; decode() advances by the embedding length)
    for task, Bsz, Pn, max_len, n_ids in (("t2i", 2, 5, 5 + 8, 5), ("mmu", 1, 7, 16, 7), ("mmu_spliced", 1, 9, 16, 4)):
        key, task = task, task.split("_")[0]
        trace = []
        orig = model.backbone.forward

        def spy(input_ids, input_embeddings, position_ids, cond, task, inference_params=None, **kw):
            trace.append((inference_params.seqlen_offset, -1 if position_ids is None else int(position_ids[0, 0]),
                          -1 if input_ids is None else int(input_ids.shape[1])))
            return orig(input_ids, input_embeddings, position_ids, cond, task, inference_params=inference_params, **kw)
        model.backbone.forward = spy
        ids = torch.zeros(Bsz, n_ids, dtype=torch.long)
        pemb = torch.randn(Bsz, Pn, 32)
        res = model.generate(input_ids=ids, input_embeddings=pemb, cond=None, max_length=max_len, temperature=1.0, top_p=0.0,
                             top_k=1, cg=False, task=task, return_dict_in_generate=True, output_scores=True)
        model.backbone.forward = orig
        out.update({f"gen.{key}.prompt_emb": pemb.numpy(), f"gen.{key}.sequences": res.sequences.numpy(),
                    f"gen.{key}.trace": np.array(trace, dtype=np.int64), f"gen.{key}.max_length": np.array(max_len),
                    f"gen.{key}.scores": torch.stack(res.scores, 1).numpy()})
        top2 = torch.stack(res.scores, 1).topk(2, dim=-1).values
        print(task, "ids", res.sequences.tolist(), "min top-1/top-2 logit margin", float((top2[..., 0] - top2[..., 1]).min()))
    # (v) round 5: the reference's repetition-penalty branch (generation.py:73-85,246-252), drawn after everything else: greedy ids under a
    # penalty of 1.3 on the t2i prompt above -- including its quirk of appending every sampled id TWICE to the returned matrix
    pemb = torch.from_numpy(out["gen.t2i.prompt_emb"])
    ids = torch.zeros(pemb.shape[0], pemb.shape[1], dtype=torch.long)
    res = model.generate(input_ids=ids, input_embeddings=pemb, cond=None, max_length=int(out["gen.t2i.max_length"]), temperature=1.0, top_p=0.0,
                         top_k=1, cg=False, task="t2i", repetition_penalty=1.3, return_dict_in_generate=True, output_scores=True)
    out["gen.t2i_rep.sequences"] = res.sequences.numpy()
    out["gen.t2i_rep.penalty"] = np.array(1.3)
    print("t2i with repetition penalty 1.3: ids", res.sequences.tolist())
    np.savez_compressed(os.path.join(HERE, "reference_model.npz"), **out)
    print("reference_model.npz:", {k: v.shape for k, v in out.items() if not k.startswith("sd.")})
    print("state dict keys:", [k[3:] for k in out if k.startswith("sd.")])


def vq_tail_reference():
    spec = importlib.util.spec_from_file_location("ref_vq_model", "/root/reference/llamagen_tokenizer/tokenizer_image/vq_model.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_vq_model"] = mod
    spec.loader.exec_module(mod)
    torch.manual_seed(0)
    cfg = dict(codebook_size=96, codebook_embed_dim=8, z_channels=32, ch=32, ch_mult=(1, 2), num_res_blocks=1)
    quantize = mod.VectorQuantizer(cfg["codebook_size"], cfg["codebook_embed_dim"], 0.25, 0.0, True, True)
    post_quant_conv = torch.nn.Conv2d(cfg["codebook_embed_dim"], cfg["z_channels"], 1)
    decoder = mod.Decoder(z_channels=cfg["z_channels"], ch=cfg["ch"], ch_mult=cfg["ch_mult"], num_res_blocks=cfg["num_res_blocks"])
    with torch.no_grad():
        # the trained codebook is not unit-norm in storage (get_codebook_entry normalises on every call): make that visible, and
        # move the GroupNorm affines / biases off their 1 / 0 initialisation
        quantize.embedding.weight.mul_(1.0 + torch.rand(cfg["codebook_size"], 1))
        for n, p in decoder.named_parameters():
            if "norm" in n:
                p.add_(0.2 * torch.randn_like(p))
    quantize.eval(); decoder.eval()
    B, side = 2, 4
    ids = torch.randint(0, cfg["codebook_size"], (B, side * side))
    shape = [B, cfg["codebook_embed_dim"], side, side]
    with torch.no_grad():
        zq = quantize.get_codebook_entry(ids.reshape(-1), shape, True)          # VQModel.decode_code (vq_model.py:52-55)
        img = decoder(post_quant_conv(zq))
    out = {"ids": ids.numpy(), "zq": zq.numpy(), "img": img.numpy(),
           "cfg": np.array([cfg["codebook_size"], cfg["codebook_embed_dim"], cfg["z_channels"], cfg["ch"], cfg["num_res_blocks"], *cfg["ch_mult"]])}
    for k, v in quantize.state_dict().items():
        out["sd.quantize." + k] = v.numpy()
    for k, v in post_quant_conv.state_dict().items():
        out["sd.post_quant_conv." + k] = v.numpy()
    for k, v in decoder.state_dict().items():
        out["sd.decoder." + k] = v.numpy()
    # key names of the full-size tokenizer (VQ_16: what vq_ds16_t2i.pt holds), shapes only: a (n_keys, 4) table in name order
    full = mod.VQ_16()
    names = sorted(k for k in full.state_dict().keys())
    out["vq16.nkeys"] = np.array([len(names)])
    for i, k in enumerate(names):
        out[f"vq16.shape.{k}"] = np.array(list(full.state_dict()[k].shape), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "vq_tail_reference.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["lora", "ops", "model", "vq"]
    if "vq" in which:
        vq_tail_reference()
    if "lora" in which:
        lora_reference()
    if "ops" in which:
        oracle_ops()
    if "model" in which:
        reference_model()
    print(os.listdir(HERE))

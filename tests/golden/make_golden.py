"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only (it imports the reference's
own Python from /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

  lora_reference.npz   inputs / state dict / outputs of the reference's models/stage2/lora.py::Linear (both task modes,
                       eval mode) -- pins omnimamba_amd.stack.TaskLoRALinear
  oracle_ops.npz       seeded input/output vectors of the CPU oracle for every op on the path (fp32) -- pins the oracle
                       against drift and is what the HIP kernels are compared with in tests/test_golden.py
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


if __name__ == "__main__":
    lora_reference()
    oracle_ops()
    print(os.listdir(HERE))

"""Developer tool: do several builds of the library compute the SAME bits on the scan (forward with final state, training forward + backward)?
usage: python tools/cmp_libs.py tag=lib.so[:ENV=V,...] ...   (first = reference; empty path = the product library)"""
import os
import subprocess
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys, torch
sys.path.insert(0, %r)
import omnimamba_amd._lib as LB
if os.environ.get("AB_LIB"):
    LB.LIB_PATH = os.environ["AB_LIB"]
    LB._LIB = LB.load(LB.LIB_PATH)
import omnimamba_amd.ssd_combined as S
dev = torch.device("cuda:0")
out = {}
for (B, L, H) in [(2, 1000, 8), (1, 2100, 4), (8, 300, 64)]:
    torch.manual_seed(B * 1000 + L)
    P, N, G = 64, 128, 1
    x = torch.randn(B, L, H, P, device=dev).bfloat16(); Bm = torch.randn(B, L, G, N, device=dev).bfloat16(); Cm = torch.randn(B, L, G, N, device=dev).bfloat16()
    dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
    dtb = torch.randn(H, device=dev) * 0.5 - 3
    y, _, fin = S.ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, return_final_states=True)
    out[f"y{B}x{L}"] = y.cpu(); out[f"fin{B}x{L}"] = fin.cpu()
    lv = [t.clone().requires_grad_() for t in (x, dt, A, Bm, Cm, D, dtb)]
    yy = S.mamba_chunk_scan_combined(lv[0], lv[1], lv[2], lv[3], lv[4], 256, D=lv[5], dt_bias=lv[6], dt_softplus=True)
    yy.backward(torch.ones_like(yy))
    out[f"yt{B}x{L}"] = yy.detach().cpu()
    for n, t in zip(["dx", "ddt", "dA", "dB", "dC", "dD", "ddtb"], lv):
        out[f"{n}{B}x{L}"] = t.grad.cpu()
torch.save(out, os.environ["AB_OUT"])
''' % ROOT
specs = []
for a in sys.argv[1:]:
    tag, rest = a.split("=", 1)
    path, _, envs = rest.partition(":")
    specs.append((tag, os.path.abspath(path) if path else "", dict(e.split("=", 1) for e in envs.split(",") if e)))
import torch
ref = None
with tempfile.TemporaryDirectory() as td:
    for tag, lib, env in specs:
        f = os.path.join(td, tag + ".pt")
        subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, AB_LIB=lib, AB_OUT=f, **env), check=True)
        cur = torch.load(f)
        if ref is None:
            ref = cur
            print(f"{tag}: reference ({len(cur)} tensors)")
            continue
        bad = []
        for k in ref:
            if not torch.equal(ref[k], cur[k]):
                e = ((ref[k].double() - cur[k].double()).norm() / ref[k].double().norm().clamp_min(1e-30)).item()
                if not (k.startswith(("dA", "dD", "ddtb", "ddt")) and e < 1e-5):   # (sums formed with float atomics: equal to rounding)
                    bad.append((k, e))
        print(f"{tag}: " + ("bit-identical" if not bad else f"DIFFERENT {bad}"), flush=True)

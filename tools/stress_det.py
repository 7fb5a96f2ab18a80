"""Developer tool: run the forward / backward scans repeatedly on identical inputs and compare bitwise (race detector)."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_fwd, ssd_scan_bwd  # noqa: E402

dev = torch.device("cuda:0")
H, P, N, G = 64, 64, 128, 1
bad = 0
for (B, L, iters) in [(2, 130, 300), (1, 64, 300), (8, 1024, 100), (8, 4096, 30), (3, 1000, 100),
                      (1, 2048, 100), (1, 8192, 60), (2, 4096, 60)]:      # the last three: split sequences (state passes + folds)
    torch.manual_seed(B * 1000 + L)
    xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
    x = xBC[..., :H * P].view(B, L, H, P)
    Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N)
    Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
    dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16()
    A = -(torch.rand(H, device=dev) * 15 + 1)
    D = torch.ones(H, device=dev)
    dtb = torch.randn(H, device=dev) * 0.5 - 3
    ref = None
    nbad = 0
    for i in range(iters):
        out, xo, fs = ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True, return_final_states=True)
        cur = (out.clone(), fs.clone())
        if ref is None:
            ref = cur
            dout = torch.randn_like(out)
        else:
            for a_, b_ in zip(ref, cur):
                if not torch.equal(a_, b_):
                    nbad += 1
                    d = (a_.float() - b_.float()).abs()
                    print(f"  fwd mismatch B={B} L={L} iter {i}: max {d.max().item():.3e} at {d.argmax().item()} nnz {(d > 0).sum().item()}", flush=True)
                    break
    refb = None
    for i in range(max(iters // 3, 10)):
        g = ssd_scan_bwd(dout, x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)
        cur = (g["dx"].clone(), g["dB"].clone(), g["dC"].clone())
        if refb is None:
            refb = cur
        else:
            for name, a_, b_ in zip(("dx", "dB", "dC"), refb, cur):
                if not torch.equal(a_, b_):
                    nbad += 1
                    d = (a_.float() - b_.float()).abs()
                    print(f"  {name} mismatch B={B} L={L} iter {i}: max {d.max().item():.3e} nnz {(d > 0).sum().item()}", flush=True)
                    break
    print(f"B={B} L={L}: {iters} fwd runs, mismatching runs: {nbad}", flush=True)
    bad += nbad
print("TOTAL mismatches", bad)

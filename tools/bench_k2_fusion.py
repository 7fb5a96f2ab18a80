"""Developer tool (round 4): what fusing the gate of the gated RMSNorm into the scan epilogue would buy.
Times, at the configs[1] shape, (a) the product forward: plain scan + norm_gated_fwd, (b) the scan with the z gate and the pre-gate
copy in its epilogue (the EXTRAS instantiation: reads z, writes y and y silu(z)) and (c) what is left of the norm after (b): a pass
that reads g and writes g * rstd * w (its lower bound: the norm kernel without gate), plus the per-row sum of squares."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.ssd_combined import ssd_scan_fwd  # noqa: E402
from omnimamba_amd.layernorm_gated import rmsnorm_fn  # noqa: E402
from tools.bench_scan import timeit  # noqa: E402

dev = torch.device("cuda:0")
H, P, N, G = 64, 64, 128, 1
for (B, L) in [(8, 4096)]:
    torch.manual_seed(0)
    zx = torch.randn(B, L, 2 * H * P + 2 * G * N, device=dev).bfloat16()
    z = zx[..., :H * P].view(B, L, H, P)
    x = zx[..., H * P:2 * H * P].view(B, L, H, P)
    Bm = zx[..., 2 * H * P:2 * H * P + G * N].view(B, L, G, N)
    Cm = zx[..., 2 * H * P + G * N:].view(B, L, G, N)
    dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16()
    A = -(torch.rand(H, device=dev) * 15 + 1)
    D = torch.ones(H, device=dev)
    dtb = torch.randn(H, device=dev) * 0.5 - 3
    w = torch.ones(H * P, device=dev)
    y = ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True)[0]
    t_scan = timeit(lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, dt_bias=dtb, dt_softplus=True), 20, 5)
    t_norm = timeit(lambda: rmsnorm_fn(y.view(B, L, H * P), w, None, z=z.reshape(B, L, H * P), eps=1e-5, norm_before_gate=False), 20, 5)
    t_scan_gate = timeit(lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dtb, dt_softplus=True, want_out_x=True), 20, 5)
    t_scan_gate_nox = timeit(lambda: ssd_scan_fwd(x, dt, A, Bm, Cm, D=D, z=z, dt_bias=dtb, dt_softplus=True), 20, 5)
    t_norm_nogate = timeit(lambda: rmsnorm_fn(y.view(B, L, H * P), w, None, eps=1e-5), 20, 5)
    print(f"B={B} L={L}:  plain scan {t_scan*1e3:.1f} us + gated norm {t_norm*1e3:.1f} us = {(t_scan+t_norm)*1e3:.1f} us")
    print(f"            scan with gate + pre-gate copy in the epilogue {t_scan_gate*1e3:.1f} us (without the pre-gate copy {t_scan_gate_nox*1e3:.1f} us)")
    print(f"            norm without gate (read g, write g rstd w: the pass that remains unless rstd moves behind out_proj) {t_norm_nogate*1e3:.1f} us")
    print(f"            fused forward = {(t_scan_gate+t_norm_nogate)*1e3:.1f} us vs {(t_scan+t_norm)*1e3:.1f} us now")

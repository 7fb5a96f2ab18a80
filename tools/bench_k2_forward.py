"""Developer tool (round 6): what the K2 fusion of the forward-only path is worth -- conv1d + SiLU of the x channels inside the scan's staging
(OmkSsdFwd.conv_weight) against conv kernel + scan -- on the fused node of one 1.3B-shaped block (B 8 x L 4096, B 1 x L 780 / 2048, bf16, no_grad),
and on the prefill of the whole 48-layer stack in bf16 (time to the first token of a 780-position MMU prompt, scripts/inference_mmu.py:137-147).
OMK_K2_FUSED is read per call, so both forms run in one process on the same box.  -> profiles/r06_k2_fusion.txt"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omnimamba_amd.ssd_combined as S  # noqa: E402
from omnimamba_amd._lib import get_lib  # noqa: E402

dev = torch.device("cuda:0")
H, P, N, G, DM = 64, 64, 128, 1, 2048
d_ssm = H * P


def t_ms(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


torch.manual_seed(0)
cw, cb = torch.randn(d_ssm + 2 * G * N, 4, device=dev) * 0.3, torch.randn(d_ssm + 2 * G * N, device=dev) * 0.1
dtb, A, D = torch.randn(H, device=dev) * 0.5 - 3, -(torch.rand(H, device=dev) * 15 + 1), torch.ones(H, device=dev)
nw, wo = torch.ones(d_ssm, device=dev), (torch.randn(DM, d_ssm, device=dev) * 0.02).bfloat16()
print("fused node of one block, forward only (conv + scan + gated norm + out_proj), microseconds; scan-side kernels from omk_ssd_last_kernels")
os.environ["OMK_K2_MIN_WGS"] = "1"      # (measure the fused form at every size; the default takes it only when the scan fills the chip)
for Bsz, L in (((8, 4096),) if os.environ.get("K2_ONLY") else ((8, 4096), (16, 2048), (6, 4096), (4, 2048), (1, 780))):
    zx = (torch.randn(Bsz, L, 2 * d_ssm + 2 * G * N + H, device=dev) * 0.8).bfloat16()
    row = []
    for mode in ("0", "1"):
        os.environ["OMK_K2_FUSED"] = mode
        cs = torch.zeros(Bsz, d_ssm + 2 * G * N, 4, dtype=torch.bfloat16, device=dev)
        with torch.no_grad():
            f = lambda: S.mamba_split_conv1d_scan_combined(zx, cw, cb, dtb, A, D, 256, return_final_states=True, rmsnorm_weight=nw, rmsnorm_eps=1e-5,
                                                           outproj_weight=wo, headdim=P, ngroups=G, norm_before_gate=False, conv_state_out=cs)
            o = f()
            row.append((t_ms(f) * 1e3, get_lib().omk_ssd_last_kernels().decode(), o[0].float().cpu(), o[1].cpu()))
    same = torch.equal(row[0][2], row[1][2]) and torch.equal(row[0][3], row[1][3])
    print(f"  B {Bsz} L {L:5d}:  separate {row[0][0]:8.1f}   fused {row[1][0]:8.1f}   ({row[1][0] - row[0][0]:+7.1f} us, results {'bit-identical' if same else 'DIFFERENT'})   [{row[1][1]}]")

if os.environ.get("K2_ONLY"):
    sys.exit(0)
# the 48-layer stack in bf16: prefill of a 780-position prompt through the cached-decode path (time to first token)
from omnimamba_amd.generation import decode  # noqa: E402
from omnimamba_amd.stack import OmniMambaLM, StackConfig  # noqa: E402
cfg = StackConfig.omnimamba_1_3b()
model = OmniMambaLM(cfg, device=dev, dtype=torch.bfloat16).eval()
for Bsz, Pn in ((1, 780), (8, 780), (16, 780)):
    ids = torch.zeros(Bsz, Pn, dtype=torch.long, device=dev)
    emb = (torch.randn(Bsz, Pn, cfg.d_model, device=dev) * 0.02).bfloat16()
    res = []
    for mode in ("0", "1"):
        os.environ["OMK_K2_FUSED"] = mode
        os.environ["OMK_PREFILL_GRAPH"] = "0"
        best, tok = 1e9, None
        for _ in range(4):
            model._decoding_cache = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = decode(ids, emb, model, Pn + 1, top_k=1, task="mmu", cg=False)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
            tok = out[:, -1].cpu()
        res.append((best * 1e3, tok))
    print(f"1.3B bf16 prefill of {Pn} positions x batch {Bsz} + first token (eager): separate {res[0][0]:7.2f} ms   fused {res[1][0]:7.2f} ms   same token: {bool((res[0][1] == res[1][1]).all())}")

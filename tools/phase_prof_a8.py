"""Developer tool: per-phase cycle breakdown of the specialised-wave forward scan (ssd_a8.hip; -DOMK_PHASE_PROF build, OMK_PROF=1):
compute waves 0..3 and helper waves 4..7 of workgroup 0.   usage: python tools/with_lib.py <prof lib> tools/phase_prof_a8.py"""
import os, sys
os.environ["OMK_PROF"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd import _capi as K
from omnimamba_amd._lib import get_lib

dev = torch.device("cuda:0")
B = int(os.environ.get("PB", "8"))
L, H, P, N, G = int(os.environ.get("PL", "4096")), 64, 64, 128, 1
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
lib = get_lib()
out = torch.empty(B, L, H, P, dtype=x.dtype, device=dev)
p = K.SsdFwd(x=K.T(x), dt=K.T(dt), A=K.T(A), Bm=K.T(Bm), Cm=K.T(Cm), D=K.T(D), z=K.T(None), dt_bias=K.T(dtb), initial_states=K.T(None), out=K.T(out),
             out_x=K.T(None), final_states=K.T(None), dt_min=0.0, dt_max=float("inf"), dt_softplus=1, chunk_size=256, force_generic=0, flags=K.SSD_NO_SPLIT | int(os.environ.get("AB_FLAGS", "0")))
ws = K.workspace(lib, "omk_ssd_scan_fwd_workspace_bytes", p, x)
for _ in range(2):
    K.run(lib, "omk_ssd_scan_fwd", p, x)
torch.cuda.synchronize()
off = ((B * H * L * 4 + 255) // 256) * 256
nC = L // 64
prof = ws[off:off + 8 * 12 * 8].view(torch.int64).cpu().view(8, 12)
cn = ["phase 1 (0)", "phase 2 (0)", "phase 1 (1)", "barrier", "phase 2 (1)"]
hn = ["build loads, commit + prefetch K", "tile 0", "commit + prefetch Q", "scalars | tile 1", "commit + prefetch U, dt", "barrier"]
print(f"B={B} L={L}: cycles per chunk")
for w in range(4):
    r = prof[w].double() / nC
    print(f"compute {w}: " + "  ".join(f"{n} {float(r[i]):6.0f}" for i, n in enumerate(cn)) + f"   sum {float(r[:5].sum()):7.0f}")
for w in range(4, 8):
    r = prof[w].double() / nC
    print(f"helper  {w}: " + "  ".join(f"{n} {float(r[i]):6.0f}" for i, n in enumerate(hn)) + f"   sum {float(r[:6].sum()):7.0f}")
core, ref = prof[0][10].item(), prof[0][11].item() & ((1 << 40) - 1)
print("HW_ID (simd = bits 5:4, wave slot = bits 3:0):", [f"w{w}: simd {(prof[w][11].item() >> 44) & 3} slot {(prof[w][11].item() >> 40) & 15}" for w in range(8)])
print(f"core clock during the loop: {core} cycles / {ref} ticks of the 100 MHz reference = {core / max(ref, 1) * 100:.0f} MHz   (loop {ref / 100:.1f} us)")

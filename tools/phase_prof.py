"""Developer tool: per-phase cycle breakdown of the forward scan (OMK_PROF=1 makes workgroup 0 dump s_memtime deltas)."""
import os, sys
os.environ["OMK_PROF"] = "1"
os.environ["OMK_SSD_NO_SPLIT"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd import _capi as K
from omnimamba_amd._lib import get_lib
import omnimamba_amd.ssd_combined as S

dev = torch.device("cuda:0")
B = int(os.environ.get("PB", "8"))
L, H, P, N, G = 4096, 64, 64, 128, 1
torch.manual_seed(0)
xBC = torch.randn(B, L, H * P + 2 * G * N, device=dev).bfloat16()
x = xBC[..., :H * P].view(B, L, H, P); Bm = xBC[..., H * P:H * P + G * N].view(B, L, G, N); Cm = xBC[..., H * P + G * N:].view(B, L, G, N)
dt = (torch.randn(B, L, H, device=dev) * 0.5).bfloat16(); A = -(torch.rand(H, device=dev) * 15 + 1); D = torch.ones(H, device=dev)
dtb = torch.randn(H, device=dev) * 0.5 - 3
lib = get_lib()
out = torch.empty(B, L, H, P, dtype=x.dtype, device=dev)
p = K.SsdFwd(x=K.T(x), dt=K.T(dt), A=K.T(A), Bm=K.T(Bm), Cm=K.T(Cm), D=K.T(D), z=K.T(None), dt_bias=K.T(dtb), initial_states=K.T(None), out=K.T(out),
             out_x=K.T(None), final_states=K.T(None), dt_min=0.0, dt_max=float("inf"), dt_softplus=1, chunk_size=256, force_generic=0)
ws = K.workspace(lib, "omk_ssd_scan_fwd_workspace_bytes", p, x)
for _ in range(2):
    K.run(lib, "omk_ssd_scan_fwd", p, x)
torch.cuda.synchronize()
off = ((B * H * L * 4 + 255) // 256) * 256
nC = L // 64
prof = ws[off:off + 4 * 12 * 8].view(torch.int64).cpu().view(4, 12)
names = ["top", "Q.S", "waitX", "G+M+MU", "S-update", "pub+commit", "epilogue", "waitY"]
print(f"B={B}: cycles per chunk, per wave (= strip)   " + " ".join(f"{n:>10s}" for n in names) + "      total")
for w in range(4):
    row = prof[w].double()[:8] / nC
    print(f"wave {w}:                         " + " ".join(f"{v:10.0f}" for v in row.tolist()) + f" {row.sum():10.0f}")
core, ref = prof[0][10].item(), prof[0][11].item()
print(f"core clock during the kernel: {core} cycles / {ref} ticks of the 100 MHz reference = {core / max(ref, 1) * 100:.0f} MHz   (kernel {ref / 100:.1f} us)")

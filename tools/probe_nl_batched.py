"""Developer probe: one variant of the fused projection per process (VARIANT env), many calls over rotating weight
copies, for `rocprofv3 --kernel-trace --stats` (tools/probe_nl_batched.sh)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.norm_linear import norm_linear  # noqa: E402

dev = torch.device("cuda:0")
var, B = os.environ.get("VARIANT", "in_plain"), int(os.environ.get("NB", "8"))
dt = torch.float32
if var.startswith("in"):
    Out, In = 8512, 2048
else:
    Out, In = 2048, 4096
Ws = [torch.randn(Out, In, device=dev, dtype=dt) * 0.02 for _ in range(12)]
x, res, z = torch.randn(B, In, device=dev, dtype=dt), torch.randn(B, In, device=dev), torch.randn(B, In, device=dev, dtype=dt)
nw = torch.ones(In, device=dev, dtype=dt)
la, lb = torch.randn(8, In, device=dev, dtype=dt) * 0.02, torch.randn(Out, 8, device=dev, dtype=dt) * 0.02
kw = dict(norm_weight=nw, eps=1e-5)
if var in ("in_res", "in_lora"):
    kw.update(residual=res, residual_out_dtype=torch.float32)
if var == "in_lora":
    kw.update(lora_a=la, lora_b=lb, lora_scale=4.0)
if var == "out_gate":
    kw.update(z=z)
for i in range(240):
    norm_linear(x, Ws[i % 12], None, **kw)
torch.cuda.synchronize()

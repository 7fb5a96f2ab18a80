"""Developer tool: which aten ops launch the stray elementwise kernels in one block step."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.mamba2 import Mamba2  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device("cuda:0")
blk = Mamba2(2048, d_state=128, headdim=64, layer_idx=0, device=dev)
u = torch.randn(8, 4096, 2048, device=dev, dtype=torch.bfloat16)
dy = torch.randn_like(u)


def step():
    ur = u.detach().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = blk(ur)
    y.backward(dy)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60, max_shapes_column_width=60))

"""Developer tool: torch-profiler table of ONE eager decode step of a 4-layer stack (which aten op launches what)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnimamba_amd.generation import InferenceParams  # noqa: E402
from omnimamba_amd.stack import OmniMambaLM, StackConfig  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

dev = torch.device("cuda:0")
cfg = StackConfig(d_model=2048, n_layer=4)
model = OmniMambaLM(cfg, device=dev, dtype=torch.float32).eval()
ip = InferenceParams(max_seqlen=64, max_batch_size=1)
emb = torch.randn(1, 8, cfg.d_model, device=dev)
with torch.no_grad():
    model(None, emb, task="t2i", inference_params=ip, num_last_tokens=1)
    ip.seqlen_offset = 8
    ids = torch.zeros(1, 1, dtype=torch.long, device=dev)
    pos = torch.full((1, 1), 8, dtype=torch.long, device=dev)
    for _ in range(3):
        model(ids, None, position_ids=pos, task="t2i", inference_params=ip, num_last_tokens=1)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        model(ids, None, position_ids=pos, task="t2i", inference_params=ip, num_last_tokens=1)
        torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=50, max_shapes_column_width=70))

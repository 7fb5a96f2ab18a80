"""Developer probe: the batched fused projection (decode, 2 - 8 sequences) at the 1.3B shapes, vector form against matrix-pipe form
(OMK_NL_MFMA is read once per process: the tool re-runs itself per setting).  Rotating weight copies (no cache reuse), calls captured in a graph."""
import os
import subprocess
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if "OMK_NL_MFMA" not in os.environ:
    for m in ("0", "1"):
        subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, OMK_NL_MFMA=m), check=False)
    sys.exit(0)
if "NL_DT" in os.environ and "OMK_NLF_WPC_LIST" in os.environ and "OMK_NLF_WPC" not in os.environ:   # one sequence: grid sizing of the fast kernel, same box
    for rnd in range(2):
        for w in os.environ["OMK_NLF_WPC_LIST"].split(","):
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, OMK_NLF_WPC=w), check=False)
    sys.exit(0)
from omnimamba_amd.norm_linear import norm_linear  # noqa: E402

dev = torch.device("cuda:0")
print(f"OMK_NL_MFMA={os.environ['OMK_NL_MFMA']} OMK_NLF_WPC={os.environ.get('OMK_NLF_WPC', '-')}")
for dt in [getattr(torch, d) for d in os.environ.get("NL_DT", "bfloat16").split(",")]:
    for var in os.environ.get("NL_VARS", "in_lora_conv,in_plain,out_gate").split(","):
        Out, In = (8512, 2048) if var.startswith("in") else (2048, 4096)
        Ws = [(torch.randn(Out, In, device=dev) * 0.02).to(dt) for _ in range(12)]
        for B in [int(b) for b in os.environ.get("NL_B", "2,4,8").split(",")]:
            x, res, z = torch.randn(B, In, device=dev).to(dt), torch.randn(B, In, device=dev), torch.randn(B, In, device=dev).to(dt)
            nw = torch.ones(In, device=dev, dtype=dt)
            la, lb = (torch.randn(8, In, device=dev) * 0.02).to(dt), (torch.randn(Out, 8, device=dev) * 0.02).to(dt)
            cst = torch.randn(B, 4, 4352, device=dev).to(dt).transpose(1, 2)
            cw, cb = torch.randn(4352, 4, device=dev).to(dt), torch.randn(4352, device=dev).to(dt)
            kw = dict(norm_weight=nw, eps=1e-5)
            if "res" in var or var == "in_lora_conv":
                kw.update(residual=res, residual_out_dtype=torch.float32)
            if "lora" in var:
                kw.update(lora_a=la, lora_b=lb, lora_scale=4.0)
            if "conv" in var:
                kw.update(conv_state=cst, conv_weight=cw, conv_bias=cb, conv_offset=4096)
            if var == "out_gate":
                kw.update(z=z)
            best = 1e9
            for i in range(12):
                norm_linear(x, Ws[i % 12], None, **kw)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()       # (a call from Python costs 15 - 22 us of host time: the kernels are timed inside a captured graph)
            with torch.cuda.graph(gr):
                for i in range(48):
                    norm_linear(x, Ws[i % 12], None, **kw)
            for _ in range(3):
                gr.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    gr.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 240 * 1e3)
            byt = Out * In * Ws[0].element_size()
            print(f"  {str(dt)[6:]:9s} {var:13s} B={B}  {best:7.2f} us   {byt / best / 1e3:7.0f} GB/s over the weights")

"""Record hipBLASLt / rocBLAS solutions for the block's GEMM shapes with PyTorch TunableOp (run on an MI355X):
    python tools/tune_gemms.py          -> omnimamba_amd/tuned/gemm_gfx950_block_1p3b.csv
"""
import os
import shutil
import sys

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "omnimamba_amd", "tuned", "gemm_gfx950_block_1p3b.csv")
tmp = "/tmp/omk_tunableop.csv"
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=tmp, OMK_GEMM_TUNING="0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "2", "--warmup", "2"]
import runpy  # noqa: E402

runpy.run_path(os.path.join(os.path.dirname(out), "..", "..", "bench.py"), run_name="__main__")
import torch  # noqa: E402

torch.cuda.tunable.write_file(tmp) if hasattr(torch.cuda.tunable, "write_file") else None
src = tmp if os.path.exists(tmp) else tmp.replace(".csv", "0.csv")
shutil.copy(src, out)
print("wrote", out)

"""Developer tool: run another tool against a different build of the library (same-box A/B).
usage: python tools/with_lib.py path/to/lib.so tools/bench_conv.py [args]"""
import os
import runpy
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omnimamba_amd._lib as LB  # noqa: E402

LB.LIB_PATH = os.path.abspath(sys.argv[1])
LB._LIB = LB.load(LB.LIB_PATH)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

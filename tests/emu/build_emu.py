"""Build the HOST (emulated) variant of the kernels: tests/emu/libomk_emu.so.  Test infrastructure only."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "omnimamba_amd", "csrc")
LIB = os.path.join(HERE, "libomk_emu.so")
CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-DOMK_EMU", "-O1", "-g", "-std=c++17", "-fPIC", "-I", HERE, "-Wno-unused-value",
         "-ffp-contract=off"]


def build(verbose=False, force=False):
    objdir = os.path.join(HERE, "obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(HERE, "hip_emu.h"), os.path.join(ROOT, "include", "omk.h")]
    hm = max(os.path.getmtime(d) for d in deps)
    jobs, objs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hm):
            jobs.append((s, o))

    def cc(j):
        cmd = [CXX, *FLAGS, "-c", j[0], "-o", j[1]]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(cc, jobs))
    if jobs or not os.path.exists(LIB):
        subprocess.run([CXX, "-shared", "-fPIC", "-Wl,-Bsymbolic", *objs, "-o", LIB], check=True)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))

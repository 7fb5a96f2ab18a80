"""Build libomnimamba_hip.so for gfx950 (MI355X) with hipcc.  Cross-compiles without a GPU.

    python -m omnimamba_amd.build            # incremental, in-tree: omnimamba_amd/lib/libomnimamba_hip.so
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
# OMK_LIB_SUFFIX: a second, developer build next to the product library (e.g. _prof with OMK_PHASE_PROF=1), used with
# tools/with_lib.py for same-box A/B runs
SUFFIX = os.environ.get("OMK_LIB_SUFFIX", "")
LIB = os.path.join(LIBDIR, f"libomnimamba_hip{SUFFIX}.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value",
         "-ffp-contract=fast"] + (["-DOMK_PHASE_PROF"] if os.environ.get("OMK_PHASE_PROF") else []) + os.environ.get("OMK_EXTRA_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "omk.h"))
    return max(os.path.getmtime(h) for h in hs)


# per-file flags (none at present; the one-wave-per-SIMD experiment that needed one left the tree in round 5)
FILE_FLAGS = {}


def _compile(src, obj, verbose):
    cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build(verbose: bool = True, force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj" + SUFFIX)
    os.makedirs(objdir, exist_ok=True)
    hm = _headers_mtime()
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(lambda j: _compile(j[0], j[1], verbose), jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)

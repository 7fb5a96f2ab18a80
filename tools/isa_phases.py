"""Per-phase instruction mix of a kernel compiled with -DOMK_ISA_MARKS (omk_platform.h: OMK_ISA_MARK leaves `; @@PHASE name` comment
lines in the ISA): static counts of every instruction between two markers, over ALL blocks of the kernel between them (a
wave-dependent branch, e.g. the strip-dependent intra phase of the forward scan, is reported as the sum over its paths and per path).

    hipcc --offload-arch=gfx950 -O3 -DOMK_ISA_MARKS -S --cuda-device-only -o k.s kernel.hip
    python tools/isa_phases.py k.s <mangled-kernel-substring>
"""
import re
import sys
from collections import Counter, OrderedDict

from isa_breakdown import classify, kernel_lines

COLS = ["VALU", "TRANS", "CVT", "MFMA", "LDS", "VMEM", "SALU", "WAIT", "NOP", "BARRIER"]


def main():
    lines = kernel_lines(sys.argv[1], sys.argv[2])
    phases = OrderedDict()
    cur = "(before the first marker: prologue)"
    phases[cur] = Counter()
    for ln in lines:
        m = re.search(r"@@PHASE (.*)", ln)
        if m:
            cur = m.group(1).strip()
            phases.setdefault(cur, Counter())
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")) or re.match(r"^\.?LBB", s):
            continue
        phases[cur][classify(s.split()[0])] += 1
    print(f"{'phase':78s} {'n':>5s} " + " ".join(f"{c:>7s}" for c in COLS))
    for k, c in phases.items():
        print(f"{k[:78]:78s} {sum(c.values()):5d} " + " ".join(f"{c.get(x, 0):7d}" for x in COLS))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()

"""Per-block instruction mix of a kernel's ISA (hipcc -S output): VALU / MFMA / LDS / VMEM / SALU / waitcnt counts per
basic block of the main loop, so that a chunk loop can be priced phase by phase without a GPU.

    python tools/isa_breakdown.py <file.s> <mangled-kernel-substring> [first_label last_label]
"""
import re
import sys
from collections import Counter, OrderedDict


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "MFMA"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith("s_barrier"):
        return "BARRIER"
    if op.startswith("s_nop"):
        return "NOP"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "TRANS"
    if op.startswith("v_cvt_pk_bf16"):
        return "CVT"
    if op.startswith("v_accvgpr"):
        return "ACCMOV"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def kernel_lines(path, key):
    out, on = [], False
    for ln in open(path):
        if not on and re.match(r"^_Z\w*:", ln) and key in ln:
            on = True
            continue
        if on:
            if ln.strip().startswith("s_endpgm"):
                out.append(ln)
                break
            out.append(ln)
    return out


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = kernel_lines(path, key)
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        blocks[cur].append((op, s))
    lo = sys.argv[3] if len(sys.argv) > 3 else None
    hi = sys.argv[4] if len(sys.argv) > 4 else None
    names = list(blocks)
    if lo:
        names = names[names.index(lo):names.index(hi) + 1]
    tot = Counter()
    print(f"{'block':12s} {'n':>5s}  " + " ".join(f"{c:>7s}" for c in ["VALU", "TRANS", "CVT", "MFMA", "LDS", "VMEM", "SALU", "WAIT", "NOP", "ACCMOV", "BARRIER"]))
    for n in names:
        c = Counter(classify(op) for op, _ in blocks[n])
        tot.update(c)
        br = [s for op, s in blocks[n] if op.startswith(("s_cbranch", "s_branch"))]
        print(f"{n:12s} {len(blocks[n]):5d}  " + " ".join(f"{c.get(k, 0):7d}" for k in ["VALU", "TRANS", "CVT", "MFMA", "LDS", "VMEM", "SALU", "WAIT", "NOP", "ACCMOV", "BARRIER"]) + "   " + "; ".join(b.split()[-1] for b in br))
    print(f"{'total':12s} {sum(tot.values()):5d}  " + " ".join(f"{tot.get(k, 0):7d}" for k in ["VALU", "TRANS", "CVT", "MFMA", "LDS", "VMEM", "SALU", "WAIT", "NOP", "ACCMOV", "BARRIER"]))
    if "--ops" in sys.argv:
        for n in names:
            c = Counter(op for op, _ in blocks[n])
            print(n, dict(c.most_common(40)))


if __name__ == "__main__":
    main()

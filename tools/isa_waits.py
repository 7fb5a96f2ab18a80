"""Developer tool: the sequence of memory requests, waits, branches and barriers of a kernel, from `hipcc -S` (one line per kernel whose mangled
name contains every given substring).  What it is for: a small load written as `cond ? to_f32(p[i]) : 0` or through a run-time dtype switch
compiles to a branch with the conversion inside, i.e. to `s_waitcnt vmcnt(0)` right behind the load -- a chain of dependent round trips
(round 5: ~4 us per decode projection call, 10 round trips at the start of every conv workgroup).
usage: python tools/isa_waits.py omnimamba_amd/csrc/conv1d.hip conv1d_bwd_cl_kernel bf16 [-D...]"""
import re
import subprocess
import sys

src, subs = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("-")]
flags = [a for a in sys.argv[2:] if a.startswith("-")]
out = "/tmp/isa_waits.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value",
                "-ffp-contract=fast", "-S", "--cuda-device-only", "-c", src, "-o", out] + flags, check=True)
t = open(out).read()
for name in re.findall(r"^(_Z\w+):", t, re.M):
    if not all(s in name for s in subs):
        continue
    i = t.index(name + ":")
    j = t.index(".Lfunc_end", i)
    keys = ("global_load", "global_store", "flat_load", "flat_store", "buffer_load", "buffer_store", "s_barrier", "s_cbranch", "s_endpgm", "v_mfma", "global_atomic")
    last, cnt, seq = None, 0, []
    for ln in t[i:j].splitlines():
        ln = ln.strip()
        k = next((k for k in keys if ln.startswith(k)), None)
        if ln.startswith("s_waitcnt") and "vmcnt" in ln:
            k = "w"
        if not k:
            continue
        tag = ln.split()[0] + ("(" + ln.split("vmcnt(")[1].split(")")[0] + ")" if k == "w" else "")
        if tag == last:
            cnt += 1
        else:
            if last:
                seq.append(f"{last} x{cnt}" if cnt > 1 else last)
            last, cnt = tag, 1
    seq.append(f"{last} x{cnt}" if cnt > 1 else last)
    print(name)
    print("  " + " | ".join(seq))

"""Developer tool: profiles/ssd_{fwd,bwd}_traffic.json from the counter summary tools/pmc_r06.sh writes (counters.txt): HBM bytes per launch
(FETCH_SIZE x 2 on gfx950 as MI355X_MICROARCH.md prescribes, WRITE_SIZE as reported; both in KB) and the MFMA-busy fraction
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles),   kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
(= 16 cycles per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16: the share of the dense bf16 matrix peak at the clock the kernel ran at).
The kernel ids the profiled run recorded (tools/prof_train_scan.py, KERNEL_IDS_OUT -> kernel_ids.json next to counters.txt) go into the
files as `kernel_ids`; bench.py compares them with omk_ssd_last_kernels() of its own timed launches and refuses a stale file.
usage: python tools/make_traffic_json.py gpurun_out/r06/pmc_scan/counters.txt profiles/r06_pmc_scan_kernels.txt"""
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ids = json.load(open(os.path.join(os.path.dirname(os.path.abspath(src)), "kernel_ids.json")))
kern, cur = {}, None
for ln in open(src):
    m = re.match(r"^\s+(\w+)\s+([0-9.]+)\s+\(n=", ln)
    if m and cur:
        kern[cur][m.group(1)] = float(m.group(2))
    elif ln.strip() and not ln.startswith(" ") and "omk::" in ln:
        cur = ln.strip()
        kern[cur] = {}
dur = {}
for ln in open(src):
    m = re.match(r"^\s+\d+ calls\s+avg\s+([0-9.]+) us\s+(.*)$", ln)
    if m:
        dur[m.group(2).strip()] = float(m.group(1))


def pick(sub, optional=False):
    ks = [k for k in kern if sub in k]
    if optional and not ks:
        return None
    assert len(ks) == 1, (sub, ks)
    return ks[0]


def traffic(k):
    return kern[k]["FETCH_SIZE"] * 1024 * 2 + kern[k]["WRITE_SIZE"] * 1024


def busy(ks):
    return sum(kern[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k in ks) / (1024 * sum(kern[k]["GRBM_GUI_ACTIVE"] / 8 for k in ks))


method = ("rocprofv3 --pmc in separate passes with --kernel-trace only (tools/pmc_r06.sh); FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md HBM section), "
          "WRITE_SIZE as reported; dt' preparation launch included; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)")
dtp = pick("ssd_dt_prep")
plain, train, dx = pick("ssd_a8_kernel<0, false, false, false, false>"), pick("ssd_a8_kernel<0, true, false, false, false>"), pick("ssd_a8_kernel<2, true, false, false, false>")
cp, fold, fin = pick("ssd_cp_kernel"), pick("ssd_cp_fold_kernel", optional=True), pick("ssd_bwd_finish_par_kernel")   # no fold launch when ssd_cp stores bf16 dB / dC itself
fwd = {
    "kernel": "ssd_a8_kernel<GS_Y, DUMP=false, KHILO=false> (+ ssd_dt_prep_vec_kernel)", "kernel_ids": ids["fwd"],
    "workload": "B=8 L=4096 H=64 P=64 N=128 bf16",
    "fetch_bytes": kern[plain]["FETCH_SIZE"] * 2048, "write_bytes": kern[plain]["WRITE_SIZE"] * 1024,
    "traffic_bytes_per_launch": traffic(plain) + traffic(dtp), "mfma_busy": round(busy([plain, dtp]), 4), "kernel_us": dur.get(plain),
    "with_window_states": {
        "kernel": "ssd_a8_kernel<GS_Y, DUMP=true, KHILO=false> (the forward of a training step)", "kernel_ids": ids["fwd_with_window_states"],
        "fetch_bytes": kern[train]["FETCH_SIZE"] * 2048, "write_bytes": kern[train]["WRITE_SIZE"] * 1024,
        "traffic_bytes_per_launch": traffic(train) + traffic(dtp), "mfma_busy": round(busy([train, dtp]), 4), "kernel_us": dur.get(train)},
    "method": method, "source": tag}
bk = {"ssd_a8_kernel<GS_DX, DUMP> (dx scan + adjoint window states)": dx, "ssd_cp_kernel": cp, "ssd_cp_fold_kernel": fold,
      "ssd_bwd_finish_par_kernel": fin, "ssd_dt_prep_vec_kernel": dtp}
bk = {n: k for n, k in bk.items() if k is not None}
bwd = {"kernel_ids": ids["bwd"], "kernels": {n: traffic(k) for n, k in bk.items()}, "kernel_us": {n: dur.get(k) for n, k in bk.items()},
       "workload": "B=8 L=4096 H=64 P=64 N=128 bf16, forward window states saved by the training forward",
       "traffic_bytes_per_launch": sum(traffic(k) for k in bk.values()), "mfma_busy": round(busy(list(bk.values())), 4),
       "mfma_busy_by_kernel": {n: round(busy([k]), 4) for n, k in bk.items()}, "method": method, "source": tag}
json.dump(fwd, open(os.path.join(ROOT, "profiles", "ssd_fwd_traffic.json"), "w"), indent=1)
json.dump(bwd, open(os.path.join(ROOT, "profiles", "ssd_bwd_traffic.json"), "w"), indent=1)
print(json.dumps(fwd, indent=1))
print(json.dumps(bwd, indent=1))

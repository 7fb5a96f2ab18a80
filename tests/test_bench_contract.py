"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload + roofline + cpu_baseline)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--min-seconds", "0", "--no-train-1p3b", "--no-decode"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "roofline_bwd", "cpu_baseline", "train_1p3b", "train_1p3b_stage2", "selscan_cfg1", "scan_target",
              "decode_1p3b", "sustained"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["unit"] == "M-elements/s" and j["value"] > 0 and abs(j["value"] - 8 * 4096 * 4096 / (j["ms_per_step"] * 1e-3) / 1e6) / j["value"] < 1e-2
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert 0.05 < rf["frac"] < 1.0 and (rf["traffic"] is None or rf["traffic"] >= 0.9 * rf["algorithmic_bytes_per_launch"])
    rb = j["roofline_bwd"]
    assert rb["bound"] == "hbm" and rb["algorithmic_bytes_per_launch"] == 8 * 4096 * 25856 and 0.02 < rb["frac"] < 1.0
    # the PMC files bench.py quotes `traffic` / `mfma_busy` from must have been recorded for the kernels this run timed:
    # bench.py nulls a stale file's numbers and says so; at HEAD the committed files must match
    for key, fn in (("roofline", "ssd_fwd_traffic.json"), ("roofline_bwd", "ssd_bwd_traffic.json")):
        tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
        want = tj["with_window_states"]["kernel_ids"] if key == "roofline" else tj["kernel_ids"]
        assert j[key]["kernel_ids"] == want, (key, j[key]["kernel_ids"], want)
        assert j[key]["traffic"] is not None and "STALE" not in (j[key]["traffic_source"] or "")
    st = j["scan_target"]                          # the north-star target shape: L = 8192, B = 8 and B = 1
    assert st["B8_L8192"]["algorithmic_bytes"] == 8 * 8192 * 17024 and 0.05 < st["B8_L8192"]["frac_of_hbm_peak"] < 1.0 and st["B1_L8192"]["launch_ms"] > 0
    assert j["sustained"] is None                  # --min-seconds 0
    c1 = j["selscan_cfg1"]                         # BASELINE configs[0]: HIP selective_scan next to the CPU restatement
    assert c1["rel_l2_vs_cpu_ref"] < 1e-3 and c1["cpu_ref"]["value"] > 0 and c1["hip_B2"]["value"] > c1["cpu_ref"]["value"]


@pytest.mark.gpu
def test_bench_times_exactly_k_steps_and_reports_the_sustained_region_separately():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--min-seconds", "1.0", "--no-train-1p3b", "--no-selscan-cfg1", "--no-scan-target", "--no-decode"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")][0])
    assert j["steps"] == 3 and j["warmup"] == 1                      # the contract: EXACTLY K timed steps
    assert abs(j["ms_per_step"] - j["timed_seconds"] / 3 * 1e3) < 0.02 * j["ms_per_step"]
    su = j["sustained"]                                              # the >= 1 s region behind it (utilisation sampling), never the headline
    assert su["steps"] > 3 and su["seconds"] >= 0.8
    assert j["roofline"]["launches_timed"] == 3 + su["steps"]        # the scan launches are HIP-event timed over both regions


@pytest.mark.parametrize("n", [1, 2])
def test_bench_launches_itself_for_n_gpus(n):
    """The driver runs `python bench.py --gpus N` WITHOUT a launcher: bench.py must re-execute itself under
    torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).  CPU check of that path with gloo."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-launch"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry_launch": True, "n_gpus": n, "ok": True}, r.stdout[-2000:]


def test_bench_under_an_external_launcher_does_not_relaunch():
    """The documented driver form: torch.distributed.run starts the ranks, bench.py reads RANK / WORLD_SIZE."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29400 + os.getpid() % 500), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


@pytest.mark.gpu
def test_bench_force_dist_runs_the_multi_rank_code_on_rccl_with_one_gpu():
    """The SCALE driver runs `python bench.py --gpus N`: launcher, nccl process group, DDP, barriers and the max over ranks.  With one
    GPU that code never ran (world size 1 skips it).  --force-dist takes exactly that path at N = 1: bench.py re-executes itself under
    torch.distributed.run, initialises RCCL, wraps the block in DDP (32 MB buckets, all-reduce of every gradient) and times K steps."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1", "--min-seconds", "0",
                        "--no-cpu-baseline", "--no-selscan-cfg1", "--no-scan-target", "--no-decode", "--no-train-1p3b"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # the RCCL banner (C stdio, flushed at exit) is not on stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["config"]["process_group"] == "nccl" and j["value"] > 0

mkdir -p gpurun_out/s4
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_ssd.py tests/test_configs_gpu.py tests/test_mamba2_module.py tests/test_context_parallel.py -m gpu -x -q > gpurun_out/s4/ssd_tests.log 2>&1; tail -3 gpurun_out/s4/ssd_tests.log
timeout 300 python bench.py --no-decode > gpurun_out/s4/bench.json 2> gpurun_out/s4/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/s4/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], 'fwd', d['roofline']['frac'], d['roofline']['launch_ms'], 'bwd', d['roofline_bwd']['frac'], d['roofline_bwd']['launch_ms'])
print(d['scan_target']['B8_L4096'])
print(d['selscan_cfg1'].get('hip_B64'), d['selscan_cfg1'].get('hip_B64_channel_last'))
PY

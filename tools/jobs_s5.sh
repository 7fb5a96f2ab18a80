mkdir -p gpurun_out/s5
timeout 600 python -m pytest tests/test_vq_tail.py tests/test_ops_ssd.py -m gpu -x -q > gpurun_out/s5/tests.log 2>&1; tail -3 gpurun_out/s5/tests.log
timeout 400 python bench.py > gpurun_out/s5/bench.json 2> gpurun_out/s5/bench.err
tail -c 900 gpurun_out/s5/bench.json

mkdir -p gpurun_out/s7
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/s7/gpu_tests.log 2>&1; tail -3 gpurun_out/s7/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s7/smoke.log 2>&1; tail -2 gpurun_out/s7/smoke.log
( time timeout 600 python bench.py > gpurun_out/s7/bench.json 2> gpurun_out/s7/bench.err ) 2> gpurun_out/s7/bench_time.txt; cat gpurun_out/s7/bench_time.txt | tail -3
bash tools/prof_bench.sh > gpurun_out/s7/prof_bench.log 2>&1
cp gpurun_out/bench_kernel_stats.csv gpurun_out/s7/ 2>/dev/null
tail -c 300 gpurun_out/s7/bench.json

python3 -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
python3 bench.py --no-cpu-baseline > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -c 300 gpurun_out/bench_now.err
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "call_of_long or calls_of_long" 2>&1 | tail -3
